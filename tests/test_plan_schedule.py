"""The plan's list schedule of the equality rows of a neighbour-row model (csrc/sg_plan.cpp) IS MuJoCo's sequential
sweep: the rows come in blocks (an element's fix row and its neighbour rows), every block exactly once, blocks of one round share no
slider, and any two blocks that share a slider keep their order."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import softgrip_amd as sg
from helpers import ROOT, model_path


SLOTS = 8   # SG_EQ_SLOTS (csrc/sg_plan.h): blocks per round, one per lane pair of an env's 16-lane group


def _schedule(scene):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libsgemu.so"))
    blob = sg.load_model(model_path(scene)).to_blob()
    out = (C.c_int * (4 * 8 * 4096))()
    nelem, nnb = C.c_int(), C.c_int()
    err = C.create_string_buffer(256)
    L.emu_plan_schedule.restype = C.c_int
    ns = L.emu_plan_schedule(blob, C.c_size_t(len(blob)), out, 8 * 4096, C.byref(nelem), C.byref(nnb), err, C.c_size_t(256))
    assert ns > 0, err.value
    return np.array(out[:4 * ns]).reshape(ns, 4), nelem.value, nnb.value


@pytest.mark.parametrize("scene,bound", [("softbox", 24), ("softcylinder", 35), ("softball", 37)])
def test_schedule_is_the_sequential_sweep(scene, bound):
    m = sg.load_model(model_path(scene))
    S, N, nnb = _schedule(scene)                      # per slot: (element e, partners of its up to three neighbour rows); N = none / idle
    e0 = m.nv - N
    # MuJoCo's row order is [fix_e, e's neighbour rows] for e = 0, 1, ...: the blocks, with their partner sliders in row order
    blocks, k = {}, 0
    for q in range(m.neq - 1):
        e1, e2 = int(m.eq_obj1id[q]) - e0, int(m.eq_obj2id[q])
        if e2 < 0:
            assert e1 == len(blocks)                   # fix rows come in element order, each opens its element's block
            blocks[e1] = []
        else:
            assert e1 == len(blocks) - 1               # a neighbour row belongs to the block of the latest fix row
            blocks[e1].append(e2 - e0); k += 1
    assert k == nnb and len(blocks) == N
    real = [(i // SLOTS, int(r[0]), [int(x) for x in r[1:]]) for i, r in enumerate(S) if r[0] < N]
    assert sorted(e for _, e, _ in real) == list(range(N))                                   # every block exactly once
    for _, e, part in real:                                                                  # with its rows' partners, in row order
        assert part == blocks[e] + [N] * (3 - len(blocks[e]))
    for r in S[S[:, 0] >= N]:
        assert (r == N).all()                                                                # idle slots touch the zero word only
    round_of = {e: rnd for rnd, e, _ in real}
    sliders = {e: {e} | set(blocks[e]) for e in range(N)}
    for rnd in set(round_of.values()):                                                       # a round's blocks share no slider
        used = [x for e in range(N) if round_of[e] == rnd for x in sliders[e]]
        assert len(used) == len(set(used)), rnd
    last = {}
    for e in range(N):                                                                       # blocks sharing a slider keep their order
        for x in sliders[e]:
            if x in last:
                assert round_of[last[x]] < round_of[e], (last[x], e, x)
            last[x] = e
    nrounds = len(S) // SLOTS
    assert bound <= nrounds <= int(1.35 * bound)      # the critical path bounds it from below; list scheduling stays close      # the critical path bounds it from below; list scheduling stays close
