"""The plan's list schedule of the equality rows of a neighbour-row model (csrc/sg_plan.cpp) IS MuJoCo's sequential
sweep: every row exactly once, rows of one round share no slider, and any two rows that share a slider keep their id order."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import softgrip_amd as sg
from helpers import ROOT, model_path


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


@pytest.mark.parametrize("scene,bound", [("softbox", 53), ("softcylinder", 69), ("softball", 73)])
def test_schedule_is_the_sequential_sweep(scene, bound):
    m = sg.load_model(model_path(scene))
    S, N, nnb = _schedule(scene)
    e0 = m.nv - N
    # the rows in MuJoCo's order: (e1, e2 or -1), and the record index the kernels use (fix row of e: e, neighbour row k: N + k)
    rows, rec, k = [], [], 0
    for q in range(m.neq - 1):
        e1, e2 = int(m.eq_obj1id[q]) - e0, int(m.eq_obj2id[q])
        if e2 < 0:
            rows.append((e1, -1)); rec.append(e1)
        else:
            rows.append((e1, e2 - e0)); rec.append(N + k); k += 1
    assert k == nnb
    real = S[S[:, 3] < N + nnb]
    assert sorted(real[:, 3].tolist()) == sorted(rec) and len(real) == len(rows)          # every row exactly once
    idle = S[S[:, 3] == N + nnb]
    assert (idle[:, 1] == N).all() and (idle[:, 2] == N).all()                               # idle slots touch the dummy word only
    round_of = {int(r[3]): int(r[0]) for r in real}
    for r in real:                                                                           # slots carry their row's sliders
        e1, e2 = rows[rec.index(int(r[3]))]
        assert int(r[1]) == e1 and int(r[2]) == (e2 if e2 >= 0 else N)
    for rnd in np.unique(real[:, 0]):                                                        # a round's rows share no slider
        used = [int(x) for row in real[real[:, 0] == rnd] for x in row[1:3] if x != N]
        assert len(used) == len(set(used)), rnd
    last = {}
    for i, (e1, e2) in enumerate(rows):                                                      # rows sharing a slider keep their order
        for e in (e1, e2):
            if e < 0:
                continue
            if e in last:
                assert round_of[rec[last[e]]] < round_of[rec[i]], (last[e], i, e)
            last[e] = i
    for i, r in enumerate(S):                                                                # typed lanes: slots 0..2 joint-fix rows, 3..7 neighbour rows
        assert int(r[0]) == i // 8
        if r[3] < N:
            assert i % 8 < 3
        elif r[3] < N + nnb:
            assert i % 8 >= 3
    nrounds = int(S[:, 0].max()) + 1
    assert bound <= nrounds <= int(1.35 * bound)      # the critical path bounds it from below; list scheduling stays close
