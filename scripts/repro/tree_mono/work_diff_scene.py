"""r05 reproducer of the r04 "dropped store" (DESIGN 4.10): the tree kernel built as ONE function (-DSGT_X_MONO: every stage pasted into
the kernel, the r04 layout) on the scene of tests/test_gpu_tree.py::test_chain_capacities_on_the_gpu[7-3-202] (two fingers of 7 links x 3
hinges = 21 dofs: the <24> instantiation), substep by substep from the oracle's states; after every launch the env's work space and
LDS block (a -DSG_DEBUG_WORK build copies the LDS block behind the work space) are held array by array against the host emulation of
the same source (tests/emu `make dbg`).  Prints the first substep at which an array differs and which words.

  python soft-grip_amd/build_native.py --ko monodbg -DSGT_X_MONO -DSG_DEBUG_WORK     # the failing layout
  python soft-grip_amd/build_native.py --ko stagedbg -DSG_DEBUG_WORK                 # the product's layout (control)
  make -C tests/emu dbg
  SOFTGRIP_LIB=soft-grip_amd/libsoftgrip_monodbg.so python scripts/repro/tree_mono/work_diff_scene.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SGT_EMU_VARIANT", "dbg")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import softgrip_amd as sg  # noqa: E402
from softgrip_amd import native  # noqa: E402
from helpers import random_gripper_xml, oracle_sim, TreeEmu  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

links, hinges, seed = (int(x) for x in os.environ.get("SCENE", "7,3,202").split(","))
nsteps = int(os.environ.get("STEPS", "3"))
rng = np.random.RandomState(seed)
open("/tmp/repro_scene.xml", "w").write(random_gripper_xml(rng, False, links=links, hinges=hinges, fingers=2))
m = sg.compile_mjcf("/tmp/repro_scene.xml", composite_neighbors=False)
nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
ks = [700.0, 400.0, 1200.0]
sims, emus = [], []
for k in ks:
    s = oracle_sim(m); s.jnt_stiffness[jids] = k; s.tendon_stiffness[0] = k; s.reset(); s.forward(); s.step(); sims.append(s)
    e = TreeEmu(m); e.set_stiffness(k, jids, [0]); e.reset(1); emus.append(e)
EL = emus[0].L
EL.temu_cws.restype = C.POINTER(C.c_double); EL.temu_cws.argtypes = [C.c_void_p]
EL.temu_cws_doubles.restype = C.c_longlong; EL.temu_cws_doubles.argtypes = [C.c_void_p]
EL.temu_layout.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
buf = C.create_string_buffer(8192); EL.temu_layout(emus[0].p, buf, 8192)
lay = [(a, int(b)) for a, b in (ln.split() for ln in buf.value.decode().strip().split("\n"))]
CS, CW = dict(lay)["CS"], dict(lay)["CW"]
regs = sorted([(o, n) for n, o in lay if n not in ("CS", "CW")])
ncw = EL.temu_cws_doubles(emus[0].p)
regs.append((ncw, "end"))
nm = native.NativeModel(m); b = native.NativeBatch(nm, 3, 0); b.set_stiffness(np.array(ks), jids, [0])
GL = nm.L
GL.sg_debug_tree_work.restype = C.c_longlong; GL.sg_debug_tree_work.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_longlong]
sens = torch.zeros(3, nm.nsensordata, dtype=torch.float64, device=b.device); flags = torch.zeros(3, dtype=torch.int32, device=b.device)
b.reset(1, sens=sens, flags=flags)
dev = dict(device=b.device, dtype=torch.float64)
gbuf = (C.c_double * ncw)()
print("library:", os.environ.get("SOFTGRIP_LIB", "product"), " CS", CS, "CW", CW, "work-space doubles", ncw)
shown = 0
for t, c in enumerate(episode_schedule()[:nsteps]):
    if c is not None:
        b.set_ctrl_broadcast(np.full(m.nu, c))
        for s in sims: s.ctrl[:] = c
        for e in emus: e.ctrl[:] = c
    for j in range(7):
        b.set_state(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                    act=torch.tensor(np.stack([s.act for s in sims]), **dev), qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
        for e, s in zip(emus, sims):
            e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.act[:] = s.act; e.warm[:] = s.qacc_warmstart
        for s in sims: s.step()
        for e in emus: e.step(1)
        b.step(1, sens=sens, flags=flags)
        st = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
        got = sens.cpu().numpy()
        for i in range(3):
            n = GL.sg_debug_tree_work(b.ptr, i, gbuf, ncw)
            assert n == ncw, (n, ncw)
            g = np.frombuffer(gbuf, dtype=np.float64, count=ncw).copy()
            h = np.ctypeslib.as_array(EL.temu_cws(emus[i].p), shape=(ncw,)).copy()
            bad = []
            for (o, name), (o2, _) in zip(regs[:-1], regs[1:]):
                a, bb = g[o:o2], h[o:o2]
                if name == "lds.red" and os.environ.get("TAP"):
                    print("      tap (tj_pos tj_vel tj_asm tj_warm tj_A ten_R ten_b t0_L0): gpu %s\n%s emu %s" % (a[8:16].tolist(), " " * 66, bb[8:16].tolist()))
                if name in ("stage", "lds.hdr", "lds.hit_cnt", "lds.csc", "lds.red"):
                    continue
                if name.startswith("lds.hit_") or name.startswith("lds.con_") or name == "lds.icnt":
                    ai, bi = a.view(np.int32), bb.view(np.int32)
                    nvalid = emus[i].ncon if "con_" in name else 10 ** 9 if name == "lds.icnt" else int(h[dict((n2, o3) for o3, n2 in regs)["lds.icnt"]:].view(np.int32)[0])
                    if name == "lds.hit_pair":
                        continue       # (reused for the streams' lists on the device)
                    ai, bi = ai[:nvalid], bi[:nvalid]
                    if (ai != bi).any():
                        k = int(np.argmax(ai != bi))
                        bad.append("%s[%d] gpu %d emu %d (%d ints off)" % (name, k, ai[k], bi[k], int((ai != bi).sum())))
                    continue
                if name == "L":
                    Kc = a.size // (CS * CS)
                    msk = np.tile(np.tril(np.ones((CS, CS), bool)).ravel(), Kc)
                    a, bb = a[:Kc * CS * CS][msk], bb[:Kc * CS * CS][msk]
                if name == "crow":
                    nc = emus[i].ncon
                    a, bb = a[:nc * CW], bb[:nc * CW]
                with np.errstate(invalid="ignore"):
                    d = np.abs(a - bb)
                    d[np.isnan(a) != np.isnan(bb)] = np.inf
                    d[np.isnan(a) & np.isnan(bb)] = 0
                tol = 1e-7 * (1 + np.abs(bb))
                if d.size and (d > tol).any():
                    idx = np.flatnonzero(d > tol)
                    k = int(idx[0])
                    extra = ""
                    if name == "crow":
                        extra = " CS %d CW %d; (contact, word): %s" % (CS, CW, [(int(x // CW), int(x % CW)) for x in idx[:12]])
                    else:
                        extra = " words %s" % idx[:16].tolist()
                    bad.append("%s[%d] gpu %r emu %r (max %.3g, %d of %d words off)%s" % (name, k, float(a[k]), float(bb[k]), np.nanmax(np.where(np.isinf(d), 0, d)), int((d > tol).sum()), d.size, extra))
            if os.environ.get("WHEREIS") and bad:     # where else does a wrong word's value live?  (a clobbered register holds SOMETHING's value)
                oswc = dict((n2, o3) for o3, n2 in regs)["lds.swc"]
                for w in range(16):
                    val = g[oswc + w]
                    if val != h[oswc + w] and np.isfinite(val) and val != 0.0:
                        def names(arr):
                            hits = np.flatnonzero(arr == val)
                            out = []
                            for x in hits[:12]:
                                k2 = max(k3 for k3 in range(len(regs)) if regs[k3][0] <= x)
                                out.append("%s[%d]" % (regs[k2][1], x - regs[k2][0]))
                            return out
                        print("      swc[%d] = %r on the GPU; the same bits elsewhere -- GPU memory: %s; emulation: %s" % (w, float(val), names(g), names(h)))
            serr = float(np.abs(got[i] - sims[i].sensordata).max())
            if bad or st["ncon"][i] != emus[i].ncon or serr > 1e-6:
                shown += 1
                print("t %d substep %d env %d: ncon gpu %d emu %d oracle %d, flags %d, max |sensor - oracle| %.3g" % (t, j, i, st["ncon"][i], emus[i].ncon, sims[i].ncon, flags[i].item(), serr))
                for x in bad: print("     ", x)
        if shown:
            print("stopping at the first substep with a difference"); sys.exit(1)
print("done: no array differs over %d env steps" % nsteps)
