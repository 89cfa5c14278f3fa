"""Debugging aid: the tree pipeline's work space on the GPU (a -DSG_DEBUG_WORK build) against the host emulation's, array by array,
after every launch of the fuzz scene's replay (states set from the oracle before every substep)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import softgrip_amd as sg
from softgrip_amd import native
import helpers
from helpers import random_gripper_xml, oracle_sim, TreeEmu
from softgrip_amd.create_dataset import episode_schedule
emu_so = os.path.join(ROOT, "tests", "emu", os.environ.get("EMU_SO", "libsgtreeemu_1ff.so"))
real = C.CDLL
C.CDLL = lambda p, *a, **k: real(emu_so if str(p).endswith("libsgtreeemu.so") else p, *a, **k)
free, neighbors = True, True
rng = np.random.RandomState(40 + 2 * int(free) + int(neighbors))
for i in range(6):
    xml = random_gripper_xml(rng, free)
    open('/tmp/g%d.xml' % i, 'w').write(xml)
    m = sg.compile_mjcf('/tmp/g%d.xml' % i, composite_neighbors=neighbors)
    nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
    jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
    ks = list(rng.uniform(300, 1400, 3))
sims, emus = [], []
for k in ks:
    s = oracle_sim(m); s.jnt_stiffness[jids] = k; s.tendon_stiffness[0] = k; s.reset(); s.forward(); s.step(); sims.append(s)
    e = TreeEmu(m); e.set_stiffness(k, jids, [0]); e.reset(1); emus.append(e)
EL = emus[0].L
EL.temu_cws.restype = C.POINTER(C.c_double); EL.temu_cws.argtypes = [C.c_void_p]
EL.temu_cws_doubles.restype = C.c_longlong; EL.temu_cws_doubles.argtypes = [C.c_void_p]
EL.temu_layout.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
buf = C.create_string_buffer(8192); EL.temu_layout(emus[0].p, buf, 8192)
lay = [(a, int(b)) for a, b in (l.split() for l in buf.value.decode().strip().split("\n"))]
CS = dict(lay)["CS"]; CW = dict(lay)["CW"]
regs = sorted([(o, n) for n, o in lay if n not in ("CS", "CW")])
ncw = EL.temu_cws_doubles(emus[0].p)
nm = native.NativeModel(m); b = native.NativeBatch(nm, 3, 0); b.set_stiffness(np.array(ks), jids, [0])
GL = nm.L
GL.sg_debug_tree_work.restype = C.c_longlong; GL.sg_debug_tree_work.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_longlong]
sens = torch.zeros(3, nm.nsensordata, dtype=torch.float64, device=b.device); flags = torch.zeros(3, dtype=torch.int32, device=b.device)
b.reset(1, sens=sens, flags=flags)
dev = dict(device=b.device, dtype=torch.float64)
gbuf = (C.c_double * ncw)()
shown = 0
for t, c in enumerate(episode_schedule()[:4]):
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
        for i in range(3):
            n = GL.sg_debug_tree_work(b.ptr, i, gbuf, ncw)
            assert n == ncw, (n, ncw)
            g = np.frombuffer(gbuf, dtype=np.float64, count=ncw).copy()
            h = np.ctypeslib.as_array(EL.temu_cws(emus[i].p), shape=(ncw,)).copy()
            bad = []
            for (o, name), (o2, _) in zip(regs[:-1], regs[1:]):
                a, bb = g[o:o2], h[o:o2]
                if name in ("stage", "lds.hdr"):   # only the first nhit records are defined; the header is the device's business
                    continue
                if name.startswith("lds.hit_") or name.startswith("lds.con_") or name == "lds.icnt":
                    ai, bi = a.view(np.int32), bb.view(np.int32)
                    if name == "lds.hit_cnt": continue   # (the device reuses it for the serial list's chain ids)
                    nvalid = emus[i].ncon if "con_" in name else 10**9 if name == "lds.icnt" else int(h[dict((n2, o3) for o3, n2 in regs)["lds.icnt"]:].view(np.int32)[0])
                    ai, bi = ai[:nvalid], bi[:nvalid]
                    if (ai != bi).any():
                        k = int(np.argmax(ai != bi))
                        bad.append("%s[%d] gpu %d emu %d (%d ints off) gpu %s emu %s" % (name, k, ai[k], bi[k], int((ai != bi).sum()), ai[:12].tolist(), bi[:12].tolist()))
                    continue
                if name == "L":   # (the register factorisation leaves the words right of the diagonal alone: never read)
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
                    k = int(np.argmax(d > tol))
                    extra = ""
                    if name == "crow":
                        idx = np.flatnonzero(d > tol)
                        extra = " CS %d CW %d; (contact, word): %s" % (CS, CW, [(int(x // CW), int(x % CW)) for x in idx[:10]])
                    bad.append("%s[%d] gpu %r emu %r (max %.3g, %d words off)%s" % (name, k, float(a[k]), float(bb[k]), np.nanmax(np.where(np.isinf(d), 0, d)), int((d > tol).sum()), extra))
            if (bad or st["ncon"][i] != emus[i].ncon) and shown < int(os.environ.get('SHOW', '2')) and (t, j) >= (3, 4):
                shown += 1
                print("t %d j %d env %d: ncon gpu %d emu %d oracle %d, flags %d" % (t, j, i, st["ncon"][i], emus[i].ncon, sims[i].ncon, flags[i].item()))
                for x in bad: print("     ", x)
        if shown:
            print("stopping at the first substep with a difference"); sys.exit(0)
print("done, differences shown:", shown)
