import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import softgrip_amd as sg
from softgrip_amd import native
from helpers import random_gripper_xml, oracle_sim
from softgrip_amd.create_dataset import episode_schedule
free, neighbors = True, True
rng = np.random.RandomState(40 + 2 * int(free) + int(neighbors))
for i in range(6):
    xml = random_gripper_xml(rng, free)
    open('/tmp/g%d.xml' % i, 'w').write(xml)
    m = sg.compile_mjcf('/tmp/g%d.xml' % i, composite_neighbors=neighbors)
    nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
    jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
    ks = list(rng.uniform(300, 1400, 3))
print("scene 5: nv", m.nv, "chain joints", nchain, "ngeom", m.ngeom)
sims = []
for k in ks:
    s = oracle_sim(m); s.jnt_stiffness[jids] = k; s.tendon_stiffness[0] = k; s.reset(); s.forward(); s.step(); sims.append(s)
nm = native.NativeModel(m); b = native.NativeBatch(nm, 3, 0); b.set_stiffness(np.array(ks), jids, [0])
sens = torch.zeros(3, nm.nsensordata, dtype=torch.float64, device=b.device); flags = torch.zeros(3, dtype=torch.int32, device=b.device)
b.reset(1, sens=sens, flags=flags)
dev = dict(device=b.device, dtype=torch.float64)
import copy
fails = {}
for t, c in enumerate(episode_schedule()[:4]):
    if c is not None:
        b.set_ctrl_broadcast(np.full(m.nu, c))
        for s in sims: s.ctrl[:] = c
    for j in range(7):
        st0 = dict(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                   act=torch.tensor(np.stack([s.act for s in sims]), **dev), qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
        w = [s.step() for s in sims]
        ref = [(s.ncon, s.nefc, s.solver_iter) for s in sims]
        refs = np.stack([s.sensordata for s in sims])
        for rep in range(int(os.environ.get("REPS", "8")) if t == 3 else 1):
            b.set_state(**st0)
            b.step(1, sens=sens, flags=flags)
            st = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
            got = [(int(a), int(bb), int(cc)) for a, bb, cc in zip(st["ncon"], st["nefc"], st["iters"])]
            err = np.abs(sens.cpu().numpy() - refs).max()
            bad = got != ref or not (err < 1e-9) or any(flags.tolist())
            if bad:
                fails[(t, j)] = fails.get((t, j), 0) + 1
                if fails[(t, j)] <= 2:
                    print(t, j, rep, "gpu", got, "oracle", ref, "flags", flags.tolist(), "err", err)
print("FAILS", fails)
