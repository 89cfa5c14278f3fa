"""One-step parity along the oracle's trajectory (diagnostic; run on the GPU box): after every env step the GPU batch is
re-seated on the oracle's state, so the error reported per step is what the kernels add in 7 substeps, not what the
dynamics amplified since the start.  usage: gpu_resync.py [scene] [nsteps] [free]   (free: no re-seating)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import softgrip_amd as sg  # noqa: E402
from oracle import oracle as O  # noqa: E402
from softgrip_amd import native  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "softbox_fix"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
free = len(sys.argv) > 3 and sys.argv[3] == "free"
ks = [700.0, 903.6948543200572, 300.0, 1400.0, 512.25, 350.0, 1000.0, 1250.0, 640.0]
m = sg.load_model(os.path.join(ROOT, "models", scene + ".sgmodel"))
nm = native.NativeModel(m)
n = len(ks)
b = native.NativeBatch(nm, n, 0)
dev = b.device
jids, tids = list(range(11, 64)), [0]
b.set_stiffness(np.array(ks), jids, tids)
om = O.OracleModel(m.to_blob())
sims = [O.OracleSim(om) for _ in ks]
for s, k in zip(sims, ks):
    s.jnt_stiffness[jids] = k
    s.tendon_stiffness[tids] = k
    s.reset(); s.forward(); s.step()
sens = torch.zeros(n, 12, dtype=torch.float64, device=dev)
flags = torch.zeros(n, dtype=torch.int32, device=dev)
touch = torch.zeros(n, dtype=torch.int32, device=dev)
b.reset(1, sens=sens, flags=flags, touch=touch)
sched = episode_schedule()
ctrl = np.zeros(2)
worst = dict(sens=0.0, q=0.0, v=0.0)
for t in range(nsteps):
    if sched[t] is not None:
        ctrl[:] = sched[t]
        b.set_ctrl_broadcast(ctrl)
        for s in sims:
            s.ctrl[:] = sched[t]
    b.step(7, sens=sens, flags=flags, touch=touch)
    ow = [max(s.step() for _ in range(7)) for s in sims]
    st = b.get_state()
    ss = b.solver_stats()
    gs, gq, gv = sens.cpu().numpy(), st["qpos"].cpu().numpy(), st["qvel"].cpu().numpy()
    es = max(np.abs(gs[e] - s.sensordata).max() for e, s in enumerate(sims))
    eq = max(np.abs(gq[e] - s.qpos).max() for e, s in enumerate(sims))
    ev = max(np.abs(gv[e] - s.qvel).max() for e, s in enumerate(sims))
    worst["sens"] = max(worst["sens"], es); worst["q"] = max(worst["q"], eq); worst["v"] = max(worst["v"], ev)
    fl = flags.cpu().numpy().tolist()
    nc = ss["ncon"].cpu().tolist()
    if t % 10 == 0 or es > 1e-8 or any(fl) or any(ow) or nc != [s.ncon for s in sims]:
        print("step %3d  dsens %.2e dq %.2e dv %.2e  ncon gpu %s oracle %s  flags gpu %s oracle %s" % (
            t, es, eq, ev, nc, [s.ncon for s in sims], fl, ow))
    if not free:
        T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=dev).contiguous()
        b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                    qacc_warmstart=T([s.qacc_warmstart for s in sims]))
print("worst one-step errors" if not free else "worst free-running errors", worst)
