"""GPU-vs-oracle step-by-step comparison (diagnostic; run on the GPU box)."""
import os
import sys
import time

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
    s.reset()
    s.forward()
    s.step()
sens = torch.zeros(n, 12, dtype=torch.float64, device=dev)
flags = torch.zeros(n, dtype=torch.int32, device=dev)
touch = torch.zeros(n, dtype=torch.int32, device=dev)
t0 = time.time()
b.reset(1, sens=sens, flags=flags, touch=touch)
torch.cuda.synchronize()
print("reset kernel wall %.3f s (includes first-launch overhead)" % (time.time() - t0))


def compare(tag):
    st = b.get_state()
    ss = b.solver_stats()
    torch.cuda.synchronize()
    worst = 0
    rows = []
    for e, s in enumerate(sims):
        ds = np.abs(sens[e].cpu().numpy() - s.sensordata).max()
        dq = np.abs(st["qpos"][e].cpu().numpy() - s.qpos).max()
        dv = np.abs(st["qvel"][e].cpu().numpy() - s.qvel).max()
        dw = np.abs(st["qacc_warmstart"][e].cpu().numpy() - s.qacc_warmstart).max()
        da = np.abs(st["act"][e].cpu().numpy() - s.act).max()
        rows.append("  env%d ncon %d/%d nefc %d/%d it %d/%d flags %d dsens %.2e dq %.2e dv %.2e dw %.2e dact %.1e" % (
            e, int(ss["ncon"][e]), s.ncon, int(ss["nefc"][e]), s.nefc, int(ss["iters"][e]), s.solver_iter, int(flags[e]), ds, dq, dv, dw, da))
        worst = max(worst, ds, dq)
    return worst, rows


w, rows = compare("reset")
print("after reset: worst %.3e" % w)
print("\n".join(rows))
sched = episode_schedule()
ctrl = np.zeros(2)
maxw = 0
shown = 0
for t in range(nsteps):
    if sched[t] is not None:
        ctrl[:] = sched[t]
        b.set_ctrl_broadcast(ctrl)
        for s in sims:
            s.ctrl[:] = sched[t]
    b.step(7, sens=sens, flags=flags, touch=touch)
    for s in sims:
        for _ in range(7):
            s.step()
    w, rows = compare(t)
    maxw = max(maxw, w)
    if t % 20 == 0 or (w > 1e-7 and shown < 6):
        print("step %d worst %.3e touch %s" % (t, w, touch.cpu().numpy().tolist()))
        print("\n".join(rows))
        if w > 1e-7:
            shown += 1
    if w > 1e-2:
        print("diverged; stopping")
        break
print("MAX worst diff over run: %.3e" % maxw)
