"""Debugging aid: where does the softcylinder episode's one outlier sample (env step 130) come from?  Replays the oracle's trajectory,
re-synchronising the GPU at every env step, and prints per-env errors and counts around the outlier; then the same step with the
oracle started from the GPU's own state (is the sample a property of the state or of the solve?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import softgrip_amd as sg
from softgrip_amd import native
from helpers import model_path, oracle_sim
from softgrip_amd.create_dataset import episode_schedule
import oracle as O
m = sg.load_model(model_path("softcylinder"), "implicit")
ks = np.linspace(300, 1400, 9)
ids, tids = O.stiffness_ids(m)
sims = []
for k in ks:
    s = oracle_sim(m); s.jnt_stiffness[ids] = k; s.tendon_stiffness[tids] = k; s.reset(); s.forward(); s.step(); sims.append(s)
nm = native.NativeModel(m); b = native.NativeBatch(nm, len(ks), 0); b.set_stiffness(ks, ids, tids)
sens = torch.zeros(len(ks), nm.nsensordata, dtype=torch.float64, device=b.device); flags = torch.zeros(len(ks), dtype=torch.int32, device=b.device)
b.reset(1, sens=sens, flags=flags)
dev = dict(device=b.device, dtype=torch.float64)
for t, c in enumerate(episode_schedule()):
    if c is not None:
        b.set_ctrl_broadcast(np.full(m.nu, c))
        for s in sims: s.ctrl[:] = c
    b.set_state(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                act=torch.tensor(np.stack([s.act for s in sims]), **dev), qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
    for s in sims:
        for _ in range(7): s.step()
    b.step(7, sens=sens, flags=flags)
    st = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
    got = sens.cpu().numpy(); want = np.stack([s.sensordata for s in sims])
    err = np.abs(got - want).max(axis=1)
    gs = b.get_state()
    qerr = np.abs(gs["qpos"].cpu().numpy() - np.stack([s.qpos for s in sims])).max(axis=1)
    if 126 <= t <= 134 or err.max() > 1e-7:
        print(t, "err", np.array2string(err, precision=2), "qerr", np.array2string(qerr, precision=2), "ncon", st["ncon"].tolist(), [s.ncon for s in sims], "iters", st["iters"].tolist(), [s.solver_iter for s in sims])
