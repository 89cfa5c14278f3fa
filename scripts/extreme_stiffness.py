import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch
import softgrip_amd as sg
from softgrip_amd import native
from softgrip_amd.create_dataset import episode_schedule
from helpers import oracle_sim, JOINT_IDS, TENDON_IDS
ks = np.array([1.0, 10.0, 100.0, 300.0, 1400.0, 1e4, 1e5, 1e6])
m = sg.load_model(ROOT + "/models/softbox.sgmodel")
nm = native.NativeModel(m)
b = native.NativeBatch(nm, len(ks), 0)
b.set_stiffness(ks, JOINT_IDS, TENDON_IDS)
sens = torch.zeros(len(ks), 12, dtype=torch.float64, device=b.device)
flags = torch.zeros(len(ks), dtype=torch.int32, device=b.device)
sims = [oracle_sim(m, k) for k in ks]
for s in sims: s.reset(); s.forward(); s.step()
b.reset(1, sens=sens, flags=flags)
ctrl = np.zeros(2); worst = np.zeros(len(ks)); fl = np.zeros(len(ks), int); ow = np.zeros(len(ks), int)
for t, c in enumerate(episode_schedule()):
    if c is not None:
        ctrl[:] = c; b.set_ctrl_broadcast(ctrl)
        for s in sims: s.ctrl[:] = c
    b.step(7, sens=sens, flags=flags)
    for i, s in enumerate(sims):
        for _ in range(7): ow[i] |= s.step()
    g = sens.cpu().numpy()
    worst = np.maximum(worst, np.abs(g - np.stack([s.sensordata for s in sims])).max(1))
    fl |= flags.cpu().numpy()
for k, w, f, o in zip(ks, worst, fl, ow): print("k %8g  max |gpu - oracle| %.2e  gpu flags %d  oracle warnings %d" % (k, w, f, o))
