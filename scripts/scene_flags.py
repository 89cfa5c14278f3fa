"""Which per-env flags a scene raises over the reference episode, and when (GPU)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import softgrip_amd as sg
from softgrip_amd import native
from softgrip_amd.create_dataset import episode_schedule

scene = sys.argv[1] if len(sys.argv) > 1 else "softball_fix"
m = sg.load_model("models/%s.sgmodel" % scene)
nm = native.NativeModel(m)
n = 64
b = native.NativeBatch(nm, n, 0)
b.set_stiffness(np.random.RandomState(0).uniform(300, 1400, n), list(range(11, 64)), [0])
sens = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
flags = torch.zeros(n, dtype=torch.int32, device="cuda:0")
touch = torch.zeros(n, dtype=torch.int32, device="cuda:0")
b.reset(1, sens=sens, flags=flags, touch=touch)
print("after reset: flags", np.unique(flags.cpu().numpy(), return_counts=True), "ncon", b.solver_stats()["ncon"].cpu().numpy()[:4])
ctrl = np.zeros(2)
seen = {}
for t, c in enumerate(episode_schedule()):
    if c is not None:
        ctrl[:] = c
        b.set_ctrl_broadcast(ctrl)
    b.step(7, sens=sens, flags=flags, touch=touch)
    f = flags.cpu().numpy()
    for v in np.unique(f[f != 0]):
        if int(v) not in seen:
            seen[int(v)] = t
            st = b.solver_stats()
            print("step %d: first flag value %d on %d envs; ncon max %d, |sens| max %.3g" % (t, v, (f == v).sum(), int(st["ncon"].max()), float(sens.abs().max())))
print("flag values first seen at step:", seen)
