"""Soak / determinism check on the GPU box: E episodes of the reference schedule at 4096 envs, twice from scratch; no env may raise
a flag, every sample must be finite, and the two runs must agree bit for bit.  usage: soak.py [scene] [episodes] [explicit|implicit]   (tendon damper; default: what bench.py uses for the scene)"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import softgrip_amd as sg  # noqa: E402
from softgrip_amd import native  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "softbox_fix"
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 4096
damper = sys.argv[3] if len(sys.argv) > 3 else ("explicit" if scene.startswith("softbox") else "implicit")
m = sg.load_model(os.path.join(ROOT, "models", scene + ".sgmodel"), damper)
nm = native.NativeModel(m)
sched = episode_schedule()
# which joints / tendons carry the per-env stiffness (bench.py stiffness_ids): the four-finger and free-ball scenes have their own
jids = list(range(65, 283)) if scene.startswith("fourfinger") else list(range(9, 227)) if scene.startswith("freeball") else list(range(11, 64))
nsd = nm.nsensordata
digests = []
for run in range(2):
    b = native.NativeBatch(nm, n, 0)
    rng = np.random.RandomState(7)
    out = torch.zeros(n, len(sched), nsd, dtype=torch.float64, device=b.device)
    flags = torch.zeros(n, dtype=torch.int32, device=b.device)
    h = hashlib.sha256()
    nbad = 0
    for ep in range(episodes):
        b.set_stiffness(rng.uniform(300, 1400, n), jids, [0])
        b.reset(1, flags=flags)
        ctrl = np.zeros(nm.nu)
        for t, c in enumerate(sched):
            if c is not None:
                ctrl[:] = c
                b.set_ctrl_broadcast(ctrl)
            b.step(7, sens=out[:, t], sens_stride=len(sched) * nsd, flags=flags)
            nbad += int((flags != 0).sum())
        a = out.cpu().numpy()
        assert np.isfinite(a).all()
        h.update(a.tobytes())
    digests.append(h.hexdigest())
    print("run %d: %d episodes x %d envs, flagged env-steps %d, max |sensor| %.3g, sha256 %s" % (run, episodes, n, nbad, np.abs(a).max(), digests[-1][:16]))
    del b
assert digests[0] == digests[1], "runs differ"
print("deterministic: the two runs agree bit for bit")
