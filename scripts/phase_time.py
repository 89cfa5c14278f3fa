"""Wall time of windows of the squeeze episode (4096 envs, default scene), for knock-out builds of the library:
SOFTGRIP_LIB=<lib> python scripts/phase_time.py [scene] -> ms per env step in the idle phase (steps 10-39), while closing (50-69),
at the squeeze peak (95-114) and after the release (170-199)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import softgrip_amd as sg  # noqa: E402
from softgrip_amd import native  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402
import torch  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "softbox"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = sg.load_model(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "%s.sgmodel" % scene), "explicit" if scene.startswith("softbox") else "implicit")
b = native.NativeBatch(native.NativeModel(m), n, 0)
b.set_stiffness(np.random.RandomState(0).uniform(300, 1400, n), list(range(11, 64)), [0])
wins = {"idle": (10, 40), "closing": (50, 70), "peak": (95, 115), "released": (170, 200)}
for rep in range(2):
    b.reset(1)
    ctrl = np.zeros(2)
    acc = {k: 0.0 for k in wins}
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
        w = [k for k, (a, z) in wins.items() if a <= t < z]
        if w:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        b.step(7)
        if w:
            torch.cuda.synchronize(); acc[w[0]] += time.perf_counter() - t0
print(os.environ.get("SOFTGRIP_LIB", "default"), scene, " ".join("%s %.3f" % (k, 1e3 * acc[k] / (wins[k][1] - wins[k][0])) for k in wins), "ms/env-step")
