"""Regression fixture of the CPU oracle: sensor rows of the reference episode (create_dataset.py schedule, 7 substeps per step) for one
stiffness, written to tests/golden/oracle_regression.npz.  These are the oracle's OWN outputs (restatement of mj_step, parity with MuJoCo
unpinned) -- the fixture pins the checker against silent changes, it is not a MuJoCo golden vector.
Run in the build container:  python scripts/gen_oracle_regression.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import softgrip_amd as sg  # noqa: E402
from helpers import model_path, oracle_sim  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

K = 903.6948543200572   # np.random.seed(0); np.random.uniform(300, 1400): the reference's first draw
out = {"stiffness": np.array(K)}
for scene, nsteps in (("softbox_fix", 200), ("softbox", 60)):
    s = oracle_sim(sg.load_model(model_path(scene)), K)
    s.reset(); s.forward(); s.step()
    rows, ncon = [], []
    for t, c in enumerate(episode_schedule()[:nsteps]):
        if c is not None:
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        rows.append(s.sensordata.copy()); ncon.append(s.ncon)
    out[scene + "_sens"] = np.array(rows)
    out[scene + "_ncon"] = np.array(ncon, dtype=np.int32)
    out[scene + "_qpos_end"] = s.qpos.copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_regression.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})
