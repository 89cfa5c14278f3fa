"""Calibration of the ensemble statistic (tests/helpers.py ENS_TOL_48) on the ball and the cylinder: the oracle against the oracle with
1e-13 added to one slider at env step 5 -- "the same system, other round-off" -- and against two wrong systems (the same data one env
step late; accelerometers 15 % off).  CPU only.  usage: python scripts/calibrate_ensemble.py [n_envs] > profiles/r04_ensemble_calibration.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import softgrip_amd as sg  # noqa: E402
from helpers import ensemble_report, model_path, oracle_episodes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ks = np.linspace(300, 1400, n)
th = os.cpu_count() or 1
for scene in ("softball", "softcylinder"):
    m = sg.load_model(model_path(scene), "implicit")
    a = oracle_episodes(m, ks, th)
    b = oracle_episodes(m, ks, th, perturb=1e-13, perturb_step=5)
    d = np.abs(a - b).max(axis=(0, 2))
    print(scene, n, "envs; max |oracle - perturbed oracle| at env steps 5 10 20 40 50 60 80 100 150 199:", " ".join("%.1e" % d[t] for t in (5, 10, 20, 40, 50, 60, 80, 100, 150, 199)))
    fmt = lambda r: " ".join("%s %.3f" % (k, r[k]) for k in sorted(r))  # noqa: E731
    print("  same system, other round-off :", fmt(ensemble_report(a, b, ks, t0=42)))
    print("  one env step late            :", fmt(ensemble_report(a[:, 1:], b[:, :-1], ks, t0=42)))
    sc = a.copy()
    sc[:, :, :6] *= 1.15
    print("  accelerometers 15 % off      :", fmt(ensemble_report(sc, b, ks, t0=42)))
