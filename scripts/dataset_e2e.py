"""End-to-end rate of the dataset host (create_dataset.main: ManEnv stepping with its per-step host checks, device -> host copy, pickling)
beside the kernel-only rate of bench.py.
usage (GPU box): python scripts/dataset_e2e.py [n_envs] [batches per scene] [scene[,scene...]] [extra create_dataset flags]
  the README's quick start = three scenes: python scripts/dataset_e2e.py 4096 1 softbox,softcylinder,softball"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from softgrip_amd import create_dataset as cd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scenes = (sys.argv[3] if len(sys.argv) > 3 else "softbox").split(",")
extra = sys.argv[4:]
paths = [os.path.join(ROOT, "models", s + ".sgmodel") for s in scenes]
with tempfile.TemporaryDirectory() as d:
    base = ["--mujoco-model-paths"] + paths + ["--n-envs", str(n), "--seed", "0", "--data-folder", d, "--data-name", "e2e"]
    cd.main(base + ["--num-batches", "1"] + extra)      # warm-up: library load, scene checks, first launches
    for f in os.listdir(d):
        os.unlink(os.path.join(d, f))
    t0 = time.time()
    cd.main(base + ["--num-batches", str(nb)] + extra)
    dt = time.time() - t0
    size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
print("create_dataset end to end (%s): %d envs x %d episode-batch(es) x %d scene(s) of 200 steps in %.2f s = %.0f env-steps/s (files: %.0f MB)%s" % (
    ", ".join(scenes), n, nb, len(scenes), dt, n * nb * len(scenes) * 200 / dt, size / 1e6, " [%s]" % " ".join(extra) if extra else ""))
