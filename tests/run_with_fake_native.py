"""TEST INFRASTRUCTURE: runs a product script (bench.py, `-m softgrip_amd.create_dataset`) with tests/fake_native.py in the place of
the HIP library, so that its rank plumbing -- self-launched ranks, stores, shards, the one JSON line -- runs on a box without a GPU.
Nothing in the product knows about this file; the ranks a script starts re-run the parent's own command line (ranks.spawn_ranks) and
so come through here again.

usage: python tests/run_with_fake_native.py <script.py | -m module> [args ...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import fake_native  # noqa: E402
from softgrip_amd import native  # noqa: E402

native.NativeModel, native.NativeBatch = fake_native.FakeModel, fake_native.FakeBatch
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.device_count = lambda: 8
torch.cuda.set_device = lambda *a, **k: None
print("run_with_fake_native: FAKE native batch (plumbing test, not a measurement)", file=sys.stderr)

if sys.argv[1] == "-m":
    sys.argv = [sys.argv[2]] + sys.argv[3:]
    runpy.run_module(sys.argv[0], run_name="__main__", alter_sys=True)
else:
    sys.argv = sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name="__main__")
