"""Multi-GPU path on CPU: world_size-2 gloo processes check the shard partition (no data-path collective) and the
max-over-ranks timing reduction bench.py uses."""
import os
import socket
import subprocess
import sys

import numpy as np

from helpers import ROOT
from softgrip_amd.create_dataset import stiffness_bin

WORKER = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from softgrip_amd.create_dataset import stiffness_bin
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = stiffness_bin(rank, world)
ks = np.random.RandomState(1000 + rank).uniform(lo, hi, 16)
t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py: time = max over ranks
n = torch.tensor([16.0])
dist.all_reduce(n)                                # whole-job env count (only used for reporting)
allk = [None] * world
dist.all_gather_object(allk, ks.tolist())
if rank == 0:
    print(json.dumps({"tmax": float(t), "n": float(n), "bins": [[min(k), max(k)] for k in allk]}))
dist.destroy_process_group()
""" % ROOT


def test_bins_partition_the_range():
    edges = [stiffness_bin(r, 8) for r in range(8)]
    assert edges[0][0] == 300.0 and edges[-1][1] == 1400.0
    for a, b in zip(edges[:-1], edges[1:]):
        assert a[1] == b[0]
    assert np.allclose([hi - lo for lo, hi in edges], 137.5)


def test_two_ranks_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(w)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert abs(res["tmax"] - 0.2) < 1e-12 and res["n"] == 32.0
    (lo0, hi0), (lo1, hi1) = res["bins"]
    assert 300 <= lo0 and hi0 <= 850 <= lo1 and hi1 <= 1400
