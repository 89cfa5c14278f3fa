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


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def test_bench_launches_its_own_ranks_gloo():
    """the REAL bench.py, `--gpus 2` with no launcher: it starts two ranks itself (torch.distributed.run as a child process), both
    run the stratified episode loop on the fake native batch, rank 0 prints the one JSON line with the whole-job aggregate"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "16",
                          "--dist-backend", "gloo", "--fake-native-for-tests"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 20 and res["warmup"] == 5 and res["scaling"] == "weak"
    assert res["config"]["steps_timed"] == 20 and "stratified over the episode" in res["config"]["timed_region"]
    assert abs(res["value"] - 2 * 16 * 20 / (res["ms_per_step"] * 20e-3)) < 1e-6 * res["value"]    # whole-job aggregate over max-rank time
    assert "split in per-rank bins" in res["config"]["workload"] and res["data"].startswith("FAKE")
    # launched the way the driver does it, the same file runs as a rank
    port = _free_port()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "3",
                          "--envs", "8", "--dist-backend", "gloo", "--fake-native-for-tests"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["config"]["steps_timed"] == 200 and "1 whole episode" in res["config"]["timed_region"]


CD_WORKER = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import fake_native
from softgrip_amd import native, create_dataset
native.NativeModel, native.NativeBatch = fake_native.FakeModel, fake_native.FakeBatch
create_dataset.main(sys.argv[1:])
"""


def test_create_dataset_main_two_ranks(tmp_path):
    """the REAL create_dataset.main under torch.distributed.run (2 ranks, fake native batch): each rank goes through ManEnv.reset()
    with its own stiffness bin, writes its parts and its shard; no collective, no shared file"""
    import pickle
    w = tmp_path / "cd_worker.py"
    w.write_text(CD_WORKER % (ROOT, os.path.join(ROOT, "tests")))
    from helpers import model_path
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), str(w), "--mujoco-model-paths", model_path("softbox"), "--n-envs", "6",
                          "--num-batches", "2", "--seed", "0", "--data-folder", str(tmp_path / "ds"), "--data-name", "sweep"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    files = sorted(os.listdir(tmp_path / "ds"))
    assert files == ["sweep.rank0.part00000.pickle", "sweep.rank0.part00001.pickle", "sweep.rank0.pickle", "sweep.rank0.summary.json",
                     "sweep.rank1.part00000.pickle", "sweep.rank1.part00001.pickle", "sweep.rank1.pickle", "sweep.rank1.summary.json"]
    k0 = pickle.load(open(tmp_path / "ds" / "sweep.rank0.pickle", "rb"))["stiffness"]
    k1 = pickle.load(open(tmp_path / "ds" / "sweep.rank1.pickle", "rb"))["stiffness"]
    assert len(k0) == 12 and len(k1) == 12 and all(300 <= k < 850 for k in k0) and all(850 <= k < 1400 for k in k1)


def test_create_dataset_launches_its_own_ranks(tmp_path):
    """`python -m softgrip_amd.create_dataset --gpus 2` with no launcher starts its own two ranks (a child
    torch.distributed.run, before anything touches a GPU), each with its own stiffness bin and shard; the parent adds the ranks'
    summary files up into one JSON line.  --total-episodes fixes the dataset size (BASELINE configs[3]: 8 x 4096 x 4)."""
    import json
    import pickle
    from helpers import model_path
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-m", "softgrip_amd.create_dataset", "--gpus", "2", "--mujoco-model-paths", model_path("softbox"),
                          "--n-envs", "5", "--total-episodes", "30", "--seed", "3", "--data-folder", str(tmp_path / "ds"), "--data-name", "cfg4",
                          "--fake-native-for-tests"], capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["episodes"] == 30 and res["env_steps"] == 30 * 200 and len(res["per_rank_env_steps_per_s"]) == 2
    assert res["envs_reset_after_a_warning"] == 0 and len(res["shard_bytes"]) == 2 and res["env_steps_per_s"] > 0
    k0 = pickle.load(open(tmp_path / "ds" / "cfg4.rank0.pickle", "rb"))["stiffness"]
    k1 = pickle.load(open(tmp_path / "ds" / "cfg4.rank1.pickle", "rb"))["stiffness"]
    assert len(k0) == 15 and len(k1) == 15 and all(300 <= k < 850 for k in k0) and all(850 <= k < 1400 for k in k1)
