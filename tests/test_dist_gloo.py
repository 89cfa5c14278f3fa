"""Multi-GPU path on CPU, world_size 2: the shard partition (no data-path collective), bench.py's default rank synchronisation -- a TCP
store, no collective library (softgrip_amd/ranks.py) -- self-launched and under torch.distributed.run, its optional torch.distributed
backend on gloo, and create_dataset's ranks.  The product scripts run over tests/fake_native.py through tests/run_with_fake_native.py."""
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


FAKE = os.path.join(ROOT, "tests", "run_with_fake_native.py")


def test_rank_group_store_barrier_and_gather(tmp_path):
    """ranks.RankGroup on its own: three ranks started by ranks.spawn_ranks, two barriers and a gather through the store; a rank that
    fails ends the job with its exit code"""
    w = tmp_path / "w.py"
    w.write_text("import sys, os, time\nsys.path.insert(0, %r)\nfrom softgrip_amd import ranks\n"
                 "if not ranks.launched_as_rank():\n    raise SystemExit(ranks.spawn_ranks(3))\n"
                 "g = ranks.RankGroup()\ng.barrier()\ntime.sleep(0.05 * g.rank)\ng.barrier()\n"
                 "v = g.gather('t', 0.5 * (g.rank + 1))\nassert v == [0.5, 1.0, 1.5], v\n"
                 "if len(sys.argv) > 1 and g.rank == 1:\n    raise SystemExit(7)\n"
                 "g.close()\nprint('rank', g.rank, 'ok', flush=True)\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(w)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and sorted(out.stdout.split("\n")[:3]) == ["rank 0 ok", "rank 1 ok", "rank 2 ok"], (out.stdout, out.stderr[-2000:])
    out = subprocess.run([sys.executable, str(w), "fail"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 7, (out.returncode, out.stderr[-2000:])


def test_bench_launches_its_own_ranks():
    """the REAL bench.py, `--gpus 2` with no launcher, on its DEFAULT rank synchronisation: the parent starts two ranks itself (child
    processes, no torch.distributed.run), they meet on a TCP store -- no process group, no collective -- run the stratified episode
    loop on the fake native batch, and rank 0 prints the one JSON line with the whole-job aggregate.  Then the same file launched the
    way the driver does it (torch.distributed.run: the ranks join the launcher's store), and once on the optional gloo backend."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, FAKE, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "16"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 20 and res["warmup"] == 5 and res["scaling"] == "weak"
    assert res["config"]["steps_timed"] == 20 and "stratified over the episode" in res["config"]["timed_region"]
    assert abs(res["value"] - 2 * 16 * 20 / (res["ms_per_step"] * 20e-3)) < 1e-6 * res["value"]    # whole-job aggregate over max-rank time
    assert "split in per-rank bins" in res["config"]["workload"] and "no collective library" in res["config"]["rank_sync"]
    assert len(res["ms_per_step_per_rank"]) == 2 and abs(max(res["ms_per_step_per_rank"]) - res["ms_per_step"]) < 1e-9
    assert "FAKE native batch" in out.stderr
    # launched the way the driver does it, the same file runs as a rank: default (store) and the optional torch.distributed backend
    for extra, sync in ([], "no collective library"), (["--dist-backend", "gloo"], "gloo"):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(_free_port()), FAKE, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "3",
                              "--envs", "8"] + extra, capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        res = json.loads(lines[0])
        assert res["n_gpus"] == 2 and res["config"]["steps_timed"] == 200 and "1 whole episode" in res["config"]["timed_region"]
        assert sync in res["config"]["rank_sync"] and len(res["ms_per_step_per_rank"]) == 2


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
    """`python -m softgrip_amd.create_dataset --gpus 2` with no launcher starts its own two ranks (child processes of a parent that
    touches no GPU), each with its own stiffness bin and shard; the ranks share nothing; the parent adds their summary files up into
    one JSON line.  --total-episodes fixes the dataset size (BASELINE configs[3]: 8 x 4096 x 4)."""
    import json
    import pickle
    from helpers import model_path
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, FAKE, "-m", "softgrip_amd.create_dataset", "--gpus", "2", "--mujoco-model-paths", model_path("softbox"),
                          "--n-envs", "5", "--total-episodes", "30", "--seed", "3", "--data-folder", str(tmp_path / "ds"), "--data-name", "cfg4"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["episodes"] == 30 and res["env_steps"] == 30 * 200 and len(res["per_rank_env_steps_per_s"]) == 2
    assert res["envs_reset_after_a_warning"] == 0 and len(res["shard_bytes"]) == 2 and res["env_steps_per_s"] > 0
    k0 = pickle.load(open(tmp_path / "ds" / "cfg4.rank0.pickle", "rb"))["stiffness"]
    k1 = pickle.load(open(tmp_path / "ds" / "cfg4.rank1.pickle", "rb"))["stiffness"]
    assert len(k0) == 15 and len(k1) == 15 and all(300 <= k < 850 for k in k0) and all(850 <= k < 1400 for k in k1)


# ---- BASELINE configs[3] at its real rank count: 8 ranks, on the CPU over the fake native batch -------------------
# What this pins is everything around the kernels that an 8-GPU run needs and a 2-rank run does not prove: eight processes meeting on one
# store and port, eight disjoint stiffness bins of width 137.5, eight shards, ONE JSON line, and a dead rank ending the job with its code.

def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_eight_ranks():
    import json
    bench = os.path.join(ROOT, "bench.py")
    args = ["--gpus", "8", "--steps", "10", "--warmup", "2", "--envs", "8", "--no-cpu-baseline", "--no-fix-variant"]
    for launcher in ([], ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]):
        out = subprocess.run([sys.executable] + launcher + [FAKE, bench] + args, capture_output=True, text=True, env=_clean_env(), timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1                                   # ONE line, from rank 0
        res = json.loads(lines[0])
        assert res["n_gpus"] == 8 and res["steps"] == 10 and res["warmup"] == 2 and res["scaling"] == "weak"
        assert len(res["ms_per_step_per_rank"]) == 8 and abs(max(res["ms_per_step_per_rank"]) - res["ms_per_step"]) < 1e-9
        assert abs(res["value"] - 8 * 8 * 10 / (res["ms_per_step"] * 10e-3)) < 1e-6 * res["value"]
        assert "configs[3]" in res["config"]["workload"] and "no collective library" in res["config"]["rank_sync"]


def test_create_dataset_eight_ranks(tmp_path):
    import json
    import pickle
    from helpers import model_path
    out = subprocess.run([sys.executable, FAKE, "-m", "softgrip_amd.create_dataset", "--gpus", "8", "--mujoco-model-paths", model_path("softbox"),
                          "--n-envs", "4", "--total-episodes", "64", "--seed", "11", "--data-folder", str(tmp_path / "ds"), "--data-name", "cfg3"],
                         capture_output=True, text=True, env=_clean_env(), timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["episodes"] == 64 and res["env_steps"] == 64 * 200 and len(res["shard_bytes"]) == 8
    assert len(res["per_rank_env_steps_per_s"]) == 8
    seen = []
    for r in range(8):
        d = pickle.load(open(tmp_path / "ds" / ("cfg3.rank%d.pickle" % r), "rb"))
        k = np.array(d["stiffness"])
        assert len(k) == 8 and len(d["data"]) == 8 and np.array(d["data"][0]).shape == (200, 12)
        assert (k >= 300 + 137.5 * r).all() and (k < 300 + 137.5 * (r + 1)).all(), (r, k)   # 8 disjoint bins of width 137.5 (SURVEY 8(d) cfg 4)
        seen.extend(k.tolist())
    assert len(set(seen)) == 64       # the ranks' RNG streams differ (seed + 1000 rank)


def test_a_killed_rank_of_eight_ends_the_job_with_its_code(tmp_path):
    """rank 5 of 8 is killed (SIGKILL, as the OOM killer or a GPU fault would) while the others wait at a barrier: the parent ends the
    seven survivors and returns 128 + 9; nobody is left behind"""
    import time
    w = tmp_path / "w.py"
    w.write_text("import sys, os, signal, time\nsys.path.insert(0, %r)\nfrom softgrip_amd import ranks\n"
                 "if not ranks.launched_as_rank():\n    raise SystemExit(ranks.spawn_ranks(8, grace_s=3.0))\n"
                 "open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                 "g = ranks.RankGroup(timeout_s=120)\ng.barrier()\n"
                 "if g.rank == 5:\n    os.kill(os.getpid(), signal.SIGKILL)\n"
                 "g.barrier()\ng.close()\n" % (ROOT, str(tmp_path)))
    t0 = time.time()
    out = subprocess.run([sys.executable, str(w)], capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert out.returncode == 128 + 9, (out.returncode, out.stderr[-2000:])
    assert time.time() - t0 < 100                      # ended by the parent, not by the store's timeout
    for r in range(8):
        pid = int(open(tmp_path / ("pid%d" % r)).read())
        try:
            os.kill(pid, 0)
            alive = True
        except OSError:
            alive = False
        assert not alive, "rank %d (pid %d) outlived the job" % (r, pid)


def test_programmatic_main_with_gpus_launches_create_dataset_ranks(monkeypatch, tmp_path):
    """create_dataset.main(argv) called from someone else's process (a script, a notebook, pytest) with --gpus N must
    start N `python -m softgrip_amd.create_dataset <argv>` ranks -- not N copies of the caller's own command line"""
    import pytest
    from softgrip_amd import create_dataset, ranks
    calls = []

    def fake_spawn(n, extra_env=None, poll=0.05, cmd=None, **kw):
        calls.append((n, cmd, extra_env))
        return 3

    monkeypatch.setattr(ranks, "spawn_ranks", fake_spawn)
    monkeypatch.delenv("RANK", raising=False)
    argv = ["--gpus", "4", "--mujoco-model-paths", "a.xml", "--data-folder", str(tmp_path), "--data-name", "x"]
    with pytest.raises(SystemExit) as e:
        create_dataset.main(argv)
    assert e.value.code == 3
    (n, cmd, extra), = calls
    assert n == 4 and cmd == [sys.executable, "-m", "softgrip_amd.create_dataset"] + argv and ROOT in extra["PYTHONPATH"]
    calls.clear()
    monkeypatch.setattr(sys, "argv", ["create_dataset"] + argv)      # the CLI entry: the ranks re-run the process's own command line
    with pytest.raises(SystemExit):
        create_dataset.main()
    assert calls[0][1] is None
