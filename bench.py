#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched soft-gripper simulator (BASELINE.json metric).

A "step" is one ManEnv.step() (7 mj_step substeps + 12-channel sensor read-out, reference
environment/manenv.py:44-53) for every env of the batch.  Workload at N GPUs: 4096 envs per
GPU of the softbox scene, stiffness drawn from U(300,1400) (BASELINE.json configs[2]; with
N > 1 each rank draws from its own stiffness bin, configs[3]), following the reference's
200-step squeeze schedule (create_dataset.py:41-60) from a fresh reset, state resident in HBM.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # the CPU baseline's OpenMP threads must not spin between steps

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes_per_env_step(nq, nv, na, nu, nsens):
    """state {qpos, qvel, qacc_warmstart, act} read + written once per ManEnv.step(),
    plus ctrl, the stiffness scalar and the sensor outputs, in fp64 words."""
    return 8 * (2 * (nq + 2 * nv + na) + nu + 1 + nsens)


def usable_cores():
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota (containers)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(model, ks, sim_step, sched, budget_envs, threads):
    """The CPU oracle (oracle/sg_oracle.c, a port -- not MuJoCo) on the host cores: `budget_envs`
    envs x one full 200-step episode, OpenMP over envs."""
    from oracle import oracle as O
    om = O.OracleModel(model.to_blob())
    sims = [O.OracleSim(om) for _ in range(budget_envs)]
    jids, tids = list(range(11, 64)), [0]
    for s, k in zip(sims, ks):
        s.jnt_stiffness[jids] = k
        s.tendon_stiffness[tids] = k
        s.reset()
        s.forward()
        s.step()
    t0 = time.perf_counter()
    for t in range(len(sched)):
        if sched[t] is not None:
            for s in sims:
                s.ctrl[:] = sched[t]
        O.step_many(om, sims, sim_step, threads)
    dt = time.perf_counter() - t0
    return budget_envs * len(sched) / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--scene", default="softbox")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL over xGMI) or gloo (for testing the multi-process path)")
    ap.add_argument("--force-device", type=int, default=-1, help="testing only: put every rank on this GPU")
    args = ap.parse_args()

    import torch
    import softgrip_amd as sg
    from softgrip_amd import native
    from softgrip_amd.create_dataset import episode_schedule, stiffness_bin

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.force_device >= 0:
        local = args.force_device
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))

    model = sg.load_model(os.path.join(ROOT, "models", args.scene + ".sgmodel"))
    nm = native.NativeModel(model)
    n = args.envs
    batch = native.NativeBatch(nm, n, local)
    dev = batch.device
    # stiffness: full paper range on one GPU, one bin per rank on several (no collective on the data path)
    if world == 1:
        ks = np.random.RandomState(0).uniform(300, 1400, n)
    else:
        lo, hi = stiffness_bin(rank, world)
        ks = np.random.RandomState(1000 + rank).uniform(lo, hi, n)
    jids, tids = list(range(11, 64)), [0]
    batch.set_stiffness(ks, jids, tids)
    sched = episode_schedule()
    T = len(sched)
    sim_step, sim_start = 7, 1
    nsd = nm.nsensordata
    out = torch.zeros(n, T, nsd, dtype=torch.float64, device=dev)
    flags = torch.zeros(n, dtype=torch.int32, device=dev)
    flags_or = torch.zeros(n, dtype=torch.int32, device=dev)
    ctrl = np.zeros(nm.nu)

    def run(nsteps, t_begin):
        """nsteps env steps following the episode schedule from position t_begin (reset at every episode start)"""
        t = t_begin
        for _ in range(nsteps):
            if t % T == 0:
                batch.reset(sim_start, flags=flags)
                ctrl[:] = 0
            if sched[t % T] is not None:
                ctrl[:] = sched[t % T]
                batch.set_ctrl_broadcast(ctrl)
            batch.step(sim_step, sens=out[:, t % T], sens_stride=T * nsd, flags=flags)
            flags_or.bitwise_or_(flags)
            t += 1
        return t

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # warmup: first W steps of an episode, then start the timed region at a fresh episode
    run(args.warmup, 0)
    barrier()
    flags_or.zero_()
    batch.profile_enable(True)
    batch.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps, 0)
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = batch.profile_read(reset=True)
    batch.profile_enable(False)
    if dist is not None:  # the only collectives of the run: a barrier per side and this MAX (timing, not data path)
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    nbad = int((flags_or != 0).sum().item())

    if rank == 0:
        value = world * n * args.steps / dt
        abytes = algorithmic_bytes_per_env_step(nm.nq, nm.nq, nm.nu, nm.nu, nsd)
        ach = abytes * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_src = None, None
        pipe = os.environ.get("SG_PIPELINE", "rows")
        tp = os.path.join(ROOT, "profiles", {"rows": "r01_v12_rows_hbm_traffic.json", "split": "r01_v4_hbm_traffic.json"}.get(pipe, "none"))
        if os.path.exists(tp) and args.scene == "softbox" and n == 4096:
            # HBM-side bytes per sg_step call from the committed rocprofv3 PMC passes of this very workload (not re-measured here)
            traffic = json.load(open(tp))["per_sg_step_call_bytes"]
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; gfx950 x2 correction on the 16-B-per-lane reads of the PGS kernel; fabric-side: Infinity-Cache hits included)" % os.path.basename(tp)
        res = {
            "metric": "env steps/sec (whole node) at batch=4096",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2]: %d envs/GPU, %s scene (nv=%d), stiffness ~ U(300,1400)%s, reference 200-step squeeze "
                                   "schedule from reset, 7 substeps per env step; %d equality rows (composite neighbour equalities %s: DESIGN.md 2, U2)" % (
                                       n, args.scene, nm.nq, " split in per-rank bins" if world > 1 else "", model.neq,
                                       "on" if (model.eq_obj2id >= 0).any() else "off"),
                       "envs_per_gpu": n, "substeps_per_step": sim_step, "physics_substeps_per_s": value * sim_step,
                       "envs_flagged_bad": nbad, "launches_timed": launches,
                       "timed_region": ("%d whole episode(s) from reset" % (args.steps // T)) if args.steps % T == 0 else
                                       ("the first %d env steps of the episode loop -- NOT the episode average (use --steps as a multiple of %d)" % (args.steps, T))},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": {"rows": "sg_chain_kernel + sg_phase_kernel + sg_pgs_rows_kernel chain of one sg_step call (rows pipeline; dominant: sg_pgs_rows_kernel)", "split": "sg_chain_kernel + sg_phase_kernel + sg_pgs_kernel chain of one sg_step call (split pipeline)"}.get(pipe, "sg_step_kernel"),
                         "avg_kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_env_step": abytes,
                         "note": "one 'launch' = the kernel chain of one sg_step call (7 substeps); the path is instruction-issue-bound (serial Gauss-Seidel per finger), not HBM-bound; traffic is mostly contact blocks re-read from L2/Infinity Cache by each sweep (DESIGN.md 4.3)"},
        }
        # secondary roofline (the binding limit is instruction issue, not HBM): VALU wavefront-instructions per env
        # step from the committed SQ-counter pass of this very command, times the measured rate, against what the chip's 1024
        # SIMDs can issue (one fp64 wavefront instruction per 4 cycles each)
        sp = os.path.join(ROOT, "profiles", "r01_v12_sq_totals.json")
        if os.path.exists(sp) and args.scene == "softbox" and n == 4096 and pipe == "rows":
            sq = json.load(open(sp))["per_env_step"]
            peak = 1024 * 2.4e9 / 4.0
            ach = sq["SQ_INSTS_VALU"] * value / world   # per GPU
            res["roofline"]["secondary"] = {
                "bound": "fp64 VALU issue", "unit": "wavefront-instructions/s", "achieved": ach, "peak": peak, "frac": ach / peak,
                "valu_insts_per_env_step": sq["SQ_INSTS_VALU"], "salu_insts_per_env_step": sq["SQ_INSTS_SALU"],
                "lds_insts_per_env_step": sq["SQ_INSTS_LDS"],
                "source": "profiles/r01_v12_sq_totals.json (rocprofv3 --pmc SQ_INSTS_*, own pass); peak = 1024 SIMDs x 2.4 GHz / 4 cycles per "
                          "fp64 wavefront instruction; the PGS kernel, 70 % of the time, can only put wavefronts on 512 SIMDs at 4096 envs"}
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            cores = usable_cores()
            envs = 16 * cores  # 16 full episodes per core: about 10-20 s of wall time
            v, cdt = cpu_baseline(model, ks, sim_step, sched, envs, cores)
            v1, cdt1 = cpu_baseline(model, ks, sim_step, sched, 8, 1)   # a 1-thread figure beside the all-cores one
            probe = []
            for mod in ("mujoco", "mujoco_py"):                          # time MuJoCo itself iff it exists on the box -- probe, never assume
                try:
                    __import__(mod)
                    probe.append(mod + ": importable (not timed: no MJCF on this box)")
                except Exception as e:  # noqa: BLE001
                    probe.append("%s: absent (%s)" % (mod, type(e).__name__))
            res["cpu_baseline"] = {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port",
                                   "sample": "%d envs x one 200-step episode (same scene/schedule/stiffness draws) on the fp64 C oracle, "
                                             "OpenMP over envs, %.1f s" % (envs, cdt),
                                   "one_thread": {"value": v1, "sample": "8 envs x one episode on 1 thread, %.1f s" % cdt1},
                                   "mujoco_probe": "; ".join(probe)}
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
