#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched soft-gripper simulator (BASELINE.json metric).

A "step" is the device work of one ManEnv.step() (reference environment/manenv.py:44-53) for every env of the batch: one
sg_step call = 7 mj_step substeps + the 12-channel sensor read-out, written straight into the [n, 200, 12] episode block --
the loop ManEnv.rollout() / create_dataset run.  (ManEnv.step() itself adds a host sync per step for its return value; the
dataset path does not take it, so it is not in the metric.)  Workload at N GPUs: 4096 envs per GPU of the softbox scene as
MuJoCo's composite documentation describes it (fix rows + neighbour equalities), stiffness drawn from
U(300,1400) (BASELINE.json configs[2]; with N > 1 each rank draws from its own stiffness bin, configs[3]), following the
reference's 200-step squeeze schedule (create_dataset.py:41-60) from a fresh reset, state resident in HBM.

Timed region: --steps S a multiple of 200 = S/200 whole episodes (resets included).  Any other S: one episode is run from
reset and S of its 200 steps, spread evenly over it, are timed one by one (sync + timer around each) -- an estimate of the
episode average, not of an episode prefix (the first 40 steps have no contacts and cost a third of the average).

`python bench.py --gpus N` without a launcher starts its own N ranks (N child processes of a parent that never touches a GPU);
started by `torch.distributed.run` it runs as a rank.  Either way the ranks share nothing but a common start and their timings, which
go through a TCP key-value store (softgrip_amd/ranks.py): NO collective library on the default path (north_star: "no RCCL collectives
required"; `--dist-backend nccl` puts the barriers and the MAX on RCCL instead).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # the CPU baseline's OpenMP threads must not spin between steps
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

device_sync = None   # torch.cuda.synchronize (set in main)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PROFILE_TAG = "r05"    # profiles/<tag>_<scene>_{hbm_traffic,sq_totals}.json: the committed rocprofv3 PMC passes of this workload


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


def stiffness_ids(scene):
    """(joint ids, tendon ids) that carry the per-env stiffness: the reference's class attributes (environment/manenv.py:12-13) for the
    two-finger scenes; the ball's 218 sliders behind the 65 gripper joints for the four-finger scene (manenv.py:11's commented ids
    belong to an older gripper file)"""
    if scene.startswith("freeball"):       # soft_experiments_softball.xml: joint 8 is the ball's free joint, joints 9 .. 226 its sliders
        return list(range(9, 227)), [0]
    return (list(range(65, 283)) if scene.startswith("fourfinger") else list(range(11, 64))), [0]


def cpu_baseline(model, ks, sim_step, sched, budget_envs, threads, scene="softbox"):
    """The CPU oracle (oracle/sg_oracle.c, a port -- not MuJoCo) on the host cores: `budget_envs`
    envs x one full 200-step episode, OpenMP over envs."""
    from oracle import oracle as O
    om = O.OracleModel(model.to_blob())
    sims = [O.OracleSim(om) for _ in range(budget_envs)]
    jids, tids = stiffness_ids(scene)
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


def stratified_steps(S, T):
    """S step indices spread evenly over an episode of T steps (the midpoints of S equal strata)"""
    return sorted({min(T - 1, int((i + 0.5) * T / S)) for i in range(S)})


class Runner:
    """one batch + its episode block; runs schedule steps and keeps the flags"""

    def __init__(self, scene, n, local, rank, world, damper=None):
        import torch
        import softgrip_amd as sg
        from softgrip_amd import native
        from softgrip_amd.create_dataset import episode_schedule, stiffness_bin
        self.torch = torch
        if damper is None:   # what ManEnv's tendon_damper="auto" ends up with: the ball / cylinder scenes only run with the implicit damper
            damper = "explicit" if scene.startswith("softbox") else "implicit"
        self.damper = damper
        self.model = sg.load_model(os.path.join(ROOT, "models", scene + ".sgmodel"), damper)
        self.nm = native.NativeModel(self.model)
        self.n = n
        self.batch = native.NativeBatch(self.nm, n, local)
        dev = self.batch.device
        # stiffness: full paper range on one GPU, one bin per rank on several (no collective on the data path)
        if world == 1:
            self.ks = np.random.RandomState(0).uniform(300, 1400, n)
        else:
            lo, hi = stiffness_bin(rank, world)
            self.ks = np.random.RandomState(1000 + rank).uniform(lo, hi, n)
        self.batch.set_stiffness(self.ks, *stiffness_ids(scene))
        self.sched = episode_schedule()
        self.T = len(self.sched)
        self.sim_step, self.sim_start = 7, 1
        self.nsd = self.nm.nsensordata
        self.out = torch.zeros(n, self.T, self.nsd, dtype=torch.float64, device=dev)
        self.flags = torch.zeros(n, dtype=torch.int32, device=dev)
        self.flags_or = torch.zeros(n, dtype=torch.int32, device=dev)
        self.ctrl = np.zeros(self.nm.nu)

    def step(self, t):
        """env step t of the episode loop (t % T == 0: reset first)"""
        T = self.T
        if t % T == 0:
            self.batch.reset(self.sim_start, flags=self.flags)
            self.ctrl[:] = 0
        if self.sched[t % T] is not None:
            self.ctrl[:] = self.sched[t % T]
            self.batch.set_ctrl_broadcast(self.ctrl)
        self.batch.step(self.sim_step, sens=self.out[:, t % T], sens_stride=T * self.nsd, flags=self.flags)
        self.flags_or.bitwise_or_(self.flags)

    def timed(self, steps, barrier, after_episode=None, profile=False):
        """-> (seconds, avg kernel ms per sg_step call over the timed steps, launches timed, description, steps timed).
        profile=False: the headline pass -- nothing but the path itself inside the timed interval.  profile=True: the same region again
        with HIP events on the launch stream around every sg_step call's kernel chain and around every solver-kernel launch (7 per
        call): the event records sit in the stream between the kernels, so that pass yields the kernel durations (roofline), not `value`
        (ADVICE r02)."""
        torch, T, b = self.torch, self.T, self.batch
        b.profile_enable(False)
        b.profile_read(reset=True)
        b.profile_read_solver(reset=True)
        if steps % T == 0:
            barrier()
            b.profile_enable(profile)
            t0 = time.perf_counter()
            for t in range(steps):
                self.step(t)
                if after_episode is not None and (t + 1) % T == 0:
                    after_episode(self)
            barrier()
            dt = time.perf_counter() - t0
            desc = "%d whole episode(s) from reset (resets included)" % (steps // T)
        else:
            pick = set(stratified_steps(steps, T))
            barrier()
            dt = 0.0
            for t in range(T):
                if t in pick:
                    device_sync()
                    b.profile_enable(profile)
                    t0 = time.perf_counter()
                    self.step(t)
                    device_sync()
                    dt += time.perf_counter() - t0
                    b.profile_enable(False)
                else:
                    self.step(t)
            barrier()
            desc = ("stratified over the episode: %d of the 200 env steps of one episode from reset, evenly spaced (steps %s), "
                    "each timed on its own between device syncs; the other steps run untimed" % (len(pick), ",".join(map(str, sorted(pick)))))
            steps = len(pick)
        kernel_ms, launches = b.profile_read(reset=True)
        self.solver_ms, self.solver_launches = b.profile_read_solver(reset=True)
        b.profile_enable(False)
        return dt, kernel_ms, launches, desc, steps

    def measure(self, steps, barrier, after_episode=None, event_pass=True):
        """headline pass (uninstrumented), then the same timed region once more with the HIP events on for the kernel durations"""
        dt, _, _, desc, nsteps = self.timed(steps, barrier, after_episode, profile=False)
        if not event_pass:
            self.solver_ms, self.solver_launches = 0.0, 0
            return dt, 0.0, 0, desc, nsteps, dt
        dtp, kernel_ms, launches, _, _ = self.timed(steps, barrier, None, profile=True)
        return dt, kernel_ms, launches, desc, nsteps, dtp


def _positive_or_none(v):
    """an occupancy query that failed returns a negative error code: not an occupancy"""
    return int(v) if v is not None and v > 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--scene", default="softbox", help="softbox (default: MuJoCo's documented composite) | softbox_fix | softball | softcylinder ...")
    ap.add_argument("--tendon-damper", default=None, choices=["explicit", "implicit"],
                    help="integration of the composite volume tendon's damper (DESIGN.md D5); default: explicit (MuJoCo's Euler) for softbox, implicit for the ball / cylinder scenes, which do not run otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fix-variant", action="store_true", help="skip the labelled secondary measurement on the fix-rows-only model")
    ap.add_argument("--with-regressor", action="store_true", help="BASELINE configs[4]: ConvNet forward + one Adam step on every finished [n,200,12] block, inside the timed region (needs --steps a multiple of 200)")
    ap.add_argument("--no-event-pass", action="store_true",
                    help="profiler runs (scripts/profile_round.sh): skip the second, HIP-event-instrumented pass over the timed region, so that the "
                         "trace holds exactly one reset + the timed steps; the line then carries no kernel durations")
    ap.add_argument("--dist-backend", default="store", choices=["store", "nccl", "gloo"],
                    help="what carries the barriers and the max-over-ranks time of an N > 1 run: store (default) = a TCP key-value store, no collective "
                         "library at all; nccl = torch.distributed over RCCL / xGMI; gloo = torch.distributed on the CPU")
    ap.add_argument("--force-device", type=int, default=-1, help="testing only: put every rank on this GPU")
    args = ap.parse_args()

    from softgrip_amd import ranks
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and not ranks.launched_as_rank():
        raise SystemExit(ranks.spawn_ranks(args.gpus))   # the parent: starts N copies of this command line as ranks, touches no GPU
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.with_regressor and args.steps % 200 != 0:
        raise SystemExit("--with-regressor needs --steps as a multiple of 200 (whole episodes)")

    import torch

    global device_sync
    device_sync = torch.cuda.synchronize

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.force_device >= 0:
        local = args.force_device
    dist, group = None, None
    if torch.cuda.device_count() <= local:
        # fail at once and on every rank: a rank that dies later would leave the others waiting in the first barrier
        raise SystemExit("bench.py: rank %d wants GPU %d but this node shows %d GPU(s)" % (rank, local, torch.cuda.device_count()))
    if world > 1:
        torch.cuda.set_device(local)
        if args.dist_backend == "store":
            group = ranks.RankGroup(rank, world)
        else:
            import torch.distributed as dist
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(args.dist_backend)

    def barrier():
        device_sync()
        if group is not None:
            group.barrier()
        if dist is not None:
            dist.barrier()
            device_sync()

    R = Runner(args.scene, args.envs, local, rank, world, args.tendon_damper)
    damper = R.damper
    n, T, nm, model = R.n, R.T, R.nm, R.model
    dev = R.batch.device

    after_episode = None
    reg = None
    if args.with_regressor:
        from softgrip_amd import convnet
        torch.manual_seed(0)
        net = convnet.ConvNet().to(dev)
        opt = convnet.make_optimizer(net)
        y = torch.tensor(R.ks, device=dev)
        reg = {"loss": []}

        def after_episode(r):
            mean, std = convnet.channel_stats(r.out)
            loss, _ = convnet.train_step(net, opt, r.out, y, mean, std, add_noise=True)
            reg["loss"].append(loss)

    # warm-up: W steps of the episode loop (kernels loaded, workspace touched); the timed region starts at a fresh reset
    for t in range(args.warmup):
        R.step(t)
    if after_episode is not None:  # warm the regressor's kernels on a throw-away copy (the episode block is not filled yet)
        import copy
        wnet = copy.deepcopy(net)
        wx = torch.randn(n, T, R.nsd, dtype=torch.float64, device=dev)
        convnet.train_step(wnet, convnet.make_optimizer(wnet), wx, y, *convnet.channel_stats(wx), add_noise=True)
        del wnet, wx
    barrier()
    R.flags_or.zero_()
    dt, kernel_ms, launches, desc, nsteps, dt_prof = R.measure(args.steps, barrier, after_episode, not args.no_event_pass)
    if group is not None:  # every rank's time through the store; the job's time is the slowest rank's
        per_rank_dt = group.gather("dt", dt)
        dt = max(per_rank_dt)
    elif dist is not None:  # --dist-backend nccl / gloo: the only collectives of the run are the barriers and this MAX (timing, not data path)
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        per_rank = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(per_rank, tt)      # per-rank times next to the MAX: a scaling run shows imbalance between GPUs
        per_rank_dt = [float(x.item()) for x in per_rank]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    else:
        per_rank_dt = [dt]
    nbad = int((R.flags_or != 0).sum().item())

    if rank == 0:
        value = world * n * nsteps / dt
        abytes = algorithmic_bytes_per_env_step(nm.nq, nm.nv, nm.nu, nm.nu, R.nsd)
        ach = abytes * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        pipe = os.environ.get("SG_PIPELINE", "rows")
        if args.scene.startswith(("fourfinger", "freeball")):
            pipe = "tree"   # outside the two-finger class: the tree pipeline is the only one that runs it
        nb_on = bool((model.eq_obj2id >= 0).any())
        res = {
            "metric": "env steps/sec (whole node) at batch=4096",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / nsteps * 1e3, "ms_per_step_with_hip_events": dt_prof / nsteps * 1e3,
            "ms_per_step_per_rank": [d / nsteps * 1e3 for d in per_rank_dt], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[%d]: %d envs/GPU, %s scene (nv=%d), stiffness ~ U(300,1400)%s, reference 200-step squeeze "
                                   "schedule from reset, 7 substeps per env step; %d equality rows (composite neighbour equalities %s: DESIGN.md 2, U2)" % (
                                       4 if args.with_regressor else (3 if world > 1 else 2), n, args.scene, nm.nq,
                                       " split in per-rank bins" if world > 1 else "", model.neq, "on" if nb_on else "off") + ("; volume tendon damper integrated %sly" % damper) + (
                                       "; + ConvNet regressor: channel stats, noise augmentation, forward and one Adam step on every finished [n,200,12] block, on device, inside the timed region" if args.with_regressor else ""),
                       "envs_per_gpu": n, "substeps_per_step": R.sim_step, "physics_substeps_per_s": value * R.sim_step,
                       "envs_flagged_bad": nbad, "launches_timed": launches, "steps_timed": nsteps, "timed_region": desc,
                       "tree_workgroups_per_cu": _positive_or_none(R.batch.tree_workgroups_per_cu() if hasattr(R.batch, "tree_workgroups_per_cu") else 0),
                       "rank_sync": {"store": "TCP key-value store (barriers + per-rank times), no collective library", "nccl": "torch.distributed on RCCL",
                                     "gloo": "torch.distributed on gloo"}[args.dist_backend] if world > 1 else None},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "traffic_source": None, "traffic_measured_in_this_run": False,
                         "kernel": {"rows": "sg_chain_kernel + sg_phase_kernel + sg_pgs_rows_kernel chain of one sg_step call (rows pipeline; dominant: sg_pgs_rows_kernel)", "split": "sg_chain_kernel + sg_phase_kernel + sg_pgs_kernel chain of one sg_step call (split pipeline)",
                                    "tree": "sg_tree_kernel: one launch per sg_step call, one env per wavefront (tree pipeline)"}.get(pipe, "sg_step_kernel"),
                         "avg_kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_env_step": abytes,
                         "note": "one 'launch' = the kernel chain of one sg_step call (7 substeps), HIP events on the launch stream over the timed steps only "
                                 "(a second pass over the same timed region: `value` is measured without the events in the stream); "
                                 "the path is instruction-issue-bound (serial Gauss-Seidel per finger), not HBM-bound; traffic is mostly contact blocks "
                                 "re-read from L2/Infinity Cache by each sweep (DESIGN.md 4.3)"},
        }
        if R.solver_launches:
            # the dominant kernel on its own: its average launch (HIP events around every launch of it in the timed steps; the committed
            # rocprofv3 --stats summary's AverageNs for the same kernel agrees) against the algorithmic bytes of one substep of one call
            sub = abytes * n / R.sim_step
            res["roofline"]["dominant_kernel"] = {
                "name": "sg_pgs_rows_kernel (one launch per physics substep)", "avg_launch_ms": R.solver_ms, "launches_timed": R.solver_launches,
                "algorithmic_bytes_per_launch": sub, "achieved": sub / (R.solver_ms * 1e-3) / 1e9, "unit": "GB/s",
                "frac": sub / (R.solver_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "share_of_call_time": R.sim_step * R.solver_ms / kernel_ms if kernel_ms else None}
        if reg is not None:
            res["config"]["regressor_loss_first_last"] = [float(reg["loss"][0]), float(reg["loss"][-1])] if reg["loss"] else None
        # Episode-average counters of this very workload from committed rocprofv3 PMC passes (profiles/<tag>_<scene>_*.json).  They
        # describe the whole 200-step episode, which is what both kinds of timed region measure or estimate; they are attached
        # only for the workload they were collected on.
        attach = n == 4096 and pipe in ("rows", "tree") and not args.with_regressor
        tp = os.path.join(ROOT, "profiles", "%s_%s_hbm_traffic.json" % (PROFILE_TAG, args.scene))
        if attach and os.path.exists(tp):
            res["roofline"]["traffic"] = json.load(open(tp))["per_sg_step_call_bytes"]
            res["roofline"]["traffic_attached_from"] = "profiles/" + os.path.basename(tp)   # NOT measured in this run: a committed PMC pass of this command
            res["roofline"]["traffic_source"] = ("profiles/%s: episode average per sg_step call (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; gfx950 x2 "
                                                 "correction on the 16-B-per-lane reads of the PGS kernel, none on the tree kernel's 8-B reads; fabric-side: Infinity-Cache hits included)" % os.path.basename(tp))
        sp = os.path.join(ROOT, "profiles", "%s_%s_sq_totals.json" % (PROFILE_TAG, args.scene))
        if attach and os.path.exists(sp):
            # secondary roofline (the binding limit is instruction issue, not HBM): VALU wavefront-instructions per env
            # step (episode average) x the measured episode-average rate, against what 1024 SIMDs can issue
            sq = json.load(open(sp))["per_env_step"]
            peak = 1024 * 2.4e9 / 4.0
            a2 = sq["SQ_INSTS_VALU"] * value / world   # per GPU
            wg = res["config"]["tree_workgroups_per_cu"]
            res["roofline"]["secondary"] = {
                "attached_from": "profiles/" + os.path.basename(sp), "counters_measured_in_this_run": False,
                "bound": "fp64 VALU issue", "unit": "wavefront-instructions/s", "achieved": a2, "peak": peak, "frac": a2 / peak,
                "valu_insts_per_env_step": sq["SQ_INSTS_VALU"], "salu_insts_per_env_step": sq["SQ_INSTS_SALU"],
                "lds_insts_per_env_step": sq["SQ_INSTS_LDS"],
                "source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_*, own pass, episode average); peak = 1024 SIMDs x 2.4 GHz / 4 cycles per "
                          "fp64 wavefront instruction; %s" % (os.path.basename(sp),
                          ("the tree kernel runs one env per wavefront, %s workgroups per CU (one wavefront each: %s of a CU's four SIMDs busy), a wavefront alone on its SIMD "
                           "issues one instruction per ~7 cycles: ~%.2f is its ceiling at this occupancy" % (wg, wg, 0.57 * min(wg or 4, 4) / 4.0)) if pipe == "tree" else
                          "the PGS kernel runs one wavefront per SIMD (1024 at 4096 envs: 4 envs per wavefront) and a wavefront alone on its SIMD issues one instruction per ~7-8 cycles, so ~0.5 is this design's ceiling for it")}
        if world == 1 and nb_on and not args.no_fix_variant and not args.with_regressor and args.scene.endswith(("softbox", "softball", "softcylinder")):
            # labelled secondary: the same workload on the fix-rows-only model (composite_neighbors=False)
            del R
            torch.cuda.empty_cache()
            R2 = Runner(args.scene + "_fix", n, local, rank, world, damper)
            for t in range(args.warmup):
                R2.step(t)
            barrier()
            dt2, km2, _, desc2, ns2, _ = R2.measure(args.steps, barrier, None, not args.no_event_pass)
            res["config"]["fix_only_variant"] = {"value": n * ns2 / dt2, "unit": "env-steps/s", "avg_kernel_ms": km2, "equality_rows": R2.model.neq,
                                                 "note": "same workload on models/%s_fix.sgmodel (composite without its neighbour equalities) -- NOT the headline" % args.scene}
            del R2
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            cores = usable_cores()
            envs = 8 * cores  # 8 full episodes per core: about 10-20 s of wall time
            ks = np.random.RandomState(0).uniform(300, 1400, n)
            from softgrip_amd.create_dataset import episode_schedule
            sched = episode_schedule()
            v, cdt = cpu_baseline(model, ks, 7, sched, envs, cores, args.scene)
            v1, cdt1 = cpu_baseline(model, ks, 7, sched, 2, 1, args.scene)   # a 1-thread figure beside the all-cores one
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
                                   "one_thread": {"value": v1, "sample": "2 envs x one episode on 1 thread, %.1f s" % cdt1},
                                   "mujoco_probe": "; ".join(probe)}
        print(json.dumps(res))
    if group is not None:
        group.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
