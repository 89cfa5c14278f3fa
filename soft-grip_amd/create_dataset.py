"""Dataset generation host: the reference's create_dataset.py (reference create_dataset.py:1-93)
driving batched envs.  Same constants, same schedule, same pickle schema
(``{"data": [ (200,12) float64 ...], "stiffness": [float ...]}``).

Multi-GPU: launched with torch.distributed.run, every rank simulates its own stiffness bin on
its own GPU and writes its own shard file -- no collective.
"""
import hashlib
import json
import os
import pickle
import sys
import time
from argparse import ArgumentParser
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .manenv import ManEnv

NUM_EPISODES = 1
MAX_ITER_PER_EP = 160
OPEN_CLOSE_DIV = 80
START_STEP = 40


def episode_schedule():
    """ctrl applied before each of the 200 env steps (reference create_dataset.py:41-56):
    40 idle, close (-0.2), toggle to +0.2 at i == 80."""
    sched = [None] * (START_STEP + MAX_ITER_PER_EP)
    sched[START_STEP] = -0.2
    for i in range(MAX_ITER_PER_EP):
        if i % OPEN_CLOSE_DIV == 0 and i > 0:
            sched[START_STEP + i] = 0.2 if sched[START_STEP + i - OPEN_CLOSE_DIV] == -0.2 else -0.2
    return sched


def stiffness_bin(rank, world, lo=300.0, hi=1400.0):
    w = (hi - lo) / world
    return lo + w * rank, lo + w * (rank + 1)


def _part_path(folder, name, ep):
    return os.path.join(folder, "{}.part{:05d}.pickle".format(name, ep))


def _atomic_pickle(path, obj):
    tmp = path + ".tmp"
    with open(tmp, "wb") as file:
        pickle.dump(obj, file)
    os.replace(tmp, path)   # a crash never leaves a half-written shard behind


def _fingerprint(args, env, rank, world, lo, hi, num_batches):
    """what a part must have been made with to belong to this run (a resumed run used to merge whatever
    <name>.partNNNNN.pickle it found): seed, batch shape, scenes (by content), stiffness range, integrator and read-out options"""
    def digest(path):
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    return {"seed": getattr(args, "seed", None), "n_envs": env.n_envs, "num_batches": num_batches, "rank": rank, "world": world,
            "scenes": [[os.path.basename(p), digest(p)] for p in args.mujoco_model_paths], "k_range": [float(lo), float(hi)],
            "sim_start": args.sim_start, "sim_step": args.sim_step, "mask_contact": bool(args.mask_contact),
            "contact_flag_mode": getattr(args, "contact_flag_mode", "intent"), "tendon_damper": getattr(args, "tendon_damper", "auto"),
            "joint_ids": list(env.joint_ids), "tendon_ids": list(env.tendon_ids)}


class ContactCapacityExceeded(RuntimeError):
    """more than --max-capacity-resets of the simulated episodes ran out of the kernels' contact capacity (64 per finger stream, 128 per
    env in the tree pipeline; the reference's MuJoCo holds nconmax = 500, soft_grip_two_fingers.xml:8, and would have carried on): such
    an env is reset with a re-drawn label like a MuJoCo warning, which is a SELECTION on the data that the reference does not make"""


class PartMismatch(RuntimeError):
    """a <name>.partNNNNN.pickle found on resume was written by a run with another configuration"""


def log_into_file(args):
    """reference create_dataset.py:20-80.  Extensions (all off by default on one GPU with one env, where the function does
    exactly what the reference does): n_envs > 1 -> every "episode" is a batch of n_envs episodes; several ranks -> each rank
    draws from its own stiffness bin and writes its own file; --num-batches B -> B episode-batches per model path;
    --incremental (always on with several ranks) -> every finished episode-batch is written at once as
    <name>.partNNNNN.pickle, a restarted run skips the parts it finds (the reference pickles once at the
    very end, create_dataset.py:75-79, and loses everything on a crash) and the final pickle is assembled from the parts."""
    assert type(args.mujoco_model_paths) is list
    num_envs = len(args.mujoco_model_paths)
    current_env = 0
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    num_batches = int(getattr(args, "num_batches", NUM_EPISODES) or NUM_EPISODES)
    incremental = bool(getattr(args, "incremental", False)) or world > 1

    env_spec = ManEnv.get_std_spec(args)
    env_spec["device"] = int(os.environ.get("LOCAL_RANK", getattr(args, "device", 0)))
    if getattr(args, "force_device", -1) >= 0:
        env_spec["device"] = args.force_device
    env = ManEnv(**env_spec)
    n = env.n_envs
    lo, hi = stiffness_bin(rank, world) if world > 1 else (300, 1400)   # stiffness sweep sharded by bin (BASELINE.json configs[3])

    os.makedirs(args.data_folder, exist_ok=True)
    name = args.data_name if world == 1 else "%s.rank%d" % (args.data_name, rank)
    path = os.path.join(args.data_folder, "{}.pickle".format(name))
    data, stiffness = list(), list()
    n_skipped = 0
    writer, pending = None, None
    fingerprint = _fingerprint(args, env, rank, world, lo, hi, num_batches)
    t_start, n_simulated, n_flagged, n_capacity = time.perf_counter(), 0, 0, 0

    for ep in range(num_batches * num_envs):
        part = _part_path(args.data_folder, name, ep)
        if incremental and os.path.exists(part):
            # a finished batch of an interrupted run: it must be this run's (same configuration), and the RNG continues from the
            # state stored with it -- a batch that re-drew labels for failed envs consumed more than its n draws, so counting
            # draws would not reproduce the uninterrupted run
            with open(part, "rb") as file:
                d = pickle.load(file)
            if d.get("config") != fingerprint:
                diff = sorted(k for k in fingerprint if d["config"].get(k) != fingerprint[k]) if isinstance(d.get("config"), dict) else []
                raise PartMismatch("%s was written by a run with another configuration (%s); remove the stale parts or use another "
                                   "--data-name" % (part, "differs in: " + ", ".join(diff) if diff else "no configuration stored with it"))
            env.rng.set_state(d["rng_state"])
            n_skipped += 1
        else:
            current_stiffness = np.array(env.reset(lo, hi), dtype=np.float64).reshape(-1).copy()   # the label is the pre-episode draw (reference :35,65)

            samples = list()
            for _ in range(START_STEP):
                readings, contact = env.step()
                readings = _mask(args, readings, contact)
                samples.append(readings)
            env.close_hand()
            for i in range(MAX_ITER_PER_EP):
                env.render()
                if i % OPEN_CLOSE_DIV == 0 and i > 0:
                    env.toggle_grip()
                readings, contact = env.step()
                readings = _mask(args, readings, contact)
                samples.append(readings)

            if n == 1:
                ep_data, ep_k = [np.array(samples)], [float(current_stiffness[0])]
            else:
                import torch
                block = torch.stack(samples, dim=1).cpu().numpy()  # [n, 200, 12]
                ep_data, ep_k = [np.array(block[e]) for e in range(n)], [float(k) for k in current_stiffness]
            if incremental:
                # the part is pickled and written by a worker thread while the next episode-batch simulates (the host is idle
                # then, waiting for the GPU); one write in flight bounds the memory, a failed write surfaces at the next hand-over
                if writer is None:
                    writer = ThreadPoolExecutor(max_workers=1)
                if pending is not None:
                    pending.result()
                pending = writer.submit(_atomic_pickle, part, {"data": ep_data, "stiffness": ep_k, "config": fingerprint,
                                                               "rng_state": env.rng.get_state()})
            else:
                data.extend(ep_data)
                stiffness.extend(ep_k)
            n_simulated += n
            n_flagged = int(getattr(env, "n_resets", 0))
            n_capacity = int(getattr(env, "n_capacity_resets", 0))
            limit = float(getattr(args, "max_capacity_resets", 0.001))
            if n_capacity > limit * n_simulated:
                if pending is not None:
                    pending.result()
                raise ContactCapacityExceeded("%d of %d simulated episodes hit the kernels' contact capacity (limit: %.3g of them, "
                                              "--max-capacity-resets); scene %s" % (n_capacity, n_simulated, limit, args.mujoco_model_paths[current_env]))

        if (ep + 1) % num_batches == 0 and num_envs > 1 and ep + 1 < num_batches * num_envs:
            # next scene (reference create_dataset.py:68-72; after the last one the reference asks load_env for an index past the
            # list, which only prints "Wrong number": nothing is reloaded at the end)
            current_env += 1
            env.load_env(current_env)

    if pending is not None:
        pending.result()
    if writer is not None:
        writer.shutdown()
    if incremental:   # assemble the final pickle from the parts (the parts stay: they are the resume state)
        for ep in range(num_batches * num_envs):
            with open(_part_path(args.data_folder, name, ep), "rb") as file:
                d = pickle.load(file)
            data.extend(d["data"])
            stiffness.extend(d["stiffness"])
        if n_skipped:
            print("resumed: {0} finished episode-batch(es) found and skipped".format(n_skipped))
    _atomic_pickle(path, {"data": data, "stiffness": stiffness})
    print("Total number of samples: {0}".format(len(data)))
    # per-rank summary (no collective: every rank writes its own; the self-launching parent adds them up, main())
    dt = time.perf_counter() - t_start
    n_steps = START_STEP + MAX_ITER_PER_EP
    summary = {"rank": rank, "world": world, "shard": os.path.basename(path), "shard_bytes": os.path.getsize(path), "episodes": len(data),
               "episodes_simulated": n_simulated, "episodes_resumed": len(data) - n_simulated, "env_steps": n_simulated * n_steps,
               "seconds": dt, "env_steps_per_s": n_simulated * n_steps / dt if dt > 0 else 0.0, "envs_reset_after_a_warning": n_flagged,
               "envs_reset_at_contact_capacity": n_capacity,
               "stiffness_bin": [float(lo), float(hi)]}
    with open(os.path.join(args.data_folder, "{}.summary.json".format(name)), "w") as file:
        json.dump(summary, file)
    return path


def _mask(args, readings, contact):
    if not args.mask_contact:
        return readings
    if isinstance(contact, (bool, np.bool_)):
        return readings if contact else np.zeros_like(readings)
    return readings * contact.to(readings.dtype).unsqueeze(-1)


def make_parser():
    parser = ArgumentParser()
    parser.add_argument('--sim-step', type=int, default=7)
    parser.add_argument('--vis', action='store_true', default=False)          # real booleans (reference uses type=bool)
    parser.add_argument('--mask-contact', action='store_true', default=False)
    parser.add_argument('--sim-start', type=int, default=1)
    parser.add_argument('--data-folder', type=str, default="./data/dataset/testing_datasets")
    parser.add_argument('--data-name', type=str, default="dataset_all_shapes")
    parser.add_argument('--mujoco-model-paths', nargs="+", required=True)
    parser.add_argument('--n-envs', type=int, default=1)
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--num-batches', type=int, default=NUM_EPISODES, help="episode-batches per model path (reference: NUM_EPISODES = 1)")
    parser.add_argument('--incremental', action='store_true', default=False,
                        help="write every finished episode-batch as <name>.partNNNNN.pickle and skip finished parts on restart")
    parser.add_argument('--contact-flag-mode', default="intent", choices=["intent", "reference"])
    parser.add_argument('--tendon-damper', default="auto", choices=["auto", "explicit", "implicit"],
                        help="integration of the composite volume tendon's damper (DESIGN.md D5); auto = explicit, implicit for scenes that need it")
    parser.add_argument('--joint-ids', type=int, nargs="+", default=None, help="joints whose stiffness is randomised (default: the reference's 11..63)")
    parser.add_argument('--tendon-ids', type=int, nargs="+", default=None, help="tendons whose stiffness is randomised (default: the reference's 0)")
    parser.add_argument('--finger-names', nargs="+", default=None, help="geom name fragments of the fingers for the contact flag (default: the reference's "
                        "['g12', 'g2']; four-finger gripper: g11 g12 g13 g2, reference manenv.py:16)")
    parser.add_argument('--n-actuated', type=int, default=None, help="actuators close_hand / loose_hand drive (default 2; four-finger gripper: 4)")
    parser.add_argument('--no-check-scene', dest="check_scene", action='store_false', default=True,
                        help="skip the load-time dry run that rejects scenes which cannot survive their own idle phase")
    parser.add_argument('--gpus', type=int, default=1,
                        help="N > 1 without a launcher: start N ranks (child processes of a parent that touches no GPU), one per GPU, each with its own "
                             "stiffness bin and shard file -- BASELINE configs[3]: --gpus 8 --n-envs 4096 --total-episodes 131072")
    parser.add_argument('--total-episodes', type=int, default=None,
                        help="fixed dataset size over all ranks and scenes: sets --num-batches to ceil(total / (ranks x n_envs x scenes))")
    parser.add_argument('--max-capacity-resets', type=float, default=0.001,
                        help="fail when more than this share of the simulated episodes hit the kernels' contact capacity (an env MuJoCo, with the "
                             "reference's nconmax = 500, would have kept: resetting it is a selection the reference does not make)")
    parser.add_argument('--force-device', type=int, default=-1, help="testing only: put every rank on this GPU")
    return parser


def _self_launch(args, argv, explicit_argv):
    """--gpus N > 1 without a launcher: N ranks as child processes (ranks.spawn_ranks: this process never touches a GPU), then the
    ranks' summary files added up into one JSON line.  The ranks share nothing: no collective, no store, no common file.
    A rank runs `python -m softgrip_amd.create_dataset <argv>` when main() was CALLED with an argv (a script, a notebook, a test: its
    own command line is not a create_dataset command), and this process's own command line when main() is the CLI entry (which keeps
    whatever wrapper the user started it under, e.g. tests/run_with_fake_native.py)."""
    from . import ranks
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.perf_counter()
    cmd = [sys.executable, "-m", "softgrip_amd.create_dataset"] + list(argv) if explicit_argv else None
    for r in range(args.gpus):   # summaries of an earlier job under the same name must not be added up as this one's
        stale = os.path.join(args.data_folder, "%s.rank%d.summary.json" % (args.data_name, r))
        if os.path.exists(stale):
            os.remove(stale)
    rc = ranks.spawn_ranks(args.gpus, {"PYTHONPATH": root + os.pathsep + os.environ.get("PYTHONPATH", "")}, cmd=cmd)
    if rc == 0:
        print(json.dumps(job_summary(args.data_folder, args.data_name, args.gpus, time.perf_counter() - t0)))
    return rc


def job_summary(folder, name, world, wall_seconds=None):
    """the whole job from the ranks' <name>.rank<r>.summary.json files (world == 1: <name>.summary.json)"""
    ranks = []
    for r in range(world):
        with open(os.path.join(folder, "%s.summary.json" % (name if world == 1 else "%s.rank%d" % (name, r)))) as f:
            ranks.append(json.load(f))
    slowest = max(r["seconds"] for r in ranks)
    steps = sum(r["env_steps"] for r in ranks)
    return {"job": "create_dataset", "n_gpus": world, "episodes": sum(r["episodes"] for r in ranks), "env_steps": steps,
            "env_steps_per_s": steps / slowest if slowest > 0 else 0.0, "slowest_rank_seconds": slowest, "wall_seconds_with_launch": wall_seconds,
            "envs_reset_after_a_warning": sum(r["envs_reset_after_a_warning"] for r in ranks),
            "envs_reset_at_contact_capacity": sum(r.get("envs_reset_at_contact_capacity", 0) for r in ranks),
            "shard_bytes": [r["shard_bytes"] for r in ranks], "per_rank_env_steps_per_s": [r["env_steps_per_s"] for r in ranks],
            "note": "end to end on the host: ManEnv.step() with its per-step flag check, device -> host copies, pickling, file writes"}


def main(argv=None):
    explicit_argv = argv is not None
    argv = sys.argv[1:] if argv is None else list(argv)
    args, _ = make_parser().parse_known_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(_self_launch(args, argv, explicit_argv))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.total_episodes is not None:
        per_round = world * args.n_envs * len(args.mujoco_model_paths)
        args.num_batches = max(1, -(-args.total_episodes // per_round))
    if args.seed is not None:
        np.random.seed(args.seed + int(os.environ.get("RANK", 0)) * 1000)
    log_into_file(args)
    if world == 1:
        print(json.dumps(job_summary(args.data_folder, args.data_name, 1)))


if __name__ == '__main__':
    main()
