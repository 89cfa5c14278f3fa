"""Dataset generation host: the reference's create_dataset.py (reference create_dataset.py:1-93)
driving batched envs.  Same constants, same schedule, same pickle schema
(``{"data": [ (200,12) float64 ...], "stiffness": [float ...]}``).

Multi-GPU: launched with torch.distributed.run, every rank simulates its own stiffness bin on
its own GPU and writes its own shard file -- no collective.
"""
import os
import pickle
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
    env = ManEnv(**env_spec)
    n = env.n_envs
    lo, hi = stiffness_bin(rank, world) if world > 1 else (300, 1400)   # stiffness sweep sharded by bin (BASELINE.json configs[3])

    os.makedirs(args.data_folder, exist_ok=True)
    name = args.data_name if world == 1 else "%s.rank%d" % (args.data_name, rank)
    path = os.path.join(args.data_folder, "{}.pickle".format(name))
    data, stiffness = list(), list()
    n_skipped = 0
    writer, pending = None, None

    for ep in range(num_batches * num_envs):
        part = _part_path(args.data_folder, name, ep)
        if incremental and os.path.exists(part):
            env.rng.uniform(lo, hi, size=n if n > 1 else None)   # consume the draws of the finished batch: the stream stays aligned
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
                pending = writer.submit(_atomic_pickle, part, {"data": ep_data, "stiffness": ep_k})
            else:
                data.extend(ep_data)
                stiffness.extend(ep_k)

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
    parser.add_argument('--no-check-scene', dest="check_scene", action='store_false', default=True,
                        help="skip the load-time dry run that rejects scenes which cannot survive their own idle phase")
    return parser


def main(argv=None):
    args, _ = make_parser().parse_known_args(argv)
    if args.seed is not None:
        np.random.seed(args.seed + int(os.environ.get("RANK", 0)) * 1000)
    log_into_file(args)


if __name__ == '__main__':
    main()
