"""Experiment: the batch as K independent part-batches on K HIP streams, so that the chain and phase kernels of one
part overlap the solver kernel of another (the solver has one wavefront per SIMD whatever the batch: its duration does not shrink
with the batch, so parts that run concurrently cost no solver time; what is won is the chain + phase time and the kernel tails).
usage (GPU box): python scripts/half_batch_overlap.py [scene] [n] [parts ...]   -> env-steps/s of a whole episode per part count"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import softgrip_amd as sg  # noqa: E402
from softgrip_amd import native  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "softbox"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
parts_list = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
m = sg.load_model(os.path.join(ROOT, "models", scene + ".sgmodel"))
nm = native.NativeModel(m)
ks = np.random.RandomState(0).uniform(300, 1400, n)
sched = episode_schedule()
ref = None
for parts in parts_list:
    sz = n // parts
    bs = [native.NativeBatch(nm, sz, 0) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    out = torch.zeros(n, len(sched), 12, dtype=torch.float64, device=bs[0].device)
    flags = torch.zeros(n, dtype=torch.int32, device=bs[0].device)
    for rep in range(2):
        for p, b in enumerate(bs):
            with torch.cuda.stream(streams[p]):
                b.set_stiffness(ks[p * sz:(p + 1) * sz], list(range(11, 64)), [0])
                b.reset(1, flags=flags[p * sz:(p + 1) * sz])
        torch.cuda.synchronize()
        t0 = time.time()
        ctrl = np.zeros(2)
        for t, c in enumerate(sched):
            for p, b in enumerate(bs):
                with torch.cuda.stream(streams[p]):
                    if c is not None:
                        ctrl[:] = c
                        b.set_ctrl_broadcast(ctrl)
                    b.step(7, sens=out[p * sz:(p + 1) * sz, t], sens_stride=len(sched) * 12, flags=flags[p * sz:(p + 1) * sz])
        torch.cuda.synchronize()
        dt = time.time() - t0
    a = out.cpu().numpy()
    if ref is None:
        ref = a
    print("%s, %d envs as %d part(s) of %d on %d stream(s): %.3f s per episode = %.0f env-steps/s; max |diff| to the first run %.3g; flagged %d" % (
        scene, n, parts, sz, parts, dt, n * len(sched) / dt, np.abs(a - ref).max(), int((flags != 0).sum())))
    del bs
