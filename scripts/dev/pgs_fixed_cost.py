"""PGS kernel launch time at idle as a function of the sweep cap: separates the per-sweep cost from the fixed part of a launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import softgrip_amd as sg
from softgrip_amd import native
n = 4096
for iters in (30, 15, 5, 1):
    m = sg.load_model(os.path.join(ROOT, "models", "softbox.sgmodel"))
    m.opt_iterations = iters
    b = native.NativeBatch(native.NativeModel(m), n, 0)
    b.set_stiffness(np.random.RandomState(0).uniform(300, 1400, n), list(range(11, 64)), [0])
    b.reset(1)
    for _ in range(5): b.step(7)
    b.profile_enable(True); b.profile_read(True); b.profile_read_solver(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): b.step(7)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, nl = b.profile_read(True); sms, snl = b.profile_read_solver(True)
    print("iterations %2d: step %.3f ms wall, kernel chain %.3f ms, solver launch %.1f us (x%d)" % (iters, dt / 30 * 1e3, ms, sms * 1e3, snl))
