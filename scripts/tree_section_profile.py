"""Per-section cycle profile of the tree pipeline's kernel (profiling build: `python soft-grip_amd/build_native.py --prof`, which
compiles the SGT_STAMP marks of csrc/sg_tree.h in): average cycles one wavefront (= one env) spends in each section of one physics
substep, for windows of the 200-step squeeze episode.

usage (GPU box): python scripts/tree_section_profile.py [scene] [n_envs]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SOFTGRIP_LIB", os.path.join(ROOT, "soft-grip_amd", "libsoftgrip_prof.so"))
import softgrip_amd as sg  # noqa: E402
from softgrip_amd import native  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

NAMES = ["0 checks", "1 kinematics, box / site poses", "2 tendons, actuators", "3 mass matrix", "4 L'DL + M^-1 columns", "5 RNE, bias, smooth acc",
         "6 sliders smooth", "7 pairs: other blocks", "8 rank + narrowphase", "9 eq / limit rows", "10 contact rows (J, W, A)",
         "11 warmstart", "12 sweep: slider limit rows", "13 sweep: contacts", "14 sweep end", "15 qacc, sensors", "16 Euler (M + hB)", "17 sweep: joint-fix rows", "18 sweep: tendon row", "19 sweep: chain limit rows", "20 pairs: object box, box flags, block lists", "21 pairs: (capsule | sphere) x box blocks"]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "fourfinger_softball_fix"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    m = sg.load_model(os.path.join(ROOT, "models", scene + ".sgmodel"), "implicit" if "box" not in scene else None)
    nm = native.NativeModel(m)
    b = native.NativeBatch(nm, n, 0)
    if nm.nboxes <= 4:
        b.set_pipeline("tree")
    L = native.lib()
    L.sg_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
    buf = (C.c_ulonglong * 48)()
    jids = list(range(65, 283)) if scene.startswith("fourfinger") else list(range(9, 227)) if scene.startswith("freeball") else list(range(11, 64))
    b.set_stiffness(np.random.RandomState(0).uniform(300, 1400, n), jids, [0])
    b.reset(1)
    ctrl = np.zeros(nm.nu)
    windows = {20: "idle (no contacts)", 60: "closing", 100: "squeeze peak", 199: "released"}
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
        if t in windows:
            L.sg_debug_sections(b.ptr, buf)  # clear
        b.step(7)
        if t in windows:
            L.sg_debug_sections(b.ptr, buf)
            v = np.array(buf[:len(NAMES)], dtype=np.float64) / (n * 7)
            st = b.solver_stats()
            print("step %d, %s: %.0f cycles per env and substep (mean ncon %.1f, sweeps %.1f)" % (t, windows[t], v.sum(), float(st["ncon"].double().mean()), float(st["iters"].double().mean())))
            for k, name in enumerate(NAMES):
                print("   %-36s %9.0f  %5.1f %%" % (name, v[k], 100 * v[k] / v.sum()))
            x = np.array(buf[40:48], dtype=np.float64)
            if x[1] or x[5]:
                print("   contact levels: %.2f update slots per pass of the lane groups; chain limit rows: %.2f row slots per pass" % (x[0] / max(x[1], 1), x[4] / max(x[5], 1)))
                if x[3]:
                    print("   one update slot: %.0f cycles to the residual (record's loads, row sums, LDS words), %.0f in the 3 x 3 block update, %.0f in the pushes and stores" % (x[2] / x[3], x[6] / x[3], x[7] / x[3]))


if __name__ == "__main__":
    main()
