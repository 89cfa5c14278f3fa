"""Per-section cycle profile of the phase kernel (profiling build: `python soft-grip_amd/build_native.py --prof`, which
compiles the SG_T stamps of csrc/sg_phase.hip and csrc/sg_rows.hip in).  Prints, for windows of the 200-step squeeze episode, the average cycles
one wavefront spends in each section of sg_phase_kernel.

usage (GPU box): SOFTGRIP_LIB=soft-grip_amd/libsoftgrip_prof.so python scripts/section_profile.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SOFTGRIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "soft-grip_amd", "libsoftgrip_prof.so"))
import softgrip_amd as sg  # noqa: E402
from softgrip_amd import native  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

NAMES = ["0 prologue/load", "1 FINISH", "2 BEGIN elements", "3 narrowphase", "4 envelope checks", "5 contact export (stores)",
         "6 eq rows+recompute_a", "7 warmstart test", "8 export rest", "9 store state"]
PGS_NAMES = {10: "loop head / imp reduction", 11: "joint-fix rows", 12: "tendon row + write-back", 13: "limit rows", 14: "contact rows",
             15: "exit", 16: "final M^-1 J' f + export"}
CHAIN_NAMES = {17: "load state", 18: "FINISH", 19: "kinematics", 20: "dynamics (M, M^-1, bias, tendon)", 21: "hand-off stores, boxes",
               22: "limit rows", 23: "store state"}


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "softbox"
    m = sg.load_model("models/%s.sgmodel" % scene, "explicit" if scene.startswith("softbox") else "implicit")
    nm = native.NativeModel(m)
    n = 4096
    b = native.NativeBatch(nm, n, 0)
    L = native.lib()
    L.sg_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
    buf = (C.c_ulonglong * 48)()
    b.set_stiffness(np.random.RandomState(0).uniform(300, 1400, n), list(range(11, 64)), [0])
    b.reset(1)
    L.sg_debug_sections(b.ptr, buf)
    ctrl = np.zeros(2)
    windows = {20: "idle (no contacts)", 60: "closing", 100: "squeeze peak", 199: "released"}
    nl = 0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
        if t in windows:
            L.sg_debug_sections(b.ptr, buf)  # clear
        b.step(7)
        if t in windows:
            L.sg_debug_sections(b.ptr, buf)
            v = np.array(buf[:10], dtype=np.float64) / (n * 8)  # 8 phase launches per sg_step call
            print("step %d, %s: %.0f cycles per wavefront and launch" % (t, windows[t], v.sum()))
            v24 = buf[24] / (n * 8.0)
            for k, name in enumerate(NAMES):
                print("   %-26s %8.0f  %5.1f %%" % (name, v[k], 100 * v[k] / (v.sum() + v24)))
            print("   %-26s %8.0f  %5.1f %%" % ("24 contact rows (build)", v24, 100 * v24 / (v.sum() + v24)))
            print("   broadphase survivors per env and launch: %.1f pairs, %.2f dense narrowphase passes" % (buf[30] / (n * 8.0), buf[31] / (n * 8.0)))
            epw = b.solver_envs_per_wavefront()  # envs per PGS wavefront: the library's own choice (sg_api.hip solver_epw)
            nwave = n // epw
            w = np.array(buf[:32], dtype=np.float64) / (nwave * 7)  # 7 PGS launches per sg_step call
            tot = sum(w[k] for k in PGS_NAMES)
            print("   sg_pgs_rows_kernel: %.0f cycles per wavefront and launch (after the prologue)" % tot)
            print("   contact passes per wavefront and launch: %.1f, contact slots swept: %.0f" % (w[29], w[28]))
            if buf[26]:  # counting build (build_native.py --count): the cycle figures of such a run are not timings
                print("   contact updates per stream and launch: %.0f, of which outside the friction cone (Newton / QCQP path): %.0f" % (
                    buf[26] / (2.0 * n * 7), buf[27] / (2.0 * n * 7)))
                print("   QCQP fallback per wavefront and launch: entered on %.0f slots, %.0f further Newton evaluations" % (
                    buf[32] / (nwave * 7.0), buf[33] / (nwave * 7.0)))
            for k, name in PGS_NAMES.items():
                print("   %-26s %8.0f  %5.1f %%" % (name, w[k], 100 * w[k] / max(tot, 1)))
            cw = np.array(buf[:32], dtype=np.float64) / ((n // 64) * 2 * 8)  # 8 chain launches per sg_step call, 64 chains per wavefront
            tot = sum(cw[k] for k in CHAIN_NAMES)
            print("   sg_chain_kernel: %.0f cycles per wavefront and launch" % tot)
            for k, name in CHAIN_NAMES.items():
                print("   %-34s %8.0f  %5.1f %%" % (name, cw[k], 100 * cw[k] / max(tot, 1)))


if __name__ == "__main__":
    main()
