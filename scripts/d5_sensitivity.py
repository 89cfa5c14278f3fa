"""D5 sensitivity: is the blow-up of the reference's ball / cylinder scenes under the explicit volume-tendon damper
a property of the restated system or of one of the places where the restatement resolved something the documentation leaves open?
The oracle runs each scene (explicit damper, composite neighbour equalities on and off) with each of its sensitivity switches flipped
(oracle/sg_oracle.c g_variant) and reports the env step of the first simulation warning.  CPU only; needs the committed model blobs.
usage: python scripts/d5_sensitivity.py > profiles/r03_d5_sensitivity.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import softgrip_amd as sg  # noqa: E402
from helpers import model_path, oracle_sim  # noqa: E402
from oracle import oracle as O  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

VARIANTS = [(0, "restatement as tested"),
            (1, "warmstart = the implicit-damping acceleration the Euler step integrates"),
            (2, "direct solref taken literally (K, B not divided by dmax^2, dmax)"),
            (4, "equality rows with the impedance at zero violation"),
            (8, "tendon equality: diagApprox = sum of its dofs' invweights"),
            (16, "no warmstart"),
            (1 | 2 | 4 | 8, "all four resolutions flipped at once")]


def episode(scene, k=700.0):
    m = sg.load_model(model_path(scene), "explicit")
    s = oracle_sim(m, k)
    s.reset(); s.forward()
    nc0 = s.ncon
    w = s.step()
    most = 0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            s.ctrl[:] = c
        for _ in range(7):
            w = s.step()
            if w:
                return "%2d contacts at reset; warning %2d at env step %3d" % (nc0, w, t)
        most = max(most, s.ncon)
    return "%2d contacts at reset; episode completes (most contacts %d)" % (nc0, most)


def main():
    L = O.lib()
    L.sgo_set_variant.argtypes = [ctypes.c_int]
    print("explicit volume-tendon damper (MuJoCo's Euler as restated), k = 700, the reference's squeeze schedule; oracle only")
    for scene in ("softball", "softball_fix", "softcylinder", "softcylinder_fix", "softbox"):
        print(scene)
        for bits, what in VARIANTS:
            L.sgo_set_variant(bits)
            print("  variant %2d  %-75s %s" % (bits, what, episode(scene)))
    L.sgo_set_variant(0)


if __name__ == "__main__":
    main()
