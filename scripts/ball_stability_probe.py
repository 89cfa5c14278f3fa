"""Why the reference's ball scene cannot produce data under the restated physics: CPU probe on the oracle.
Compiles variants of the reference's soft_experiments_softball_adjusted_for_2_fingers.xml (other composite spacing = other radius; with /
without the neighbour equalities; solver sweeps; the volume tendon's damper switched off) and runs the reference's squeeze schedule
until the first simulation warning.  Needs the reference's MJCF (build container only); writes nothing into the reference.
usage: python scripts/ball_stability_probe.py > profiles/r02_ball_stability_probe.txt"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import softgrip_amd as sg  # noqa: E402
from helpers import oracle_sim  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

REF = "/root/reference/data/gripper"
tmp = tempfile.mkdtemp()
for f in os.listdir(REF):
    if f.endswith(".xml"):
        shutil.copy(os.path.join(REF, f), tmp)
src = open(os.path.join(tmp, "soft_experiments_softball_adjusted_for_2_fingers.xml")).read()


def episode(spacing, nb, iters=30, tendon_damping=None, scene=None):
    p = os.path.join(tmp, "probe.xml")
    open(p, "w").write(src.replace('spacing="0.31"', 'spacing="%g"' % spacing))
    m = sg.compile_mjcf(scene or p, composite_neighbors=nb)
    m.opt_iterations = iters
    if tendon_damping is not None:
        m.tendon_damping[0] = tendon_damping
    s = oracle_sim(m, 700.0)
    s.reset(); s.forward()
    nc0, mx = s.ncon, 0
    w = s.step()
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            s.ctrl[:] = c
        for _ in range(7):
            w = s.step()
            if w:
                return "%2d contacts at reset; warning %d at env step %d (most contacts until then: %d)" % (nc0, w, t, mx)
        mx = max(mx, s.ncon)
    return "%2d contacts at reset; episode completes (most contacts: %d)" % (nc0, mx)


print("ball, composite spacing 0.31 (the reference's: radius 0.93, fingers' inner faces 0.75 from the axis), 0.27 and 0.26 (radius 0.78: free at reset)")
for sp in (0.31, 0.27, 0.26):
    for nb in (True, False):
        print("  spacing %.2f, neighbour rows %-3s: %s" % (sp, "on" if nb else "off", episode(sp, nb)))
print("radius 0.78, neighbour rows on, 1000 sweeps:          ", episode(0.26, True, iters=1000))
print("radius 0.78, neighbour rows off, 1000 sweeps:         ", episode(0.26, False, iters=1000))
print("radius 0.78, neighbour rows on, volume tendon damper 0:", episode(0.26, True, tendon_damping=0.0))
print("radius 0.78, neighbour rows off, volume tendon damper 0:", episode(0.26, False, tendon_damping=0.0))
print("radius 0.93, neighbour rows off, volume tendon damper 0:", episode(0.31, False, tendon_damping=0.0))
print("radius 0.93, neighbour rows on, volume tendon damper 0: ", episode(0.31, True, tendon_damping=0.0))
box = os.path.join(tmp, "soft_experiments_softbox_adjusted_for_2_fingers.xml")
print("box (benchmark scene), neighbour rows on:              ", episode(0.3, True, scene=box))
print("box (benchmark scene), neighbour rows off:             ", episode(0.3, False, scene=box))
shutil.rmtree(tmp)
