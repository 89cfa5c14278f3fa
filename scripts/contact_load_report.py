"""Why one scene's solver launch is longer than another's: event counts of the rows solver over a whole episode
from the counting build (`python soft-grip_amd/build_native.py --count` -> libsoftgrip_count.so: atomics inside the contact update,
so nothing here is a timing): contacts per env, the contact SLOTS a solver wavefront sweeps per pass (the longest of its 8 streams:
what its time follows), updates outside the friction cone, entries into mju_QCQP2's Newton iteration per wavefront slot and the
evaluations it then runs (the slowest stream's), next to the scene's equality schedule.
usage (GPU box): python scripts/contact_load_report.py [scene ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SOFTGRIP_LIB"] = os.path.join(ROOT, "soft-grip_amd", "libsoftgrip_count.so")
import softgrip_amd as sg  # noqa: E402
from softgrip_amd import native  # noqa: E402
from softgrip_amd.create_dataset import episode_schedule  # noqa: E402

n = 4096
for scene in (sys.argv[1:] or ["softbox", "softball", "softcylinder"]):
    m = sg.load_model(os.path.join(ROOT, "models", scene + ".sgmodel"), None if scene.startswith("softbox") else "implicit")
    nm = native.NativeModel(m)
    b = native.NativeBatch(nm, n, 0)
    L = native.lib()
    L.sg_debug_sections.argtypes = [C.c_void_p, C.c_void_p]
    buf = (C.c_ulonglong * 48)()
    b.set_stiffness(np.random.RandomState(0).uniform(300, 1400, n), list(range(11, 64)), [0])
    b.reset(1)
    L.sg_debug_sections(b.ptr, buf)
    ctrl = np.zeros(2)
    ncs, its = [], []
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
        b.step(7)
        st = b.solver_stats()
        ncs.append(st["ncon"].cpu().numpy()); its.append(st["iters"].cpu().numpy())
    L.sg_debug_sections(b.ptr, buf)
    v = np.array(buf[:48], dtype=np.float64)
    ncs, its = np.stack(ncs), np.stack(its)
    passes = v[29]                       # contact passes with a contact somewhere in the wavefront (one per sweep and pass of the streams)
    launches = 200 * 7
    print("%s: %d equality rows, %d sliders" % (scene, m.neq, nm.nelem))
    print("   contacts per env (end of each env step): mean %.1f over the episode, %.1f over the squeeze (steps 45 - 120), max %d; sweeps per solve: mean %.1f" % (
        ncs.mean(), ncs[45:120].mean(), ncs.max(), its.mean()))
    print("   contact slots swept per wavefront and pass (its longest stream): %.1f; passes per launch and wavefront: %.1f  -> %.0f slots per launch and wavefront" % (
        v[28] / max(passes, 1), passes / (launches * (n / 4)), v[28] / (launches * (n / 4))))
    print("   contact updates: %.3g per launch and env; outside the friction cone: %.1f %%; wavefront slots that enter the Newton iteration: %.1f %% (%.2f evaluations each)" % (
        v[26] / (launches * n), 100 * v[27] / max(v[26], 1), 100 * v[32] / max(v[28], 1), v[33] / max(v[32], 1)))
    del b, nm
