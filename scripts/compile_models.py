"""Compiles the reference's MJCF scenes into model blobs (models/*.sgmodel).

The MJCF files live only in /root/reference (never on the GPU box), so the compiled blobs --
pure data: masses, poses, constraint parameters -- are committed, together with this script.
Run in the build container:  python scripts/compile_models.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import softgrip_amd as sg  # noqa: E402

REF = "/root/reference/data/gripper"
SCENES = {
    "softbox": "soft_experiments_softbox_adjusted_for_2_fingers.xml",
    "softcylinder": "soft_experiments_softcylinder_adjusted_for_2_fingers.xml",
    "softball": "soft_experiments_softball_adjusted_for_2_fingers.xml",
}

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "models"), exist_ok=True)
    # <scene>.sgmodel: the composite as MuJoCo's documentation describes it -- fix rows, neighbour equalities, tendon row (the
    # default; rows pipeline only); <scene>_fix.sgmodel: the same scene without the neighbour equalities (opt-in)
    for suffix, nb in (("", True), ("_fix", False)):
        for name, xml in SCENES.items():
            m = sg.compile_mjcf(os.path.join(REF, xml), composite_neighbors=nb)
            out = os.path.join(ROOT, "models", name + suffix + ".sgmodel")
            with open(out, "wb") as f:
                f.write(m.to_blob())
            print(name + suffix, "nv", m.nv, "neq", m.neq, "->", out, os.path.getsize(out), "bytes")
