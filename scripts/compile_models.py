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

def fourfinger_scene(tmpdir):
    """the reference's FOUR-finger gripper (data/gripper/soft_grip_four_fingers.xml; its ids survive only as comments
    in environment/manenv.py:11,16) squeezing the soft ball.  The reference has no experiment file for the pair that is not also a
    free-joint scene, so this one is the two-finger ball experiment with its <include> switched to the four-finger gripper --
    generated here from the reference's files (copied to a scratch directory; nothing is written into the reference)."""
    import shutil
    for f in os.listdir(REF):
        if f.endswith(".xml"):
            shutil.copy(os.path.join(REF, f), tmpdir)
    x = open(os.path.join(tmpdir, "soft_experiments_softball_adjusted_for_2_fingers.xml")).read()
    assert "soft_grip_two_fingers.xml" in x
    path = os.path.join(tmpdir, "soft_experiments_softball_four_fingers.xml")
    with open(path, "w") as f:
        f.write(x.replace("soft_grip_two_fingers.xml", "soft_grip_four_fingers.xml"))
    return path


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "models"), exist_ok=True)
    import tempfile
    # the free-floating ball (reference data/gripper/soft_experiments_softball.xml: <freejoint/> on the composite's body)
    for suffix, nb in (("_fix", False), ("", True)):
        m = sg.compile_mjcf(os.path.join(REF, "soft_experiments_softball.xml"), composite_neighbors=nb)
        out = os.path.join(ROOT, "models", "freeball" + suffix + ".sgmodel")
        with open(out, "wb") as f:
            f.write(m.to_blob())
        print("freeball" + suffix, "nq", m.nq, "nv", m.nv, "neq", m.neq, "->", out, os.path.getsize(out), "bytes")
    with tempfile.TemporaryDirectory() as tmp:   # the four-finger gripper (tree pipeline)
        for suffix, nb in (("_fix", False), ("", True)):   # fix rows only / with the composite's neighbour equalities (651 rows)
            m = sg.compile_mjcf(fourfinger_scene(tmp), composite_neighbors=nb)
            out = os.path.join(ROOT, "models", "fourfinger_softball" + suffix + ".sgmodel")
            with open(out, "wb") as f:
                f.write(m.to_blob())
            print("fourfinger_softball" + suffix, "nv", m.nv, "neq", m.neq, "->", out, os.path.getsize(out), "bytes")
    # <scene>.sgmodel: the composite as MuJoCo's documentation describes it -- fix rows, neighbour equalities, tendon row (the
    # default; rows pipeline only); <scene>_fix.sgmodel: the same scene without the neighbour equalities (opt-in)
    for suffix, nb in (("", True), ("_fix", False)):
        for name, xml in SCENES.items():
            m = sg.compile_mjcf(os.path.join(REF, xml), composite_neighbors=nb)
            out = os.path.join(ROOT, "models", name + suffix + ".sgmodel")
            with open(out, "wb") as f:
                f.write(m.to_blob())
            print(name + suffix, "nv", m.nv, "neq", m.neq, "->", out, os.path.getsize(out), "bytes")
