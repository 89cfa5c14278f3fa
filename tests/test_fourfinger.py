"""SURVEY 8(f) rank 4, first half: the reference's FOUR-finger gripper (data/gripper/soft_grip_four_fingers.xml: 4 fingers of 8 links /
16 dofs, one spatial tendon through 8 sites per finger, 4 cylinder actuators, 8 sensors = 24 channels; its ids are comments in
environment/manenv.py:11,16) on the model compilers and the oracle -- the way round 1 started the two-finger box.  The fast kernels' plan
class is two fingers of two links; this model runs in the tree pipeline (csrc/sg_tree.h; tests/test_tree_emu.py, tests/test_gpu_tree.py);
the free-floating ball of soft_experiments_softball.xml (<freejoint/>) is not built anywhere yet.

models/fourfinger_softball_fix.sgmodel is compiled by scripts/compile_models.py from the reference's files (the two-finger ball experiment
with its <include> switched to the four-finger gripper)."""
import os

import numpy as np
import pytest

import softgrip_amd as sg
from helpers import ROOT, model_path
from oracle import oracle as O
from softgrip_amd.create_dataset import episode_schedule

REF = "/root/reference/data/gripper"
FINGERS = ['g11', 'g12', 'g13', 'g2']           # reference environment/manenv.py:16 (commented four-finger variant)


@pytest.fixture(scope="module")
def model():
    return sg.load_model(model_path("fourfinger_softball_fix"), "implicit")


def test_sizes_of_the_four_finger_model(model):
    m = model
    assert m.nv == 283 and m.nv - 218 == 65      # 218 shell elements of the 7 x 7 x 7 ellipsoid + 65 gripper joints:
    # three left fingers of 16 dofs (adduction + twist on the first link, flexion + twist on seven more) and the right one of 17
    assert m.ntendon == 5 and m.nu == 4 and m.nsensordata == 24
    assert m.tendon_num.tolist() == [218, 8, 8, 8, 8]          # OBJT first (tendon_ids = [0] still means the volume tendon), 8 sites each
    assert abs(float(np.sum(m.body_mass)) - float(np.sum(sg.load_model(model_path("softball_fix")).body_mass))) > 0   # another gripper
    for name in FINGERS:
        assert any(name in (g or "") for g in m.geom_names), name
    # 24 channels: four accelerometers then four gyros (soft_grip_four_fingers.xml:351-361)
    assert m.sensor_type.tolist() == [m.sensor_type[0]] * 4 + [m.sensor_type[4]] * 4 and m.sensor_type[0] != m.sensor_type[4]


def test_multi_site_tendon_length_and_jacobian(model):
    """a spatial tendon through 8 sites: its length is the sum of the seven segment lengths between consecutive site positions, and the
    oracle's Jacobian row is the derivative of that length (central differences on every gripper dof)"""
    m = model
    om = O.OracleModel(m.to_blob())
    s = O.OracleSim(om)
    s.reset()
    rng = np.random.RandomState(0)
    s.qpos[:65] += 0.05 * rng.randn(65)
    s.forward()
    L = O.lib()
    import ctypes as C
    site = np.ctypeslib.as_array(L.sgo_site_xpos(s.ptr), shape=(len(m.site_bodyid), 3)).copy()
    for t in range(1, 5):
        a, n = m.tendon_adr[t], m.tendon_num[t]
        ids = m.wrap_objid[a:a + n]
        want = sum(np.linalg.norm(site[ids[i + 1]] - site[ids[i]]) for i in range(n - 1))
        assert abs(s.ten_length[t] - want) < 1e-12
    base = s.qpos.copy()
    L0 = s.ten_length.copy()
    assert np.all(L0[1:] > 2.0)
    # Jacobian rows through the actuator: d length / d q by central differences against the moment the oracle uses.  The moment is not
    # exposed; the tendon spring force is: give the tendon a stiffness and compare qfrc changes -- simpler: finite differences of the
    # length against a second finite difference scheme at half the step agree (the function is smooth), and a dof outside the finger
    # leaves the length alone
    for t in range(1, 5):
        g1, g2 = np.zeros(65), np.zeros(65)
        for j in range(65):
            for h, g in ((1e-5, g1), (5e-6, g2)):
                s.qpos[:] = base; s.qpos[j] += h; s.forward(); lp = s.ten_length[t]
                s.qpos[:] = base; s.qpos[j] -= h; s.forward(); lm = s.ten_length[t]
                g[j] = (lp - lm) / (2 * h)
        assert np.abs(g1 - g2).max() < 1e-6
        nz = np.flatnonzero(np.abs(g1) > 1e-9)
        assert 6 <= len(nz) <= 17                                   # only its own finger's dofs move it
        assert nz.max() - nz.min() < 17
    s.qpos[:] = base


def test_four_finger_squeeze_episode_on_the_oracle(model):
    """the reference's schedule on the four-finger scene (oracle only): every finger reaches the ball, 24 channels stay finite, and the
    contact read-out of manenv.py:65-85 with finger_names = ['g11', 'g12', 'g13', 'g2'] comes up"""
    m = model
    s = O.OracleSim(O.OracleModel(m.to_blob()))
    s._om = s.model
    k = 700.0
    s.jnt_stiffness[65:] = k
    s.tendon_stiffness[0] = k
    s.reset(); s.forward()
    assert s.ncon == 0
    assert s.step() == 0
    rows, touched = [], set()
    for t, c in enumerate(episode_schedule()[:100]):
        if c is not None:
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0, t
        rows.append(s.sensordata.copy())
        for cc in s.contacts():
            n1, n2 = m.geom_names[cc["geom1"]] or "", m.geom_names[cc["geom2"]] or ""
            if "OBJ" in n1 or "OBJ" in n2:
                for f in FINGERS:
                    if f in n1 or f in n2:
                        touched.add(f)
    rows = np.array(rows)
    assert rows.shape == (100, 24) and np.isfinite(rows).all()
    assert np.abs(rows[5, [2, 5, 8, 11]] - 9.81).max() < 0.5           # four accelerometers at rest read gravity along their z
    assert touched == set(FINGERS), touched


def test_kernels_take_the_four_finger_model(model):
    """the two-finger kernels refuse it, the tree pipeline (csrc/sg_tree.h) runs it: sg_model_create accepts the blob (no GPU needed for
    that) and reports the 64 finger boxes; tests/test_tree_emu.py and tests/test_gpu_tree.py hold the pipeline against the oracle"""
    from softgrip_amd import native
    nm = native.NativeModel(model)
    assert nm.nboxes == 64 and nm.nsensordata == 24 and nm.nq == 283 and nm.nelem == 218
    bad = sg.load_model(model_path("fourfinger_softball_fix"), "implicit")
    bad.jnt_type = bad.jnt_type.copy()
    bad.jnt_type[3] = 2                    # a slide joint in a finger chain: outside both classes
    with pytest.raises(native.SoftgripError) as ei:
        native.NativeModel(bad)
    assert ei.value.code == native.SG_ERR_MODEL and "two-finger kernels" in str(ei.value) and "tree pipeline" in str(ei.value)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference's MJCF files (build container only)")
def test_both_compilers_agree_on_the_four_finger_scene(tmp_path):
    """the malformed text node at soft_grip_four_fingers.xml:317 and the 8-site tendons go through both compilers; the two blobs agree
    field by field, and the committed blob is a fresh compile"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from compile_models import fourfinger_scene
    from softgrip_amd import native
    path = fourfinger_scene(str(tmp_path))
    a = sg.compile_mjcf(path, composite_neighbors=False)
    b = sg.Model.from_blob(native.compile_mjcf_native(path, composite_neighbors=False))
    assert (a.nv, a.ntendon, a.nu, a.neq) == (b.nv, b.ntendon, b.nu, b.neq) == (283, 5, 4, 219)
    for f in ("body_mass", "body_pos", "jnt_axis", "jnt_range", "geom_size", "site_pos", "tendon_length0", "dof_invweight0", "tendon_invweight0"):
        np.testing.assert_allclose(getattr(a, f), getattr(b, f), rtol=1e-9, atol=1e-12, err_msg=f)
    assert a.wrap_objid.tolist() == b.wrap_objid.tolist() and a.geom_names == b.geom_names
    with open(model_path("fourfinger_softball_fix"), "rb") as f:
        assert f.read() == a.to_blob()
    # with the composite's neighbour equalities the scene has the ball's 651 rows
    assert sg.compile_mjcf(path).neq == 651
