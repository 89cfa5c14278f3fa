"""SURVEY 8(f) rank 4, second half: the free joint (reference data/gripper/soft_experiments_softball.xml:8, `<freejoint/>` on the body
that carries the ball's composite) in the Python MJCF compiler and the oracle -- the way the four-finger gripper started.  7 positions
(world position + quaternion), 6 dofs (linear velocity in the world frame, angular velocity in the body frame); the blob gains the
address maps jnt_qposadr / jnt_dofadr / dof_jntid for such models only.  Known answers on a scene of this repo's own
(tests/data/free_body.xml), then the reference's scene on the oracle.  The two-finger kernels refuse a free joint; the tree
pipeline's object block runs it (tests/test_tree_emu.py, tests/test_gpu_tree.py)."""
import os

import numpy as np
import pytest

import softgrip_amd as sg
from helpers import ROOT, oracle_sim
from softgrip_amd.create_dataset import episode_schedule
from softgrip_amd.mjcf import quat_mul, quat_to_mat

REF = "/root/reference/data/gripper/soft_experiments_softball.xml"


@pytest.fixture(scope="module")
def brick():
    return sg.compile_mjcf(os.path.join(ROOT, "tests", "data", "free_body.xml"))


def test_sizes_and_address_maps(brick):
    m = brick
    assert (m.nq, m.nv, m.njnt) == (14, 12, 2) and m.has_free_joint
    assert m.jnt_qposadr.tolist() == [0, 7] and m.jnt_dofadr.tolist() == [0, 6] and m.dof_jntid.tolist() == [0] * 6 + [1] * 6
    np.testing.assert_allclose(m.qpos0[:3], [0, 0, 5])
    np.testing.assert_allclose(m.qpos0[3:7], np.array([0.9, 0.1, -0.3, 0.2]) / np.linalg.norm([0.9, 0.1, -0.3, 0.2]))
    # mj_setConst for a free body: 1 / m for the translations, the mean of 1 / I_k for the rotations
    I = 2.0 / 3 * np.array([0.2 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.2 ** 2])
    np.testing.assert_allclose(m.dof_invweight0[:3], 0.5, rtol=1e-12)
    np.testing.assert_allclose(m.dof_invweight0[3:6], np.mean(1 / I), rtol=1e-12)
    m2 = sg.Model.from_blob(m.to_blob())
    assert (m2.nq, m2.nv, m2.njnt) == (14, 12, 2) and m2.to_blob() == m.to_blob()


def test_free_fall_and_resting_sphere(brick):
    """semi-implicit Euler on a free body under gravity: v_k = -g h k, z_k = z_0 - g h^2 k (k + 1) / 2 to round-off, the orientation
    untouched; the sphere (0.1 mm above the plane) settles at its radius minus the soft contact's static penetration, carried by one
    frictionless contact"""
    s = oracle_sim(brick)
    s.reset(); s.forward()
    h, g, n = 0.002, 9.81, 400
    q0 = s.qpos.copy()
    for _ in range(n):
        assert s.step() == 0
    np.testing.assert_allclose(s.qvel[:3], [0, 0, -g * h * n], atol=1e-12)
    np.testing.assert_allclose(s.qpos[2], q0[2] - g * h * h * n * (n + 1) / 2, atol=1e-12)
    np.testing.assert_allclose(s.qpos[3:7], q0[3:7], atol=1e-15)
    assert np.abs(s.qvel[3:6]).max() == 0
    assert s.ncon == 1 and abs(s.qvel[8]) < 1e-6                      # the marble rests
    assert 0.19 < s.qpos[9] < 0.2 and np.abs(s.qpos[7:9] - [2, 0]).max() < 1e-9
    f = s.efc_force()
    assert abs(f[-1] - 0.5 * g) < 1e-6                                 # the contact carries the weight


def test_gyroscopic_bias_and_mass_matrix(brick):
    """a tumbling brick: the oracle's bias force on the rotational dofs is w x I w in the body frame, on the translations -m g; its mass
    matrix is diag(m, m, m, I_1, I_2, I_3) (body-frame rotations about the principal axes); both independent of the pose"""
    s = oracle_sim(brick)
    s.reset()
    w = np.array([1.3, -0.7, 2.1])
    s.qvel[3:6] = w
    s.qvel[:3] = [0.4, 0.0, -0.2]
    s.forward()
    I = 2.0 / 3 * np.array([0.2 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.2 ** 2])
    np.testing.assert_allclose(s.qfrc_bias[3:6], np.cross(w, I * w), atol=1e-13)
    np.testing.assert_allclose(s.qfrc_bias[:3], [0, 0, 2.0 * 9.81], atol=1e-12)
    np.testing.assert_allclose(np.diag(s.qM)[:6], [2, 2, 2, *I], rtol=1e-12)
    assert np.abs(s.qM[:6, :6] - np.diag(np.diag(s.qM)[:6])).max() < 1e-14
    # ... and the Python compiler's own mass matrix (NumPy, another implementation) agrees on the whole model
    M, _ = brick.mass_matrix(s.qpos.copy())
    np.testing.assert_allclose(np.tril(s.qM), np.tril(M), atol=1e-13)


def test_torque_free_tumbling_conserves_angular_momentum(brick):
    """no gravity torque on a free body: L = R I w is conserved by the exact motion; the first-order integrator (Euler on w, exact
    quaternion step) keeps it to O(h) -- and halving the step halves the drift.  The energy of rotation stays within O(h) too."""
    I = 2.0 / 3 * np.array([0.2 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.3 ** 2, 0.1 ** 2 + 0.2 ** 2])

    def run(nsub):
        m = sg.Model.from_blob(brick.to_blob())
        m.opt_timestep = 0.002 / nsub
        s = oracle_sim(m)
        s.reset()
        s.qvel[3:6] = [0.2, 3.0, 0.1]          # near the unstable middle axis (I_y lies between I_x and I_z): it tumbles
        L = []
        for _ in range(400 * nsub):
            R = quat_to_mat(s.qpos[3:7])
            L.append(R @ (I * s.qvel[3:6]))
            assert s.step() == 0
        return np.array(L), s

    L1, s1 = run(1)
    L2, _ = run(2)
    d1, d2 = np.abs(L1[-1] - L1[0]).max(), np.abs(L2[-1] - L2[0]).max()
    assert d1 < 5e-3 * np.abs(L1[0]).max() and 0.35 < d2 / d1 < 0.65, (d1, d2)
    assert abs(np.linalg.norm(s1.qpos[3:7]) - 1) < 1e-14
    assert np.abs(s1.qvel[3:6] - [0.2, 3.0, 0.1]).max() > 0.3          # the unstable axis is leaving


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference's MJCF files (build container only)")
def test_reference_free_ball_scene_on_the_oracle():
    """soft_experiments_softball.xml as the reference ships it: the ball's composite on a free body (8 gripper dofs + 6 + 218 sliders).
    Sizes, then the whole squeeze schedule on the oracle, no warning: everything stays finite, the quaternion normalised; the kernels
    refuse the model with a reason."""
    m = sg.compile_mjcf(REF, composite_neighbors=False)
    assert (m.nq, m.nv, m.njnt, m.neq) == (233, 232, 227, 219)
    j = int(np.flatnonzero(m.jnt_type == 0)[0])
    assert m.jnt_dofadr[j] == 8 and m.jnt_qposadr[j] == 8 and m.body_geomnum[m.jnt_bodyid[j]] == 1     # the free body carries OBJGcenter
    assert sg.compile_mjcf(REF).neq == 651
    m.opt_implicit_tendon_damping = 1                      # D5, as every ball scene
    s = oracle_sim(m)
    elem_joints = list(range(9, 227))                      # joint ids of the 218 sliders (the free joint is joint 8)
    s.jnt_stiffness[elem_joints] = 700.0
    s.tendon_stiffness[0] = 700.0
    s.reset(); s.forward()
    assert s.ncon > 20                                     # the shell starts inside the fingers, as in the two-finger ball scene
    p0 = s.qpos[8:11].copy()
    assert s.step() == 0
    track, ncons = [], []
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0, t
        assert np.isfinite(s.sensordata).all() and abs(np.linalg.norm(s.qpos[11:15]) - 1) < 1e-12
        track.append(s.qpos[8:11].copy()); ncons.append(s.ncon)
    track = np.array(track)
    # the fingers push the free ball out of their way and squeeze it upwards; released, it drops back onto them and stays in the gripper
    assert np.abs(track - p0).max() > 0.3 and track[:, 2].max() > p0[2] + 0.4 and track[-1, 2] < track[:, 2].max() - 0.2
    assert np.all(np.abs(track[:, 1]) < 0.3) and np.all((track[:, 2] > 0.5) & (track[:, 2] < 2.0)) and max(ncons) >= 30
    # the kernels: the two-finger plans refuse a free joint, the tree pipeline's object block takes it (tests/test_tree_emu.py,
    # tests/test_gpu_tree.py hold it against this oracle)
    from softgrip_amd import native
    nm = native.NativeModel(m)
    assert (nm.nq, nm.nv) == (233, 232)


def test_both_compilers_agree_on_free_joint_scenes():
    """the library's own MJCF compiler (csrc/sg_mjcf.cpp, sg_mjcf_compile) takes <freejoint/> like mjcf.py: sizes, address maps and the
    dof tree equal, reals to 1e-12 relative -- on the own scene and (build container) on the reference's free ball"""
    from softgrip_amd import native
    paths = [os.path.join(ROOT, "tests", "data", "free_body.xml")] + ([REF] if os.path.exists(REF) else [])
    for path in paths:
        a = sg.compile_mjcf(path, composite_neighbors=False)
        b = sg.Model.from_blob(native.compile_mjcf_native(path, composite_neighbors=False))
        assert (a.nq, a.nv, a.njnt, a.neq) == (b.nq, b.nv, b.njnt, b.neq) and b.has_free_joint
        for f in ("dof_parentid", "jnt_qposadr", "jnt_dofadr", "dof_jntid", "jnt_type", "jnt_bodyid", "eq_obj1id", "wrap_objid", "body_jntadr"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        for f in ("body_mass", "body_pos", "body_imat", "qpos0", "qpos_spring", "dof_damping", "dof_armature", "dof_invweight0", "body_invweight0",
                  "tendon_length0", "tendon_invweight0", "jnt_axis", "geom_pos"):
            x, y = np.asarray(getattr(a, f), float), np.asarray(getattr(b, f), float)
            if x.size:
                np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-14 * max(1.0, float(np.abs(x).max())), err_msg=f)
        assert abs(a.meaninertia - b.meaninertia) < 1e-14
