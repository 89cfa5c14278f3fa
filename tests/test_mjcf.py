"""MJCF compiler known answers (analytic items of App. B.9)."""
import os

import numpy as np
import pytest

import softgrip_amd as sg
from helpers import REF_XML, model_path

SIZES = {"softbox": (121, 118, 120, 110), "softcylinder": (203, 200, 202, 192), "softball": (229, 226, 228, 218)}


@pytest.mark.parametrize("scene", list(SIZES))
def test_blob_sizes_and_mass(scene):
    m = sg.load_model(model_path(scene + "_fix"))   # the composite without its neighbour equalities (opt-in variant)
    nbody, nv, ngeom, nelem = SIZES[scene]
    assert (m.nbody, m.nv, m.ngeom) == (nbody, nv, ngeom)
    assert m.neq == nelem + 1 and m.ntendon == 3 and (m.eq_obj2id < 0).all()
    nnb = {"softbox": 216, "softcylinder": 380, "softball": 432}[scene]   # the documented composite, the DEFAULT model
    mn = sg.load_model(model_path(scene))
    assert mn.neq == nelem + nnb + 1 and (mn.eq_obj2id >= 0).sum() == nnb and mn.eq_type[-1] == 3 and mn.eq_obj2id[0] == -1
    assert mn.nv == nv and np.array_equal(mn.body_mass, m.body_mass) and m.nu == 2 and m.nsensordata == 12
    assert abs(m.body_mass.sum() - 0.45) < 1e-14              # settotalmass
    assert m.opt_timestep == 0.005 and m.opt_iterations == 30 and m.opt_tolerance == 1e-7
    assert m.tendon_names[0] == "OBJT"                        # the tendon the reference randomises (manenv.py:13)
    np.testing.assert_allclose(m.tendon_length0[1:], 1.287012043, atol=1e-9)
    L, J = m.tendon_length_jac(m.qpos0)
    np.testing.assert_allclose(J[1, 0], 0.27660969, atol=1e-8)
    np.testing.assert_allclose(J[2, 5], -0.27660969, atol=1e-8)
    # element sliders: dof_invweight0 = 1/m, body_invweight0 = 1/(3m), OBJT invweight = sum 1/m
    e0 = nv - nelem
    me = m.body_mass[-nelem:]
    np.testing.assert_allclose(m.dof_invweight0[e0:], 1 / me, rtol=1e-12)
    np.testing.assert_allclose(m.body_invweight0[-nelem:, 0], 1 / (3 * me), rtol=1e-12)
    np.testing.assert_allclose(m.tendon_invweight0[0], (1 / me).sum(), rtol=1e-12)


def test_softbox_element_mass_and_ids():
    m = sg.load_model(model_path("softbox_fix"))
    np.testing.assert_allclose(m.body_mass[11], 1.9458618e-4, rtol=1e-7)
    assert m.geom_names[:10] == ["ground", "", "", "g121", "g122", "g123", "g21", "g22", "g23", "OBJGcenter"]
    assert m.jnt_names[8] == "OBJJ0_0_0" and m.body_names[10] == "" and m.body_names[11] == "OBJB0_0_0"
    # joints 11..63 (what set_new_stiffness writes) are object sliders 3..55
    assert all(n.startswith("OBJJ") for n in m.jnt_names[11:64])
    # direct-format solref of the composite equalities: k = 100/0.97^2, b = 10/0.97 (App. B.9 item 6)
    assert tuple(m.eq_solref[0]) == (-100.0, -10.0) and m.eq_solimp[0][1] == 0.97


@pytest.mark.skipif(not os.path.exists(REF_XML % "softbox"), reason="reference MJCF only exists in the build container")
@pytest.mark.parametrize("scene", list(SIZES))
def test_committed_blob_matches_reference_mjcf(scene):
    fresh = sg.compile_mjcf(REF_XML % scene).to_blob()           # default: with the composite's neighbour equalities
    with open(model_path(scene), "rb") as f:
        assert f.read() == fresh
    fresh = sg.compile_mjcf(REF_XML % scene, composite_neighbors=False).to_blob()
    with open(model_path(scene + "_fix"), "rb") as f:
        assert f.read() == fresh


def test_blob_roundtrip():
    m = sg.load_model(model_path("softbox_fix"))
    m2 = sg.Model.from_blob(m.to_blob())
    assert m2.to_blob() == m.to_blob()


def test_rejects_unsupported():
    import tempfile
    xml = "<mujoco><worldbody><body><freejoint/><geom type='sphere' size='1'/></body></worldbody></mujoco>"
    with tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False) as f:
        f.write(xml)
    with pytest.raises(ValueError):
        sg.compile_mjcf(f.name)
    os.unlink(f.name)


@pytest.mark.parametrize("scene,count,spacing,centre", [("softbox", (4, 5, 7), 0.3, (1.7, 0.0, 1.0)), ("softball", (7, 7, 7), 0.31, (1.7, 0.0, 1.0)),
                                                         ("softcylinder", (6, 8, 6), 0.3, (1.5, -0.04, 1.0))])
def test_composite_shell_geometry(scene, count, spacing, centre):
    """composite expansion: shell grid points in ix-outer / iz-inner order; box: grid * half-size, ellipsoid: unit
    direction * half-size; every element's slider axis and capsule axis point from the centre to the element, and the capsule is shifted
    inwards by radius + half-length so that its outer tip lies on the shell surface."""
    m = sg.load_model(model_path(scene))
    half = np.array([0.5 * spacing * (c - 1) for c in count])
    pts = []
    for ix in range(count[0]):
        for iy in range(count[1]):
            for iz in range(count[2]):
                if ix in (0, count[0] - 1) or iy in (0, count[1] - 1) or iz in (0, count[2] - 1):
                    p = np.array([2.0 * ix / (count[0] - 1) - 1, 2.0 * iy / (count[1] - 1) - 1, 2.0 * iz / (count[2] - 1) - 1])
                    pts.append(p * half if scene == "softbox" else p / np.linalg.norm(p) * half)
    pts = np.array(pts)
    n = len(pts)
    assert n == m.nv - 8
    np.testing.assert_allclose(m.body_pos[-n:], pts, atol=1e-14)                 # relative to the static parent at `centre`
    np.testing.assert_allclose(m.body_pos[10], centre, atol=1e-14)
    radial = pts / np.linalg.norm(pts, axis=1)[:, None]
    r, hl = m.geom_size[-1, 0], m.geom_size[-1, 1]
    # capsule centres sit radius + half-length inside the surface point, along the radial direction (geom_pos = (0, 0, -(r + hl)) in a frame whose z is radial)
    np.testing.assert_allclose(m.geom_pos[-n:], np.tile([0, 0, -(r + hl)], (n, 1)), atol=1e-15)
    from softgrip_amd.mjcf import quat_to_mat
    zaxis = np.array([np.asarray(quat_to_mat(q)).reshape(3, 3)[:, 2] for q in m.body_quat[-n:]])
    np.testing.assert_allclose(zaxis, radial, atol=1e-12)
    np.testing.assert_allclose(m.jnt_axis[-n:], np.tile([0, 0, 1.0], (n, 1)), atol=0)
    assert (m.jnt_type[-n:] == 2).all() and m.geom_size[9, 0] == 2 * r            # slide joints; centre sphere = twice the element radius
