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


# ---- the library's own MJCF compiler (csrc/sg_mjcf.cpp: sg_mjcf_compile / sg_model_compile) against mjcf.py ----
_SHELL_PARTS = """<mujoco>
  <compiler angle="radian" inertiafromgeom="auto" settotalmass="0.8"/>
  <option timestep="0.004" iterations="25" tolerance="1e-6" solver="PGS" cone="elliptic" gravity="0 0 -9.81"/>
  <size nconmax="300" njmax="900"/>
  <default>
    <geom friction="0.9 0.01 0.001" solimp="0.8 0.9"/>
    <joint damping="2.5" armature="0.01"/>
    <default class="finger">
      <geom type="box" density="800" condim="3"/>
      <joint type="hinge" limited="true" range="-0.4 0.6" solreflimit="0.01 1"/>
      <default class="tip">
        <geom rgba="1 0 0 1" margin="0.002"/>
      </default>
    </default>
  </default>
  <worldbody>
    <geom name="floor" type="plane" size="5 5 0.1" condim="1"/>
    <body name="base" pos="0 0 1.5" childclass="finger">
      <geom size="0.3 0.3 0.1"/>
      <site name="anchor" pos="0.1 0 -0.1"/>
      <body name="link1" pos="0.4 0 0" quat="0.9 0 0.1 0">
        <joint name="h1" axis="0 1 0" pos="-0.1 0 0"/>
        <joint name="h2" axis="1 0 0" range="-0.01 0.01"/>
        <geom name="l1" size="0.2 0.05 0.1" pos="0.1 0 0"/>
        <site name="s1" pos="0.25 0 0.02"/>
        <site name="imu" pos="0.1 0 0"/>
        <body name="link2" pos="0.4 0 0">
          <joint name="h3" axis="0 1 0" springref="0.1" stiffness="3"/>
          <geom name="l2" class="tip" size="0.15 0.05 0.1" pos="0.1 0 0" quat="1 0 0 0.2"/>
          <geom name="l2b" type="capsule" size="0.04 0.1" pos="0.2 0 0" mass="0.03"/>
        </body>
      </body>
    </body>
  </worldbody>
  <tendon>
    <spatial name="pull" stiffness="10" damping="0.5"><site site="anchor"/><site site="s1"/></spatial>
    <fixed name="couple"><joint joint="h1" coef="1"/><joint joint="h3" coef="-0.5"/></fixed>
  </tendon>
  <actuator><cylinder tendon="pull" timeconst="0.7" diameter="0.3" bias="0 -2 0"/></actuator>
  <sensor><accelerometer name="acc" site="imu"/><gyro name="gyr" site="imu"/></sensor>
</mujoco>
"""
_SHELL_MAIN = """<mujoco model="shell test">
  <!-- a scene of this repo's own making: every feature of the subset, small enough to read -->
  <include file="shell_parts.xml"/>
  <worldbody>
    <body pos="1.2 0.1 0.9">
      <composite prefix="OBJ" type="%s" count="%s" spacing="0.2">
        <geom type="capsule" size=".05 0.08" mass="0.002" contype="0" conaffinity="1"/>
        <skin texcoord="true"/>
        <joint kind="main" stiffness="400" damping="30" solreffix="-80 -8" solimpfix="0.9 0.96 0.00001 0.8 2"/>
        <tendon kind="main" stiffness="300" damping="20" solreffix="-50 -5"/>
      </composite>
    </body>
  </worldbody>
</mujoco>
"""


def _compare_models(py, nat, rtol=1e-11):
    for f in sg.Model._FIELDS_I32:
        np.testing.assert_array_equal(np.asarray(getattr(py, f)).ravel(), np.asarray(getattr(nat, f)).ravel(), err_msg=f)
    for f in sg.Model._FIELDS_F64:
        a, b = np.asarray(getattr(py, f), dtype=float).ravel(), np.asarray(getattr(nat, f), dtype=float).ravel()
        assert a.shape == b.shape, f
        np.testing.assert_allclose(b, a, rtol=rtol, atol=1e-16, err_msg=f)
    for k in ("body_names", "jnt_names", "geom_names", "site_names", "tendon_names", "sensor_names"):
        assert getattr(py, k) == getattr(nat, k), k
    assert (py.opt_timestep, py.opt_iterations, py.opt_tolerance, py.opt_impratio, py.nconmax, py.njmax) == (
        nat.opt_timestep, nat.opt_iterations, nat.opt_tolerance, nat.opt_impratio, nat.nconmax, nat.njmax)
    np.testing.assert_allclose(nat.meaninertia, py.meaninertia, rtol=rtol)
    np.testing.assert_array_equal(nat.opt_gravity, py.opt_gravity)


@pytest.mark.parametrize("ctype,count,neighbors", [("box", "3 4 3", True), ("ellipsoid", "4 4 5", True), ("cylinder", "4 5 3", True),
                                                   ("box", "2 2 2", False)])
def test_native_compiler_matches_python_on_own_scene(tmp_path, ctype, count, neighbors):
    """sg_mjcf_compile (C++, what sg_model_compile runs) against mjcf.py on a scene of this repo's own making that uses the whole
    subset: nested <include>, nested default classes + childclass, plane / box / capsule geoms with density or mass,
    settotalmass, two hinges on one body, springref, spatial and fixed tendons, a cylinder actuator given by diameter, both sensor
    kinds and a composite shell of each type with its fix, neighbour and tendon equalities.  Integers and names equal, reals to
    1e-11 relative (the two take the mass matrix's inverse in different orders)."""
    from softgrip_amd import native
    (tmp_path / "shell_parts.xml").write_text(_SHELL_PARTS)
    main = tmp_path / "main.xml"
    main.write_text(_SHELL_MAIN % (ctype, count))
    py = sg.Model.from_blob(sg.compile_mjcf(str(main), composite_neighbors=neighbors).to_blob())
    nat = sg.Model.from_blob(native.compile_mjcf_native(str(main), composite_neighbors=neighbors, implicit_tendon_damping=not neighbors))
    _compare_models(py, nat)
    assert nat.opt_implicit_tendon_damping == (0 if neighbors else 1) and py.opt_implicit_tendon_damping == 0
    n = [int(c) for c in count.split()]
    nshell = n[0] * n[1] * n[2] - max(n[0] - 2, 0) * max(n[1] - 2, 0) * max(n[2] - 2, 0)
    assert nat.nv == 3 + nshell and nat.ntendon == 3 and nat.tendon_names[0] == "OBJT" and nat.nu == 1 and nat.nsensordata == 6
    assert (nat.neq > nshell + 1) == neighbors and nat.eq_type[-1] == 3
    assert tuple(nat.eq_solref[0]) == (-80.0, -8.0) and tuple(nat.eq_solref[-1]) == (-50.0, -5.0) and nat.eq_solimp[0][3] == 0.8
    np.testing.assert_allclose(nat.body_mass.sum(), 0.8, rtol=1e-13)
    np.testing.assert_allclose(nat.actuator_gain[0], np.pi * 0.3 ** 2 / 4, rtol=1e-15)


@pytest.mark.parametrize("name", ["arm2", "boxbox", "capbox", "capbox_slide", "hinge_sensor", "limit", "slider", "tendon", "volume_tendon", "mini_gripper"])
def test_native_compiler_matches_python_on_test_scenes(name):
    from softgrip_amd import native
    from helpers import ROOT
    path = os.path.join(ROOT, "tests", "data", name + ".xml")
    _compare_models(sg.Model.from_blob(sg.compile_mjcf(path).to_blob()), sg.Model.from_blob(native.compile_mjcf_native(path)))


@pytest.mark.skipif(not os.path.exists(REF_XML % "softbox"), reason="reference MJCF only exists in the build container")
@pytest.mark.parametrize("scene", list(SIZES))
def test_native_compiler_matches_committed_blobs(scene):
    """the three reference scenes through the native compiler: the committed models/*.sgmodel (compiled by mjcf.py) field by field,
    and sg_model_compile accepts them (plan built: same nq / nelem as from the committed blob)"""
    import ctypes as C
    from softgrip_amd import native
    for suffix, nb in (("", True), ("_fix", False)):
        nat = sg.Model.from_blob(native.compile_mjcf_native(REF_XML % scene, composite_neighbors=nb))
        _compare_models(sg.load_model(model_path(scene + suffix)), nat)
    L = native.lib()
    ptr = C.c_void_p()
    native.check(L.sg_model_compile(os.fsencode(REF_XML % scene), 0, C.byref(ptr)))
    assert L.sg_model_nq(ptr) == SIZES[scene][1] and L.sg_model_nelem(ptr) == SIZES[scene][3]
    L.sg_model_destroy(ptr)


_COMPOSITE_XML = ("<mujoco><compiler angle='radian'/><option solver='PGS' cone='elliptic'/><worldbody><body pos='0 0 1'>"
                  "<composite type='box' count='%s' spacing='.3'><geom type='capsule' size='.02 .05' mass='.01'/></composite>"
                  "</body></worldbody></mujoco>")


def test_native_compiler_errors_are_reported(tmp_path):
    """same refusals as mjcf.py, as SG_ERR_MODEL with a message (no exception crosses the C ABI)"""
    from softgrip_amd import native
    cases = {"free.xml": ("<mujoco><compiler angle='radian'/><option solver='PGS' cone='elliptic'/><worldbody><body><body><freejoint/>"
                          "<geom type='sphere' size='1'/></body></body></worldbody></mujoco>", "free joint must be the only joint of a child of the world"),
             "newton.xml": ("<mujoco><compiler angle='radian'/><worldbody/></mujoco>", "solver='PGS'"),
             "degree.xml": ("<mujoco><option solver='PGS' cone='elliptic'/><compiler angle='degree'/></mujoco>", "radian"),
             "broken.xml": ("<mujoco><worldbody><body></worldbody></mujoco>", "XML error"),
             "mesh.xml": ("<mujoco><compiler angle='radian'/><option solver='PGS' cone='elliptic'/><worldbody><geom type='mesh'/>"
                          "</worldbody></mujoco>", "unsupported geom type"),
             # an include cycle used to recurse until the stack overflowed, a huge composite count hung the compile
             "cycle.xml": ("<mujoco><include file='cycle.xml'/></mujoco>", "include cycle"),
             "cycle_a.xml": ("<mujoco><include file='cycle_b.xml'/></mujoco>", "include cycle"),
             "huge.xml": (_COMPOSITE_XML % "2000 2000 2000", r"whole numbers in \[2, 64\]"),
             "huge2.xml": (_COMPOSITE_XML % "1e300 4 4", r"whole numbers in \[2, 64\]"),
             "frac.xml": (_COMPOSITE_XML % "4.5 4 4", r"whole numbers in \[2, 64\]"),
             "many.xml": (_COMPOSITE_XML % "20 20 20", "at most 256")}
    (tmp_path / "cycle_b.xml").write_text("<mujoco><include file='cycle_a.xml'/></mujoco>")
    for fn, (xml, what) in cases.items():
        (tmp_path / fn).write_text(xml)
        with pytest.raises(native.SoftgripError, match=what) as ei:
            native.compile_mjcf_native(str(tmp_path / fn))
        assert ei.value.code == -2
        if fn.startswith(("cycle", "huge", "frac", "many")):   # the Python compiler refuses the same files with the same words
            with pytest.raises(ValueError, match=what):
                sg.compile_mjcf(str(tmp_path / fn))
    with pytest.raises(native.SoftgripError, match="cannot open"):
        native.compile_mjcf_native(str(tmp_path / "missing.xml"))


def test_own_scene_in_the_plan_class_is_accepted_by_sg_model_compile():
    """tests/data/mini_gripper.xml: a scene of this repo's own making inside the kernels' plan class (two 4-dof finger chains, a
    3 x 4 x 3 shell, other dimensions / masses / time step than the reference's) goes XML -> native compiler -> plan on the CPU;
    the GPU tests run it (the reference's XML files do not exist on the GPU box)."""
    import ctypes as C
    from softgrip_amd import native
    from helpers import ROOT
    path = os.path.join(ROOT, "tests", "data", "mini_gripper.xml")
    L = native.lib()
    for flags, neq in ((0, 34 + 64 + 1), (native.SG_COMPILE_NO_NEIGHBORS, 35), (native.SG_COMPILE_IMPLICIT_TENDON_DAMPER, 99)):
        ptr = C.c_void_p()
        native.check(L.sg_model_compile(os.fsencode(path), flags, C.byref(ptr)))
        assert (L.sg_model_nq(ptr), L.sg_model_nelem(ptr), L.sg_model_nu(ptr), L.sg_model_nsensordata(ptr)) == (42, 34, 2, 12)
        L.sg_model_destroy(ptr)
        m = sg.Model.from_blob(native.compile_mjcf_native(path, composite_neighbors=not (flags & 1), implicit_tendon_damping=bool(flags & 2)))
        assert m.neq == neq and m.opt_timestep == 0.004 and m.opt_iterations == 20


def _random_scene(rng):
    """a random scene inside the MJCF subset: a tree of bodies up to three deep with boxes / capsules / spheres (mass or density),
    hinges and slides (some limited, damped, with armature / springref), sites, optionally a composite shell, spatial and fixed
    tendons over what exists, a cylinder actuator and sensors"""
    f = lambda lo, hi: "%.6g" % rng.uniform(lo, hi)  # noqa: E731
    names = {"joint": [], "site": [], "hinge": []}
    out = ["<mujoco>", "<compiler angle=\"radian\" settotalmass=\"%s\"/>" % f(0.2, 3) if rng.rand() < 0.5 else "<compiler angle=\"radian\"/>",
           "<option timestep=\"%s\" solver=\"PGS\" cone=\"elliptic\" iterations=\"%d\"/>" % (f(0.001, 0.01), rng.randint(5, 50)),
           "<default><geom friction=\"%s %s\"/><default class=\"c1\"><joint damping=\"%s\"/><geom density=\"%s\"/></default></default>" % (
               f(0.5, 1.5), f(0.001, 0.01), f(0, 3), f(100, 2000)), "<worldbody>", "<geom type=\"plane\" size=\"1 1 1\"/>"]

    def body(depth):
        out.append("<body pos=\"%s %s %s\" quat=\"%s %s %s %s\"%s>" % (f(-1, 1), f(-1, 1), f(0, 2), f(0.5, 1), f(-.5, .5), f(-.5, .5), f(-.5, .5),
                                                                    " childclass=\"c1\"" if rng.rand() < 0.3 else ""))
        for _ in range(rng.randint(0, 3)):
            jn = "j%d" % len(names["joint"])
            typ = "hinge" if rng.rand() < 0.7 else "slide"
            out.append("<joint name=\"%s\" type=\"%s\" axis=\"%s %s %s\" pos=\"%s 0 0\"%s%s armature=\"%s\" springref=\"%s\" stiffness=\"%s\"/>" % (
                jn, typ, f(-1, 1), f(-1, 1), f(0.2, 1), f(-.2, .2), " limited=\"true\" range=\"%s %s\"" % (f(-1, 0), f(0, 1)) if rng.rand() < 0.5 else "",
                " damping=\"%s\"" % f(0, 2) if rng.rand() < 0.5 else "", f(0, 0.05), f(-.1, .1), f(0, 5)))
            names["joint"].append(jn)
            if typ == "hinge":
                names["hinge"].append(jn)
        for _ in range(rng.randint(1, 3)):
            typ = ["box", "capsule", "sphere"][rng.randint(3)]
            size = {"box": "%s %s %s" % (f(.05, .3), f(.05, .3), f(.05, .3)), "capsule": "%s %s" % (f(.03, .1), f(.05, .3)), "sphere": f(.05, .2)}[typ]
            out.append("<geom type=\"%s\" size=\"%s\" pos=\"%s %s %s\" quat=\"%s %s 0 %s\"%s/>" % (
                typ, size, f(-.3, .3), f(-.3, .3), f(-.3, .3), f(.5, 1), f(-.5, .5), f(-.5, .5), " mass=\"%s\"" % f(0.01, 0.5) if rng.rand() < 0.5 else ""))
        for _ in range(rng.randint(0, 2)):
            sn = "s%d" % len(names["site"])
            out.append("<site name=\"%s\" pos=\"%s %s %s\"/>" % (sn, f(-.3, .3), f(-.3, .3), f(-.3, .3)))
            names["site"].append(sn)
        if depth < 3:
            for _ in range(rng.randint(0, 2 if depth else 3)):
                body(depth + 1)
        out.append("</body>")
    for _ in range(rng.randint(1, 3)):
        body(1)
    if rng.rand() < 0.6:
        ctype = ["box", "ellipsoid", "cylinder"][rng.randint(3)]
        out.append("<body pos=\"2 0 1\"><composite prefix=\"S\" type=\"%s\" count=\"%d %d %d\" spacing=\"%s\"><geom type=\"capsule\" size=\".03 .04\" mass=\"%s\"/>"
                   "<joint kind=\"main\" stiffness=\"%s\" damping=\"%s\" solreffix=\"-100 -10\"/><tendon kind=\"main\" damping=\"%s\" solimpfix=\"0.8 0.9 0.01\"/></composite></body>" % (
                       ctype, rng.randint(2, 5), rng.randint(2, 5), rng.randint(2, 5), f(.1, .3), f(.001, .01), f(10, 900), f(1, 90), f(0, 5)))
    out.append("</worldbody>")
    ten = []
    if len(names["site"]) >= 2:
        ten.append("<spatial name=\"sp\" stiffness=\"%s\" damping=\"%s\"><site site=\"%s\"/><site site=\"%s\"/></spatial>" % (f(0, 50), f(0, 2), names["site"][0], names["site"][-1]))
    if len(names["joint"]) >= 2:
        ten.append("<fixed name=\"fx\"><joint joint=\"%s\" coef=\"%s\"/><joint joint=\"%s\" coef=\"%s\"/></fixed>" % (names["joint"][0], f(-2, 2), names["joint"][-1], f(-2, 2)))
    if ten:
        out.append("<tendon>" + "".join(ten) + "</tendon>")
        out.append("<actuator><cylinder tendon=\"%s\" timeconst=\"%s\" area=\"%s\"/></actuator>" % ("sp" if len(names["site"]) >= 2 else "fx", f(.1, 2), f(1, 500)))
    if names["site"]:
        out.append("<sensor><accelerometer site=\"%s\"/><gyro name=\"g\" site=\"%s\"/></sensor>" % (names["site"][0], names["site"][-1]))
    out.append("</mujoco>")
    return "\n".join(out)


def test_native_compiler_matches_python_on_random_scenes(tmp_path):
    """40 random scenes inside the subset (seeded): the two compilers agree field by field on every one, or refuse the same ones
    (a random scene may have a moving body without mass)"""
    from softgrip_amd import native
    rng = np.random.RandomState(20260)
    compiled = 0
    for i in range(40):
        path = tmp_path / ("r%d.xml" % i)
        path.write_text(_random_scene(rng))
        try:
            py = sg.Model.from_blob(sg.compile_mjcf(str(path)).to_blob())
        except (ValueError, np.linalg.LinAlgError) as e:
            with pytest.raises(native.SoftgripError):
                native.compile_mjcf_native(str(path))
            continue
        _compare_models(py, sg.Model.from_blob(native.compile_mjcf_native(str(path))), rtol=1e-9)
        compiled += 1
    assert compiled >= 25
