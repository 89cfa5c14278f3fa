"""The TREE pipeline's source (csrc/sg_tree.h: grippers outside the two-finger class) compiled for the host
(tests/emu/sg_tree_emu.cpp: a parallel loop becomes a serial loop) against the oracle: the reference's four-finger gripper
(soft_grip_four_fingers.xml) and, as a cross-check of the general code on models the fast kernels also run, the two-finger scenes."""
import os

import numpy as np
import pytest

import softgrip_amd as sg
from helpers import JOINT_IDS, TENDON_IDS, TreeEmu, model_path, oracle_sim, random_gripper_xml
from softgrip_amd.create_dataset import episode_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FF_JOINTS = list(range(65, 283))      # reference environment/manenv.py:11 (commented four-finger ids): the ball's sliders


def _pair(scene, k, damper, jids, tids):
    m = sg.load_model(model_path(scene), damper)
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = k
    s.tendon_stiffness[tids] = k
    s.reset(); s.forward(); s.step()
    e = TreeEmu(m)
    e.set_stiffness(k, jids, tids)
    assert e.reset(1) == 0
    return m, e, s


@pytest.mark.parametrize("scene,damper,tol", [("softbox_fix", None, 1e-9), ("softball_fix", "implicit", 1e-8)])
def test_two_finger_scenes_free_running(scene, damper, tol):
    """the general code on the two-finger class: a whole squeeze episode free-running, contacts, row and sweep counts exact"""
    m, e, s = _pair(scene, 903.6948543200572, damper, JOINT_IDS, TENDON_IDS)
    np.testing.assert_allclose(e.qpos, s.qpos, atol=1e-14)
    worst = 0.0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        worst = max(worst, np.abs(e.sensordata - s.sensordata).max())
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
    assert worst < tol
    np.testing.assert_allclose(e.qpos, s.qpos, atol=tol)
    np.testing.assert_allclose(e.act, s.act, atol=1e-14)


def test_four_finger_episode_step_by_step():
    """soft_grip_four_fingers.xml closing on the ball (4 chains of 16 / 17 dofs, 64 boxes, 8-site tendons, 218 limited sliders:
    437 - 531 rows): along the oracle's trajectory -- the emulation re-seated on the oracle's state after every env step, which this
    scene needs (218 active limit rows make it amplify round-off: free-running the two part by 1e-5 after 30 steps) -- sensors of all
    24 channels, contact, row and sweep counts and the touch bits at every step"""
    m, e, s = _pair("fourfinger_softball_fix", 700.0, "implicit", FF_JOINTS, [0])
    boxes = [g for g in range(m.ngeom) if m.geom_type[g] == 6 and m.body_weldid[m.geom_bodyid[g]] != 0]
    worst, touched = 0.0, 0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.warm[:] = s.qacc_warmstart; e.act[:] = s.act
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        worst = max(worst, np.abs(e.sensordata - s.sensordata).max())
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
        np.testing.assert_allclose(e.qvel, s.qvel, atol=1e-10)
        want = 0
        for cc in s.contacts():
            for g, o in ((cc["geom1"], cc["geom2"]), (cc["geom2"], cc["geom1"])):
                if g in boxes and "OBJ" in (m.geom_names[o] or ""):
                    want |= 1 << boxes.index(g)
        assert e.touch_bits() == want, t
        touched |= want
    assert worst < 1e-9
    assert bin(touched).count("1") >= 8          # boxes of all four fingers reach the ball
    for f in range(4):
        assert touched >> (16 * f) & 0xFFFF, f


def test_four_finger_first_steps_free_running():
    """free-running until the round-off amplification sets in: the first 15 env steps (105 substeps) at 1e-9"""
    m, e, s = _pair("fourfinger_softball_fix", 1100.0, "implicit", FF_JOINTS, [0])
    for t, c in enumerate(episode_schedule()[:15]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        assert np.abs(e.sensordata - s.sensordata).max() < 1e-9
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
    np.testing.assert_allclose(e.qpos, s.qpos, atol=1e-11)


@pytest.mark.parametrize("scene,damper,n_steps", [("softbox", None, 110), ("softball", "implicit", 40)])
def test_default_models_with_neighbour_equalities(scene, damper, n_steps):
    """the composite as MuJoCo's documentation describes it (fix rows AND neighbour equalities: 327 / 651 equality rows) in the tree
    pipeline: the equality BLOCKS [fix_e, e's neighbour rows] run by the plan's list schedule, 64 blocks a round, against the oracle's
    sequential sweep -- along the oracle's trajectory (these models amplify round-off: DESIGN.md 2), sensors, contact, row and sweep
    counts at every env step"""
    m, e, s = _pair(scene, 700.0, damper, JOINT_IDS, TENDON_IDS)
    assert m.neq in (327, 651)
    worst = 0.0
    for t, c in enumerate(episode_schedule()[:n_steps]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.warm[:] = s.qacc_warmstart; e.act[:] = s.act
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
        worst = max(worst, np.abs(e.sensordata - s.sensordata).max())
        np.testing.assert_allclose(e.qvel, s.qvel, atol=1e-8)
    assert worst < 1e-7, worst


def test_default_softbox_first_steps_free_running():
    """... and free-running until the round-off amplification of the neighbour-row model sets in: 40 env steps at 1e-9"""
    m, e, s = _pair("softbox", 903.6948543200572, None, JOINT_IDS, TENDON_IDS)
    for t, c in enumerate(episode_schedule()[:40]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
        assert np.abs(e.sensordata - s.sensordata).max() < 1e-9
    np.testing.assert_allclose(e.qpos, s.qpos, atol=1e-10)


def test_tree_plan_refuses_what_it_cannot_run():
    m = sg.load_model(model_path("fourfinger_softball_fix"), "implicit")
    m.jnt_type = m.jnt_type.copy()
    m.jnt_type[3] = 2                      # a slide joint in a finger chain
    with pytest.raises(RuntimeError) as ei:
        TreeEmu(m)
    assert "hinge" in str(ei.value)


def test_free_ball_with_neighbour_equalities():
    """models/freeball.sgmodel: the free-floating ball as MuJoCo's documentation describes the composite (651 equality rows) -- the
    equality blocks [fix_e, e's neighbour rows] swept one after the other with the body's acceleration carried along, a neighbour row
    pushing the body through both of its sliders -- along the oracle's trajectory for 100 env steps: sensors, contact, row and sweep
    counts"""
    m = sg.load_model(model_path("freeball"), "implicit")
    assert m.neq == 651 and m.has_free_joint
    jids = list(range(9, 227))
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = 700.0
    s.tendon_stiffness[0] = 700.0
    s.reset(); s.forward(); s.step()
    e = TreeEmu(m)
    e.set_stiffness(700.0, jids, [0])
    assert e.reset(1) == 0
    np.testing.assert_allclose(e.qvel, s.qvel, atol=1e-12)
    worst = 0.0
    for t, c in enumerate(episode_schedule()[:100]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.warm[:] = s.qacc_warmstart; e.act[:] = s.act
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
        worst = max(worst, np.abs(e.sensordata - s.sensordata).max())
    assert worst < 1e-8, worst


# ---- every pair kind of the candidate-pair table, and the serial contact list (a slider under both fingers, both chains in one row) ----
@pytest.mark.parametrize("kind", ["thin", "fingers", "stop", "shelf", "rest", "sledge"])
def test_special_scenes_pair_kinds_and_the_serial_list(tmp_path, kind):
    """the scenes of tests/test_emu_vs_oracle.py that leave the fast kernels' two pair kinds, on the tree pipeline's source: box - box
    (finger x finger: both chains in one row; finger x static block), plane - box, plane - capsule, static box - capsule, and one
    slider carrying contacts of both fingers -- the cases in which the sweep runs the contacts as one serial list instead of one
    stream per chain.  Sensors, contact, row and sweep counts per env step; the oracle's contact list confirms the special pair."""
    from test_emu_vs_oracle import elements_touching_both_fingers, general_path_scene, special_contacts, thin_shell_scene
    path = thin_shell_scene(tmp_path / "thin.xml") if kind == "thin" else general_path_scene(kind, tmp_path / (kind + ".xml"))
    m = sg.compile_mjcf(path, composite_neighbors=False)
    jids = list(range(8, m.nv))
    s = oracle_sim(m)
    k = 640.0
    s.jnt_stiffness[jids] = k
    s.tendon_stiffness[0] = k
    s.reset(); s.forward(); s.step()
    e = TreeEmu(m)
    e.set_stiffness(k, jids, [0])
    assert e.reset(1) == 0
    worst, special = 0.0, 0
    for t, c in enumerate(episode_schedule()[:150]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0, t
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
        special += elements_touching_both_fingers(m, s.contacts()) if kind == "thin" else special_contacts(m, s.contacts(), kind)
        worst = max(worst, np.abs(e.sensordata - s.sensordata).max())
        if kind in ("shelf", "rest", "sledge"):   # dozens of standing contacts from the first step on: these scenes amplify round-off
            e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.warm[:] = s.qacc_warmstart; e.act[:] = s.act
    assert special > (40 if kind == "thin" else 100), special
    assert worst < 1e-7, worst


def test_pair_with_other_contact_parameters_is_flagged_not_dropped(tmp_path):
    """a candidate pair whose mixed contact parameters differ from the finger / object pairs' (the plan keeps one set) cannot
    become rows in the tree pipeline -- it must then raise SG_FLAG_UNSUPPORTED_PAIR once it is within reach, not vanish: the "stop"
    scene's static block with a friction of its own.  No flag while the finger is away; flagged no later than the substep in which the
    oracle (which mixes parameters per pair) reports a contact on the block."""
    from test_emu_vs_oracle import general_path_scene, special_contacts
    path = general_path_scene("stop", tmp_path / "stop.xml")
    x = open(path).read().replace('<geom name="stop" class="link"', '<geom name="stop" class="link" friction="2.5 0.005 0.0001"')
    assert 'friction="2.5' in x
    open(path, "w").write(x)
    m = sg.compile_mjcf(path, composite_neighbors=False)
    jids = list(range(8, m.nv))
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = 640.0
    s.tendon_stiffness[0] = 640.0
    s.reset(); s.forward(); s.step()
    e = TreeEmu(m)
    e.set_stiffness(640.0, jids, [0])
    assert e.reset(1) == 0
    first_flag = first_contact = None
    for t, c in enumerate(episode_schedule()[:120]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        fl = e.step(7)
        assert fl in (0, 32), (t, fl)                       # SG_FLAG_UNSUPPORTED_PAIR and nothing else
        if fl and first_flag is None:
            first_flag = t
        if special_contacts(m, s.contacts(), "stop") and first_contact is None:
            first_contact = t
        if first_flag is not None and first_contact is not None:
            break
        e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.warm[:] = s.qacc_warmstart; e.act[:] = s.act   # (the flagged env follows the oracle)
    assert first_contact is not None and first_flag is not None, (first_flag, first_contact)
    assert 40 < first_flag <= first_contact, (first_flag, first_contact)   # not during the idle phase, not after the contact


# ---- the free object (soft_experiments_softball.xml:8): the composite's elements on a body with a free joint ----
def _free_mini(tmp_path, far):
    """the own scene with a <freejoint/> on the object's body; far: lifted out of the gripper's reach and tilted (no contacts ever)"""
    import os
    from helpers import ROOT
    x = open(os.path.join(ROOT, "tests", "data", "mini_gripper.xml")).read()
    # near: the gripper lowered to the ground, the object resting on it between the fingers (the "rest" scene of test_emu_vs_oracle.py)
    new = '<body pos="1.15 0 6.0" quat="0.9 0.2 -0.1 0.3">' if far else '<body pos="1.15 0 0.236">'
    x = x.replace('<body pos="1.15 0 1.0">\n      <composite', new + '\n      <freejoint/>\n      <composite')
    if not far:
        x = x.replace('<body pos="0 0 1.0">', '<body pos="0 0 0.3">')
    path = str(tmp_path / ("free_far.xml" if far else "free_near.xml"))
    with open(path, "w") as f:
        f.write(x)
    return sg.compile_mjcf(path, composite_neighbors=False)


def test_free_object_tumbling_without_contacts(tmp_path):
    """a composite on a free body, thrown, spinning, its sliders moving: the object block of the tree pipeline (arrow-shaped mass matrix
    through a 6 x 6 Schur complement, star-shaped RNE, the joint-fix rows one after the other, quaternion integration) against the
    oracle's general tree algorithms, free-running for 150 substeps: 1e-12"""
    m = _free_mini(tmp_path, far=True)
    assert (m.nq, m.nv, m.njnt) == (49, 48, 43)
    fj = int(np.flatnonzero(m.jnt_type == 0)[0])
    jids = list(range(fj + 1, m.njnt))
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = 640.0
    s.tendon_stiffness[0] = 640.0
    e = TreeEmu(m)
    e.set_stiffness(640.0, jids, [0])
    s.reset()
    s.qvel[fj:fj + 3] = [0.3, -0.2, 0.5]
    s.qvel[fj + 3:fj + 6] = [1.0, -2.0, 0.7]
    s.qvel[fj + 6:] = 0.05 * np.random.RandomState(0).randn(m.nv - fj - 6)
    s.qpos[fj + 7:] += 0.01 * np.random.RandomState(1).randn(m.nq - fj - 7)
    e.qpos[:] = s.qpos; e.qvel[:] = s.qvel
    for t in range(150):
        assert s.step() == 0 and e.step(1) == 0
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
    np.testing.assert_allclose(e.qpos, s.qpos, atol=1e-12)
    np.testing.assert_allclose(e.qvel, s.qvel, atol=1e-12)
    assert abs(np.linalg.norm(e.qpos[fj + 3:fj + 7]) - 1) < 1e-14 and np.abs(e.qpos[fj + 3:fj + 7] - m.qpos0[fj + 3:fj + 7]).max() > 0.1


def test_free_object_squeezed_by_the_own_gripper(tmp_path):
    """the same object resting on the ground between the fingers (plane - capsule contacts with the capsule axis as tangent hint),
    squeezed and pushed around by them -- every contact carries six object columns and all of them go through the body; free-running
    over the schedule, counts exact"""
    m = _free_mini(tmp_path, far=False)
    fj = int(np.flatnonzero(m.jnt_type == 0)[0])
    jids = list(range(fj + 1, m.njnt))
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = 640.0
    s.tendon_stiffness[0] = 640.0
    s.reset(); s.forward(); s.step()
    e = TreeEmu(m)
    e.set_stiffness(640.0, jids, [0])
    assert e.reset(1) == 0
    worst, ground, fingers = 0.0, 0, 0
    for t, c in enumerate(episode_schedule()[:140]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
        worst = max(worst, np.abs(e.sensordata - s.sensordata).max())
        names = [(m.geom_names[cc["geom1"]][:4], m.geom_names[cc["geom2"]][:4]) for cc in s.contacts()]
        ground += sum(1 for n in names if n[0] == "grou")
        fingers += sum(1 for n in names if n[1][:1] == "f" or n[0][:1] == "f")
    assert ground > 100 and fingers > 100 and worst < 1e-7, (ground, fingers, worst)
    np.testing.assert_allclose(e.qpos, s.qpos, atol=1e-7)


@pytest.mark.skipif(not __import__("os").path.exists("/root/reference/data/gripper/soft_experiments_softball.xml"), reason="needs the reference's MJCF files (build container only)")
def test_reference_free_ball_scene_free_running():
    """soft_experiments_softball.xml as the reference ships it (8 gripper dofs + 6 + 218 sliders, the shell inside the fingers at the
    start, D5 damper): the whole squeeze episode free-running against the oracle -- sensors 1e-7, contact / row / sweep counts exact"""
    m = sg.compile_mjcf("/root/reference/data/gripper/soft_experiments_softball.xml", composite_neighbors=False)
    m.opt_implicit_tendon_damping = 1
    jids = list(range(9, 227))
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = 700.0
    s.tendon_stiffness[0] = 700.0
    s.reset(); s.forward(); s.step()
    e = TreeEmu(m)
    e.set_stiffness(700.0, jids, [0])
    assert e.reset(1) == 0
    worst = 0.0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        assert e.step(7) == 0
        assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), t
        worst = max(worst, np.abs(e.sensordata - s.sensordata).max())
    assert worst < 1e-7, worst
    np.testing.assert_allclose(e.qpos, s.qpos, atol=1e-8)


# ---- fuzzing the tree class: random grippers against the oracle ----
@pytest.mark.parametrize("free,neighbors", [(False, False), (False, True), (True, False), (True, True)])
def test_random_grippers_step_by_step(tmp_path, free, neighbors):
    """16 seeded random grippers of the tree class per variant (object fixed / on a free joint, fix rows only / with the composite's
    neighbour equalities), 40 env steps of the squeeze schedule each, along the oracle's trajectory, re-seated after every SUBSTEP: a
    random scene need not be a stable one (seed 20's third gripper multiplies a perturbation by 5 every substep until it is flagged:
    1e-15 in a position is 1e-6 in an accelerometer seven substeps later, on both sides alike), so the comparison is of one substep's
    arithmetic at a time -- sensors, positions, velocities, contact / row / sweep counts.  A scene whose random geometry leaves the
    class (too many contacts at once) or blows up may be flagged -- by BOTH sides, in the same substep, or not at all."""
    rng = np.random.RandomState(20 + 2 * int(free) + int(neighbors))
    touched = ran = 0
    for i in range(16):
        path = tmp_path / ("g%d.xml" % i)
        path.write_text(random_gripper_xml(rng, free))
        m = sg.compile_mjcf(str(path), composite_neighbors=neighbors)
        nchain = int(np.flatnonzero(m.jnt_type != 3)[0])          # joints before the first non-hinge one: the fingers'
        jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
        s = oracle_sim(m)
        k = rng.uniform(300, 1400)
        s.jnt_stiffness[jids] = k
        s.tendon_stiffness[0] = k
        s.reset(); s.forward(); s.step()
        e = TreeEmu(m)
        e.set_stiffness(k, jids, [0])
        assert e.reset(1) == 0, i
        np.testing.assert_allclose(e.qvel, s.qvel, atol=1e-9, err_msg=str(i))
        most, flagged, errs = 0, False, [0.0]
        for t, c in enumerate(episode_schedule()[:40]):
            if c is not None:
                e.ctrl[:] = c
                s.ctrl[:] = c
            for j in range(7):
                e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.warm[:] = s.qacc_warmstart; e.act[:] = s.act
                w, f = s.step(), e.step(1)
                if w or f:
                    # both flag the scene, or neither -- but for the kernel's own capacity (SGT_MAXCON = 128 contacts an env;
                    # the oracle holds the model's nconmax = 300), which it reports as CONTACTFULL
                    assert (w and f) or (f == 8 and s.ncon > 128), (i, t, j, w, f, s.ncon)
                    flagged = True
                    break
                assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), (i, t, j)
                scale = 1.0 + np.abs(s.sensordata).max() + np.abs(s.qacc_warmstart).max()   # (an accelerometer sample is a sum of |qacc| r terms that may cancel)
                err = max(np.abs(e.sensordata - s.sensordata).max(), np.abs(e.qvel - s.qvel).max(), 100 * np.abs(e.qpos - s.qpos).max()) / scale
                assert err < 1e-7, (i, t, j, err)     # (a sweep that stops at the iteration cap next to a cone boundary: 2e-9 seen once)
                errs.append(err)
                most = max(most, s.ncon)
            if flagged:
                break
        else:
            ran += 1
        assert np.percentile(errs, 95) < 1e-10, (i, np.percentile(errs, 95))   # the rule: one substep's arithmetic agrees to 1e-10 of the signal
        touched += most > 0
    assert ran >= 12 and touched >= 8, (ran, touched)


def test_oracle_same_chain_contact_block_against_numpy(tmp_path):
    """A contact between two boxes of ONE finger (links 2 and 5 of a curled chain) names the dofs they share twice in the oracle's
    sparse row; the fuzz above found the oracle's A = J M^-1 J' + R using only one of the two entries (the kernel source had it
    right).  The oracle's block is checked against dense numpy algebra on the host compiler's mass matrix and Jacobians."""
    rng = np.random.RandomState(20)
    for _ in range(7):
        xml = random_gripper_xml(rng, False)
        k = rng.uniform(300, 1400)
    path = tmp_path / "g.xml"
    path.write_text(xml)
    m = sg.compile_mjcf(str(path), composite_neighbors=False)
    nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
    jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = k
    s.tendon_stiffness[0] = k
    s.reset(); s.forward(); s.step()
    sched = episode_schedule()
    seen = 0
    for t in range(3):
        if sched[t] is not None:
            s.ctrl[:] = sched[t]
        for _ in range(7):
            q = s.qpos.copy()
            assert s.step() == 0
            AR, b, ty, ids, mu = s.constraint_problem()
            for ci, cd in enumerate(s.contacts()):
                b1, b2 = int(m.geom_bodyid[cd["geom1"]]), int(m.geom_bodyid[cd["geom2"]])
                if not (0 < b1 < b2 and b2 < nchain):
                    continue
                anc = b2
                while anc > b1:
                    anc = int(m.body_parentid[anc])
                if anc != b1:
                    continue                 # not on one chain
                rows = np.flatnonzero((ids == ci) & (ty == ty.max()))
                M, _ = m.mass_matrix(q)
                kin = m.kinematics(q)
                F = cd["frame"].reshape(3, 3)
                J = F @ (m._jac_point(kin, b2, cd["pos"])[0] - m._jac_point(kin, b1, cd["pos"])[0])
                A = J @ np.linalg.solve(M, J.T)
                got = AR[np.ix_(rows, rows)]
                assert np.abs(got - np.diag(np.diag(got)) - (A - np.diag(np.diag(A)))).max() < 1e-9 * np.abs(A).max()
                R = np.diag(got) - np.diag(A)
                assert (R > 0).all() and np.ptp(R) < 1e-9 * R[0]     # what is left on the diagonal is the regulariser (impratio 1, equal friction)
                seen += 1
    assert seen >= 4


@pytest.mark.parametrize("links,hinges,stride,seed", [(7, 3, 24, 202), (6, 3, 20, 204), (2, 2, 4, 206)])
def test_chain_capacities_and_kernel_instantiations(tmp_path, links, hinges, stride, seed):
    """the kernel is instantiated for three unroll capacities of the per-chain loops (8, 20, 24: csrc/sg_tree.hip) and the library picks
    the smallest one that holds the model's padded chain stride; the random scenes above stop at 15 dofs a chain, the four-finger
    gripper has 17 -- here chains of 21 (stride 24: the full capacity), 18 (20) and 4 dofs (4) step along the oracle, substep by substep"""
    rng = np.random.RandomState(seed)        # (seeds whose scenes stay within the kernel's 128 contacts)
    path = tmp_path / "g.xml"
    path.write_text(random_gripper_xml(rng, False, links=links, hinges=hinges, fingers=2))
    m = sg.compile_mjcf(str(path), composite_neighbors=False)
    nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
    assert nchain == 2 * links * hinges and -(-links * hinges // 4) * 4 == stride
    jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
    s = oracle_sim(m)
    s.jnt_stiffness[jids] = 700.0
    s.tendon_stiffness[0] = 700.0
    s.reset(); s.forward(); s.step()
    e = TreeEmu(m)
    e.set_stiffness(700.0, jids, [0])
    assert e.reset(1) == 0
    most = 0
    for t, c in enumerate(episode_schedule()[:30]):
        if c is not None:
            e.ctrl[:] = c
            s.ctrl[:] = c
        for j in range(7):
            e.qpos[:] = s.qpos; e.qvel[:] = s.qvel; e.warm[:] = s.qacc_warmstart; e.act[:] = s.act
            w, f = s.step(), e.step(1)
            assert w == 0 and f == 0, (t, j, w, f)
            assert (e.ncon, e.nefc, e.iters) == (s.ncon, s.nefc, s.solver_iter), (t, j)
            scale = 1.0 + np.abs(s.sensordata).max() + np.abs(s.qacc_warmstart).max()
            assert max(np.abs(e.sensordata - s.sensordata).max(), np.abs(e.qvel - s.qvel).max()) < 1e-8 * scale, (t, j)
            most = max(most, s.ncon)
    assert most > 0


def test_checking_layout_gives_the_same_numbers(tmp_path):
    """scripts/sanitize/run_emu_oracle.sh runs the emulation in its checking layout (SGT_EMU_SEPARATE: every array of the env's LDS block
    and work space a heap block of its own, so that ASan sees one-past-the-end of ANY array).  This keeps that layout compiling and
    equal to the plain one: the free ball's first env steps, bit for bit."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "libsgtreeemu_sep.so")
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-std=c++17", "-Wno-unknown-pragmas", "-DSGT_EMU_SEPARATE", "-o", so,
                           os.path.join(ROOT, "tests", "emu", "sg_tree_emu.cpp"), os.path.join(ROOT, "soft-grip_amd", "csrc", "sg_plan.cpp")])
    m = sg.load_model(model_path("freeball_fix"), "implicit")
    plain = TreeEmu(m)
    real = C.CDLL
    try:
        C.CDLL = lambda p, *a, **k: real(so if str(p).endswith("libsgtreeemu.so") else p, *a, **k)
        sep = TreeEmu(m)
    finally:
        C.CDLL = real
    for e in (plain, sep):
        e.set_stiffness(700.0, list(range(9, 227)), [0])
        e.reset(1)
    for t_ in range(3):
        for e in (plain, sep):
            e.ctrl[:] = 0.0 if t_ < 1 else 1.0
            e.step(7)
        assert (plain.ncon, plain.nefc, plain.iters, plain.flags) == (sep.ncon, sep.nefc, sep.iters, sep.flags)
        np.testing.assert_array_equal(plain.sensordata, sep.sensordata)
        np.testing.assert_array_equal(plain.qpos, sep.qpos)


def test_block_update_with_precomputed_constants_is_the_generic_update():
    """sg_math.h contact_block_update_pre (what the tree sweep runs since r05: the friction block's inverse and eigen-decomposition built
    once per contact, mju_QCQP2's Newton iteration in eigen-coordinates) against contact_block_update (the oracle's formula) on 200 000 random 3 x 3
    blocks -- anisotropic friction, contacts without a normal force yet, nearly singular friction blocks: forces and cost change equal
    to 1e-9 relative, a third of the updates ending ON the cone"""
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libsgtreeemu.so"))
    L.temu_block_update_check.argtypes = [C.c_int, C.c_uint] + [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_int)] * 2
    mdf, mrel, ns, nr = C.c_double(), C.c_double(), C.c_int(), C.c_int()
    bad = L.temu_block_update_check(200000, 7, C.byref(mdf), C.byref(mrel), C.byref(ns), C.byref(nr))
    assert bad == 0 and mrel.value < 1e-9, (bad, mdf.value, mrel.value)
    assert ns.value > 20000 and nr.value < 150000, (ns.value, nr.value)
