"""GPU parity of the TREE pipeline (csrc/sg_tree.h / sg_tree.hip) through the C ABI: the reference's four-finger
gripper (soft_grip_four_fingers.xml on the ball: 283 dofs, 64 finger boxes, 24 sensor channels) against the oracle, and the same
kernel on the two-finger scenes against the oracle and against the fast kernels."""
import numpy as np
import pytest

import softgrip_amd as sg
from helpers import JOINT_IDS, TENDON_IDS, model_path, oracle_sim
from softgrip_amd.create_dataset import episode_schedule

pytestmark = pytest.mark.gpu
FF_JOINTS = list(range(65, 283))
FINGERS = ['g11', 'g12', 'g13', 'g2']           # reference environment/manenv.py:16


def _torch():
    import torch
    return torch


def _batch(model, ks, jids, tids, pipeline=None):
    from softgrip_amd import native
    torch = _torch()
    nm = native.NativeModel(model)
    b = native.NativeBatch(nm, len(ks), 0)
    if pipeline:
        b.set_pipeline(pipeline)
    b.set_stiffness(np.asarray(ks, dtype=np.float64), jids, tids)
    sens = torch.zeros(len(ks), nm.nsensordata, dtype=torch.float64, device=b.device)
    flags = torch.zeros(len(ks), dtype=torch.int32, device=b.device)
    return nm, b, sens, flags


def _oracles(model, ks, jids, tids):
    sims = []
    for k in ks:
        s = oracle_sim(model)
        s.jnt_stiffness[jids] = k
        s.tendon_stiffness[tids] = k
        sims.append(s)

    def start(s):
        s.reset(); s.forward(); s.step()
    _each(sims, start)
    return sims


def _each(sims, fn):
    """fn(sim) for every oracle env, on threads (the C oracle runs without the GIL; every env has its own model and data)"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(len(sims), os.cpu_count() or 1))) as ex:
        return list(ex.map(fn, sims))


def _step7(sims):
    """7 substeps of every oracle env; -> per env the OR of the warnings mj_step raised (0 = none)"""
    def go(s):
        w = 0
        for _ in range(7):
            w |= s.step()
        return w
    return _each(sims, go)


@pytest.mark.parametrize("scene", ["fourfinger_softball_fix", "fourfinger_softball"])
def test_four_finger_episode_matches_oracle(scene):
    """(fix rows only: 219 equality rows; with the composite's neighbour equalities: 651 -- the blocks run by the plan's 64-slot schedule)
    the whole squeeze schedule, 5 envs over the stiffness range, along the oracle's trajectories (the batch re-seated on the oracles'
    states after every env step: 218 active limit rows make this scene amplify round-off, tests/test_tree_emu.py): every sensor
    sample of the 24 channels at 1e-7, contact / row / sweep counts and the 64 touch bits exact at every step"""
    torch = _torch()
    m = sg.load_model(model_path(scene), "implicit")
    ks = [300.0, 575.0, 850.0, 1125.0, 1400.0] if scene.endswith("_fix") else [400.0, 650.0, 1000.0, 1210.0, 1390.0]
    nm, b, sens, flags = _batch(m, ks, FF_JOINTS, [0])
    assert nm.nboxes == 64 and nm.nsensordata == 24
    sims = _oracles(m, ks, FF_JOINTS, [0])
    b.reset(1, sens=sens, flags=flags)
    assert int(flags.abs().sum()) == 0
    st = b.get_state()
    np.testing.assert_allclose(st["qpos"].cpu().numpy(), np.stack([s.qpos for s in sims]), atol=1e-12)
    boxes = [g for g in range(m.ngeom) if m.geom_type[g] == 6 and m.body_weldid[m.geom_bodyid[g]] != 0]
    worst, touched = 0.0, 0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            b.set_ctrl_broadcast(np.full(4, c))
            for s in sims:
                s.ctrl[:] = c
        dev = dict(device=b.device, dtype=torch.float64)
        b.set_state(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                    act=torch.tensor(np.stack([s.act for s in sims]), **dev),
                    qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
        b.step(7, sens=sens, flags=flags)                       # (asynchronous: the oracles step while the kernel runs)
        assert not any(_step7(sims)), t
        assert int(flags.abs().sum()) == 0, t
        got = sens.cpu().numpy()
        worst = max(worst, np.abs(got - np.stack([s.sensordata for s in sims])).max())
        stats = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
        words = b.touch_words(2).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        for i, s in enumerate(sims):
            assert (stats["ncon"][i], stats["nefc"][i], stats["iters"][i]) == (s.ncon, s.nefc, s.solver_iter), (t, i)
            want = 0
            for cc in s.contacts():
                for g, o in ((cc["geom1"], cc["geom2"]), (cc["geom2"], cc["geom1"])):
                    if g in boxes and "OBJ" in (m.geom_names[o] or ""):
                        want |= 1 << boxes.index(g)
            assert int(words[i, 0]) | (int(words[i, 1]) << 32) == want, (t, i)
            touched |= want
    assert worst < 1e-7, worst
    for f in range(4):
        assert touched >> (16 * f) & 0xFFFF, f
    print("four-finger episode (%s): max |sensor - oracle| = %.2e" % (scene, worst))


def test_four_finger_free_running_window():
    """the four-finger scene FREE-RUNNING on the GPU against the oracle until its round-off amplification sets in (218
    active limit rows: two correct runs part by 1e-5 after 30 env steps, tests/test_tree_emu.py): 5 envs over the stiffness range,
    the first 15 env steps (105 substeps) at 1e-7 on all 24 channels with contact / row / sweep counts exact, and still within
    north_star's 1e-4 at step 25."""
    m = sg.load_model(model_path("fourfinger_softball_fix"), "implicit")
    ks = [300.0, 575.0, 850.0, 1125.0, 1400.0]
    nm, b, sens, flags = _batch(m, ks, FF_JOINTS, [0])
    sims = _oracles(m, ks, FF_JOINTS, [0])
    b.reset(1, sens=sens, flags=flags)
    errs = []
    for t in range(25):
        b.step(7, sens=sens, flags=flags)
        assert not any(_step7(sims)), t
        assert int(flags.abs().sum()) == 0, t
        errs.append(np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max())
        if t < 15:
            stats = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
            for i, s in enumerate(sims):
                assert (stats["ncon"][i], stats["nefc"][i], stats["iters"][i]) == (s.ncon, s.nefc, s.solver_iter), (t, i)
    assert max(errs[:15]) < 1e-7 and max(errs) < 1e-4, errs
    q = b.get_state()["qpos"].cpu().numpy()
    np.testing.assert_allclose(q, np.stack([s.qpos for s in sims]), atol=1e-6)
    print("four-finger free-running: max |sensor - oracle| %.2e over 15 env steps, %.2e over 25" % (max(errs[:15]), max(errs)))


@pytest.mark.parametrize("scene,jids,nu,steps", [("fourfinger_softball_fix", FF_JOINTS, 4, 40), ("freeball_fix", list(range(9, 227)), 2, 14),
                                                 ("fourfinger_softball", FF_JOINTS, 4, 24), ("freeball", list(range(9, 227)), 2, 14)])
def test_tree_full_size_properties(scene, jids, nu, steps):
    """the two tree workloads at BASELINE size (4096 envs, the batch the bench lines are quoted on), the fingers
    closing from the first step so that contacts, limit rows and the 30-sweep solves are in the window: identical parameters give
    bit-identical trajectories wherever the env sits in the batch, a permutation of the stiffnesses permutes the outputs, no env is
    flagged, the state stays finite (the free ball: unit quaternions).  Since r05 also the DEFAULT models of the two scenes (651
    equality rows: the sweep's neighbour-row instantiation at 4096 envs); `freeball` -- whose stiff envs leave the pipeline's
    envelope later in an episode (test_free_ball_episode_matches_oracle) -- over the steps before its first flag, at least 8."""
    torch = _torch()
    n = 4096
    rng = np.random.RandomState(0)
    ks = rng.uniform(300, 1400, n)
    ks[1::2] = ks[0::2]                        # pairs of identical envs
    perm = rng.permutation(n)
    m = sg.load_model(model_path(scene), "implicit")
    outs = []
    for kk in (ks, ks[perm]):
        nm, b, sens, flags = _batch(m, kk, jids, [0])
        b.reset(1, sens=sens, flags=flags)
        fl = flags.clone()
        b.set_ctrl_broadcast(np.full(nu, -0.2))
        most = 0
        snap = None
        for t in range(steps):
            b.step(7, sens=sens, flags=flags)
            fl |= flags
            if scene == "freeball":
                if int((fl != 0).sum()):
                    assert t >= 8, (t, torch.nonzero(fl).flatten()[:8].tolist())
                    break
                st = b.get_state()
                snap = (t, sens.cpu().numpy(), st["qpos"].cpu().numpy(), st["qvel"].cpu().numpy())
            if t % 5 == 4 or t == steps - 1:
                most = max(most, int(b.solver_stats()["ncon"].max()))
        assert most > 0, "no contact in the window"
        if scene == "freeball":
            outs.append(snap)
        else:
            assert int((fl != 0).sum()) == 0, (scene, torch.nonzero(fl).flatten()[:8].tolist())
            st = b.get_state()
            outs.append((steps - 1, sens.cpu().numpy(), st["qpos"].cpu().numpy(), st["qvel"].cpu().numpy()))
        del b, nm
        torch.cuda.empty_cache()
    (t1, s1, q1, v1), (t2, s2, q2, v2) = outs
    assert t1 == t2
    assert np.array_equal(s1[0::2], s1[1::2]) and np.array_equal(q1[0::2], q1[1::2]) and np.array_equal(v1[0::2], v1[1::2])
    assert np.array_equal(s1[perm], s2) and np.array_equal(q1[perm], q2) and np.array_equal(v1[perm], v2)
    assert np.isfinite(s1).all() and np.isfinite(q1).all()
    if scene.startswith("freeball"):
        assert np.abs(np.linalg.norm(q1[:, 11:15], axis=1) - 1).max() < 1e-12


@pytest.mark.parametrize("scene,damper", [("softbox_fix", None), ("softball_fix", "implicit")])
def test_tree_pipeline_on_two_finger_scenes(scene, damper):
    """the tree kernel on models the fast kernels run too: a whole free-running episode against the oracle (sensors 1e-7, counts
    exact) and against the rows pipeline"""
    torch = _torch()
    m = sg.load_model(model_path(scene), damper)
    ks = [320.0, 903.6948543200572, 1390.0]
    nm, b, sens, flags = _batch(m, ks, JOINT_IDS, TENDON_IDS, "tree")
    nm2, b2, sens2, flags2 = _batch(m, ks, JOINT_IDS, TENDON_IDS, "rows")
    sims = _oracles(m, ks, JOINT_IDS, TENDON_IDS)
    touch, touch2 = torch.zeros(3, dtype=torch.int32, device=b.device), torch.zeros(3, dtype=torch.int32, device=b.device)
    b.reset(1, sens=sens, flags=flags)
    b2.reset(1, sens=sens2, flags=flags2)
    worst = worst2 = 0.0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            for x in (b, b2):
                x.set_ctrl_broadcast(np.full(2, c))
            for s in sims:
                s.ctrl[:] = c
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
        b.step(7, sens=sens, flags=flags, touch=touch)
        b2.step(7, sens=sens2, flags=flags2, touch=touch2)
        assert int(flags.abs().sum()) == 0 and int(flags2.abs().sum()) == 0
        got = sens.cpu().numpy()
        worst = max(worst, np.abs(got - np.stack([s.sensordata for s in sims])).max())
        worst2 = max(worst2, np.abs(got - sens2.cpu().numpy()).max())
        st, st2 = b.solver_stats(), b2.solver_stats()
        for i, s in enumerate(sims):
            assert (int(st["ncon"][i]), int(st["nefc"][i]), int(st["iters"][i])) == (s.ncon, s.nefc, s.solver_iter), (t, i)
        assert torch.equal(st["ncon"], st2["ncon"]) and torch.equal(touch, touch2), t
    assert worst < 1e-7 and worst2 < 1e-7, (worst, worst2)


def test_default_two_finger_model_in_the_tree_pipeline():
    """models/softbox.sgmodel (327 equality rows: the benchmark model) through pipeline 3 against the rows pipeline and the oracle, along
    the oracle's trajectory for 120 env steps: sensors 1e-7, counts exact, both pipelines"""
    torch = _torch()
    m = sg.load_model(model_path("softbox"))
    ks = [320.0, 903.6948543200572, 1390.0]
    nm, b, sens, flags = _batch(m, ks, JOINT_IDS, TENDON_IDS, "tree")
    nm2, b2, sens2, flags2 = _batch(m, ks, JOINT_IDS, TENDON_IDS, "rows")
    sims = _oracles(m, ks, JOINT_IDS, TENDON_IDS)
    b.reset(1, sens=sens, flags=flags)
    b2.reset(1, sens=sens2, flags=flags2)
    dev = dict(device=b.device, dtype=torch.float64)
    worst = worst2 = 0.0
    for t, c in enumerate(episode_schedule()[:120]):
        if c is not None:
            for x in (b, b2):
                x.set_ctrl_broadcast(np.full(2, c))
            for s in sims:
                s.ctrl[:] = c
        st = dict(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                  act=torch.tensor(np.stack([s.act for s in sims]), **dev), qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
        b.set_state(**st); b2.set_state(**st)
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
        b.step(7, sens=sens, flags=flags)
        b2.step(7, sens=sens2, flags=flags2)
        assert int(flags.abs().sum()) == 0 and int(flags2.abs().sum()) == 0
        ref = np.stack([s.sensordata for s in sims])
        worst = max(worst, np.abs(sens.cpu().numpy() - ref).max())
        worst2 = max(worst2, np.abs(sens2.cpu().numpy() - ref).max())
        st1, st2 = b.solver_stats(), b2.solver_stats()
        for i, s in enumerate(sims):
            assert (int(st1["ncon"][i]), int(st1["nefc"][i]), int(st1["iters"][i])) == (s.ncon, s.nefc, s.solver_iter), (t, i)
            assert (int(st2["ncon"][i]), int(st2["nefc"][i]), int(st2["iters"][i])) == (s.ncon, s.nefc, s.solver_iter), (t, i)
    assert worst < 1e-7 and worst2 < 1e-7, (worst, worst2)


def test_four_finger_manenv_contact_flag_and_masked_reset():
    """ManEnv on the four-finger scene with the reference's commented ids (finger_names g11 / g12 / g13 / g2, four actuators): the
    squeeze brings the contact flag up for every env, an env given a NaN state is flagged, reset alone (masked) with a re-drawn
    stiffness, and the others are untouched"""
    torch = _torch()
    from softgrip_amd.manenv import ManEnv
    np.random.seed(3)
    env = ManEnv(1, 7, [model_path("fourfinger_softball_fix")], is_vis=False, n_envs=6, tendon_damper="implicit",
                 joint_ids=FF_JOINTS, tendon_ids=[0], finger_names=FINGERS, n_actuated=4)
    assert env.nmodel.nboxes == 64 and all(bits != 0 for bits in env._finger_bits)
    env.reset()
    flag_seen = torch.zeros(6, dtype=torch.bool, device=env.env.device)
    for t, c in enumerate(episode_schedule()[:100]):
        if c is not None:
            env.close_hand() if c < 0 else env.loose_hand()
        sens, flag = env.step()
        assert sens.shape == (6, 24) and bool(torch.isfinite(sens).all())
        flag_seen |= flag
    assert bool(flag_seen.all())          # all four fingers on the ball at some step, in every env
    # one env breaks
    st = env.env.get_state()
    before = st["qpos"].clone()
    k_before = env.stiffness.copy()
    st["qpos"][2, 70] = float("nan")
    env.env.set_state(qpos=st["qpos"])
    sens, flag = env.step()
    assert env.n_resets == 1 and env.stiffness[2] != k_before[2] and np.array_equal(np.delete(env.stiffness, 2), np.delete(k_before, 2))
    after = env.env.get_state()["qpos"]
    assert bool(torch.isfinite(after).all())
    assert float((after[[0, 1, 3, 4, 5]] - before[[0, 1, 3, 4, 5]]).abs().max()) < 0.05     # they went on with their episode


@pytest.mark.parametrize("scene", ["freeball_fix", "freeball"])
def test_free_ball_episode_matches_oracle(scene):
    """the reference's free-floating ball (soft_experiments_softball.xml: the composite on a body with a free joint, nq = 233, nv = 232)
    in the tree pipeline's object block on the GPU: FREE-RUNNING against the oracle for the first 40 (default model: 15) env steps (280
    substeps with up to 39 contacts: sensors 1e-6), then along the oracle's trajectories (the batch re-seated after every env step: late
    in the episode a contact at its threshold decides differently after 1300 substeps of round-off) -- sensors 1e-7, contact / row /
    sweep counts exact at all 200 steps, unit quaternions.

    freeball_fix: 4 envs over the stiffness range, none ever flagged.  freeball (the default model, neighbour equalities; r05: 7 envs, the
    whole episode): that model flings the ball out of the gripper for the stiffer envs, and where it lands it collects more contacts
    than the pipeline holds (oracle: 191 / 193 / 268 at k = 1200 / 900 / 750, env steps 52 / 83 / 114) or the oracle itself raises a
    warning (k = 1400, step 77).  An env is compared for as long as it lives, and the step at which the GPU flags it must be the step at
    which the oracle warns or exceeds the pipeline's 128 contacts -- neither earlier nor later; at least three envs run all 200 steps."""
    torch = _torch()
    m = sg.load_model(model_path(scene), "implicit")
    fix = scene.endswith("_fix")
    assert (m.nq, m.nv, m.njnt) == (233, 232, 227) and m.neq == (219 if fix else 651)
    n_free = 40 if fix else 15
    jids = list(range(9, 227))              # joint ids of the ball's 218 sliders (joint 8 is the free joint)
    ks = [300.0, 700.0, 1050.0, 1400.0] if fix else [300.0, 450.0, 600.0, 750.0, 900.0, 1200.0, 1400.0]
    n = len(ks)
    nm, b, sens, flags = _batch(m, ks, jids, [0])
    assert (nm.nq, nm.nv) == (233, 232)
    sims = _oracles(m, ks, jids, [0])
    b.reset(1, sens=sens, flags=flags)
    assert int(flags.abs().sum()) == 0
    worst_free = worst = 0.0
    dev = dict(device=b.device, dtype=torch.float64)
    died = [None] * n

    def go(s):
        w = mx = 0
        for _ in range(7):
            w |= s.step()
            mx = max(mx, s.ncon)
        return w, mx

    for t, c in enumerate(episode_schedule()):
        if c is not None:
            b.set_ctrl_broadcast(np.full(2, c))
            for s in sims:
                s.ctrl[:] = c
        if t >= n_free:      # (an env that has died sits on its oracle's last state from here on and is no longer looked at)
            b.set_state(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                        act=torch.tensor(np.stack([s.act for s in sims]), **dev),
                        qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
        b.step(7, sens=sens, flags=flags)
        live = [i for i in range(n) if died[i] is None]
        res = dict(zip(live, _each([sims[i] for i in live], go)))
        fl = flags.cpu().numpy()
        got = sens.cpu().numpy()
        stats = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
        for i in live:
            w, mx = res[i]
            if w or mx > 128:
                assert fl[i] != 0 and t >= n_free and not fix, (t, i, w, mx, fl[i])     # the GPU flags the env at this very step
                if not w:
                    assert fl[i] == 8, (t, i, fl[i])                                    # SG_FLAG_CONTACTFULL: more than the pipeline's 128 contacts
                died[i] = (t, w, mx)
                continue
            assert fl[i] == 0, (t, i, fl[i], w, mx)
            err = np.abs(got[i] - sims[i].sensordata).max()
            if t < n_free:
                worst_free = max(worst_free, err)
            else:
                worst = max(worst, err)
            assert (stats["ncon"][i], stats["nefc"][i], stats["iters"][i]) == (sims[i].ncon, sims[i].nefc, sims[i].solver_iter), (t, i)
        if t == n_free - 1:
            q = b.get_state()["qpos"].cpu().numpy()
            np.testing.assert_allclose(q, np.stack([s.qpos for s in sims]), atol=1e-7)
    st = b.get_state()
    q = st["qpos"].cpu().numpy()
    assert q.shape == (n, 233) and st["qvel"].shape == (n, 232)
    alive = [i for i in range(n) if died[i] is None]
    assert len(alive) >= (n if fix else 3), died
    assert np.abs(np.linalg.norm(q[alive][:, 11:15], axis=1) - 1).max() < 1e-12
    assert worst_free < 1e-6 and worst < 1e-7, (worst_free, worst)
    print("free ball episode (%s): max |sensor - oracle| = %.2e free-running (%d steps), %.2e re-seated; envs that left the pipeline's "
          "envelope (env step, oracle warning, most contacts): %s" % (scene, worst_free, n_free, worst, {ks[i]: d for i, d in enumerate(died) if d}))


def test_model_outside_both_classes_is_refused_with_both_reasons():
    from softgrip_amd import native
    m = sg.load_model(model_path("fourfinger_softball_fix"), "implicit")
    m.jnt_type = m.jnt_type.copy()
    m.jnt_type[3] = 2                      # a slide joint in a finger chain
    with pytest.raises(native.SoftgripError) as ei:
        native.NativeModel(m)
    assert ei.value.code == native.SG_ERR_MODEL and "two-finger kernels" in str(ei.value) and "tree pipeline" in str(ei.value)


def test_datasets_from_the_four_finger_and_the_free_ball_scenes(tmp_path):
    """the dataset host (create_dataset: reference create_dataset.py:23-79) on the two scenes of SURVEY 8(f) rank 4: one episode-batch of
    8 envs each; the four-finger rows are [200, 24] (reference manenv.py:11,16: the commented ids), the free ball's [200, 12]; all finite;
    each scene's first env reproduces the oracle's first rows for its label"""
    import pickle
    from softgrip_amd import create_dataset as cd
    cases = [("fourfinger_softball_fix", FF_JOINTS, 24, ["--finger-names"] + FINGERS + ["--n-actuated", "4"]),
             ("freeball_fix", list(range(9, 227)), 12, [])]
    for scene, jids, width, extra in cases:
        args = cd.make_parser().parse_args(["--mujoco-model-paths", model_path(scene), "--n-envs", "8", "--data-folder", str(tmp_path), "--data-name", scene,
                                            "--tendon-damper", "implicit", "--joint-ids"] + [str(j) for j in jids] + ["--tendon-ids", "0"] + extra)
        np.random.seed(11)
        d = pickle.load(open(cd.log_into_file(args), "rb"))
        X = np.array(d["data"])
        assert X.shape == (8, 200, width) and np.isfinite(X).all() and len(d["stiffness"]) == 8
        m = sg.load_model(model_path(scene), "implicit")
        s = oracle_sim(m)
        s.jnt_stiffness[jids] = d["stiffness"][0]
        s.tendon_stiffness[0] = d["stiffness"][0]
        s.reset(); s.forward(); s.step()
        for t in range(8):
            for _ in range(7):
                assert s.step() == 0
            assert np.abs(X[0, t] - s.sensordata).max() < 1e-6, (scene, t)


@pytest.mark.parametrize("free,neighbors", [(False, False), (False, True), (True, False), (True, True)])
def test_random_grippers_on_the_gpu(tmp_path, free, neighbors):
    """the CPU fuzz of tests/test_tree_emu.py on the device: 10 seeded random grippers per variant (1 - 4 fingers, 1 - 5 links, other chain
    strides, LDS layouts and workgroup counts per CU than the committed scenes), 3 envs over the stiffness range each, 40 env steps
    re-seated on the oracles after every substep; counts exact, sensors / velocities to 1e-7 of the signal (95 % of the substeps to
    1e-10); a scene that leaves the class or blows up is flagged by both sides in the same substep (or by the kernel alone for its own
    capacity of 128 contacts)"""
    from helpers import random_gripper_xml
    torch = _torch()
    import os
    nscene = int(os.environ.get("SG_FUZZ_SCENES", "10"))      # (a one-off sweep: SG_FUZZ_SCENES=100 SG_FUZZ_SEED=1000 pytest -k random_grippers_on_the_gpu)
    rng = np.random.RandomState(int(os.environ.get("SG_FUZZ_SEED", "40")) + 2 * int(free) + int(neighbors))
    ran = nabs = 0
    for i in range(nscene):
        path = tmp_path / ("g%d.xml" % i)
        path.write_text(random_gripper_xml(rng, free))
        m = sg.compile_mjcf(str(path), composite_neighbors=neighbors)
        nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
        jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
        ks = list(rng.uniform(300, 1400, 3))
        nm, b, sens, flags = _batch(m, ks, jids, [0])
        sims = _oracles(m, ks, jids, [0])
        b.reset(1, sens=sens, flags=flags)
        assert int(flags.abs().sum()) == 0, i
        dev = dict(device=b.device, dtype=torch.float64)
        errs, stop = [0.0], False
        for t, c in enumerate(episode_schedule()[:40]):
            if c is not None:
                b.set_ctrl_broadcast(np.full(m.nu, c))
                for s in sims:
                    s.ctrl[:] = c
            for j in range(7):
                b.set_state(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                            act=torch.tensor(np.stack([s.act for s in sims]), **dev),
                            qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
                w = [s.step() for s in sims]
                b.step(1, sens=sens, flags=flags)
                f = flags.cpu().numpy()
                if any(w) or f.any():
                    for e in range(3):
                        assert (bool(w[e]) == bool(f[e])) or (f[e] == 8 and not w[e] and sims[e].ncon > 128), (i, t, j, e, w, f)
                    stop = True
                    break
                got, st = sens.cpu().numpy(), b.get_state()
                stats = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
                qv = st["qvel"].cpu().numpy()
                for e, s in enumerate(sims):
                    assert (stats["ncon"][e], stats["nefc"][e], stats["iters"][e]) == (s.ncon, s.nefc, s.solver_iter), (i, t, j, e)
                    scale = 1.0 + np.abs(s.sensordata).max() + np.abs(s.qacc_warmstart).max()   # (an accelerometer sample is a sum of |qacc| r terms that may cancel)
                    aerr = np.abs(got[e] - s.sensordata).max()
                    err = max(aerr, np.abs(qv[e] - s.qvel).max()) / scale
                    assert err < 1e-7, (i, t, j, e, err)
                    if np.abs(s.qacc_warmstart).max() < 1e3:     # where the scene is not blowing up, north_star's absolute bound
                        assert aerr < 1e-4, (i, t, j, e, aerr)   # (1e-7 of the signal is <= 1e-4 there; measured: 1e-9 and below)
                        nabs += 1
                    errs.append(err)
            if stop:
                break
        else:
            ran += 1
        assert np.percentile(errs, 95) < 1e-10, (i, np.percentile(errs, 95))
        del b, nm
    assert ran >= 0.6 * nscene, ran
    assert nabs > 100 * nscene, nabs     # the absolute bound was exercised on most substeps
    print("gpu fuzz (free %s, neighbour rows %s): %d scenes, %d ran their 40 steps unflagged, %d (env, substep) samples under the absolute 1e-4 bound" % (free, neighbors, nscene, ran, nabs))


@pytest.mark.parametrize("links,hinges,seed", [(7, 3, 202), (6, 3, 204), (2, 2, 206)])
def test_chain_capacities_on_the_gpu(tmp_path, links, hinges, seed):
    """the three instantiations of the tree kernel (unroll capacities 24 / 20 / 8: csrc/sg_tree.hip) on chains of 21, 18 and 4 dofs --
    the scenes of tests/test_tree_emu.py::test_chain_capacities_and_kernel_instantiations, 3 envs, 30 env steps re-seated per substep"""
    from helpers import random_gripper_xml
    torch = _torch()
    rng = np.random.RandomState(seed)
    path = tmp_path / "g.xml"
    path.write_text(random_gripper_xml(rng, False, links=links, hinges=hinges, fingers=2))
    m = sg.compile_mjcf(str(path), composite_neighbors=False)
    nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
    jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
    ks = [700.0, 400.0, 1200.0]
    nm, b, sens, flags = _batch(m, ks, jids, [0])
    sims = _oracles(m, ks, jids, [0])
    b.reset(1, sens=sens, flags=flags)
    assert int(flags.abs().sum()) == 0
    dev = dict(device=b.device, dtype=torch.float64)
    most = 0
    for t, c in enumerate(episode_schedule()[:30]):
        if c is not None:
            b.set_ctrl_broadcast(np.full(m.nu, c))
            for s in sims:
                s.ctrl[:] = c
        for j in range(7):
            b.set_state(qpos=torch.tensor(np.stack([s.qpos for s in sims]), **dev), qvel=torch.tensor(np.stack([s.qvel for s in sims]), **dev),
                        act=torch.tensor(np.stack([s.act for s in sims]), **dev),
                        qacc_warmstart=torch.tensor(np.stack([s.qacc_warmstart for s in sims]), **dev))
            w = [s.step() for s in sims]
            b.step(1, sens=sens, flags=flags)
            f = flags.cpu().numpy()
            assert w[0] == 0 and f[0] == 0, (t, j, w, f)   # (env 0 is the emulation test's scene and stiffness: within the kernel's capacity throughout)
            most = max(most, sims[0].ncon)
            got, qv = sens.cpu().numpy(), b.get_state()["qvel"].cpu().numpy()
            stats = {k: v.cpu().numpy() for k, v in b.solver_stats().items()}
            for e, s in enumerate(sims):
                if w[e] or f[e]:
                    assert (w[e] and f[e]) or (f[e] == 8 and s.ncon > 128), (t, j, e, w, f)
                    continue
                assert (stats["ncon"][e], stats["nefc"][e], stats["iters"][e]) == (s.ncon, s.nefc, s.solver_iter), (t, j, e)
                scale = 1.0 + np.abs(s.sensordata).max() + np.abs(s.qacc_warmstart).max()
                assert max(np.abs(got[e] - s.sensordata).max(), np.abs(qv[e] - s.qvel).max()) < 1e-7 * scale, (t, j, e)
    assert most > 0


@pytest.mark.gpu
@pytest.mark.parametrize("scene,jids,want", [("fourfinger_softball_fix", FF_JOINTS, 4), ("freeball_fix", list(range(9, 227)), 4), ("freeball", list(range(9, 227)), 3)])
def test_tree_kernel_workgroups_per_cu(scene, jids, want):
    """the occupancy the r04 numbers rest on: the env's LDS block is small enough for four workgroups per CU -- one
    wavefront per SIMD -- on the four-finger gripper (37.6 KB) and on the free ball (37.8 KB); the free ball's neighbour-row model
    keeps C_e in LDS for its blocks (48 KB: three).  sg_tree_workgroups_per_cu asks the runtime
    (hipOccupancyMaxActiveBlocksPerMultiprocessor with the launch's LDS bytes); a two-finger batch on the rows pipeline answers 0."""
    from softgrip_amd import native
    m = sg.load_model(model_path(scene), "implicit")
    nm, b, sens, flags = _batch(m, [700.0, 900.0], jids, [0])
    assert b.tree_workgroups_per_cu() == want
    m2 = sg.load_model(model_path("softbox_fix"))
    b2 = native.NativeBatch(native.NativeModel(m2), 2, 0)
    assert b2.tree_workgroups_per_cu() == 0


def test_contact_capacity_overflow_is_loud(tmp_path):
    """the reference's MuJoCo holds nconmax = 500 contacts (soft_grip_two_fingers.xml:8), the tree pipeline 128 per
    env.  An env that runs out of them is flagged SG_FLAG_CONTACTFULL and reset with a re-drawn label like a MuJoCo warning -- but it is
    COUNTED APART (ManEnv.n_capacity_resets), and a dataset job in which more than --max-capacity-resets (default 0.1 %) of the episodes
    end that way fails instead of silently selecting its data.  Driven past the capacity for real: the free ball's default model at
    k = 1200 lands with 191 contacts at env step 52 (test_free_ball_episode_matches_oracle)."""
    import types
    from softgrip_amd import ManEnv
    from softgrip_amd import create_dataset as cd
    jids = list(range(9, 227))
    np.random.seed(0)
    env = ManEnv(1, 7, [model_path("freeball")], is_vis=False, n_envs=3, joint_ids=jids, tendon_damper="implicit")
    env.reset()
    env.set_stiffness_values([450.0, 1200.0, 600.0])
    for t, c in enumerate(episode_schedule()[:60]):
        if c is not None:
            (env.close_hand if c < 0 else env.loose_hand)()
        env.step()
        assert (env.n_capacity_resets, env.n_resets) == ((0, 0) if t < 52 else (1, 1)), t
    assert env.stiffness[1] != 1200.0 and env.stiffness[0] == 450.0 and env.stiffness[2] == 600.0      # the label was re-drawn, the others went on
    # the dataset job: refuses by default, obeys an explicit allowance
    args = types.SimpleNamespace(mujoco_model_paths=[model_path("freeball")], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                 data_folder=str(tmp_path), data_name="fb", n_envs=16, device=0, joint_ids=jids, tendon_damper="implicit")
    np.random.seed(3)
    with pytest.raises(cd.ContactCapacityExceeded):
        cd.log_into_file(args)
    args.max_capacity_resets = 1.0
    np.random.seed(3)
    cd.log_into_file(args)
    import json
    s = json.load(open(tmp_path / "fb.summary.json"))
    assert s["envs_reset_at_contact_capacity"] >= 1 and s["envs_reset_after_a_warning"] >= s["envs_reset_at_contact_capacity"]
