"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle."""
import numpy as np
import pytest

import softgrip_amd as sg
from helpers import JOINT_IDS, TENDON_IDS, library_for, model_path, oracle_sim
from softgrip_amd.create_dataset import episode_schedule

pytestmark = pytest.mark.gpu

TOL_SENSOR = 1e-7   # abs, fp64 path; north_star allows 1e-4 against MuJoCo-CPU


def _gpu_batch(scene, ks, pipeline=None, damper=None):
    import torch
    from softgrip_amd import native
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    m = sg.load_model(model_path(scene), damper)
    nm = native.NativeModel(m, library_for(pipeline))
    b = native.NativeBatch(nm, len(ks), 0)
    if pipeline is not None:
        b.set_pipeline(pipeline)
    b.set_stiffness(np.asarray(ks, dtype=np.float64), JOINT_IDS, TENDON_IDS)
    return m, nm, b


def _bufs(b, n):
    import torch
    return (torch.zeros(n, 12, dtype=torch.float64, device=b.device), torch.zeros(n, dtype=torch.int32, device=b.device),
            torch.zeros(n, dtype=torch.int32, device=b.device))


def _touch_bit_of_geom(m):
    """geom id -> bit of `touch_out` (include/softgrip.h: bit 2*chain + box), from the model alone: the moving boxes, grouped by the
    root moving body of their finger, in body order"""
    chains = {}
    for g in range(m.ngeom):
        b = m.geom_bodyid[g]
        if m.body_weldid[b] == 0 or m.geom_type[g] != 6:
            continue
        while m.body_weldid[m.body_parentid[b]] != 0:
            b = m.body_parentid[b]
        chains.setdefault(b, []).append(g)
    return {g: 2 * c + k for c, root in enumerate(sorted(chains)) for k, g in enumerate(chains[root])}


def _expected_touch(m, contacts, bit_of):
    """the set of (finger box, OBJ* geom) pairs in the oracle's contact list, as touch bits (what manenv.py:71-83 looks at)"""
    t = 0
    for c in contacts:
        for a, b in ((c["geom1"], c["geom2"]), (c["geom2"], c["geom1"])):
            if a in bit_of and m.geom_names[b].startswith("OBJ"):
                t |= 1 << bit_of[a]
    return t


def _reference_flag(m, contacts, fingers_left, obj_name="OBJ"):
    """replay of reference environment/manenv.py:65-85 on a contact list; `fingers_left` is the list the reference aliases
    (pass a fresh copy of finger_names for the intended semantics, a persistent list for the reference's actual behaviour)"""
    flag = False
    for c in contacts:
        n1, n2 = m.geom_names[c["geom1"]] or None, m.geom_names[c["geom2"]] or None
        if n1 is not None and n2 is not None:
            if obj_name in n1 or obj_name in n2:
                for f in fingers_left:
                    if f in n1 or f in n2:
                        fingers_left.remove(f)
        if len(fingers_left) == 0:
            flag = True
            break
    return flag


# the 9-env cylinder episode (test_episode_matches_oracle): how many of its 1800 (env, step) samples may sit on a knife edge of the ORACLE --
# each one proven to be an outcome the oracle itself produces under a round-off-sized perturbation -- and the draws per amplitude
CYL_MAX_EVENTS = 8
CYL_DRAWS = 96
FREE_RUN_STEPS = 47  # env steps over which the neighbour-row scene is also compared free-running (contact starts at ~45; the error grows 10x every 5 steps from there: 1e-10 at 47, 7e-10 at 50, 3e-8 at 60)


@pytest.mark.parametrize("scene,pipeline,damper", [
    ("softbox_fix", "rows", None), ("softbox_fix", "split", None), ("softbox_fix", "fused", None), ("softbox", "rows", None),
    ("softbox_fix", "rows", "implicit"), ("softbox", "rows", "implicit"),
    ("softball_fix", "rows", "implicit"), ("softball_fix", "fused", "implicit"), ("softcylinder_fix", "rows", "implicit"),
    ("softcylinder_fix", "split", "implicit"), ("softball", "rows", "implicit"), ("softcylinder", "rows", "implicit")])
def test_episode_matches_oracle(scene, pipeline, damper):
    """every kernel pipeline against the oracle over the whole reference episode, 9 envs so that the PGS kernel runs a full
    and a partial wavefront.  softbox = the scene as compiled by default, with the
    composite's neighbour equalities (rows pipeline only); *_fix = the fix-rows-only variant (all three
    pipelines).  damper = "implicit": the volume tendon's damper integrated implicitly -- the only way the
    reference's ball and cylinder scenes, which start in deep penetration, get through an episode.

    The *_fix scenes are compared free-running over the 200 steps.  With the neighbour rows the squeeze is sensitive to
    round-off (two runs that differ in the last bit part by a factor ~10 every 5 env steps once the fingers touch), so
    those scenes are compared (a) free-running up to FREE_RUN_STEPS (softbox: no contact before) and (b) over the whole episode step
    by step along the oracle's trajectory: after every env step the batch is re-seated on the oracle's state, and what is bounded
    is the error the kernels add in one env step (7 substeps) -- every step of the episode, contact sets and iteration counts
    exactly."""
    import os
    import torch
    from oracle import oracle as O
    ks = [700.0, 903.6948543200572, 300.0, 1400.0, 512.25, 350.0, 1000.0, 1250.0, 640.0]   # 9 envs, every scene (r04: the oracle's envs step on threads)
    reseat = not scene.endswith("_fix")
    # free-running window of the neighbour-row models: up to first contact for the box; the ball / cylinder touch the fingers from the
    # first step on but sit nearly still through the idle phase (a 1e-13 perturbation at step 5 is at 1e-9 at step 20, 2e-8 at step 40,
    # then x 10 per 5 steps) -- their first 20 steps are compared free-running too
    free_run = FREE_RUN_STEPS if scene == "softbox" else 20
    m, nm, b = _gpu_batch(scene, ks, pipeline, damper)
    sens, flags, touch = _bufs(b, len(ks))
    om = O.OracleModel(m.to_blob())
    sims = [O.OracleSim(om) for _ in ks]
    threads = min(len(ks), os.cpu_count() or 1)
    for s, k in zip(sims, ks):
        s._om = om
        s.jnt_stiffness[JOINT_IDS] = k
        s.tendon_stiffness[TENDON_IDS] = k
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    np.testing.assert_allclose(sens.cpu().numpy(), np.stack([s.sensordata for s in sims]), atol=1e-12)
    ctrl = np.zeros(2)
    worst, events, knife = 0.0, 0, []
    bit_of = _touch_bit_of_geom(m)
    touched = 0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        if scene == "softcylinder" and reseat:
            pre = [(s.qpos.copy(), s.qvel.copy(), s.qacc_warmstart.copy(), s.act.copy()) for s in sims]
        b.step(7, sens=sens, flags=flags, touch=touch)
        assert O.step_many(om, sims, 7, threads) == 0, t
        got = sens.cpu().numpy()
        want_s = np.stack([s.sensordata for s in sims])
        st = b.solver_stats()
        counts = list(zip(st["ncon"].cpu().tolist(), st["nefc"].cpu().tolist(), st["iters"].cpu().tolist()))
        same = np.array([counts[e] == (s.ncon, s.nefc, s.solver_iter) for e, s in enumerate(sims)])
        err_e = np.abs(got - want_s).max(axis=1)
        if scene == "softcylinder" and reseat:
            # The cylinder's squeeze is the most violent of the scenes (sensor spikes of 1e2 .. 1e3): a few times an episode a
            # contact sits at its threshold so closely that round-off -- the same numbers summed in another order than the oracle sums
            # them -- decides it the other way within the seven substeps, and that env step differs by an impact (1e0 .. 1e2).  Such a
            # sample is NOT excused and NOT held against the kernels' own arithmetic (r04 did that: a self-comparison): it is held against
            # THE ORACLE'S OWN OUTCOME SET.  The oracle is restarted from its pre-step state with a perturbation of round-off size added
            # to every position and velocity -- seeded draws at 1e-13, then 1e-12, then 1e-11 (the test's own state tolerance is 1e-9 /
            # 1e-7) -- and the GPU's row must be one of the outcomes the oracle itself produces: sensors to 1e-6 relative, (ncon, nefc,
            # sweeps) and the touch bits exactly.  Measured (r05): the steps are several-way knife edges -- env 6 at step 129 has four
            # distinct outcomes among 48 draws at 1e-13 (one is the GPU's); at steps 130 / 132 / 135 about 60 of 64 perturbed draws give
            # the GPU's row and the UNPERTURBED oracle is the odd one out.  At most CYL_MAX_EVENTS of the 9 x 200 samples may need this.
            for e in np.flatnonzero(~((err_e < TOL_SENSOR) & same)):
                ok, seen = False, []
                for amp in (1e-13, 1e-12, 1e-11):
                    rng = np.random.RandomState(1000 * t + int(e))
                    for _ in range(CYL_DRAWS):
                        s2 = O.OracleSim(om)
                        s2._om = om
                        s2.jnt_stiffness[JOINT_IDS] = ks[e]
                        s2.tendon_stiffness[TENDON_IDS] = ks[e]
                        s2.reset(); s2.forward()
                        s2.ctrl[:] = ctrl
                        q, v, w, a = pre[e]
                        s2.qpos[:] = q + amp * rng.uniform(-1, 1, q.shape)
                        s2.qvel[:] = v + amp * rng.uniform(-1, 1, v.shape)
                        s2.qacc_warmstart[:] = w
                        s2.act[:] = a
                        assert O.step_many(om, [s2], 7, 1) == 0
                        d = np.abs(s2.sensordata - got[e]).max() / (1 + np.abs(got[e]).max())
                        seen.append(float(d))
                        if d < 1e-6 and counts[e] == (s2.ncon, s2.nefc, s2.solver_iter) and \
                                int(touch[e].item()) == _expected_touch(m, s2.contacts(), bit_of):
                            ok = True
                            break
                    if ok:
                        break
                assert ok, ("the GPU's row is none of the oracle's outcomes under round-off-sized perturbations", t, e, got[e], sorted(seen)[:4])
                knife.append((t, int(e), amp, len(seen)))
                events += 1
                same[e] = False     # (its state is not compared below: the batch is re-seated on the unperturbed oracle's state anyway)
            assert events <= CYL_MAX_EVENTS, (t, knife)
        else:
            assert same.all(), (t, counts)
        worst = max(worst, np.abs(got[same] - want_s[same]).max()) if same.any() else worst
        assert worst < TOL_SENSOR, (t, worst)
        assert int(flags.abs().sum()) == 0
        # the contact read-out (reference manenv.py:65-85 reads data.contact after the 7 substeps): touch_out == the (finger box,
        # OBJ*) pairs of the oracle's contact list, every step
        want = [_expected_touch(m, s.contacts(), bit_of) for s in sims]
        have = touch.cpu().tolist()
        assert [have[e] for e in range(len(sims)) if same[e]] == [want[e] for e in range(len(sims)) if same[e]], (t, have, want)
        touched |= want[0]
        if reseat and t >= free_run:
            gs = b.get_state()
            for e, s in enumerate(sims):
                if same[e]:
                    np.testing.assert_allclose(gs["qpos"][e].cpu().numpy(), s.qpos, atol=1e-9)
                    np.testing.assert_allclose(gs["qvel"][e].cpu().numpy(), s.qvel, atol=1e-7)
            T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=b.device).contiguous()  # noqa: E731
            b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                        qacc_warmstart=T([s.qacc_warmstart for s in sims]))
    assert worst < TOL_SENSOR, worst
    if knife:
        print("cylinder knife-edge samples proven against the oracle's outcome set (env step, env, amplitude, draws needed):", knife)
    assert (touched & 0b0011) and (touched & 0b1100), touched   # both fingers did touch the object during the squeeze
    st = b.get_state()
    for e, s in enumerate(sims):
        if not same[e]:
            continue
        np.testing.assert_allclose(st["qpos"][e].cpu().numpy(), s.qpos, atol=1e-9)
        np.testing.assert_allclose(st["qvel"][e].cpu().numpy(), s.qvel, atol=1e-7)
        np.testing.assert_allclose(st["act"][e].cpu().numpy(), s.act, atol=1e-13)


@pytest.mark.parametrize("epw,n", [(8, 21), (4, 21), (8, 8)])
def test_solver_envs_per_wavefront_variants_match_oracle(epw, n):
    """the solver kernel instantiation with 8 envs per wavefront (sg_pgs_rows_kernel<*, false, 8>) is what a fix-rows-only
    batch of >= 8185 envs runs, and no parity test reached it (they use a few envs: 4 per wavefront).  Forced here through
    sg_set_solver_envs_per_wavefront on 21 envs -- two full wavefronts and a ragged one of 5 -- over the whole episode, free-running
    against the oracle, contact and sweep counts exactly; 4 per wavefront on the same batch for symmetry."""
    import os
    from oracle import oracle as O
    ks = np.linspace(300.0, 1400.0, n)
    m, nm, b = _gpu_batch("softbox_fix", ks)
    assert b.solver_envs_per_wavefront() == 4          # automatic choice below 8185 envs
    b.set_solver_envs_per_wavefront(epw)
    assert b.solver_envs_per_wavefront() == epw
    sens, flags, touch = _bufs(b, n)
    om = O.OracleModel(m.to_blob())
    sims = [O.OracleSim(om) for _ in ks]
    for s, k in zip(sims, ks):
        s.jnt_stiffness[JOINT_IDS] = k
        s.tendon_stiffness[TENDON_IDS] = k
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    worst = 0.0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        assert O.step_many(om, sims, 7, min(16, os.cpu_count() or 1)) == 0
        worst = max(worst, np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max())
        assert worst < TOL_SENSOR, (t, worst)
        assert int(flags.abs().sum()) == 0
        if t % 10 == 0 or t > 190:
            st = b.solver_stats()
            assert st["ncon"].cpu().tolist() == [s.ncon for s in sims] and st["iters"].cpu().tolist() == [s.solver_iter for s in sims], t
    # a model with neighbour rows always runs 4 envs of 16 lanes
    from softgrip_amd import native
    _, _, b2 = _gpu_batch("softbox", [700.0])
    with pytest.raises(native.SoftgripError):
        b2.set_solver_envs_per_wavefront(8)
    assert b2.solver_envs_per_wavefront() == 4


@pytest.mark.parametrize("scene,pipeline", [("softcylinder_fix", "rows"), ("softball_fix", "rows"), ("softcylinder_fix", "split"), ("softball_fix", "split"),
                                            ("softcylinder_fix", "fused"), ("softball_fix", "fused"),
                                            ("softcylinder", "rows"), ("softball", "rows")])
def test_other_scenes_first_substeps(scene, pipeline):
    """R = 3 / 4 kernel instantiations; these scenes start in deep penetration (chaotic), so only the first substeps
    are compared point-wise"""
    ks = [700.0, 400.0]
    m, nm, b = _gpu_batch(scene, ks, pipeline)
    sens, flags, touch = _bufs(b, len(ks))
    sims = [oracle_sim(m, k) for k in ks]
    for s in sims:
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    for n in range(2):
        b.step(1, sens=sens, flags=flags, touch=touch)
        for s in sims:
            s.step()
        st = b.get_state()
        for e, s in enumerate(sims):
            np.testing.assert_allclose(st["qpos"][e].cpu().numpy(), s.qpos, atol=1e-11)
            np.testing.assert_allclose(sens[e].cpu().numpy(), s.sensordata, atol=1e-6)
        assert b.solver_stats()["ncon"].cpu().tolist() == [s.ncon for s in sims]


@pytest.mark.parametrize("scene", ["softbox", "softbox_fix", "softball", "softcylinder"])
def test_full_size_properties(scene):
    """BASELINE size (4096 envs), on the product's default models of all three reference scenes (the ball and the cylinder with the
    implicit tendon damper, as ManEnv loads them) and on the box's fix-rows-only variant: size-independent properties -- identical
    parameters give bit-identical trajectories wherever the env sits in the batch, and a permutation of the stiffnesses permutes the
    outputs (bit-exact determinism holds whatever the system's sensitivity to round-off)."""
    import torch
    n = 4096
    rng = np.random.RandomState(0)
    ks = rng.uniform(300, 1400, n)
    ks[1::2] = ks[0::2]                        # pairs of identical envs
    perm = rng.permutation(n)
    damper = None if scene.startswith("softbox") else "implicit"
    m, nm, b = _gpu_batch(scene, ks, None, damper)
    _, _, b2 = _gpu_batch(scene, ks[perm], None, damper)
    outs = []
    for batch in (b, b2):
        sens, flags, touch = _bufs(batch, n)
        batch.reset(1, sens=sens, flags=flags, touch=touch)
        batch.set_ctrl_broadcast(np.array([-0.2, -0.2]))
        fl = torch.zeros(n, dtype=torch.int32, device=batch.device)
        for _ in range(60):
            batch.step(7, sens=sens, flags=flags, touch=touch)
            fl |= flags
        assert int((fl != 0).sum()) == 0
        outs.append((sens.cpu().numpy(), batch.get_state()["qpos"].cpu().numpy()))
    s1, q1 = outs[0]
    s2, q2 = outs[1]
    assert np.array_equal(s1[0::2], s1[1::2]) and np.array_equal(q1[0::2], q1[1::2])
    assert np.array_equal(s1[perm], s2) and np.array_equal(q1[perm], q2)
    assert np.isfinite(s1).all()
    if scene.startswith("softbox"):
        assert np.abs(s1[:, 2] - 9.81).max() < 5.0       # accelerometer z stays near gravity (the ball / cylinder are thrown about: no such bound)


def test_pipelines_agree_at_full_size():
    """the three pipelines run the same Gauss-Seidel sweep: on the scene they all support they agree to round-off on every env
    of a BASELINE-size batch"""
    n = 4096
    ks = np.random.RandomState(0).uniform(300, 1400, n)
    outs = []
    for pipeline in ("rows", "fused"):
        _, _, b = _gpu_batch("softbox_fix", ks, pipeline)
        sens, flags, touch = _bufs(b, n)
        b.reset(1, sens=sens, flags=flags, touch=touch)
        b.set_ctrl_broadcast(np.array([-0.2, -0.2]))
        for _ in range(60):
            b.step(7, sens=sens, flags=flags, touch=touch)
        outs.append(sens.cpu().numpy())
    assert np.abs(outs[0] - outs[1]).max() < 1e-8


def test_neighbour_row_model_refuses_other_pipelines():
    """in the test build, which has them; the product build has no fused / split pipeline at all and says so"""
    from softgrip_amd import native
    m = sg.load_model(model_path("softbox"))
    b = native.NativeBatch(native.NativeModel(m, library_for("fused")), 1, 0)
    with pytest.raises(native.SoftgripError):
        b.set_pipeline("fused")
    m, nm, b = _gpu_batch("softbox_fix", [700.0])
    with pytest.raises(native.SoftgripError, match="not part of this build"):
        b.set_pipeline("split")


@pytest.mark.parametrize("scene", ["softbox", "softbox_fix"])
def test_cfg2_uniform_stiffness_batch_is_bit_identical_across_envs(scene):
    """SURVEY 8(d) cfg 2: B = 1024, every env k = 700 (the XML value), the whole 200-step squeeze.  Identical envs must give
    bit-identical trajectories whatever lane, quad or wavefront they sit in -- free-running, on both model variants -- and env 0 is
    checked against the oracle at every step: free-running on the fix-rows-only variant, and on the default model (which amplifies
    round-off from step ~47 on) with the oracle RE-SEATED on the batch's state after every env step, so that each step's
    error is bounded along the product's own trajectory (the episode test re-seats the batch on the oracle's instead)."""
    n = 1024
    m, nm, b = _gpu_batch(scene, np.full(n, 700.0))
    sens, flags, touch = _bufs(b, n)
    s0 = oracle_sim(m, 700.0)
    s0.reset(); s0.forward(); s0.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    reseat = scene == "softbox"
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            s0.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for _ in range(7):
            assert s0.step() == 0
        got = sens[:1].cpu().numpy()
        assert np.abs(got[0] - s0.sensordata).max() < TOL_SENSOR, t
        if reseat or t in (39, 48, 60, 100, 119, 150, 199):
            assert b.solver_stats()["ncon"][:1].cpu().tolist() == [s0.ncon]
        if t in (39, 48, 60, 100, 119, 150, 199):
            got = sens.cpu().numpy()
            assert np.array_equal(got, np.broadcast_to(got[0], got.shape)), "envs differ at step %d" % t
            assert int(flags.abs().sum()) == 0
            tc = touch.cpu().numpy()
            assert (tc == tc[0]).all()
        if reseat and t >= FREE_RUN_STEPS:
            st = b.get_state()
            s0.qpos[:] = st["qpos"][0].cpu().numpy(); s0.qvel[:] = st["qvel"][0].cpu().numpy()
            s0.act[:] = st["act"][0].cpu().numpy(); s0.qacc_warmstart[:] = st["qacc_warmstart"][0].cpu().numpy()
    st = b.get_state()
    assert bool((st["qpos"] == st["qpos"][0]).all()) and bool((st["qvel"] == st["qvel"][0]).all())


@pytest.mark.parametrize("scene,n,t_pw,t_stat", [("softbox", 256, FREE_RUN_STEPS - 3, FREE_RUN_STEPS - 3), ("softball", 48, 20, 42), ("softcylinder", 48, 20, 42)])
def test_default_model_ensemble_matches_oracle_over_the_whole_episode(scene, n, t_pw, t_stat):
    """what the product writes into the later rows of a default-model dataset, for all three reference scenes
    (the ball and the cylinder with the implicit tendon damper, as ManEnv loads them; fewer envs: their oracle is 6 x slower).  Envs on
    a fine stiffness grid, GPU free-running against the oracle free-running over the whole 200-step episode.  The first rows point-wise
    (1e-7 up to first contact for the box; the ball / cylinder are in contact from the first step on and multiply a perturbation by ~100
    over their idle phase: 1e-7 for 20 steps, north_star's 1e-4 up to the start of the squeeze); from there on the restated system amplifies
    round-off and the two runs are two samples of the same chaotic squeeze, so the comparison is statistical: per step and
    channel the mean / spread / quantiles over the sweep, and per env the regressor-relevant features (mean and spread of every channel
    over the squeeze, and their rank correlation with the label), all within the sampling error calibrated in
    tests/test_oracle_kat.py::test_ensemble_statistic_is_calibrated (helpers.ENS_TOL / ENS_TOL_48)."""
    import os
    import torch
    from helpers import ENS_TOL, ENS_TOL_48, assert_ensembles_match, oracle_episodes
    ks = np.linspace(300.0, 1400.0, n)
    m, nm, b = _gpu_batch(scene, ks, None, None if scene == "softbox" else "implicit")
    sched = episode_schedule()
    out = torch.zeros(n, len(sched), 12, dtype=torch.float64, device=b.device)
    flags = torch.zeros(n, dtype=torch.int32, device=b.device)
    bad = torch.zeros_like(flags)
    b.reset(1, flags=flags)
    ctrl = np.zeros(2)
    for t, c in enumerate(sched):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
        b.step(7, sens=out[:, t], sens_stride=len(sched) * 12, flags=flags)
        bad |= flags
    assert int((bad != 0).sum()) == 0
    got = out.cpu().numpy()
    want = oracle_episodes(m, ks, threads=min(16, os.cpu_count() or 1))
    # box: the softest envs of the fine grid touch a step or two before the nine stiffnesses FREE_RUN_STEPS was fitted to: point-wise up
    # to step 43, and still within 1e-5 at FREE_RUN_STEPS (measured 1.5e-7 there); the statistical comparison takes over from t_stat
    assert np.abs(got[:, :t_pw] - want[:, :t_pw]).max() < TOL_SENSOR
    if scene == "softbox":
        assert np.abs(got[:, :t_stat + 3] - want[:, :t_stat + 3]).max() < 1e-5
    else:
        assert np.abs(got[:, :t_stat] - want[:, :t_stat]).max() < 1e-4
    rep = assert_ensembles_match(got, want, ks, t0=t_stat, tol=ENS_TOL if n >= 256 else ENS_TOL_48)
    print("ensemble parity, %s, steps %d..199: %s" % (scene, t_stat, rep))


def test_implicit_tendon_damper_deviation_quantified():
    """what the D5 flag (the volume tendon's damper integrated implicitly -- the only way the ball, the cylinder and
    the four-finger scene run) does where both integrators are stable: the box scene, idle phase, explicit vs implicit, on the GPU and on
    the oracle.  The fingers do not feel the object before first contact, so through the idle phase finger state and all 12 sensor
    channels are EQUAL (0, not small); what differs is the shell: slider positions by 1.6e-5 m, slider velocities by 1.7e-4 m/s
    (the shell's settling under the tendon's damper).  From first contact (env step 41) the two runs are two nearby initial
    conditions of the chaotic squeeze -- sensors 0.2 .. 0.7 apart at step 41, O(1) from step 43 -- with the same contact counts up
    to step 44.  The GPU's deviation equals the oracle's (same numbers to 1e-9): the flag does the same thing on both sides."""
    import torch
    ks = [300.0, 700.0, 1400.0]
    runs = {}
    for damper in ("explicit", "implicit"):
        m, nm, b = _gpu_batch("softbox", ks, None, damper)
        sims = [oracle_sim(m, k) for k in ks]
        for s in sims:
            s.reset(); s.forward(); s.step()
        sens, flags, touch = _bufs(b, len(ks))
        b.reset(1, sens=sens, flags=flags, touch=touch)
        rows = []
        for t in range(45):
            if t == 40:
                b.set_ctrl_broadcast(np.array([-0.2, -0.2]))
                for s in sims:
                    s.ctrl[:] = -0.2
            b.step(7, sens=sens, flags=flags, touch=touch)
            for s in sims:
                for _ in range(7):
                    assert s.step() == 0
            st = b.get_state()
            rows.append(dict(gs=sens.cpu().numpy().copy(), gq=st["qpos"].cpu().numpy(), gv=st["qvel"].cpu().numpy(), ncon=b.solver_stats()["ncon"].cpu().numpy(),
                             os=np.stack([s.sensordata for s in sims]), oq=np.stack([s.qpos for s in sims]), ov=np.stack([s.qvel for s in sims])))
            assert int(flags.abs().sum()) == 0
        runs[damper] = rows
    E, I = runs["explicit"], runs["implicit"]
    dq = max(np.abs(E[t]["gq"][:, 8:] - I[t]["gq"][:, 8:]).max() for t in range(41))
    dv = max(np.abs(E[t]["gv"][:, 8:] - I[t]["gv"][:, 8:]).max() for t in range(41))
    for t in range(41):                                            # idle phase: the fingers and the sensors do not know about the flag
        assert np.array_equal(E[t]["gs"], I[t]["gs"]) and np.array_equal(E[t]["gq"][:, :8], I[t]["gq"][:, :8]), t
        assert np.array_equal(E[t]["os"], I[t]["os"]), t
    assert 1e-6 < dq < 5e-5 and 1e-5 < dv < 5e-4, (dq, dv)         # ... the shell does: measured 1.6e-5 m, 1.7e-4 m/s
    for t in range(41):                                            # and the GPU's deviation is the oracle's
        assert np.abs((E[t]["gq"] - I[t]["gq"]) - (E[t]["oq"] - I[t]["oq"])).max() < 1e-9, t
        assert np.abs((E[t]["gv"] - I[t]["gv"]) - (E[t]["ov"] - I[t]["ov"])).max() < 1e-8, t
    first = [np.abs(E[t]["gs"] - I[t]["gs"]).max() for t in range(41, 45)]
    assert 0.05 < first[0] < 2.0 and max(first) < 20.0, first      # first contact: O(0.1 .. 1) apart at once, like any 1e-5 change of the contact geometry
    assert all(np.array_equal(E[t]["ncon"], I[t]["ncon"]) for t in range(44))
    print("D5 on the box scene: idle phase sensors equal; sliders |dq| %.2e |dv| %.2e; sensors at steps 41..44: %s" % (dq, dv, ["%.2f" % x for x in first]))


def test_state_roundtrip_and_masked_reset():
    import torch
    ks = [700.0, 800.0, 900.0]
    m, nm, b = _gpu_batch("softbox_fix", ks)
    sens, flags, touch = _bufs(b, 3)
    b.reset(1, sens=sens, flags=flags, touch=touch)
    b.set_ctrl_broadcast(np.array([-0.2, -0.2]))
    for _ in range(50):
        b.step(7, sens=sens, flags=flags, touch=touch)
    st = b.get_state()
    ref = sens.clone()
    b.step(7, sens=sens, flags=flags, touch=touch)
    after = sens.clone()
    b.set_state(**{k: v for k, v in st.items()})
    b.step(7, sens=sens, flags=flags, touch=touch)
    assert torch.equal(sens, after)                                   # restart from a saved state is exact
    # masked reset touches only the selected env
    st1 = b.get_state()
    mask = torch.tensor([0, 1, 0], dtype=torch.uint8, device=b.device)
    b.reset(1, sens=sens, flags=flags, touch=touch, mask=mask)
    st2 = b.get_state()
    assert torch.equal(st1["qpos"][0], st2["qpos"][0]) and torch.equal(st1["qpos"][2], st2["qpos"][2])
    assert float(st2["qpos"][1].abs().max()) < 1e-3 and float(st2["ctrl"][1].abs().max()) == 0.0
    assert float(st2["ctrl"][0][0]) == -0.2
    assert ref.shape == (3, 12)


def test_manenv_and_dataset_on_gpu(tmp_path):
    import pickle
    import types
    from softgrip_amd import create_dataset as cd
    np.random.seed(0)
    args = types.SimpleNamespace(mujoco_model_paths=[model_path("softbox_fix")], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                 data_folder=str(tmp_path), data_name="ds", n_envs=3, device=0)
    path = cd.log_into_file(args)
    d = pickle.load(open(path, "rb"))
    assert len(d["data"]) == 3 and np.array(d["data"][0]).shape == (200, 12)
    s = oracle_sim(sg.load_model(model_path("softbox_fix")), d["stiffness"][1])
    s.reset(); s.forward(); s.step()
    ref = []
    for c in episode_schedule():
        if c is not None:
            s.ctrl[:] = c
        for _ in range(7):
            s.step()
        ref.append(s.sensordata.copy())
    assert np.abs(np.array(d["data"][1]) - np.array(ref)).max() < TOL_SENSOR


def test_bad_env_is_reset_like_mujoco_exception():
    """reference manenv.py:47-51: a simulation warning makes ManEnv.step() reset the env (re-drawing its stiffness) and go on"""
    import torch
    from softgrip_amd import ManEnv
    np.random.seed(0)
    env = ManEnv(1, 7, [model_path("softbox_fix")], is_vis=False, n_envs=4)
    k0 = env.reset().copy()
    env.close_hand()
    for _ in range(20):
        env.step()
    st = env.env.get_state()
    st["qvel"][2, 10] = float("nan")                          # corrupt env 2
    env.env.set_state(qvel=st["qvel"].contiguous())
    readings, contact = env.step()
    assert torch.isfinite(readings).all()
    assert env.stiffness[2] != k0[2] and np.array_equal(env.stiffness[[0, 1, 3]], k0[[0, 1, 3]])
    q = env.env.get_state()["qpos"]
    assert float(q[2].abs().max()) < 1e-3 and float(q[0].abs().max()) > 1e-2     # env 2 restarted, the others carried on
    # raw flags: the kernels report the failure as data
    st = env.env.get_state()
    st["qpos"][1, 0] = 1e11
    env.env.set_state(qpos=st["qpos"].contiguous())
    flags = torch.zeros(4, dtype=torch.int32, device=env.env.device)
    env.env.step(7, flags=flags)
    assert flags.cpu().tolist() == [0, 1, 0, 0]


def test_config5_online_regressor_at_full_size():
    """BASELINE configs[4] at its stated size: 4096 envs (default scene) -> the on-device [4096, 200, 12] block -> per-channel statistics,
    noise augmentation, ConvNet forward and one Adam step (PyTorch-ROCm) without leaving the GPU; `bench.py --with-regressor` times
    exactly this loop"""
    import torch
    from softgrip_amd import ManEnv, convnet
    np.random.seed(1)
    n = 4096
    env = ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=n)
    env.set_new_stiffness()
    out, flags = env.rollout(episode_schedule())
    assert out.shape == (n, 200, 12) and out.is_cuda and out.dtype == torch.float64 and int((flags != 0).sum()) == 0
    assert bool(torch.isfinite(out).all())
    torch.manual_seed(0)
    net = convnet.ConvNet().to(out.device)
    opt = convnet.make_optimizer(net)
    y = torch.tensor(env.stiffness, device=out.device)
    mean, std = convnet.channel_stats(out)
    assert mean.shape == (1, 1, 12) and float(std.min()) > 0
    l0, pred = convnet.train_step(net, opt, out, y, mean, std, add_noise=True)
    assert pred.shape == (n,) and torch.isfinite(l0) and float(pred.min()) >= 300 and float(pred.max()) <= 1400
    for _ in range(8):
        l1, _ = convnet.train_step(net, opt, out, y, mean, std)
    assert float(l1) < float(l0)
    # the signal carries the label: after a few steps the regressor separates soft from stiff objects better than chance
    net.eval()
    with torch.no_grad():
        p = convnet.normalize_predictions(net((out - mean) / std))
    assert p.shape == (n,) and bool(torch.isfinite(p).all())


def test_regressor_step_numerics_at_full_size():
    """f2, numerically: one training step of the ConvNet on the GPU (float32, as the reference computes:
    NeuralNets.py:22) against an fp64 evaluation on the CPU -- same weights, same batch of n = 4096 simulated episodes [4096, 200, 12]
    (a real rollout of the default scene, normalised per channel as functions/utils.py:40-41), same labels.  Compared: the forward
    output and the loss in training mode, EVERY gradient tensor, the moving statistics, the weights after the Adam step, and the
    forward output in inference mode.  The CPU side is torch's CPU backend on a .double() copy, itself held against the NumPy
    restatement and finite differences in tests/test_convnet_dataset.py; here the NumPy restatement also evaluates a 64-sample slice
    in inference mode.  Tolerances are float32's: 1e-4 relative on outputs, 1e-2 in norm on every gradient tensor."""
    import copy
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import np_convnet as npc
    from softgrip_amd import ManEnv, convnet
    np.random.seed(4)
    n = 4096
    env = ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=n)
    env.set_new_stiffness()
    x, flags = env.rollout(episode_schedule())
    assert int((flags != 0).sum()) == 0
    y = torch.tensor(env.stiffness, device=x.device)
    mean, std = convnet.channel_stats(x)
    torch.manual_seed(1)
    net = convnet.ConvNet()
    with torch.no_grad():   # moving statistics and biases away from their initial values, so that inference mode tests something
        for name, b in net.named_buffers():
            b.copy_(0.05 * torch.randn(b.shape) if name.endswith("mean") else 0.8 + 0.4 * torch.rand(b.shape))
        for name, p in net.named_parameters():
            if name.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape))
    ref = copy.deepcopy(net).double()
    net = net.to(x.device)
    state0 = {k: v.detach().double().numpy().copy() for k, v in ref.state_dict().items()}

    # ---- inference mode, before the step
    net.eval(); ref.eval()
    xc, yc, mc, sc = x.cpu(), y.cpu(), mean.cpu(), std.cpu()
    with torch.no_grad():
        e_gpu = convnet.normalize_predictions(net((x - mean) / std)).cpu().double().numpy()
        e_ref = convnet.normalize_predictions(ref((xc - mc) / sc)).numpy()
    assert np.abs(e_gpu - e_ref).max() < 1e-4 * 1400
    xs = ((xc[:64] - mc) / sc).numpy()
    np.testing.assert_allclose(npc.predictions(npc.forward(state0, xs, training=False)), e_ref[:64], rtol=1e-9)

    # ---- one training step (no noise: the two sides must see the same batch)
    opt, ropt = convnet.make_optimizer(net), convnet.make_optimizer(ref)
    l_gpu, p_gpu = convnet.train_step(net, opt, x, y, mean, std)
    l_ref, p_ref = convnet.train_step(ref, ropt, xc, yc, mc, sc)
    assert abs(float(l_gpu) - float(l_ref)) < 1e-4 * float(l_ref)
    assert np.abs(p_gpu.cpu().double().numpy() - p_ref.numpy()).max() < 1e-4 * 1400
    g_ref = {k: v.grad for k, v in ref.named_parameters()}
    # a bias in front of a BatchNormalization (directly, or through the average pool and a dense layer: conv3) has NO gradient -- the
    # batch mean takes it out -- so its fp64 gradient is round-off (1e-14) and its float32 one is noise (1e-8): bounded, not compared
    shadowed = {"conv1.bias", "conv2.bias", "conv3.bias", "fc1.bias", "fc2.bias", "fc3.bias"}
    worst = 0.0
    for k, v in net.named_parameters():
        g, r = v.grad.cpu().double(), g_ref[k]
        if k in shadowed:
            assert float(r.abs().max()) < 1e-10 and float(g.abs().max()) < 1e-3, k   # measured: 7e-14 in fp64, 4e-5 in float32
            continue
        # MAE's gradient is sign(pred - y) / n per sample: one sample within float32 error of its label flips a sign and moves every
        # upstream gradient by 2 / n = 5e-4 relative -- hence 1e-2 in norm and 5e-2 on the worst entry, not 1e-6
        l2 = float((g - r).norm() / r.norm())
        mx = float((g - r).abs().max() / r.abs().max())
        worst = max(worst, l2)
        assert l2 < 1e-2 and mx < 5e-2, (k, l2, mx)
    bufs = dict(ref.named_buffers())
    for k, v in net.named_buffers():
        np.testing.assert_allclose(v.cpu().double().numpy(), bufs[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    # the Adam step: exactly Adam's first step on the GPU's own gradients, and the same weights as the fp64 side wherever the
    # gradient is large enough for its sign to be beyond float32 doubt (the first step moves every weight by ~lr * sign(g))
    after = dict(ref.named_parameters())
    for k, v in net.named_parameters():
        w1 = v.detach().cpu().double().numpy()
        g = v.grad.cpu().double().numpy()
        np.testing.assert_allclose(w1, npc.adam_first_step(state0[k], g), rtol=0, atol=2e-7 * max(1.0, float(np.abs(state0[k]).max())), err_msg=k)
        if k in shadowed:
            continue
        r = g_ref[k].numpy()
        sure = np.abs(r) > 1e-2 * np.abs(r).max()
        assert sure.any()
        assert np.abs(w1 - after[k].detach().numpy())[sure].max() < 1e-5, k
    # ---- inference mode after the step (updated weights and moving statistics)
    net.eval(); ref.eval()
    with torch.no_grad():
        e_gpu = convnet.normalize_predictions(net((x - mean) / std)).cpu().double().numpy()
        e_ref = convnet.normalize_predictions(ref((xc - mc) / sc)).numpy()
    # looser than before the step: Adam's first step moves EVERY weight by ~lr, also the six bias vectors whose gradient is zero in
    # exact arithmetic (float32 noise decides their sign), and in inference mode the moving statistics no longer cancel a bias shift:
    # measured 0.94 on predictions of ~700
    assert np.abs(e_gpu - e_ref).max() < 3e-3 * 1400
    print("regressor step on the GPU vs fp64: worst relative gradient error (L2, per tensor) %.2e" % worst)


@pytest.mark.parametrize("scene,n", [("softbox_fix", 1), ("softbox_fix", 65), ("softbox", 1), ("softbox", 13)])
def test_ragged_batches(scene, n):
    """batch sizes that fill neither a PGS wavefront (8 envs) nor a chain wavefront (64 chains): every env against the oracle
    through reset, the idle phase and the first contacts (45 env steps)"""
    ks = np.linspace(320.0, 1380.0, n)
    m, nm, b = _gpu_batch(scene, ks)
    sens, flags, touch = _bufs(b, n)
    sims = [oracle_sim(m, k) for k in ks]
    for s in sims:
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    for t, c in enumerate(episode_schedule()[:45]):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
    assert np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max() < TOL_SENSOR
    assert int(flags.abs().sum()) == 0
    assert b.solver_stats()["ncon"].cpu().tolist() == [s.ncon for s in sims]


@pytest.mark.parametrize("scene,damper", [("softbox_fix", None), ("softbox", None), ("softball", "implicit"), ("softcylinder", "implicit")])
def test_runs_are_bit_reproducible_and_flag_free(scene, damper):
    """two runs from scratch of two consecutive episodes (fresh stiffness draws, reset in between) at 1024 envs: no flag, all finite, and
    the same bits -- nothing on the path depends on scheduling (scripts/soak.py does this at 4096 envs and more episodes).  The ball
    and cylinder scenes with the implicit volume-tendon damper: the whole stiffness range, no env flagged."""
    import torch
    n, outs = 1024, []
    sched = episode_schedule()
    for run in range(2):
        rng = np.random.RandomState(7)
        m, nm, b = _gpu_batch(scene, rng.uniform(300, 1400, n), damper=damper)
        out = torch.zeros(n, len(sched), 12, dtype=torch.float64, device=b.device)
        flags = torch.zeros(n, dtype=torch.int32, device=b.device)
        acc = []
        for ep in range(2):
            if ep:
                b.set_stiffness(rng.uniform(300, 1400, n), JOINT_IDS, TENDON_IDS)
            b.reset(1, flags=flags)
            ctrl = np.zeros(2)
            for t, c in enumerate(sched):
                if c is not None:
                    ctrl[:] = c
                    b.set_ctrl_broadcast(ctrl)
                b.step(7, sens=out[:, t], sens_stride=len(sched) * 12, flags=flags)
                assert int((flags != 0).sum()) == 0, (ep, t)
            acc.append(out.cpu().numpy().copy())
        outs.append(np.stack(acc))
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])


def test_stiffness_far_outside_the_paper_range():
    """k = 1 ... 1e4 (the reference draws 300 ... 1400): same parity as inside the range; k = 1e5 makes the explicit element springs
    unstable at h = 5 ms -- MuJoCo would raise its bad-qacc warning, the oracle and the kernels both flag the env (data, not an error)"""
    ks = [1.0, 10.0, 100.0, 1e4, 1e5]
    m, nm, b = _gpu_batch("softbox_fix", ks)
    sens, flags, touch = _bufs(b, len(ks))
    sims = [oracle_sim(m, k) for k in ks]
    for s in sims:
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    worst = np.zeros(len(ks))
    gflag, oflag = np.zeros(len(ks), int), np.zeros(len(ks), int)
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for i, s in enumerate(sims):
            for _ in range(7):
                oflag[i] |= s.step()
        gflag |= flags.cpu().numpy()
        ok = (gflag == 0) & (oflag == 0)
        d = np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max(1)
        worst[ok] = np.maximum(worst[ok], d[ok])
    assert (gflag[:4] == 0).all() and (oflag[:4] == 0).all() and worst[:4].max() < TOL_SENSOR
    assert gflag[4] != 0 and oflag[4] != 0


def _oracle_episode_with_flags(m, k, mode):
    """(sensor rows [200,12], contact flag per step) of one episode on the oracle, the flag by replaying the reference's own logic
    (manenv.py:65-85) on the oracle's contact list: "intent" = a fresh finger list per call, "reference" = the list the reference
    aliases and never refills (one list per env, as in a fresh process per env)"""
    s = oracle_sim(m, k)
    s.reset(); s.forward(); s.step()
    left = ['g12', 'g2']
    _reference_flag(m, s.contacts(), left if mode == "reference" else list(left))   # reset() ends in step() -> get_sensor_sensordata()
    rows, fl = [], []
    for c in episode_schedule():
        if c is not None:
            s.ctrl[:] = c
        for _ in range(7):
            assert s.step() == 0
        rows.append(s.sensordata.copy())
        fl.append(_reference_flag(m, s.contacts(), left if mode == "reference" else ['g12', 'g2']))
    return np.array(rows), np.array(fl)


@pytest.mark.parametrize("mode", ["intent", "reference"])
def test_contact_flag_and_masked_dataset_match_reference_logic(tmp_path, mode):
    """a5 / f3: ManEnv's contact flag in both modes == the reference's get_sensor_sensordata logic replayed on the oracle's contacts,
    at every step of the episode; and a --mask-contact dataset written on the GPU == the oracle's rows masked by that flag
    (reference create_dataset.py:43-44,57-58).  softbox_fix: the variant that can be compared free-running over 200 steps."""
    import pickle
    import types
    from softgrip_amd import create_dataset as cd
    np.random.seed(11)
    args = types.SimpleNamespace(mujoco_model_paths=[model_path("softbox_fix")], sim_start=1, sim_step=7, vis=False, mask_contact=True,
                                 data_folder=str(tmp_path), data_name="masked_" + mode, n_envs=3, device=0, contact_flag_mode=mode)
    d = pickle.load(open(cd.log_into_file(args), "rb"))
    m = sg.load_model(model_path("softbox_fix"))
    nmasked = 0
    for e in range(3):
        rows, fl = _oracle_episode_with_flags(m, d["stiffness"][e], mode)
        want = rows * fl[:, None]
        got = np.array(d["data"][e])
        assert np.array_equal(np.abs(got).sum(1) == 0, ~fl), "env %d: masked steps differ" % e
        assert np.abs(got - want).max() < TOL_SENSOR
        nmasked += int((~fl).sum())
        assert fl.any() and not fl.all()
    if mode == "reference":   # the aliased list is never refilled: once both fingers have touched, any contact keeps the flag up
        rows, fl_i = _oracle_episode_with_flags(m, d["stiffness"][0], "intent")
        rows, fl_r = _oracle_episode_with_flags(m, d["stiffness"][0], "reference")
        assert fl_r.sum() >= fl_i.sum()


def test_default_scene_dataset_first_steps(tmp_path):
    """the dataset path on the DEFAULT model (composite with its neighbour equalities): rows 0..46 of every env against the oracle
    (beyond, the restated system amplifies round-off), the stored labels are the draws"""
    import pickle
    import types
    from softgrip_amd import create_dataset as cd
    np.random.seed(0)
    draws = np.random.uniform(300, 1400, size=3)
    np.random.seed(0)
    args = types.SimpleNamespace(mujoco_model_paths=[model_path("softbox")], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                 data_folder=str(tmp_path), data_name="ds", n_envs=3, device=0)
    d = pickle.load(open(cd.log_into_file(args), "rb"))
    assert d["stiffness"] == draws.tolist()
    m = sg.load_model(model_path("softbox"))
    assert m.neq == 327
    for e in range(3):
        s = oracle_sim(m, d["stiffness"][e])
        s.reset(); s.forward(); s.step()
        for t, c in enumerate(episode_schedule()[:FREE_RUN_STEPS]):
            if c is not None:
                s.ctrl[:] = c
            for _ in range(7):
                s.step()
            assert np.abs(np.array(d["data"][e])[t] - s.sensordata).max() < TOL_SENSOR, (e, t)


@pytest.mark.parametrize("scene", ["softball", "softcylinder"])
def test_ball_and_cylinder_scenes_need_the_implicit_tendon_damper(scene, tmp_path, capsys):
    """the reference's other two scenes start 0.14 / 0.30 deep in penetration (45 / 37 contacts at reset).  With the volume
    tendon's damper integrated explicitly (MuJoCo's Euler as restated) the start diverges within a few env steps -- oracle and
    kernels alike -- so tendon_damper="explicit" refuses them at load time instead of looping through resets.
    The default, "auto", reloads them with the implicit damper, says so, and a dataset comes out: compared here
    with the oracle over the first 12 env steps of two envs (free-running, hence the short window: the neighbour-row model
    is sensitive to round-off)."""
    from softgrip_amd import ManEnv, SimulationError
    from softgrip_amd import create_dataset as cd
    import pickle
    with pytest.raises(SimulationError, match="does not survive its own idle phase with the explicit tendon damper"):
        ManEnv(1, 7, [model_path(scene)], is_vis=False, n_envs=4, tendon_damper="explicit")
    s = oracle_sim(sg.load_model(model_path(scene)), 700.0)
    s.reset(); s.forward()
    w = s.step()
    for _ in range(70):
        w |= s.step()
    assert w != 0                                    # the oracle agrees: explicit diverges

    args = cd.make_parser().parse_args(["--mujoco-model-paths", model_path(scene), "--n-envs", "4", "--seed", "11",
                                        "--data-folder", str(tmp_path), "--data-name", "d"])
    np.random.seed(11)
    path = cd.log_into_file(args)
    assert "reloading it with tendon_damper=\"implicit\"" in capsys.readouterr().out
    with open(path, "rb") as f:
        d = pickle.load(f)
    assert len(d["data"]) == 4 and np.array(d["data"][0]).shape == (200, 12) and np.isfinite(np.array(d["data"])).all()
    m = sg.load_model(model_path(scene), "implicit")
    for e in (0, 3):
        s = oracle_sim(m, d["stiffness"][e])
        s.reset(); s.forward(); s.step()
        for t, c in enumerate(episode_schedule()[:12]):
            for _ in range(7):
                assert s.step() == 0
            assert np.abs(np.array(d["data"][e])[t] - s.sensordata).max() < 1e-6, (e, t)


def test_default_scene_far_outside_the_paper_range_reseated():
    """the default model (neighbour rows; the solver tracks the tendon row's sum and offset instead of recomputing them) at k = 1, 10 and
    1e4, through reset, idle phase, contact onset and the first 40 steps of the squeeze: per-step parity along the oracle's trajectory
    (re-seated after every env step, as in test_softbox_episode_matches_oracle), contact and sweep counts exactly"""
    import torch
    ks = [1.0, 10.0, 1e4]
    m, nm, b = _gpu_batch("softbox", ks)
    sens, flags, touch = _bufs(b, len(ks))
    sims = [oracle_sim(m, k) for k in ks]
    for s in sims:
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=b.device).contiguous()  # noqa: E731
    for t, c in enumerate(episode_schedule()[:80]):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
        assert np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max() < TOL_SENSOR, t
        assert int(flags.abs().sum()) == 0
        st = b.solver_stats()
        assert st["ncon"].cpu().tolist() == [s.ncon for s in sims] and st["iters"].cpu().tolist() == [s.solver_iter for s in sims]
        b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                    qacc_warmstart=T([s.qacc_warmstart for s in sims]))


def test_three_scene_dataset_like_the_reference_run(tmp_path, capsys):
    """the reference's own usage: one create_dataset run over the box, cylinder and ball scenes (create_dataset.py:23,68-72).  One
    episode-batch per scene, in order; the box keeps MuJoCo's explicit tendon damper, the other two are reloaded with the implicit one
    (announced); every sample finite; each scene's first env reproduces the oracle's first rows for its label."""
    import pickle
    from softgrip_amd import create_dataset as cd
    scenes = ["softbox", "softcylinder", "softball"]
    args = cd.make_parser().parse_args(["--mujoco-model-paths"] + [model_path(s) for s in scenes] + [
        "--n-envs", "6", "--data-folder", str(tmp_path), "--data-name", "three"])
    np.random.seed(5)
    draws = np.random.uniform(300, 1400, size=18)
    np.random.seed(5)
    d = pickle.load(open(cd.log_into_file(args), "rb"))
    assert capsys.readouterr().out.count("reloading it with tendon_damper=\"implicit\"") == 2
    assert len(d["data"]) == 18 and d["stiffness"] == draws.tolist()
    X = np.array(d["data"])
    assert X.shape == (18, 200, 12) and np.isfinite(X).all()
    for i, scene in enumerate(scenes):
        m = sg.load_model(model_path(scene), "explicit" if scene == "softbox" else "implicit")
        s = oracle_sim(m, d["stiffness"][6 * i])
        s.reset(); s.forward(); s.step()
        for t in range(10):
            for _ in range(7):
                assert s.step() == 0
            assert np.abs(X[6 * i, t] - s.sensordata).max() < 1e-6, (scene, t)


@pytest.mark.parametrize("neighbors,pipeline", [(False, "rows"), (False, "fused"), (False, "split"), (True, "rows")])
def test_own_scene_through_the_native_compiler(neighbors, pipeline):
    """tests/data/mini_gripper.xml -- a scene of this repo's own making inside the plan class (34 shell elements, two finger chains,
    h = 0.004, 20 sweeps; not one of the reference's) -- compiled by the library's own MJCF compiler (sg_mjcf_compile, what
    sg_model_compile runs), stepped through the squeeze schedule on the GPU (R = 1 / NSL = 8 kernel instantiations) and compared with
    the oracle built from the same blob: free-running for the fix-rows-only variant, re-seated per step with the neighbour rows."""
    import os
    import torch
    from helpers import ROOT
    from softgrip_amd import native
    from oracle import oracle as O
    path = os.path.join(ROOT, "tests", "data", "mini_gripper.xml")
    m = sg.Model.from_blob(native.compile_mjcf_native(path, composite_neighbors=neighbors))
    assert m.nv == 42 and m.neq == (99 if neighbors else 35)
    jids, tids = list(range(8, 42)), [0]
    ks = [640.0, 300.0, 1400.0, 905.5, 512.25]
    nm = native.NativeModel(m, library_for(pipeline))
    b = native.NativeBatch(nm, len(ks), 0)
    b.set_pipeline(pipeline)
    b.set_stiffness(np.asarray(ks), jids, tids)
    sens, flags, touch = _bufs(b, len(ks))
    om = O.OracleModel(m.to_blob())
    sims = [O.OracleSim(om) for _ in ks]
    for s, k in zip(sims, ks):
        s.jnt_stiffness[jids] = k
        s.tendon_stiffness[tids] = k
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    worst, most = 0.0, 0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
        worst = max(worst, np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max())
        assert worst < TOL_SENSOR, (t, worst)
        assert int(flags.abs().sum()) == 0
        st = b.solver_stats()
        assert st["ncon"].cpu().tolist() == [s.ncon for s in sims]
        assert st["iters"].cpu().tolist() == [s.solver_iter for s in sims]
        most = max(most, max(s.ncon for s in sims))
        if neighbors:
            T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=b.device).contiguous()  # noqa: E731
            b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                        qacc_warmstart=T([s.qacc_warmstart for s in sims]))
    assert most >= 6


def test_c_abi_from_plain_c_matches_the_python_host(tmp_path):
    """examples/c_rollout.c: the boundary used the way a non-Python caller would -- gcc, include/softgrip.h, libsoftgrip.so and the
    HIP runtime, nothing else: sg_model_compile on an XML file, sg_batch_create, sg_set_stiffness / sg_reset / sg_set_ctrl / sg_step
    over the squeeze schedule.  Its sensor rows must be bit-identical to the Python host's on the same inputs."""
    import os
    import subprocess
    from helpers import ROOT
    from softgrip_amd import native
    exe = str(tmp_path / "c_rollout")
    libdir = os.path.join(ROOT, "soft-grip_amd")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "examples", "c_rollout.c"), "-o", exe, "-L", libdir, "-lsoftgrip", "-L", "/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    xml = os.path.join(ROOT, "tests", "data", "mini_gripper.xml")
    n = 8
    out = subprocess.run([exe, xml, str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[-1] == "flagged 0" and len(lines) == 201
    rows = np.array([[float(x) for x in ln.split()[1:]] for ln in lines[:200]])       # [200, 24]: env 0 and env n - 1
    m = sg.Model.from_blob(native.compile_mjcf_native(xml))
    b = native.NativeBatch(native.NativeModel(m), n, 0)
    b.set_stiffness(np.array([300.0 + 1100.0 * e / (n - 1) for e in range(n)]), list(range(m.nv - 34, m.nv)), [0])
    sens, flags, touch = _bufs(b, n)
    b.reset(1, sens=sens, flags=flags, touch=touch)
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            b.set_ctrl_broadcast(np.array([c, c]))
        b.step(7, sens=sens, flags=flags, touch=touch)
        got = sens.cpu().numpy()
        assert np.array_equal(rows[t, :12], got[0]) and np.array_equal(rows[t, 12:], got[n - 1]), t


def test_dataset_from_an_xml_scene_with_its_own_id_set(tmp_path):
    """create_dataset on tests/data/mini_gripper.xml given as an XML path (compiled by mjcf.py at load) with --joint-ids for its
    34-element shell: labels = the seeded draws, rows = the oracle's for those labels (fix-rows-only is not selectable from the command
    line, so the default neighbour-row model is compared over its first steps)"""
    import os
    import pickle
    from helpers import ROOT
    from oracle import oracle as O
    from softgrip_amd import create_dataset as cd
    xml = os.path.join(ROOT, "tests", "data", "mini_gripper.xml")
    args = cd.make_parser().parse_args(["--mujoco-model-paths", xml, "--n-envs", "3", "--data-folder", str(tmp_path), "--data-name", "mini",
                                        "--joint-ids"] + [str(j) for j in range(8, 42)] + ["--tendon-ids", "0"])
    np.random.seed(2)
    draws = np.random.uniform(300, 1400, size=3)
    np.random.seed(2)
    d = pickle.load(open(cd.log_into_file(args), "rb"))
    assert d["stiffness"] == draws.tolist() and np.array(d["data"]).shape == (3, 200, 12) and np.isfinite(np.array(d["data"])).all()
    m = sg.compile_mjcf(xml)
    om = O.OracleModel(m.to_blob())
    for e in range(3):
        s = O.OracleSim(om)
        s.jnt_stiffness[8:] = d["stiffness"][e]
        s.tendon_stiffness[0] = d["stiffness"][e]
        s.reset(); s.forward(); s.step()
        for t, c in enumerate(episode_schedule()[:55]):
            if c is not None:
                s.ctrl[:] = c
            for _ in range(7):
                assert s.step() == 0
            assert np.abs(np.array(d["data"][e])[t] - s.sensordata).max() < 1e-6, (e, t)


@pytest.mark.parametrize("neighbors", [False, True])
def test_random_scenes_in_the_plan_class_on_the_gpu(tmp_path, neighbors):
    """Fuzzing the kernels themselves: 10 seeded variants of tests/data/mini_gripper.xml (tests/test_emu_vs_oracle.py
    _plan_class_variant: box / ellipsoid / cylinder shells of 26 - 68 elements, other link sizes, masses, time steps, sweep counts, so
    other template instantiations and equality schedules), compiled by the native compiler, 3 envs each, 100 env steps through idle,
    closing and squeeze, step by step along the oracle's trajectory (re-seated after every env step: several of these scenes amplify
    round-off): sensors, contact counts and sweep counts at every step."""
    import torch
    from oracle import oracle as O
    from softgrip_amd import native
    from test_emu_vs_oracle import _plan_class_variant
    rng = np.random.RandomState(23)
    touched = left_envelope = 0
    for i in range(10):
        path = tmp_path / ("v%d.xml" % i)
        path.write_text(_plan_class_variant(rng))
        m = sg.Model.from_blob(native.compile_mjcf_native(str(path), composite_neighbors=neighbors))
        jids, tids = list(range(8, m.nv)), [0]
        ks = rng.uniform(300, 1400, 3)
        b = native.NativeBatch(native.NativeModel(m), 3, 0)
        b.set_stiffness(ks, jids, tids)
        sens, flags, touch = _bufs(b, 3)
        om = O.OracleModel(m.to_blob())
        sims = [O.OracleSim(om) for _ in ks]
        for s, k in zip(sims, ks):
            s.jnt_stiffness[jids] = k
            s.tendon_stiffness[tids] = k
            s.reset(); s.forward(); s.step()
        b.reset(1, sens=sens, flags=flags, touch=touch)
        ctrl = np.zeros(2)
        most = 0
        T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=b.device).contiguous()  # noqa: E731
        for t, c in enumerate(episode_schedule()[:100]):
            if c is not None:
                ctrl[:] = c
                b.set_ctrl_broadcast(ctrl)
                for s in sims:
                    s.ctrl[:] = c
            b.step(7, sens=sens, flags=flags, touch=touch)
            for s in sims:
                for _ in range(7):
                    assert s.step() == 0, (i, t)
            # (until r02 a random scene could leave the kernels' envelope -- a slider reaching a static geom, finger boxes about to touch
            # -- and was then reported as SG_FLAG_UNSUPPORTED_PAIR; since r03 the general contact path takes those substeps)
            assert int(flags.abs().sum()) == 0, (i, t, flags.cpu().tolist())
            assert np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max() < TOL_SENSOR, (i, t)
            st = b.solver_stats()
            assert st["ncon"].cpu().tolist() == [s.ncon for s in sims] and st["iters"].cpu().tolist() == [s.solver_iter for s in sims], (i, t)
            b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                        qacc_warmstart=T([s.qacc_warmstart for s in sims]))
            most = max(most, max(s.ncon for s in sims))
        touched += most > 0
        del b
    assert touched >= 7 and left_envelope == 0, (touched, left_envelope)


@pytest.mark.parametrize("neighbors,pipeline", [(False, "rows"), (False, "split"), (False, "fused"), (True, "rows")])
def test_one_slider_under_both_fingers_on_the_gpu(tmp_path, neighbors, pipeline):
    """the solver's `shared` path (tests/test_emu_vs_oracle.py thin_shell_scene: a thin shell whose edge capsules touch BOTH fingers, so the
    two finger streams of an env meet on one slider and are swept one after the other): 130 env steps, 4 envs, against the oracle --
    free-running without the neighbour rows, re-seated with them; the oracle's contact list confirms the situation occurs"""
    import torch
    from oracle import oracle as O
    from softgrip_amd import native
    from test_emu_vs_oracle import elements_touching_both_fingers, thin_shell_scene
    m = sg.Model.from_blob(native.compile_mjcf_native(thin_shell_scene(tmp_path / "thin.xml"), composite_neighbors=neighbors))
    jids, tids = list(range(8, m.nv)), [0]
    ks = [600.0, 300.0, 1400.0, 950.0]
    b = native.NativeBatch(native.NativeModel(m, library_for(pipeline)), len(ks), 0)
    b.set_pipeline(pipeline)
    b.set_stiffness(np.asarray(ks), jids, tids)
    sens, flags, touch = _bufs(b, len(ks))
    om = O.OracleModel(m.to_blob())
    sims = [O.OracleSim(om) for _ in ks]
    for s, k in zip(sims, ks):
        s.jnt_stiffness[jids] = k
        s.tendon_stiffness[tids] = k
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    shared = 0
    T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=b.device).contiguous()  # noqa: E731
    for t, c in enumerate(episode_schedule()[:130]):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
        assert int(flags.abs().sum()) == 0, t
        assert np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max() < TOL_SENSOR, t
        st = b.solver_stats()
        assert st["ncon"].cpu().tolist() == [s.ncon for s in sims] and st["iters"].cpu().tolist() == [s.solver_iter for s in sims], t
        shared += sum(elements_touching_both_fingers(m, s.contacts()) > 0 for s in sims)
        if neighbors:
            b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                        qacc_warmstart=T([s.qacc_warmstart for s in sims]))
    assert shared > 100


@pytest.mark.parametrize("kind,neighbors", [("fingers", False), ("stop", False), ("shelf", False), ("rest", False), ("sledge", False),
                                            ("fingers", True), ("shelf", True), ("rest", True)])
def test_general_contact_path_on_the_gpu(tmp_path, kind, neighbors):
    """a11: the collision pairs outside the fast path's two kinds, in the HIP kernels.  Five variants of the own
    scene (tests/test_emu_vs_oracle.py general_path_scene) in which such a pair becomes active -- the finger tips closing on each
    other (box - box, both finger chains in one constraint row), a static block in a finger's way (box - box against a static box),
    the object resting on a static block (static box - capsule), the object resting on the ground (plane - capsule) and the
    finger tips resting on it (plane - box) -- compiled by the native compiler, 5 envs, 150 env steps against the oracle: sensors,
    contact counts (= the oracle's whole contact list, in its order: the rows and the Gauss-Seidel sweep depend on it) and sweep
    counts at every step, no flag.  The two box - box scenes free-running without the neighbour rows; the scenes with dozens of
    standing contacts from the first step on (shelf, rest, sledge: they amplify round-off, 1e-6 after 115 steps) and the neighbour-row
    variants re-seated per step."""
    import torch
    from oracle import oracle as O
    from softgrip_amd import native
    from test_emu_vs_oracle import general_path_scene, special_contacts
    m = sg.Model.from_blob(native.compile_mjcf_native(general_path_scene(kind, tmp_path / (kind + ".xml")), composite_neighbors=neighbors))
    jids, tids = list(range(8, m.nv)), [0]
    ks = [640.0, 300.0, 1400.0, 950.0, 512.25]
    b = native.NativeBatch(native.NativeModel(m), len(ks), 0)
    b.set_stiffness(np.asarray(ks), jids, tids)
    sens, flags, touch = _bufs(b, len(ks))
    om = O.OracleModel(m.to_blob())
    sims = [O.OracleSim(om) for _ in ks]
    for s, k in zip(sims, ks):
        s.jnt_stiffness[jids] = k
        s.tendon_stiffness[tids] = k
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    special = 0
    bit_of = _touch_bit_of_geom(m)
    T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=b.device).contiguous()  # noqa: E731
    for t, c in enumerate(episode_schedule()[:150]):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
        assert int(flags.abs().sum()) == 0, (t, flags.cpu().tolist())
        assert np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max() < TOL_SENSOR, t
        st = b.solver_stats()
        assert st["ncon"].cpu().tolist() == [s.ncon for s in sims] and st["iters"].cpu().tolist() == [s.solver_iter for s in sims], t
        assert st["nefc"].cpu().tolist() == [s.nefc for s in sims], t
        assert touch.cpu().tolist() == [_expected_touch(m, s.contacts(), bit_of) for s in sims], t
        special += sum(special_contacts(m, s.contacts(), kind) for s in sims)
        if neighbors or kind in ("shelf", "rest", "sledge"):
            b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                        qacc_warmstart=T([s.qacc_warmstart for s in sims]))
    assert special > 300, special
    # the other pipelines have no general path: there the pair is reported as data, never silently dropped
    if not neighbors and kind == "stop":
        b2 = native.NativeBatch(native.NativeModel(m, library_for("split")), 2, 0)
        b2.set_pipeline("split")
        b2.set_stiffness(np.asarray(ks[:2]), jids, tids)
        s2, f2, t2 = _bufs(b2, 2)
        b2.reset(1, sens=s2, flags=f2, touch=t2)
        b2.set_ctrl_broadcast(np.array([-0.2, -0.2]))
        seen = 0
        for _ in range(60):
            b2.step(7, sens=s2, flags=f2, touch=t2)
            seen |= int(f2.max())
        assert seen == 32


@pytest.mark.parametrize("neighbors", [False, True])
def test_general_contact_path_on_a_four_round_model(tmp_path, neighbors):
    """r05s: the phase kernel of models with 193 .. 256 sliders (four element rounds: `sg_phase_kernel<4, ...>`) keeps its pair list and per-slot
    slider pushes in the work space instead of LDS (SG_PHASE_SLIM) -- in the general pass (`<4, ..., GEN = true>`) too, which no
    reference scene runs.  The "rest" scene (the object on the ground plane: every env on the general path in every substep) with a 6 x 9 x 6
    shell (212 sliders), 4 envs, 90 env steps through idle and closing, re-seated on the oracle per step: sensors, contact / row / sweep
    counts and touch bits at every step, no flag."""
    import torch
    from oracle import oracle as O
    from softgrip_amd import native
    from test_emu_vs_oracle import general_path_scene, special_contacts
    path = general_path_scene("rest", tmp_path / "rest212.xml")
    x = open(path).read()
    assert 'count="3 4 3" spacing="0.2"' in x
    open(path, "w").write(x.replace('count="3 4 3" spacing="0.2"', 'count="6 9 6" spacing="0.08"'))
    m = sg.Model.from_blob(native.compile_mjcf_native(path, composite_neighbors=neighbors))
    assert m.nv - 8 == 212
    jids, tids = list(range(8, m.nv)), [0]
    ks = [640.0, 300.0, 1400.0, 950.0]
    b = native.NativeBatch(native.NativeModel(m), len(ks), 0)
    b.set_stiffness(np.asarray(ks), jids, tids)
    sens, flags, touch = _bufs(b, len(ks))
    om = O.OracleModel(m.to_blob())
    sims = [O.OracleSim(om) for _ in ks]
    for s, k in zip(sims, ks):
        s.jnt_stiffness[jids] = k
        s.tendon_stiffness[tids] = k
        s.reset(); s.forward(); s.step()
    b.reset(1, sens=sens, flags=flags, touch=touch)
    ctrl = np.zeros(2)
    special = fast = 0
    bit_of = _touch_bit_of_geom(m)
    T = lambda a: torch.tensor(np.stack(a), dtype=torch.float64, device=b.device).contiguous()  # noqa: E731
    for t, c in enumerate(episode_schedule()[:90]):
        if c is not None:
            ctrl[:] = c
            b.set_ctrl_broadcast(ctrl)
            for s in sims:
                s.ctrl[:] = c
        b.step(7, sens=sens, flags=flags, touch=touch)
        for s in sims:
            for _ in range(7):
                assert s.step() == 0
        assert int(flags.abs().sum()) == 0, (t, flags.cpu().tolist())
        assert np.abs(sens.cpu().numpy() - np.stack([s.sensordata for s in sims])).max() < TOL_SENSOR, t
        st = b.solver_stats()
        assert st["ncon"].cpu().tolist() == [s.ncon for s in sims] and st["iters"].cpu().tolist() == [s.solver_iter for s in sims], t
        assert st["nefc"].cpu().tolist() == [s.nefc for s in sims], t
        assert touch.cpu().tolist() == [_expected_touch(m, s.contacts(), bit_of) for s in sims], t
        sp = sum(special_contacts(m, s.contacts(), "rest") for s in sims)
        special += sp
        fast += sum(s.ncon for s in sims) - sp
        b.set_state(qpos=T([s.qpos for s in sims]), qvel=T([s.qvel for s in sims]), act=T([s.act for s in sims]),
                    qacc_warmstart=T([s.qacc_warmstart for s in sims]))
    assert special > 1000 and fast > 100, (special, fast)   # plane - capsule contacts throughout, finger contacts once the hand closes


@pytest.mark.parametrize("neighbors", [False, True])
def test_general_contact_path_takes_the_whole_batch(tmp_path, neighbors):
    """VERDICT r03 3(d) / ADVICE r03: the general contact path used to take 256 envs per substep and flag the rest (which 256: the order
    of arrival).  The "rest" scene -- the object resting on the ground plane, so EVERY env is on that path in EVERY substep -- at 1 500
    envs of one stiffness: no env flagged, all envs bit-identical whatever block of the general pass served them, env 0 equal to the
    oracle step by step, twice from reset with the same bits."""
    import torch
    from oracle import oracle as O
    from softgrip_amd import native
    from test_emu_vs_oracle import general_path_scene, special_contacts
    m = sg.Model.from_blob(native.compile_mjcf_native(general_path_scene("rest", tmp_path / "rest.xml"), composite_neighbors=neighbors))
    jids, tids = list(range(8, m.nv)), [0]
    n = 1500
    b = native.NativeBatch(native.NativeModel(m), n, 0)
    b.set_stiffness(np.full(n, 640.0), jids, tids)
    sens, flags, touch = _bufs(b, n)
    runs = []
    for rep in range(2):
        s = O.OracleSim(O.OracleModel(m.to_blob()))
        s._om = s.model
        s.jnt_stiffness[jids] = 640.0
        s.tendon_stiffness[tids] = 640.0
        s.reset(); s.forward(); s.step()
        b.reset(1, sens=sens, flags=flags, touch=touch)
        ctrl = np.zeros(2)
        rows, special = [], 0
        for t, c in enumerate(episode_schedule()[:60]):
            if c is not None:
                ctrl[:] = c
                b.set_ctrl_broadcast(ctrl)
                s.ctrl[:] = c
            b.step(7, sens=sens, flags=flags, touch=touch)
            for _ in range(7):
                assert s.step() == 0
            assert int(flags.abs().sum()) == 0, (t, torch.nonzero(flags).flatten()[:8].tolist())
            x = sens.cpu().numpy()
            assert (x == x[0]).all(), t                                          # identical envs: bit-identical, wherever they were served
            st = b.solver_stats()
            assert (st["ncon"] == s.ncon).all() and (st["nefc"] == s.nefc).all() and (st["iters"] == s.solver_iter).all(), t
            assert np.abs(x[0] - s.sensordata).max() < TOL_SENSOR, t
            special += special_contacts(m, s.contacts(), "rest")
            rows.append(x[0].copy())
            qp = torch.tensor(np.tile(s.qpos, (n, 1)), dtype=torch.float64, device=b.device)      # (dozens of standing contacts: re-seated per step)
            qv = torch.tensor(np.tile(s.qvel, (n, 1)), dtype=torch.float64, device=b.device)
            qa = torch.tensor(np.tile(s.act, (n, 1)), dtype=torch.float64, device=b.device)
            qw = torch.tensor(np.tile(s.qacc_warmstart, (n, 1)), dtype=torch.float64, device=b.device)
            b.set_state(qpos=qp, qvel=qv, act=qa, qacc_warmstart=qw)
        assert special >= 60 * 4, special
        runs.append(np.stack(rows))
    assert (runs[0] == runs[1]).all()


def test_bench_two_ranks_with_the_real_library(tmp_path):
    """the multi-process path with the real kernels on its DEFAULT rank synchronisation: `bench.py --gpus 2` starts its own two ranks
    (children of a parent that never touches the GPU), both placed on this box's one GPU (--force-device 0); they meet on a TCP store --
    no process group, no collective library --; per-rank stiffness bins, barrier + max-over-ranks timing, one JSON line from rank 0
    with the whole-job rate.  Then once launched the way the driver does it (torch.distributed.run)."""
    import json
    import os
    import subprocess
    import sys
    from helpers import ROOT
    from softgrip_amd import ranks
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--gpus", "2", "--envs", "512", "--steps", "10", "--warmup", "2", "--force-device", "0", "--no-cpu-baseline"]
    for launcher in ([], ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(ranks.free_port())]):
        out = subprocess.run([sys.executable] + launcher + [os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 512
        assert d["config"]["envs_flagged_bad"] == 0 and d["value"] > 1e4 and "per-rank bins" in d["config"]["workload"]
        assert "no collective library" in d["config"]["rank_sync"] and len(d["ms_per_step_per_rank"]) == 2
        assert abs(d["value"] - 2 * 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


# BASELINE configs[3] is 8 ranks.  This pool kills a job that has more than 6 processes on its GPU at once ("process guard": pytest + 5
# ranks were counted as 7 and killed, r05), and the pytest process itself holds the device -- so the rehearsal with the REAL library
# runs 4 ranks on the one GPU; the same files run at world_size 8 on the CPU (tests/test_dist_gloo.py: the store, the port, the bins
# and the exit-code path at the real rank count).
RANKS_ON_ONE_GPU = 4


def test_bench_many_ranks_with_the_real_library():
    """configs[3] rehearsal: `bench.py --gpus N` self-launched and under torch.distributed.run, N ranks with the
    real library sharing this box's one GPU; one JSON line, N per-rank times, the whole-job aggregate over the slowest rank"""
    import json
    import os
    import subprocess
    import sys
    from helpers import ROOT
    from softgrip_amd import ranks
    N = RANKS_ON_ONE_GPU
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--gpus", str(N), "--envs", "512", "--steps", "10", "--warmup", "2", "--force-device", "0", "--no-cpu-baseline", "--no-fix-variant"]
    for launcher in ([], ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(N), "--master-addr", "127.0.0.1", "--master-port", str(ranks.free_port())]):
        out = subprocess.run([sys.executable] + launcher + [os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == N and d["steps"] == 10 and d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 512
        assert d["config"]["envs_flagged_bad"] == 0 and d["value"] > 1e4 and "per-rank bins" in d["config"]["workload"]
        assert "no collective library" in d["config"]["rank_sync"] and len(d["ms_per_step_per_rank"]) == N
        assert abs(max(d["ms_per_step_per_rank"]) - d["ms_per_step"]) < 1e-9
        assert abs(d["value"] - N * 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_create_dataset_many_ranks_with_the_real_library(tmp_path):
    """configs[3] rehearsal, the dataset job: `python -m softgrip_amd.create_dataset --gpus N` starts its own N ranks on the one GPU, each
    with its own stiffness bin and shard; N shards whose labels fall in N disjoint bins, episode rows finite, and shard r's first
    episode equal to the oracle's for its label (first 40 steps: the idle phase, before the default model amplifies round-off)"""
    import json
    import os
    import pickle
    import subprocess
    import sys
    from helpers import ROOT
    N = RANKS_ON_ONE_GPU
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-m", "softgrip_amd.create_dataset", "--gpus", str(N), "--mujoco-model-paths", model_path("softbox"),
                          "--n-envs", "64", "--total-episodes", str(N * 64 * 2), "--seed", "5", "--force-device", "0", "--data-folder", str(tmp_path / "ds"),
                          "--data-name", "cfg3"], capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == N and res["episodes"] == N * 64 * 2 and res["env_steps"] == N * 64 * 2 * 200 and len(res["shard_bytes"]) == N
    assert res["envs_reset_after_a_warning"] == 0
    w = 1100.0 / N
    m = sg.load_model(model_path("softbox"))
    for r in range(N):
        d = pickle.load(open(tmp_path / "ds" / ("cfg3.rank%d.pickle" % r), "rb"))
        k = np.array(d["stiffness"])
        assert len(k) == 128 and (k >= 300 + w * r).all() and (k < 300 + w * (r + 1)).all() and len(set(k.tolist())) == 128
        rows = np.array(d["data"])
        assert rows.shape == (128, 200, 12) and np.isfinite(rows).all()
        s = oracle_sim(m, float(k[0]))
        s.reset(); s.forward(); s.step()
        for t in range(40):
            for _ in range(7):
                s.step()
            assert np.abs(rows[0, t] - s.sensordata).max() < TOL_SENSOR, (r, t)


def test_config0_one_env_one_episode_on_the_device(tmp_path):
    """BASELINE configs[0] with the HIP path in MuJoCo's place: ManEnv(n_envs=1) through create_dataset.log_into_file,
    the reference's own shape (reference create_dataset.py:33-65, manenv.py:44-53,65-85): step() -> (ndarray(12,) float64, bool),
    reset() -> a Python float, the pickle one (200, 12) array and one float label; all 200 rows against the oracle at 1e-7 (softbox_fix: the
    variant that can be compared free-running), and the contact flag of every step against the oracle's contact list"""
    import pickle
    import types
    from softgrip_amd import ManEnv
    from softgrip_amd import create_dataset as cd
    np.random.seed(0)
    args = types.SimpleNamespace(mujoco_model_paths=[model_path("softbox_fix")], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                 data_folder=str(tmp_path), data_name="one")
    path = cd.log_into_file(args)                      # no n_envs attribute at all: the reference's argparse has none
    d = pickle.load(open(path, "rb"))
    assert set(d) == {"data", "stiffness"} and len(d["data"]) == 1 and len(d["stiffness"]) == 1
    assert type(d["stiffness"][0]) is float and d["stiffness"][0] == 903.6948543200572        # seed 0's first U(300, 1400) draw (SURVEY 8(d) cfg 1)
    rows = d["data"][0]
    assert isinstance(rows, np.ndarray) and rows.shape == (200, 12) and rows.dtype == np.float64
    m = sg.load_model(model_path("softbox_fix"))
    s = oracle_sim(m, d["stiffness"][0])
    s.reset(); s.forward(); s.step()
    # the same episode once more by hand, call by call, for the per-call return types and the contact flag
    np.random.seed(0)
    env = ManEnv(1, 7, [model_path("softbox_fix")], is_vis=False)
    k = env.reset()
    assert isinstance(k, float) and k == d["stiffness"][0]
    nflag = 0
    for t, c in enumerate(episode_schedule()):
        if c is not None:
            s.ctrl[:] = c
            (env.close_hand if c < 0 else env.loose_hand)()
        for _ in range(7):
            s.step()
        readings, contact = env.step()
        assert isinstance(readings, np.ndarray) and readings.shape == (12,) and readings.dtype == np.float64 and type(contact) is bool
        assert np.abs(rows[t] - s.sensordata).max() < TOL_SENSOR, t
        assert np.abs(readings - s.sensordata).max() < TOL_SENSOR, t
        assert contact == _reference_flag(m, s.contacts(), list(ManEnv.finger_names)), t   # the reference's loop (manenv.py:65-85) on the oracle's contact list
        nflag += contact
        r2, c2 = env.get_sensor_sensordata()
        assert np.array_equal(r2, readings) and c2 == contact
    assert 0 < nflag < 200


def test_cached_scene_reload_starts_like_a_fresh_sim():
    """load_env() of a scene seen before reuses its batch -- and must still look like the reference's fresh MjSim:
    sensordata and the contact read-out empty until the first reset / step"""
    from softgrip_amd import ManEnv
    np.random.seed(1)
    env = ManEnv(1, 7, [model_path("softbox_fix"), model_path("softbox")], is_vis=False, n_envs=3)
    env.reset()
    env.close_hand()
    for _ in range(60):
        env.step()
    r, c = env.get_sensor_sensordata()
    assert float(r.abs().max()) > 0 and bool(c.any())
    first = env.env
    env.load_env(1)
    env.load_env(0)
    assert env.env is first                                  # the cached batch
    r, c = env.get_sensor_sensordata()
    assert float(r.abs().max()) == 0.0 and not bool(c.any())
    env.max_cached_scenes = 1
    env.close()
    assert list(env._scenes) == [model_path("softbox_fix")]
