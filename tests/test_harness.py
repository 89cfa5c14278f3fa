"""Host logic (ManEnv mirror + dataset rollout) against the fixture captured from the REFERENCE's own Python
(scripts/gen_harness_fixture.py -> tests/golden/harness_fixture.json).  A fake native batch records the calls,
so no GPU is needed."""
import json
import os
import pickle
import types

import numpy as np
import pytest
import torch

import softgrip_amd as sg
from helpers import ROOT, model_path
from softgrip_amd import create_dataset as cd
from softgrip_amd import manenv, native

FX = json.load(open(os.path.join(ROOT, "tests", "golden", "harness_fixture.json")))


from fake_native import FakeBatch, FakeModel  # noqa: E402


@pytest.fixture
def fake_native(monkeypatch):
    FakeBatch.log = []
    monkeypatch.setattr(native, "NativeModel", FakeModel)
    monkeypatch.setattr(native, "NativeBatch", FakeBatch)
    return FakeBatch


def _args(tmp, n_envs=1):
    return types.SimpleNamespace(mujoco_model_paths=[model_path("softbox_fix")], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                 data_folder=str(tmp), data_name="fx", n_envs=n_envs, device=0,
                                 check_scene=False)   # the load-time dry run (an extension) would add its steps to the call log


def test_constants_and_class_attrs():
    assert {k: getattr(cd, k) for k in FX["constants"]} == FX["constants"]
    s = FX["set_new_stiffness"]
    assert manenv.ManEnv.joint_ids == s["joint_ids_attr"] and manenv.ManEnv.tendon_ids == s["tendon_ids_attr"]
    assert manenv.ManEnv.finger_names == s["finger_names"] and manenv.ManEnv.obj_name == s["obj_name"]
    assert s["joint_ids_changed"] == s["joint_ids_attr"]          # exactly ids 11..63


def test_rng_draws_match_reference():
    for seed in (0, 1, 1234):
        np.random.seed(seed)
        assert [float(np.random.uniform(300, 1400)) for _ in range(4)] == FX["uniform_300_1400_seed%d" % seed]
        np.random.seed(seed)   # a batch draw consumes the stream exactly like successive scalar draws
        assert np.random.uniform(300, 1400, size=4).tolist() == FX["uniform_300_1400_seed%d" % seed]


def test_episode_schedule_matches_reference_step_log(fake_native, tmp_path):
    np.random.seed(0)
    path = cd.log_into_file(_args(tmp_path))
    sub = [e for e in fake_native.log if e[0] == "substep"]
    runs = []
    for _, c0, c1 in sub:
        if runs and runs[-1][0] == [c0, c1]:
            runs[-1][1] += 1
        else:
            runs.append([[c0, c1], 1])
    assert runs == FX["mj_step_ctrl_runs"] and len(sub) == FX["n_mj_step"]
    st = [e for e in fake_native.log if e[0] == "stiffness"]
    assert st[0][1] == FX["set_new_stiffness"]["joint_ids_attr"] and st[0][2] == FX["set_new_stiffness"]["tendon_ids_attr"]
    d = pickle.load(open(path, "rb"))
    assert sorted(d.keys()) == FX["pickle_keys"]
    x = np.array(d["data"][0])
    assert list(x.shape) == FX["sample_shape"] and str(x.dtype) == FX["sample_dtype"]
    assert d["stiffness"] == FX["stiffness_seed0"]
    # the fake sensors count substeps: sample t holds 1 + 7 (t + 1), as with the reference against the stub simulator
    np.testing.assert_array_equal(x[:, 0], 1 + 7 * (np.arange(200) + 1))
    # fused schedule used by rollout()/bench.py is the same control sequence
    sched = cd.episode_schedule()
    ctrl, seq = 0.0, []
    for c in sched:
        ctrl = ctrl if c is None else c
        seq += [ctrl] * 7
    assert [c0 for _, c0, _ in sub[1:]] == seq


def test_batched_dataset_schema(fake_native, tmp_path):
    np.random.seed(0)
    path = cd.log_into_file(_args(tmp_path, n_envs=4))
    d = pickle.load(open(path, "rb"))
    assert len(d["data"]) == 4 and len(d["stiffness"]) == 4
    assert d["stiffness"] == FX["uniform_300_1400_seed0"]          # env e gets the e-th draw of the reference's stream
    assert all(np.array(x).shape == (200, 12) and np.array(x).dtype == np.float64 for x in d["data"])


def test_manenv_single_env_return_types(fake_native):
    np.random.seed(0)
    env = manenv.ManEnv(1, 7, [model_path("softbox_fix")], is_vis=False)
    k = env.reset()
    assert isinstance(k, float) and k == FX["stiffness_seed0"][0]
    r, c = env.step()
    assert isinstance(r, np.ndarray) and r.shape == (12,) and r.dtype == np.float64 and isinstance(c, bool)
    assert sorted(manenv.ManEnv.get_std_spec(types.SimpleNamespace(sim_start=1, sim_step=7, mujoco_model_paths=["a"], vis=False)).keys()) == FX["std_spec_keys"]
    env.close_hand(); assert env.is_closing and env.env.ctrl.tolist() == [-0.2, -0.2]
    env.toggle_grip(); assert not env.is_closing and env.env.ctrl.tolist() == [0.2, 0.2]
    env.toggle_grip(); assert env.is_closing
    env.load_env(5)   # out of range: prints, like the reference


def test_contact_flag_modes(fake_native):
    env = manenv.ManEnv(1, 7, [model_path("softbox_fix")], is_vis=False, n_envs=3)
    bits = env._chain_geom_bits()
    assert bits == {0: "g122", 1: "g123", 2: "g22", 3: "g23"}
    assert env._finger_bits == [0b0011, 0b1100]                    # 'g12' -> left boxes, 'g2' -> right boxes
    env._touch[:] = torch.tensor([0b0001, 0b0101, 0], dtype=torch.int32)
    assert env._contact_flags().tolist() == [False, True, False]
    # reference mode: the per-env list is never refilled (manenv.py:70-83 aliasing) -> after both fingers touched once,
    # the flag follows ncon > 0
    env2 = manenv.ManEnv(1, 7, [model_path("softbox_fix")], is_vis=False, n_envs=1, contact_flag_mode="reference")
    env2.env.solver_stats = lambda: dict(ncon=torch.tensor([3], dtype=torch.int32))
    env2._touch[:] = 0b0001
    assert env2._contact_flags().tolist() == [False]
    env2._touch[:] = 0b0100
    assert env2._contact_flags().tolist() == [True]
    env2._touch[:] = 0
    assert env2._contact_flags().tolist() == [True]                # ncon > 0 is enough from now on
    env2.env.solver_stats = lambda: dict(ncon=torch.tensor([0], dtype=torch.int32))
    assert env2._contact_flags().tolist() == [False]


def test_sharded_dataset_by_stiffness_bin(fake_native, tmp_path, monkeypatch):
    """configs[3]: every rank draws its envs' stiffness from its own bin and writes its own shard (no collective)"""
    paths = []
    for rank in range(2):
        monkeypatch.setenv("RANK", str(rank)); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("LOCAL_RANK", "0")
        np.random.seed(100 + rank)
        paths.append(cd.log_into_file(_args(tmp_path, n_envs=8)))
    assert [os.path.basename(p) for p in paths] == ["fx.rank0.pickle", "fx.rank1.pickle"]
    k0 = pickle.load(open(paths[0], "rb"))["stiffness"]
    k1 = pickle.load(open(paths[1], "rb"))["stiffness"]
    assert len(k0) == 8 and len(k1) == 8
    assert all(300 <= k < 850 for k in k0) and all(850 <= k < 1400 for k in k1)


def test_incremental_shards_and_resume(fake_native, tmp_path):
    """every finished episode-batch is on disk at once; a restarted run skips finished parts, keeps the RNG
    stream aligned and ends with the same dataset as an uninterrupted run"""
    def args(folder):
        a = _args(folder, n_envs=4)
        a.num_batches, a.incremental = 3, True
        return a
    full = tmp_path / "full"; part = tmp_path / "part"
    np.random.seed(5)
    ref = pickle.load(open(cd.log_into_file(args(full)), "rb"))
    assert len(ref["data"]) == 12 and sorted(os.listdir(full)) == ["fx.part00000.pickle", "fx.part00001.pickle", "fx.part00002.pickle", "fx.pickle", "fx.summary.json"]
    # interrupted run: only the first part exists (as after a crash during batch 1)
    os.makedirs(part)
    import shutil
    shutil.copy(full / "fx.part00000.pickle", part / "fx.part00000.pickle")
    fake_native.log.clear()
    np.random.seed(5)
    got = pickle.load(open(cd.log_into_file(args(part)), "rb"))
    assert got["stiffness"] == ref["stiffness"]
    assert all(np.array_equal(a, b) for a, b in zip(got["data"], ref["data"]))
    assert sum(1 for e in fake_native.log if e[0] == "reset") == 2        # only the two missing batches were simulated


def test_resume_refuses_foreign_parts_and_restores_the_rng(fake_native, tmp_path, monkeypatch):
    """(a) a part left by a run with another seed / batch size / scene list is refused, not merged; (b) a batch in which an
    env failed and had its label re-drawn consumed MORE than n draws -- the resumed run continues from the RNG state stored with the
    part, so its labels equal the uninterrupted run's (counting n draws per skipped batch did not)"""
    import shutil

    def args(folder, **kw):
        a = _args(folder, n_envs=4)
        a.num_batches, a.incremental, a.seed = 3, True, 5
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    calls = {"n": 0, "on": True}
    orig = fake_native._advance

    def failing(self, n, sens, flags, touch):          # env 2 fails once, in the first batch
        orig(self, n, sens, flags, touch)
        calls["n"] += 1
        if calls["on"] and calls["n"] == 30 and flags is not None:
            flags[2] = 4
    monkeypatch.setattr(fake_native, "_advance", failing)
    full = tmp_path / "full"
    np.random.seed(5)
    ref = pickle.load(open(cd.log_into_file(args(full)), "rb"))
    d0 = pickle.load(open(full / "fx.part00000.pickle", "rb"))
    assert set(d0) == {"data", "stiffness", "config", "rng_state"} and d0["config"]["seed"] == 5 and d0["config"]["n_envs"] == 4
    assert set(ref) == {"data", "stiffness"}                       # the final pickle keeps the reference's schema
    # (b) resume after the batch with the re-draw
    calls["on"] = False
    part = tmp_path / "part"
    os.makedirs(part)
    shutil.copy(full / "fx.part00000.pickle", part / "fx.part00000.pickle")
    np.random.seed(5)
    got = pickle.load(open(cd.log_into_file(args(part)), "rb"))
    assert got["stiffness"] == ref["stiffness"]
    # (a) foreign parts
    for kw, what in ((dict(seed=6), "seed"), (dict(n_envs=3), "n_envs"), (dict(mask_contact=True), "mask_contact")):
        other = tmp_path / ("other_" + what)
        os.makedirs(other)
        shutil.copy(full / "fx.part00000.pickle", other / "fx.part00000.pickle")
        with pytest.raises(cd.PartMismatch, match=what):
            cd.log_into_file(args(other, **kw))
    legacy = tmp_path / "legacy"
    os.makedirs(legacy)
    pickle.dump({"data": d0["data"], "stiffness": d0["stiffness"]}, open(legacy / "fx.part00000.pickle", "wb"))
    with pytest.raises(cd.PartMismatch, match="no configuration stored"):
        cd.log_into_file(args(legacy))
    summary = __import__("json").load(open(full / "fx.summary.json"))
    assert summary["episodes"] == 12 and summary["envs_reset_after_a_warning"] == 1 and summary["env_steps"] == 12 * 200


def test_labels_are_the_pre_episode_draw_after_a_mid_episode_failure(fake_native, tmp_path, monkeypatch):
    """reference create_dataset.py:35,65: the stored label is what reset() returned, also when an env failed and was re-drawn
    mid-episode (manenv.py:50-51) -- for one env and for a batch alike; the re-draw stays inside the range of the last draw"""
    calls = {"n": 0}
    orig = fake_native._advance

    def failing(self, n, sens, flags, touch):
        orig(self, n, sens, flags, touch)
        calls["n"] += 1
        if calls["n"] == 12 and flags is not None and not self.nmodel.model.opt_implicit_tendon_damping:
            flags[-1] = 32     # (the load-time check runs on a one-env probe batch since r05)
    monkeypatch.setattr(fake_native, "_advance", failing)
    with pytest.raises(manenv.SimulationError, match="explicit tendon damper .flags UNSUPPORTED_PAIR"):
        manenv.ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=2, tendon_damper="explicit")
    calls["n"] = 0
    manenv.ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=2, check_scene=False, tendon_damper="explicit")


def test_auto_tendon_damper_reloads_a_failing_scene_implicitly(fake_native, monkeypatch, capsys):
    """tendon_damper="auto" (default): explicit first; a scene that fails the dry run under it is reloaded with the implicit
    damper and checked again; a scene that fails under both is refused"""
    orig = fake_native._advance
    env = manenv.ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=2)
    assert env.tendon_damper == "explicit" and env.model.opt_implicit_tendon_damping == 0
    assert manenv.ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=2, tendon_damper="implicit").tendon_damper == "implicit"

    def explicit_fails(self, n, sens, flags, touch):
        orig(self, n, sens, flags, touch)
        if flags is not None and not self.nmodel.model.opt_implicit_tendon_damping:
            flags[0] = 4
    monkeypatch.setattr(fake_native, "_advance", explicit_fails)
    env = manenv.ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=2)
    assert env.tendon_damper == "implicit" and env.nmodel.model.opt_implicit_tendon_damping == 1
    assert "reloading it with tendon_damper=\"implicit\"" in capsys.readouterr().out

    def always_fails(self, n, sens, flags, touch):
        orig(self, n, sens, flags, touch)
        if flags is not None:
            flags[0] = 4
    monkeypatch.setattr(fake_native, "_advance", always_fails)
    with pytest.raises(manenv.SimulationError, match="implicit tendon damper"):
        manenv.ManEnv(1, 7, [model_path("softbox")], is_vis=False, n_envs=2)


def test_multi_scene_loop_matches_reference(fake_native, tmp_path, monkeypatch):
    """three model paths in one run (reference create_dataset.py:23,68-72), against the fixture generated by the reference's own
    log_into_file on a stub simulator: which scene is loaded when (each once, in order, nothing reloaded at the end), one reset and
    1401 mj_steps per scene, one sample per scene, the labels = the first three draws of the seeded stream"""
    M = FX["multi_scene"]
    loads = []
    real = manenv.load_model

    def logging_load(path, tendon_damper=None):
        loads.append(os.path.basename(path))
        return real(path, tendon_damper)
    monkeypatch.setattr(manenv, "load_model", logging_load)
    paths = [model_path("softbox_fix"), model_path("softcylinder_fix"), model_path("softball_fix")]
    args = _args(tmp_path)
    args.mujoco_model_paths = paths
    np.random.seed(0)
    path = cd.log_into_file(args)
    assert loads == [os.path.basename(p) for p in paths] and len(loads) == len(M["loads"])
    assert sum(1 for e in fake_native.log if e[0] == "substep") == M["n_mj_step"]
    assert sum(1 for e in fake_native.log if e[0] == "reset") == M["n_reset"]
    d = pickle.load(open(path, "rb"))
    assert len(d["data"]) == M["n_samples"] and d["stiffness"] == M["stiffness_seed0"]
    assert [float(np.array(x)[0, 0]) for x in d["data"]] == M["first_sensor_of_each_sample"]


def test_load_env_of_a_scene_seen_before_reuses_its_batch(fake_native, monkeypatch, capsys):
    """the reference's loop switches scene after every episode (create_dataset.py:68-72, NUM_EPISODES = 1).  The
    second load_env of a scene must not compile the model, create a batch or dry-run the idle phase again -- its batch and its
    load-time verdict (here: "needs the implicit damper") are kept -- and must leave what a fresh MjSim would: the state after
    mj_resetData, no per-env stiffness, ctrl 0."""
    orig = fake_native._advance

    def ball_fails_explicit(self, n, sens, flags, touch):
        orig(self, n, sens, flags, touch)
        if flags is not None and "softball" in self.nmodel.model_path and not self.nmodel.model.opt_implicit_tendon_damping:
            flags[0] = 4
    loads, made = [], []
    real_load, real_init = manenv.load_model, fake_native.__init__

    def logging_load(path, tendon_damper=None):
        loads.append((os.path.basename(path), tendon_damper))
        m = real_load(path, tendon_damper)
        m._path = path
        return m

    def counting_init(self, nmodel, n_envs, device=0):
        nmodel.model_path = nmodel.model._path
        made.append((os.path.basename(nmodel.model_path), n_envs))
        real_init(self, nmodel, n_envs, device)
    monkeypatch.setattr(manenv, "load_model", logging_load)
    monkeypatch.setattr(fake_native, "__init__", counting_init)
    monkeypatch.setattr(fake_native, "_advance", ball_fails_explicit)
    paths = [model_path("softbox_fix"), model_path("softball_fix")]
    env = manenv.ManEnv(1, 7, paths, is_vis=False, n_envs=2)
    env.load_env(1)
    assert env.tendon_damper == "implicit" and "reloading it" in capsys.readouterr().out
    n_loads, n_made = len(loads), len(made)
    # box: the one-env probe of the load-time check, then the batch; ball: the probe under the explicit damper (fails), the probe and the batch under the implicit one
    assert n_loads == 3 and made == [("softbox_fix.sgmodel", 1), ("softbox_fix.sgmodel", 2), ("softball_fix.sgmodel", 1), ("softball_fix.sgmodel", 1), ("softball_fix.sgmodel", 2)]
    for _ in range(3):                                 # the episode loop: box, ball, box, ball ...
        env.load_env(0)
        assert env.tendon_damper == "explicit" and env.model.opt_implicit_tendon_damping == 0
        env.reset(); env.close_hand(); env.step()
        n_sub = sum(1 for e in fake_native.log if e[0] == "substep")
        env.load_env(1)
        assert sum(1 for e in fake_native.log if e[0] == "substep") == n_sub          # no dry run: not one substep
        assert env.tendon_damper == "implicit" and env.model.opt_implicit_tendon_damping == 1
        assert fake_native.log[-1] == ("reset", 0) and fake_native.log[-2][0] == "stiffness" and fake_native.log[-2][1:3] == ([], [])
        assert np.isnan(env.stiffness).all() and (env._ctrl == 0).all() and (env.env.ctrl == 0).all()
        env.reset(); env.step()
    assert len(loads) == n_loads and len(made) == n_made and capsys.readouterr().out == ""


def test_stiffness_id_sets_can_be_overridden_and_are_checked(fake_native):
    """the reference hard-codes joints 11..63 and tendon 0 (manenv.py:12-13) -- the defaults; another scene layout passes its own ids,
    and ids that do not fit the loaded scene are refused at load time instead of failing inside the library"""
    env = manenv.ManEnv(1, 7, [model_path("softbox_fix")], is_vis=False, n_envs=2, check_scene=False, joint_ids=range(8, 118), tendon_ids=[0])
    env.reset()
    st = [e for e in fake_native.log if e[0] == "stiffness"][-1]
    assert st[1] == list(range(8, 118)) and st[2] == [0]
    assert manenv.ManEnv.joint_ids == list(range(11, 64))           # the class attributes stay the reference's
    with pytest.raises(ValueError, match="do not fit"):
        manenv.ManEnv(1, 7, [model_path("softbox_fix")], is_vis=False, check_scene=False, joint_ids=[5, 118])


def test_contact_capacity_resets_are_counted_apart_and_fail_the_job(fake_native, tmp_path, monkeypatch):
    """an env flagged SG_FLAG_CONTACTFULL is reset like a MuJoCo warning (manenv.py:50-51) but counted in n_capacity_resets; the dataset
    job fails when more than --max-capacity-resets of its episodes end that way"""
    import types
    from softgrip_amd import create_dataset as cd
    orig = fake_native._advance
    hit = {"n": 0}

    def advance(self, n, sens, flags, touch):
        orig(self, n, sens, flags, touch)
        if flags is not None and self.nsub == 1 + 7 * 50 and hit["n"] == 0:     # env step 50 of the first episode-batch: env 2 runs out of contacts
            flags[2] = native.SG_FLAG_CONTACTFULL
            hit["n"] = 1

    monkeypatch.setattr(fake_native, "_advance", advance)
    args = types.SimpleNamespace(mujoco_model_paths=[model_path("softbox_fix")], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                 data_folder=str(tmp_path), data_name="cap", n_envs=4, device=0, check_scene=False)
    np.random.seed(0)
    with pytest.raises(cd.ContactCapacityExceeded):
        cd.log_into_file(args)
    hit["n"] = 0
    args.max_capacity_resets = 0.5
    np.random.seed(0)
    cd.log_into_file(args)
    import json
    s = json.load(open(tmp_path / "cap.summary.json"))
    assert s["envs_reset_at_contact_capacity"] == 1 and s["envs_reset_after_a_warning"] == 1
