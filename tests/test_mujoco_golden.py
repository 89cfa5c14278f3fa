"""Parity of the oracle with MuJoCo itself -- armed by tests/golden/mujoco_*.npz, which only a machine with a compatible
MuJoCo can produce (scripts/capture_mujoco_golden.py; SURVEY.md 8(c) gate iv).  No capture is committed yet, so this test
SKIPS and the project's parity with MuJoCo is UNPINNED; what is pinned meanwhile: test_oracle_kat.py,
test_mjcf.py, test_harness.py and the HIP-vs-oracle tests."""
import glob
import json
import os

import numpy as np
import pytest

import softgrip_amd as sg
from helpers import ROOT, model_path, oracle_sim

CAPTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mujoco_*.npz")))
TOL = 1e-4  # north_star: sensors within 1e-4 of MuJoCo-CPU


def test_capture_script_schedule_is_the_reference_schedule():
    """the capture script re-implements the 200-step schedule without importing this package; it must equal the one the
    harness fixture (generated from the reference's own create_dataset.py) pins"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cap", os.path.join(ROOT, "scripts", "capture_mujoco_golden.py"))
    cap = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cap)
    from softgrip_amd.create_dataset import episode_schedule
    assert cap.schedule() == episode_schedule()
    assert (cap.SIM_START, cap.SIM_STEP, cap.JOINT_IDS, cap.TENDON_IDS) == (1, 7, list(range(11, 64)), [0])


def _scene_of(meta):
    return meta["xml"].replace("soft_experiments_", "").replace("_adjusted_for_2_fingers.xml", "")


def check_capture(path, model_for_scene=None):
    """the comparison the armed test runs on a capture: model counts, total mass, then every sensor row of every captured episode.
    models/<scene>.sgmodel is the default (neighbour equalities, 327 / 573 / 651 rows); a capture whose `neq` equals the fix-rows-only
    variant's (111 / 193 / 219) settles SURVEY U2 the other way and is held against models/<scene>_fix.sgmodel"""
    d = np.load(path)
    meta = json.loads(str(d["meta"]))
    scene = _scene_of(meta)
    load = model_for_scene or (lambda name: sg.load_model(model_path(name)))
    m = load(scene)
    variant = scene
    fix = load(scene + "_fix")
    if meta["counts"].get("neq") == fix.neq:
        m, variant = fix, scene + "_fix"   # MuJoCo did not create the neighbour equalities
        # (with them the squeeze amplifies round-off, DESIGN 2: expect agreement at TOL over the first ~60 env steps only)
    for k, v in meta["counts"].items():
        assert getattr(m, k, v) == v, "compiled model differs from MuJoCo's in %s" % k
    assert abs(float(np.sum(m.body_mass)) - meta["total_mass"]) < 1e-9
    from softgrip_amd.create_dataset import episode_schedule
    n_steps = d["sensordata"].shape[1] - 1
    for i, k in enumerate(d["stiffness"]):
        s = oracle_sim(m, float(k))
        s.reset(); s.forward(); s.step()
        np.testing.assert_allclose(s.sensordata, d["sensordata"][i, 0], atol=TOL)
        for t, c in enumerate(episode_schedule()[:n_steps]):
            if c is not None:
                s.ctrl[:] = c
            for _ in range(7):
                s.step()
            np.testing.assert_allclose(s.sensordata, d["sensordata"][i, t + 1], atol=TOL, err_msg="env step %d, k=%g" % (t, k))
    if "stage_qpos" in d:
        bad = check_stages(d, m)
        assert not bad, "stage intermediates differ from MuJoCo's: " + "; ".join(bad)
    return variant


def check_stages(d, m, tol=1e-6):
    """The capture's stage intermediates (scripts/capture_mujoco_golden.py stage_snapshot: one forward pass, the first with a contact)
    against the oracle seated on the SAME entering state -- stage by stage, every stage reported, so that one capture says which of
    the open resolutions (DESIGN D1 / D2) is off instead of one sensor mismatch saying that something is:
    contacts (order, geoms, distance, position, normal: D1, D2) -> row order and count -> A + R = J M^-1 J' + R (the rows' Jacobians,
    the regularisers: impedance, diagApprox: U1, U3 - U6) -> b = J qacc_smooth - aref (solref -> K, B; the smooth dynamics) -> the
    solver's forces -> qacc.  Returns the list of stages that differ (empty = all agree)."""
    k = float(d["stiffness"][int(d["stage_where"][0])])
    s = oracle_sim(m, k)
    s.reset()
    s.qpos[:] = d["stage_qpos"]; s.qvel[:] = d["stage_qvel"]; s.act[:] = d["stage_act"]; s.ctrl[:] = d["stage_ctrl"]
    s.qacc_warmstart[:] = d["stage_qacc_warmstart"]
    s.forward()
    bad = []

    def close(name, got, want, t=tol):
        got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
        if got.shape != want.shape:
            bad.append("%s: shape %s vs %s" % (name, got.shape, want.shape))
        elif got.size and not np.allclose(got, want, rtol=t, atol=t):
            bad.append("%s: max |d| %.3g" % (name, np.abs(got - want).max()))
    con = s.contacts()
    geoms = [[c["geom1"], c["geom2"]] for c in con]
    if geoms != d["stage_con_geom"].tolist():
        bad.append("contact list (geom pairs in order): %d vs %d contacts, first difference at %s" % (
            len(geoms), len(d["stage_con_geom"]), next((i for i, (a, b) in enumerate(zip(geoms, d["stage_con_geom"].tolist())) if a != b), "the end")))
    else:
        close("contact dist", [c["dist"] for c in con], d["stage_con_dist"])
        close("contact pos", [c["pos"] for c in con], d["stage_con_pos"])
        close("contact normal", [c["frame"][:3] for c in con], d["stage_con_frame"][:, :3])
    AR, b, ty, ids, mu = s.constraint_problem()
    if ty.tolist() != d["stage_efc_type"].tolist():
        bad.append("row types / order: %d vs %d rows" % (len(ty), len(d["stage_efc_type"])))
        return bad          # the row-wise comparisons below need the same rows
    J, M, R = d["stage_efc_J"], d["stage_qM"], d["stage_efc_R"]
    close("A + R = J M^-1 J' + R", AR, J @ np.linalg.solve(M, J.T) + np.diag(R))
    close("R (diagonal only)", np.diag(AR) - np.einsum("ij,ji->i", J, np.linalg.solve(M, J.T)), R)
    close("b = J qacc_smooth - aref", b, J @ d["stage_qacc_smooth"] - d["stage_efc_aref"])
    close("efc_force", s.efc_force(), d["stage_efc_force"], 1e-5)
    if len(d["stage_qacc"]) == len(s.qacc):
        close("qacc", s.qacc, d["stage_qacc"], 1e-5)
    return bad


@pytest.mark.skipif(not CAPTURES, reason="MuJoCo parity not yet measured: no tests/golden/mujoco_*.npz (run scripts/capture_mujoco_golden.py "
                                         "on a machine with MuJoCo 2.x / mujoco_py and commit its output)")
@pytest.mark.parametrize("path", CAPTURES or ["none"])
def test_oracle_matches_mujoco_capture(path):
    check_capture(path)


def _synthetic_stages(m, k, n_steps):
    """stage_* arrays in the capture script's layout from the ORACLE's first forward pass with a contact.  The oracle does not expose J, M,
    R and aref one by one, so they are fabricated consistently with what it does expose: J = sqrtm(A + R) against a unit mass matrix
    and R = 0 give J M^-1 J' + R = A + R; aref = -b with qacc_smooth = 0 gives b -- enough to execute check_stages' every line"""
    from scipy.linalg import sqrtm
    from softgrip_amd.create_dataset import episode_schedule
    s = oracle_sim(m, float(k))
    s.reset(); s.forward(); s.step()
    for t, c in enumerate(episode_schedule()[:n_steps]):
        if c is not None:
            s.ctrl[:] = c
        for sub in range(7):
            ent = dict(qpos=s.qpos.copy(), qvel=s.qvel.copy(), act=s.act.copy(), ctrl=s.ctrl.copy(), qacc_warmstart=s.qacc_warmstart.copy())
            s.step()
            if s.ncon > 0:
                AR, b, ty, ids, mu = s.constraint_problem()
                con = s.contacts()
                st = {"stage_" + kk: v for kk, v in ent.items()}
                st.update(stage_where=np.array([0, t, sub], dtype=np.int32), stage_con_geom=np.array([[c["geom1"], c["geom2"]] for c in con], dtype=np.int32),
                          stage_con_dist=np.array([c["dist"] for c in con]), stage_con_pos=np.array([c["pos"] for c in con]),
                          stage_con_frame=np.array([c["frame"] for c in con]), stage_efc_type=ty.astype(np.int32), stage_efc_id=ids.astype(np.int32),
                          stage_efc_J=np.real(sqrtm(AR)), stage_qM=np.eye(len(b)), stage_efc_R=np.zeros(len(b)), stage_efc_aref=-b,
                          stage_qacc_smooth=np.zeros(len(b)), stage_efc_force=s.efc_force(), stage_qacc=s.qacc.copy())
                return st
    raise AssertionError("no contact within %d env steps" % n_steps)


def _synthetic_capture(path, m, ks, n_steps, perturb=0.0, stages=None):
    """a file of exactly the layout scripts/capture_mujoco_golden.py writes, filled by the ORACLE instead of MuJoCo -- it pins nothing
    about MuJoCo; it exists so that the armed path above is executed before the day a real capture arrives"""
    from softgrip_amd.create_dataset import episode_schedule
    sens = np.zeros((len(ks), n_steps + 1, 12))
    for i, k in enumerate(ks):
        s = oracle_sim(m, float(k))
        s.reset(); s.forward(); s.step()
        sens[i, 0] = s.sensordata
        for t, c in enumerate(episode_schedule()[:n_steps]):
            if c is not None:
                s.ctrl[:] = c
            for _ in range(7):
                s.step()
            sens[i, t + 1] = s.sensordata
    sens[:, n_steps // 2:, 3] += perturb
    meta = {"mujoco": "synthetic (oracle)", "xml": "soft_experiments_softbox_adjusted_for_2_fingers.xml",
            "counts": {k: int(getattr(m, k)) for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ntendon", "neq", "nsensordata")},
            "total_mass": float(np.sum(m.body_mass))}
    np.savez_compressed(path, stiffness=np.array(ks, dtype=float), sensordata=sens, meta=json.dumps(meta), **(stages or {}))


@pytest.mark.parametrize("variant", ["softbox", "softbox_fix"])
def test_armed_path_runs_on_a_synthetic_capture(tmp_path, variant):
    """the armed comparison must work the day a capture arrives.  Both U2 outcomes: a capture with 327 equality
    rows is held against the default model, one with 111 against the fix-rows-only variant; a capture that differs from the oracle by
    more than 1e-4 fails, and one with other model counts fails on the counts."""
    m = sg.load_model(model_path(variant))
    p = str(tmp_path / "mujoco_synth.npz")
    _synthetic_capture(p, m, [700.0, 903.6948543200572], 14)
    assert check_capture(p) == variant
    _synthetic_capture(p, m, [700.0], 6, perturb=3e-4)
    with pytest.raises(AssertionError, match="env step"):
        check_capture(p)
    d = dict(np.load(p))
    meta = json.loads(str(d["meta"]))
    meta["counts"]["ngeom"] += 1
    d["meta"] = json.dumps(meta)
    np.savez_compressed(p, **d)
    with pytest.raises(AssertionError, match="ngeom"):
        check_capture(p)


def test_stage_comparison_runs_on_a_synthetic_capture(tmp_path):
    """the capture script stores the intermediates of the first forward pass with a contact; check_stages compares
    them with the oracle stage by stage.  Executed here on stages written by the oracle itself (all stages agree), then with single
    stages falsified -- a swapped pair of contacts, one regulariser 1 % off, a reference acceleration off -- each of which must be
    NAMED, alone, in the report: that is what lets one real capture resolve U1 - U6 / D1 / D2 individually."""
    m = sg.load_model(model_path("softbox_fix"))
    st = _synthetic_stages(m, 700.0, 60)
    assert len(st["stage_con_geom"]) >= 1 and st["stage_where"][1] >= 40
    p = str(tmp_path / "mujoco_synth.npz")
    _synthetic_capture(p, m, [700.0], 3, stages=st)
    assert check_capture(p) == "softbox_fix"
    base = dict(np.load(p))
    assert check_stages(base, m) == []

    def report(**changes):
        d = dict(base)
        d.update(changes)
        return check_stages(d, m)
    R = base["stage_efc_R"].copy(); R[5] += 0.01 * base["stage_efc_J"][5] @ base["stage_efc_J"][5]
    rep = report(stage_efc_R=R)
    assert len(rep) == 2 and rep[0].startswith("A + R") and rep[1].startswith("R (diagonal only)"), rep
    aref = base["stage_efc_aref"].copy(); aref[-1] += 1e-3
    rep = report(stage_efc_aref=aref)
    assert len(rep) == 1 and rep[0].startswith("b = J qacc_smooth - aref"), rep
    dist = base["stage_con_dist"].copy(); dist[0] -= 1e-4
    rep = report(stage_con_dist=dist)
    assert len(rep) == 1 and rep[0].startswith("contact dist"), rep
    if len(base["stage_con_geom"]) > 1 and base["stage_con_geom"][0].tolist() != base["stage_con_geom"][1].tolist():
        g = base["stage_con_geom"].copy(); g[[0, 1]] = g[[1, 0]]
        rep = report(stage_con_geom=g)
        assert len(rep) == 1 and rep[0].startswith("contact list"), rep
    rep = report(stage_efc_type=base["stage_efc_type"][:-3])
    assert rep and rep[-1].startswith("row types / order"), rep
    d = dict(base); d["stage_efc_force"] = base["stage_efc_force"] * 1.01
    np.savez_compressed(p, **d)
    with pytest.raises(AssertionError, match="efc_force"):
        check_capture(p)
