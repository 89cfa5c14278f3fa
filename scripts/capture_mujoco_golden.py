"""Golden capture for the day a compatible MuJoCo is at hand (SURVEY.md 8(c) gate iv).

Neither MuJoCo nor mujoco_py exists in the build container or on the GPU boxes, so parity of the oracle with MuJoCo itself is
UNPINNED.  This script is the other half of that gate: run it on any machine that has the reference checkout and
either `mujoco_py` (2.0.2.x, what the reference used) or the `mujoco` bindings of a release that still compiles
`<composite type="box|ellipsoid">` (2.x), commit the .npz it writes under tests/golden/, and tests/test_mujoco_golden.py
turns from "skipped: MuJoCo parity not yet measured" into the real parity test (oracle vs MuJoCo, 1e-4 abs on every sensor
sample, plus the compiled-model counts the MJCF compiler must reproduce).

usage: capture_mujoco_golden.py --xml <reference>/data/gripper/soft_experiments_softbox_adjusted_for_2_fingers.xml \
           --stiffness 700 903.6948543200572 300 1400 --out tests/golden/mujoco_softbox.npz
       (once per scene: softbox, softball, softcylinder -- the ball / cylinder captures answer DESIGN.md D5: meta["d5"] holds neq,
       the contacts at reset, the env step of the first simulation warning and the volume tendon's equality force in the idle phase)

Since r04 the capture also holds the STAGE INTERMEDIATES of one forward pass -- the first substep of the first stiffness in which a
contact exists -- as `stage_*` arrays: the state entering that substep (so that the oracle can be seated on exactly it), the contact
list in MuJoCo's order (geoms, dist, pos, frame, friction, solref, solimp, dim), and per constraint row its type / id, position,
margin, R, D, aref, force and its DENSE Jacobian row, plus qM, qacc_smooth, qfrc_bias, qacc.  tests/test_mujoco_golden.py
(check_stages) then compares stage by stage: contact order and geometry, the row order and count, R (the
impedance / diagApprox resolutions U1, U3 - U6), aref, J and A = J M^-1 J' + R, qacc_smooth and the solver's forces -- one
capture resolves the open items one by one instead of as one sensor mismatch.

It drives MuJoCo exactly as reference environment/manenv.py:44-109 and create_dataset.py:33-72 do: stiffness on joints 11..63 and
tendon 0, reset + forward + sim_start(1) steps, then 200 env steps of sim_step(7) substeps under the 40 idle / close at 40 /
toggle at 120 schedule.
"""
import argparse
import json

import numpy as np

JOINT_IDS, TENDON_IDS = list(range(11, 64)), [0]
SIM_START, SIM_STEP, N_STEPS, START_STEP, OPEN_CLOSE_DIV = 1, 7, 200, 40, 80


class PyBackend:  # mujoco_py 2.0.x
    def __init__(self, xml):
        import mujoco_py
        self.mj = mujoco_py
        self.model = mujoco_py.load_model_from_path(xml)
        self.sim = mujoco_py.MjSim(self.model)
        self.data = self.sim.data
        self.version = "mujoco_py " + getattr(mujoco_py, "__version__", "?")

    def reset(self): self.sim.reset()
    def forward(self): self.sim.forward()
    def step(self): self.sim.step()     # mujoco_py raises MujocoException on a simulation warning (what reference manenv.py:50 catches)
    def iters(self): return int(self.data.solver_iter)
    def warnings(self): return [int(w.number) for w in self.data.warning]
    def is_sparse(self): return bool(self.mj.functions.mj_isSparse(self.model))
    def full_M(self):
        M = np.zeros(self.model.nv * self.model.nv)
        self.mj.functions.mj_fullM(self.model, M, self.data.qM)
        return M.reshape(self.model.nv, self.model.nv)


class NewBackend:  # `mujoco` bindings
    def __init__(self, xml):
        import mujoco
        self.mj = mujoco
        self.model = mujoco.MjModel.from_xml_path(xml)
        self.data = mujoco.MjData(self.model)
        self.version = "mujoco " + mujoco.__version__

    def reset(self): self.mj.mj_resetData(self.model, self.data)
    def forward(self): self.mj.mj_forward(self.model, self.data)
    def step(self): self.mj.mj_step(self.model, self.data)
    def iters(self): return int(np.sum(np.atleast_1d(self.data.solver_niter)))
    def warnings(self): return [int(w.number) for w in self.data.warning]
    def is_sparse(self): return bool(self.mj.mj_isSparse(self.model))
    def full_M(self):
        M = np.zeros((self.model.nv, self.model.nv))
        self.mj.mj_fullM(self.model, M, self.data.qM)
        return M


def backend(xml):
    try:
        return PyBackend(xml)
    except ImportError:
        return NewBackend(xml)


def dense_J(B):
    """efc_J of the last forward pass as a dense [nefc, nv] array, whatever the Jacobian's storage"""
    d, nv, n = B.data, B.model.nv, int(B.data.nefc)
    J = np.zeros((n, nv))
    flat = np.asarray(d.efc_J).ravel()
    if B.is_sparse():
        nnz, adr, col = np.asarray(d.efc_J_rownnz).ravel(), np.asarray(d.efc_J_rowadr).ravel(), np.asarray(d.efc_J_colind).ravel()
        for r in range(n):
            J[r, col[adr[r]:adr[r] + nnz[r]]] = flat[adr[r]:adr[r] + nnz[r]]
    else:
        J[:] = flat[:n * nv].reshape(n, nv)
    return J


def stage_snapshot(B, entering):
    """everything one forward pass leaves behind (called right after the mj_step that ran it; `entering`: the state it started from)"""
    d, n, nc = B.data, int(B.data.nefc), int(B.data.ncon)
    st = {"stage_" + k: np.array(v, dtype=float) for k, v in entering.items()}
    con = [d.contact[i] for i in range(nc)]
    st.update({
        "stage_con_geom": np.array([[c.geom1, c.geom2] for c in con], dtype=np.int32).reshape(nc, 2),
        "stage_con_dist": np.array([c.dist for c in con]), "stage_con_pos": np.array([np.asarray(c.pos) for c in con]).reshape(nc, 3),
        "stage_con_frame": np.array([np.asarray(c.frame) for c in con]).reshape(nc, 9),
        "stage_con_friction": np.array([np.asarray(c.friction) for c in con]).reshape(nc, 5),
        "stage_con_solref": np.array([np.asarray(c.solref) for c in con]).reshape(nc, 2),
        "stage_con_solimp": np.array([np.asarray(c.solimp)[:5] for c in con]).reshape(nc, -1),
        "stage_con_dim": np.array([c.dim for c in con], dtype=np.int32), "stage_con_efc_address": np.array([c.efc_address for c in con], dtype=np.int32),
        "stage_con_includemargin": np.array([c.includemargin for c in con]),
    })
    for name in ("efc_type", "efc_id"):
        st["stage_" + name] = np.asarray(getattr(d, name)).ravel()[:n].astype(np.int32)
    for name in ("efc_pos", "efc_margin", "efc_R", "efc_D", "efc_aref", "efc_force", "efc_vel", "efc_diagApprox", "efc_frictionloss"):
        if hasattr(d, name):
            st["stage_" + name] = np.asarray(getattr(d, name)).ravel()[:n].astype(float)
    J = dense_J(B)
    st["stage_efc_J"] = J
    st["stage_efc_J_nnz"] = (J != 0).sum(1).astype(np.int32)
    st["stage_qM"] = B.full_M()
    for name in ("qacc_smooth", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc", "qfrc_constraint", "ten_length", "sensordata"):
        if hasattr(d, name):
            st["stage_" + name] = np.array(getattr(d, name), dtype=float).ravel()
    return st


def schedule():
    """ctrl value to set before env step t (None = unchanged): reference create_dataset.py:46-60"""
    out, closing = [], True
    for t in range(N_STEPS):
        c = None
        if t == START_STEP:
            c, closing = -0.2, False
        elif t > START_STEP and (t - START_STEP) % OPEN_CLOSE_DIV == 0:
            c = 0.2 if not closing else -0.2
            closing = not closing
        out.append(c)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--xml", required=True)
    ap.add_argument("--stiffness", type=float, nargs="+", default=[700.0, 903.6948543200572, 300.0, 1400.0])
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    B = backend(args.xml)
    m = B.model
    meta = {"mujoco": B.version, "xml": args.xml.split("/")[-1],
            "counts": {k: int(getattr(m, k)) for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ntendon", "neq", "nsensordata")},
            "total_mass": float(np.sum(m.body_mass)), "tendon_length0": [float(x) for x in m.tendon_length0],
            "dof_invweight0": [float(x) for x in m.dof_invweight0], "tendon_invweight0": [float(x) for x in m.tendon_invweight0],
            "body_invweight0": np.asarray(m.body_invweight0).tolist(), "opt": {"timestep": float(m.opt.timestep), "iterations": int(m.opt.iterations),
                                                                            "tolerance": float(m.opt.tolerance), "impratio": float(m.opt.impratio)}}
    # what settles SURVEY U2 and DESIGN D5 at once: the number of equality rows the composite compiles to, the
    # contacts the scene starts with, whether a warning fires in the idle phase (ctrl = 0) and the force on the volume tendon's
    # equality row (the last equality) during it
    d5 = {"neq": int(m.neq), "ncon_at_reset": [], "first_warning_env_step": [], "warning_counts": [], "tendon_row_force_idle": []}
    sens = np.zeros((len(args.stiffness), N_STEPS + 1, m.nsensordata))
    ncon = np.zeros((len(args.stiffness), N_STEPS + 1), dtype=np.int32)
    nefc = np.zeros_like(ncon)
    iters = np.zeros_like(ncon)
    qpos = np.zeros((len(args.stiffness), N_STEPS + 1, m.nq))
    stages = {}
    for i, k in enumerate(args.stiffness):
        for j in JOINT_IDS:
            m.jnt_stiffness[j] = k
        for t in TENDON_IDS:
            m.tendon_stiffness[t] = k
        B.reset(); B.forward()
        d5["ncon_at_reset"].append(int(B.data.ncon))
        first_warning, tforce = None, []
        for _ in range(SIM_START):
            B.step()
        sens[i, 0], ncon[i, 0], nefc[i, 0], iters[i, 0], qpos[i, 0] = B.data.sensordata, B.data.ncon, B.data.nefc, B.iters(), B.data.qpos
        for t, c in enumerate(schedule()):
            if c is not None:
                B.data.ctrl[:] = c
            try:
                for sub in range(SIM_STEP):
                    entering = None
                    if i == 0 and not stages:
                        entering = {"qpos": np.array(B.data.qpos), "qvel": np.array(B.data.qvel), "act": np.array(B.data.act), "ctrl": np.array(B.data.ctrl),
                                    "qacc_warmstart": np.array(B.data.qacc_warmstart)}
                    B.step()
                    if entering is not None and int(B.data.ncon) > 0:     # the forward pass of this mj_step ran on `entering` and found a contact
                        stages = stage_snapshot(B, entering)
                        stages["stage_where"] = np.array([i, t, sub], dtype=np.int32)
            except Exception as err:  # noqa: BLE001 -- mujoco_py.builder.MujocoException: the reference would reset() here
                first_warning = first_warning if first_warning is not None else t
                print("k = %g: %s at env step %d" % (k, type(err).__name__, t))
                break
            if first_warning is None and any(B.warnings()):
                first_warning = t
            if t < START_STEP and m.neq > 0 and B.data.nefc >= m.neq:
                tforce.append(float(B.data.efc_force[m.neq - 1]))
            sens[i, t + 1], ncon[i, t + 1], nefc[i, t + 1], iters[i, t + 1], qpos[i, t + 1] = B.data.sensordata, B.data.ncon, B.data.nefc, B.iters(), B.data.qpos
        d5["first_warning_env_step"].append(first_warning)
        d5["warning_counts"].append(B.warnings())
        d5["tendon_row_force_idle"].append(tforce)
    meta["d5"] = d5
    np.savez_compressed(args.out, stiffness=np.array(args.stiffness), sensordata=sens, ncon=ncon, nefc=nefc, iters=iters, qpos=qpos,
                        meta=json.dumps(meta), **stages)
    print("wrote", args.out, meta["mujoco"], meta["counts"])


if __name__ == "__main__":
    main()
