"""Generates tests/golden/harness_fixture.json by importing the REFERENCE's own ManEnv and
create_dataset.log_into_file (from /root/reference, build container only) against a stub
``mujoco_py`` that records every call.  Pins the Python-caller rows a2-a7 of SURVEY.md 8(a):
control schedule, step counts, RNG draws, randomised id sets, pickle schema.

The stub emulates NumPy < 1.24 for the one line of the reference that needs a ragged object
array (environment/manenv.py:53); nothing of the reference is copied into this repo.
"""
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "harness_fixture.json")


def make_stub(log):
    mj = types.ModuleType("mujoco_py")
    builder = types.ModuleType("mujoco_py.builder")

    class MujocoException(Exception):
        pass

    builder.MujocoException = MujocoException
    mj.builder = builder

    class Model:
        def __init__(self, path):
            self.path = path
            self.jnt_stiffness = np.full(118, 700.0)
            self.tendon_stiffness = np.array([700.0, 0.0, 0.0])

        def geom_id2name(self, i):
            return "g%d" % i

    class Data:
        def __init__(self):
            self.ctrl = np.zeros(2)
            self.sensordata = np.zeros(12)
            self.ncon = 0
            self.contact = []

    class MjSim:
        def __init__(self, model):
            self.model, self.data = model, Data()
            self.nstep = 0

        def step(self):
            self.nstep += 1
            log["steps"].append((float(self.data.ctrl[0]), float(self.data.ctrl[1])))
            self.data.sensordata[:] = self.nstep

        def reset(self):
            log["calls"].append("reset")
            self.data.ctrl[:] = 0

        def forward(self):
            log["calls"].append("forward")

    def load_model_from_path(p):
        log["calls"].append("load " + p)
        return Model(p)

    mj.load_model_from_path = load_model_from_path
    mj.MjSim = MjSim
    mj.MjViewer = lambda sim: None
    sys.modules["mujoco_py"] = mj
    sys.modules["mujoco_py.builder"] = builder


def main():
    log = {"steps": [], "calls": []}
    make_stub(log)
    sys.path.insert(0, REF)
    # NumPy >= 1.24 refuses the ragged (ndarray(12), bool) -> array conversion of manenv.py:53: emulate the old behaviour
    real_asanyarray = np.asanyarray

    def asanyarray(x, *a, **k):
        if isinstance(x, tuple) and len(x) == 2 and isinstance(x[0], np.ndarray) and isinstance(x[1], (bool, np.bool_)):
            out = np.empty(2, dtype=object)
            out[0], out[1] = x
            return out
        return real_asanyarray(x, *a, **k)

    np.asanyarray = asanyarray
    import create_dataset as ref_cd
    from environment import ManEnv as RefManEnv

    fixture = {}
    for seed in (0, 1, 1234):
        np.random.seed(seed)
        fixture["uniform_300_1400_seed%d" % seed] = [float(np.random.uniform(300, 1400)) for _ in range(4)]
    with tempfile.TemporaryDirectory() as td:
        args = types.SimpleNamespace(mujoco_model_paths=["a.xml"], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                     data_folder=td, data_name="fx")
        np.random.seed(0)
        ref_cd.tqdm = lambda x: x
        ref_cd.log_into_file(args)
        with open(os.path.join(td, "fx.pickle"), "rb") as f:
            d = pickle.load(f)
    steps = log["steps"]
    runs = []
    for c in steps:
        if runs and runs[-1][0] == list(c):
            runs[-1][1] += 1
        else:
            runs.append([list(c), 1])
    fixture["mj_step_ctrl_runs"] = runs
    fixture["n_mj_step"] = len(steps)
    fixture["calls_before_first_step"] = [c for c in log["calls"] if not c.startswith("load ")][:2]
    fixture["pickle_keys"] = sorted(d.keys())
    fixture["sample_shape"] = list(np.array(d["data"][0]).shape)
    fixture["sample_dtype"] = str(np.array(d["data"][0]).dtype)
    fixture["stiffness_seed0"] = [float(x) for x in d["stiffness"]]
    fixture["constants"] = {k: getattr(ref_cd, k) for k in ("NUM_EPISODES", "MAX_ITER_PER_EP", "OPEN_CLOSE_DIV", "START_STEP")}
    # which model entries set_new_stiffness touches
    np.random.seed(0)
    env = RefManEnv(1, 7, ["a.xml"], is_vis=False)
    k = env.set_new_stiffness()
    js = env.env.model.jnt_stiffness
    fixture["set_new_stiffness"] = {"value": float(k), "joint_ids_changed": [int(i) for i in np.flatnonzero(js != 700.0)],
                                    "tendon0": float(env.env.model.tendon_stiffness[0]),
                                    "joint_ids_attr": list(RefManEnv.joint_ids), "tendon_ids_attr": list(RefManEnv.tendon_ids),
                                    "finger_names": ["g12", "g2"], "obj_name": RefManEnv.obj_name}
    fixture["std_spec_keys"] = sorted(RefManEnv.get_std_spec(
        types.SimpleNamespace(sim_start=1, sim_step=7, mujoco_model_paths=["a"], vis=False)).keys())
    # the multi-scene loop (reference create_dataset.py:23,68-72): three model paths in one run -- which scene is loaded when, how many
    # mj_steps each gets, the labels
    log["steps"].clear()
    log["calls"].clear()
    with tempfile.TemporaryDirectory() as td:
        args = types.SimpleNamespace(mujoco_model_paths=["a.xml", "b.xml", "c.xml"], sim_start=1, sim_step=7, vis=False, mask_contact=False,
                                     data_folder=td, data_name="fx3")
        np.random.seed(0)
        ref_cd.log_into_file(args)
        with open(os.path.join(td, "fx3.pickle"), "rb") as f:
            d3 = pickle.load(f)
    fixture["multi_scene"] = {"loads": [c[5:] for c in log["calls"] if c.startswith("load ")], "n_mj_step": len(log["steps"]),
                              "n_reset": log["calls"].count("reset"), "n_samples": len(d3["data"]),
                              "stiffness_seed0": [float(x) for x in d3["stiffness"]],
                              "first_sensor_of_each_sample": [float(np.array(x)[0, 0]) for x in d3["data"]]}
    with open(OUT, "w") as f:
        json.dump(fixture, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
