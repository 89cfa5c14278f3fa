import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_XML = "/root/reference/data/gripper/soft_experiments_%s_adjusted_for_2_fingers.xml"
JOINT_IDS, TENDON_IDS = list(range(11, 64)), [0]


def model_path(scene):
    return os.path.join(ROOT, "models", scene + ".sgmodel")


class Emu:
    """ctypes wrapper of tests/emu/libsgemu.so (lane-serial run of the kernels' math)."""

    def __init__(self, blob, nv):
        so = os.path.join(ROOT, "tests", "emu", "libsgemu.so")
        subprocess.check_call(["make", "-C", os.path.dirname(so)], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.emu_new.restype = C.c_void_p
        L.emu_new.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.emu_free.argtypes = [C.c_void_p]
        L.emu_substep.argtypes = [C.c_void_p, C.c_int]
        L.emu_reset.argtypes = [C.c_void_p]
        L.emu_sensordata.restype = C.POINTER(C.c_double)
        L.emu_sensordata.argtypes = [C.c_void_p]
        L.emu_set_ctrl.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.emu_set_jnt_stiffness.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.emu_set_tendon_stiffness.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.emu_get_state.argtypes = [C.c_void_p] + [np.ctypeslib.ndpointer(np.float64)] * 4
        L.emu_set_state.argtypes = [C.c_void_p] + [np.ctypeslib.ndpointer(np.float64)] * 4
        for f in ("emu_ncon", "emu_nefc", "emu_iters"):
            getattr(L, f).argtypes = [C.c_void_p]
        err = C.create_string_buffer(256)
        self.L, self.nv = L, nv
        self.p = L.emu_new(blob, len(blob), err, 256)
        if not self.p:
            raise RuntimeError(err.value.decode())

    def __del__(self):
        if getattr(self, "p", None):
            self.L.emu_free(self.p)

    def set_stiffness(self, k, joint_ids=JOINT_IDS, tendon_ids=TENDON_IDS):
        for j in joint_ids:
            self.L.emu_set_jnt_stiffness(self.p, j, k)
        for t in tendon_ids:
            self.L.emu_set_tendon_stiffness(self.p, t, k)

    def reset(self):
        self.L.emu_reset(self.p)

    def substep(self, integrate=True):
        return self.L.emu_substep(self.p, int(integrate))

    def set_ctrl(self, v):
        for u in range(2):
            self.L.emu_set_ctrl(self.p, u, v)

    @property
    def sensordata(self):
        return np.ctypeslib.as_array(self.L.emu_sensordata(self.p), shape=(12,)).copy()

    def state(self):
        q, v, w, a = (np.zeros(self.nv) for _ in range(3)), None, None, None
        q, v, w = np.zeros(self.nv), np.zeros(self.nv), np.zeros(self.nv)
        a = np.zeros(2)
        self.L.emu_get_state(self.p, q, v, w, a)
        return q, v, w, a

    def set_state(self, qpos, qvel, warm, act):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        self.L.emu_set_state(self.p, c(qpos), c(qvel), c(warm), c(act))

    @property
    def ncon(self):
        return self.L.emu_ncon(self.p)


def oracle_sim(model, k=None):
    from oracle import oracle as O
    om = O.OracleModel(model.to_blob())
    s = O.OracleSim(om)
    s._om = om
    if k is not None:
        s.jnt_stiffness[JOINT_IDS] = k
        s.tendon_stiffness[TENDON_IDS] = k
    return s
