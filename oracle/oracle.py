"""ctypes binding of the CPU oracle (oracle/sg_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  PARITY UNPINNED (see sg_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WARN_BADQPOS, WARN_BADQVEL, WARN_BADQACC, WARN_CONTACTFULL, WARN_CNSTRFULL, WARN_UNSUPPORTED_PAIR = 1, 2, 4, 8, 16, 32


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "sg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.sgo_model_load.restype = vp
        L.sgo_model_load.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.sgo_model_free.argtypes = [vp]
        L.sgo_data_new.restype = vp
        L.sgo_data_new.argtypes = [vp]
        L.sgo_data_free.argtypes = [vp]
        for f in ("sgo_nv", "sgo_nq", "sgo_njnt", "sgo_nu", "sgo_nsensordata", "sgo_ntendon"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp]
        L.sgo_reset.argtypes = [vp, vp]
        for f in ("sgo_forward", "sgo_step"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp, vp]
        for f in ("sgo_qpos", "sgo_qvel", "sgo_act", "sgo_ctrl", "sgo_qacc", "sgo_qacc_warmstart", "sgo_sensordata",
                  "sgo_jnt_stiffness", "sgo_tendon_stiffness", "sgo_ten_length", "sgo_qfrc_bias", "sgo_qM",
                  "sgo_site_xpos", "sgo_efc_force"):
            getattr(L, f).restype = dp
            getattr(L, f).argtypes = [vp]
        for f in ("sgo_efc_AR", "sgo_efc_b"):
            getattr(L, f).restype = dp
            getattr(L, f).argtypes = [vp]
        for f in ("sgo_efc_type", "sgo_efc_id"):
            getattr(L, f).restype = ip
            getattr(L, f).argtypes = [vp]
        L.sgo_contact_friction.argtypes = [vp, C.c_int, dp]
        for f in ("sgo_ncon", "sgo_nefc", "sgo_solver_iter"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp]
        L.sgo_contact.argtypes = [vp, C.c_int, ip, ip, dp, dp, dp]
        L.sgo_step_many.restype = C.c_int
        L.sgo_step_many.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int]
        _LIB = L
    return _LIB


class OracleModel:
    def __init__(self, blob: bytes):
        L = lib()
        err = C.create_string_buffer(256)
        self.ptr = L.sgo_model_load(blob, len(blob), err, 256)
        if not self.ptr:
            raise RuntimeError("oracle: " + err.value.decode())
        self.nv, self.nu = L.sgo_nv(self.ptr), L.sgo_nu(self.ptr)
        self.nq, self.njnt = L.sgo_nq(self.ptr), L.sgo_njnt(self.ptr)      # = nv unless the model has a free joint
        self.nsensordata, self.ntendon = L.sgo_nsensordata(self.ptr), L.sgo_ntendon(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None) and _LIB is not None:
            _LIB.sgo_model_free(self.ptr)
            self.ptr = None


class OracleSim:
    """One env: the oracle's counterpart of mujoco_py.MjSim (reference environment/manenv.py:28)."""

    def __init__(self, model: OracleModel):
        L = lib()
        self.model = model
        self.ptr = L.sgo_data_new(model.ptr)

        def view(fn, n):
            return np.ctypeslib.as_array(getattr(L, fn)(self.ptr), shape=(n,))

        m = model
        self.qpos, self.qvel, self.qacc = view("sgo_qpos", m.nq), view("sgo_qvel", m.nv), view("sgo_qacc", m.nv)
        self.qacc_warmstart = view("sgo_qacc_warmstart", m.nv)
        self.act, self.ctrl = view("sgo_act", m.nu), view("sgo_ctrl", m.nu)
        self.sensordata = view("sgo_sensordata", m.nsensordata)
        self.jnt_stiffness, self.tendon_stiffness = view("sgo_jnt_stiffness", m.njnt), view("sgo_tendon_stiffness", m.ntendon)
        self.ten_length, self.qfrc_bias = view("sgo_ten_length", m.ntendon), view("sgo_qfrc_bias", m.nv)
        self.qM = np.ctypeslib.as_array(L.sgo_qM(self.ptr), shape=(m.nv, m.nv))

    def __del__(self):
        if getattr(self, "ptr", None) and _LIB is not None:
            _LIB.sgo_data_free(self.ptr)
            self.ptr = None

    def reset(self):
        lib().sgo_reset(self.model.ptr, self.ptr)

    def forward(self):
        return lib().sgo_forward(self.model.ptr, self.ptr)

    def step(self):
        return lib().sgo_step(self.model.ptr, self.ptr)

    @property
    def ncon(self):
        return lib().sgo_ncon(self.ptr)

    @property
    def nefc(self):
        return lib().sgo_nefc(self.ptr)

    @property
    def solver_iter(self):
        return lib().sgo_solver_iter(self.ptr)

    def efc_force(self):
        return np.ctypeslib.as_array(lib().sgo_efc_force(self.ptr), shape=(self.nefc,)).copy()

    def constraint_problem(self):
        """(A + R, b, row types, contact id per row, friction per contact) of the last forward pass"""
        n = self.nefc
        AR = np.ctypeslib.as_array(lib().sgo_efc_AR(self.ptr), shape=(n, n)).copy()
        b = np.ctypeslib.as_array(lib().sgo_efc_b(self.ptr), shape=(n,)).copy()
        ty = np.ctypeslib.as_array(lib().sgo_efc_type(self.ptr), shape=(n,)).copy()
        ids = np.ctypeslib.as_array(lib().sgo_efc_id(self.ptr), shape=(n,)).copy()
        mu = np.zeros((self.ncon, 5))
        for i in range(self.ncon):
            buf = (C.c_double * 5)()
            lib().sgo_contact_friction(self.ptr, i, buf)
            mu[i] = np.array(buf)
        return AR, b, ty, ids, mu

    def contacts(self):
        out = []
        g1, g2, dist = C.c_int(), C.c_int(), C.c_double()
        pos, frame = (C.c_double * 3)(), (C.c_double * 9)()
        for i in range(self.ncon):
            lib().sgo_contact(self.ptr, i, C.byref(g1), C.byref(g2), C.byref(dist), pos, frame)
            out.append(dict(geom1=g1.value, geom2=g2.value, dist=dist.value, pos=np.array(pos), frame=np.array(frame)))
        return out


def step_many(model: OracleModel, sims, nsteps, nthreads):
    arr = (C.c_void_p * len(sims))(*[s.ptr for s in sims])
    return lib().sgo_step_many(model.ptr, arr, len(sims), nsteps, nthreads)
