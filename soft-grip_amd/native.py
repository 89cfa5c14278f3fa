"""ctypes binding of libsoftgrip.so (include/softgrip.h).  There is no CPU fallback: every
compute entry point needs the HIP library and a GPU, and fails loudly otherwise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsoftgrip.so")
_LIB = None

SG_OK, SG_ERR_INVALID, SG_ERR_MODEL, SG_ERR_NO_DEVICE, SG_ERR_HIP, SG_ERR_NOMEM = 0, -1, -2, -3, -4, -5
# per-env flags (include/softgrip.h: sg_flag)
SG_FLAG_BADQPOS, SG_FLAG_BADQVEL, SG_FLAG_BADQACC, SG_FLAG_CONTACTFULL, SG_FLAG_CNSTRFULL, SG_FLAG_UNSUPPORTED_PAIR = 1, 2, 4, 8, 16, 32
FLAG_BADQPOS, FLAG_BADQVEL, FLAG_BADQACC, FLAG_CONTACTFULL, FLAG_CNSTRFULL, FLAG_UNSUPPORTED_PAIR = 1, 2, 4, 8, 16, 32

# every symbol include/softgrip.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "sg_last_error", "sg_version", "sg_model_create", "sg_model_destroy", "sg_model_nq", "sg_model_nu",
    "sg_model_nsensordata", "sg_model_ntendon", "sg_model_nelem", "sg_batch_create", "sg_batch_destroy",
    "sg_batch_nenvs", "sg_batch_device", "sg_set_stiffness", "sg_set_ctrl", "sg_reset", "sg_step", "sg_get_state",
    "sg_set_state", "sg_get_solver_stats", "sg_set_pipeline", "sg_profile_enable", "sg_profile_read", "sg_profile_read_solver",
    "sg_model_compile", "sg_mjcf_compile", "sg_blob_free", "sg_set_solver_envs_per_wavefront", "sg_solver_envs_per_wavefront",
    "sg_get_touch_words", "sg_model_nboxes", "sg_model_nv", "sg_model_njnt", "sg_tree_workgroups_per_cu",
]
SG_COMPILE_NO_NEIGHBORS, SG_COMPILE_IMPLICIT_TENDON_DAMPER = 1, 2


class SoftgripError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("softgrip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """the product library (soft-grip_amd/libsoftgrip.so; SOFTGRIP_LIB: a profiling / experiment build of it)"""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsoftgrip.so is missing (%s).  Build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no CPU fallback for the simulator." % LIB_PATH)
        _LIB = load_library(os.environ.get("SOFTGRIP_LIB", LIB_PATH))
    return _LIB


def load_library(path):
    """a build of the library as a configured ctypes handle.  `NativeModel(model, library=...)` binds a model -- and the batches made
    from it -- to another build than the product's (the cross-check tests load the test build with r01's pipelines this way)."""
    import torch  # noqa: F401  -- first: PyTorch-ROCm ships its own HIP runtime, and the process must end up with ONE (loading
    # libsoftgrip.so first would pull in /opt/rocm's copy and torch would then find no devices)
    L = C.CDLL(path)
    vp, dp, ip, i64 = C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong
    L.sg_last_error.restype = C.c_char_p
    L.sg_version.restype = C.c_char_p
    L.sg_model_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.sg_model_destroy.argtypes = [vp]
    L.sg_model_destroy.restype = None
    L.sg_model_compile.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.sg_mjcf_compile.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.sg_blob_free.argtypes = [vp]
    L.sg_blob_free.restype = None
    L.sg_get_touch_words.argtypes = [vp, ip, C.c_int, vp]
    for f in ("sg_model_nq", "sg_model_nv", "sg_model_njnt", "sg_model_nu", "sg_model_nsensordata", "sg_model_ntendon", "sg_model_nelem", "sg_model_nboxes"):
        getattr(L, f).argtypes = [vp]
    L.sg_batch_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.sg_batch_destroy.argtypes = [vp]
    L.sg_batch_destroy.restype = None
    L.sg_batch_nenvs.argtypes = [vp]
    L.sg_batch_device.argtypes = [vp]
    L.sg_set_stiffness.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int, vp]
    L.sg_set_ctrl.argtypes = [vp, vp, C.c_int, vp]
    L.sg_reset.argtypes = [vp, vp, C.c_int, dp, ip, ip, vp]
    L.sg_step.argtypes = [vp, C.c_int, dp, i64, ip, ip, vp]
    L.sg_get_state.argtypes = [vp, dp, dp, dp, dp, dp, vp]
    L.sg_set_state.argtypes = [vp, dp, dp, dp, dp, dp, vp]
    L.sg_get_solver_stats.argtypes = [vp, ip, ip, ip, vp]
    L.sg_set_pipeline.argtypes = [vp, C.c_int]
    L.sg_set_solver_envs_per_wavefront.argtypes = [vp, C.c_int]
    L.sg_solver_envs_per_wavefront.argtypes = [vp]
    L.sg_tree_workgroups_per_cu.argtypes = [vp]
    L.sg_profile_enable.argtypes = [vp, C.c_int]
    L.sg_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.sg_profile_read_solver.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    return L


def check(code, L=None):
    if code != SG_OK:
        raise SoftgripError(code, (L or lib()).sg_last_error().decode())


def compile_mjcf_native(xml_path, composite_neighbors=True, implicit_tendon_damping=False):
    """The library's own MJCF compiler (csrc/sg_mjcf.cpp, ``sg_mjcf_compile``): XML file -> blob bytes.  The Python host uses
    mjcf.py; this is what a caller without Python gets from ``sg_model_compile`` (tests/test_mjcf.py compares the two)."""
    flags = (0 if composite_neighbors else SG_COMPILE_NO_NEIGHBORS) | (SG_COMPILE_IMPLICIT_TENDON_DAMPER if implicit_tendon_damping else 0)
    blob, n = C.c_void_p(), C.c_size_t()
    check(lib().sg_mjcf_compile(os.fsencode(xml_path), flags, C.byref(blob), C.byref(n)))
    try:
        return C.string_at(blob, n.value)
    finally:
        lib().sg_blob_free(blob)


class NativeModel:
    """sg_model handle built from a compiled ``mjcf.Model``."""

    def __init__(self, model, library=None):
        self.model = model
        self.L = L = library or lib()
        blob = model.to_blob()
        self.ptr = C.c_void_p()
        check(L.sg_model_create(blob, len(blob), C.byref(self.ptr)), L)
        self.nq = L.sg_model_nq(self.ptr)
        self.nv = L.sg_model_nv(self.ptr)        # == nq unless the model has a free joint (7 positions, 6 dofs)
        self.nu = L.sg_model_nu(self.ptr)
        self.nsensordata = L.sg_model_nsensordata(self.ptr)
        self.ntendon = L.sg_model_ntendon(self.ptr)
        self.nelem = L.sg_model_nelem(self.ptr)
        self.nboxes = L.sg_model_nboxes(self.ptr)     # moving finger boxes = bits of the contact read-out

    def __del__(self):
        if getattr(self, "ptr", None) and getattr(self, "L", None) is not None:
            self.L.sg_model_destroy(self.ptr)
            self.ptr = None


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class NativeBatch:
    """sg_batch handle; all array arguments are torch tensors on the batch's device."""

    def __init__(self, nmodel: NativeModel, n_envs: int, device: int = 0):
        import torch
        self.torch = torch
        self.nmodel, self.n, self.device_index = nmodel, n_envs, device
        self.L = nmodel.L
        self.ptr = C.c_void_p()
        self._check(self.L.sg_batch_create(nmodel.ptr, n_envs, device, C.byref(self.ptr)))
        self.device = torch.device("cuda", device)

    def _check(self, code):
        check(code, self.L)

    def __del__(self):
        if getattr(self, "ptr", None) and getattr(self, "L", None) is not None:
            self.L.sg_batch_destroy(self.ptr)
            self.ptr = None

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def set_stiffness(self, k, jnt_ids, ten_ids):
        k = np.ascontiguousarray(k, dtype=np.float64)
        assert k.shape == (self.n,)
        ja = (C.c_int * len(jnt_ids))(*jnt_ids)
        ta = (C.c_int * len(ten_ids))(*ten_ids)
        self._check(self.L.sg_set_stiffness(self.ptr, k.ctypes.data_as(C.c_void_p), 1, ja, len(jnt_ids), ta, len(ten_ids), self._stream()))

    def set_ctrl_broadcast(self, ctrl):
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        assert c.shape == (self.nmodel.nu,)
        self._check(self.L.sg_set_ctrl(self.ptr, c.ctypes.data_as(C.c_void_p), 1, self._stream()))

    def set_ctrl(self, ctrl_t):
        assert ctrl_t.is_cuda and ctrl_t.dtype == self.torch.float64 and ctrl_t.shape == (self.n, self.nmodel.nu) and ctrl_t.is_contiguous()
        self._check(self.L.sg_set_ctrl(self.ptr, _ptr(ctrl_t), 0, self._stream()))

    def reset(self, sim_start, sens=None, flags=None, touch=None, mask=None):
        self._check(self.L.sg_reset(self.ptr, _ptr(mask), sim_start, _ptr(sens), _ptr(flags), _ptr(touch), self._stream()))

    def step(self, n_substeps, sens=None, sens_stride=0, flags=None, touch=None):
        self._check(self.L.sg_step(self.ptr, n_substeps, _ptr(sens), sens_stride, _ptr(flags), _ptr(touch), self._stream()))

    def get_state(self):
        t, m = self.torch, self.nmodel
        kw = dict(dtype=t.float64, device=self.device)
        out = dict(qpos=t.empty(self.n, m.nq, **kw), qvel=t.empty(self.n, m.nv, **kw), act=t.empty(self.n, m.nu, **kw),
                   qacc_warmstart=t.empty(self.n, m.nv, **kw), ctrl=t.empty(self.n, m.nu, **kw))
        self._check(self.L.sg_get_state(self.ptr, _ptr(out["qpos"]), _ptr(out["qvel"]), _ptr(out["act"]), _ptr(out["qacc_warmstart"]),
                                 _ptr(out["ctrl"]), self._stream()))
        return out

    def set_state(self, qpos=None, qvel=None, act=None, qacc_warmstart=None, ctrl=None):
        for x in (qpos, qvel, act, qacc_warmstart, ctrl):
            assert x is None or (x.is_cuda and x.dtype == self.torch.float64 and x.is_contiguous())
        self._check(self.L.sg_set_state(self.ptr, _ptr(qpos), _ptr(qvel), _ptr(act), _ptr(qacc_warmstart), _ptr(ctrl), self._stream()))

    def solver_stats(self):
        t = self.torch
        out = [t.empty(self.n, dtype=t.int32, device=self.device) for _ in range(3)]
        self._check(self.L.sg_get_solver_stats(self.ptr, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), self._stream()))
        return dict(ncon=out[0], nefc=out[1], iters=out[2])

    def set_pipeline(self, name):
        self._check(self.L.sg_set_pipeline(self.ptr, {"fused": 0, "split": 1, "rows": 2, "tree": 3}[name]))

    def touch_words(self, nwords=2):
        """[n, nwords] int32: bit g of an env's words = moving finger box g touches an object geom (sg_get_touch_words)"""
        t = self.torch
        out = t.empty(self.n, nwords, dtype=t.int32, device=self.device)
        self._check(self.L.sg_get_touch_words(self.ptr, _ptr(out), nwords, self._stream()))
        return out

    def set_solver_envs_per_wavefront(self, epw):
        self._check(self.L.sg_set_solver_envs_per_wavefront(self.ptr, int(epw)))

    def solver_envs_per_wavefront(self):
        return self.L.sg_solver_envs_per_wavefront(self.ptr)

    def tree_workgroups_per_cu(self):
        """tree pipeline: workgroups (= envs) per CU the runtime grants the kernel with this model's LDS block; 0 on the rows pipeline"""
        return self.L.sg_tree_workgroups_per_cu(self.ptr)

    def profile_enable(self, on=True):
        self._check(self.L.sg_profile_enable(self.ptr, int(on)))

    def profile_read(self, reset=True):
        ms, n = C.c_double(), C.c_longlong()
        self._check(self.L.sg_profile_read(self.ptr, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_read_solver(self, reset=True):
        ms, n = C.c_double(), C.c_longlong()
        self._check(self.L.sg_profile_read_solver(self.ptr, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value
