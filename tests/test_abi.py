"""The C-ABI library loads and exports every symbol include/softgrip.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

import softgrip_amd as sg
from helpers import ROOT, model_path
from softgrip_amd import native


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "softgrip.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z_]+)\s*\(", hdr))
    assert declared == set(native.SYMBOLS)
    L = native.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.sg_version()


def test_model_create_and_errors():
    m = sg.load_model(model_path("softbox_fix"))
    nm = native.NativeModel(m)
    assert (nm.nq, nm.nu, nm.nsensordata, nm.ntendon, nm.nelem) == (118, 2, 12, 3, 110)
    p = C.c_void_p()
    assert native.lib().sg_model_create(b"garbage" * 10, 70, C.byref(p)) == native.SG_ERR_MODEL
    assert b"blob" in native.lib().sg_last_error()
    # a model outside the supported class is refused with a reason
    bad = sg.compile_mjcf(os.path.join(ROOT, "tests", "data", "slider.xml"))
    with pytest.raises(native.SoftgripError) as ei:
        native.NativeModel(bad)
    assert ei.value.code == native.SG_ERR_MODEL


@pytest.mark.parametrize("scene,nelem", [("softbox", 110), ("softcylinder", 192), ("softball", 218)])
def test_neighbour_row_models_are_accepted(scene, nelem):
    """the plan builder takes the composite's neighbour equalities (two-joint rows, right after their element's fix row) and
    refuses what it cannot schedule: a neighbour row with a polynomial other than q1 = q2"""
    m = sg.load_model(model_path(scene))
    nm = native.NativeModel(m)
    assert nm.nelem == nelem
    import copy
    bad = copy.copy(m)
    bad.eq_data = m.eq_data.copy()
    k = int((m.eq_obj2id >= 0).nonzero()[0][0])
    bad.eq_data[k, 1] = 2.0
    with pytest.raises(native.SoftgripError) as ei:
        native.NativeModel(bad)
    assert ei.value.code == native.SG_ERR_MODEL and "polycoef" in str(ei.value)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    nm = native.NativeModel(sg.load_model(model_path("softbox_fix")))
    with pytest.raises(native.SoftgripError) as ei:
        native.NativeBatch(nm, 4, 0)
    assert ei.value.code == native.SG_ERR_NO_DEVICE
