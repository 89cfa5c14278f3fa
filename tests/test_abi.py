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


def test_asymmetric_box_counts_leave_the_two_finger_class(tmp_path):
    """the two-finger kernels number the finger boxes 2 * chain + box; with another count per chain that is not the geom-id
    order the C ABI documents for touch_out / sg_get_touch_words (and ManEnv._chain_geom_bits enumerates).  The two-finger plan
    therefore takes chains with exactly two boxes only; a gripper with three boxes on one finger -- or one -- runs in the tree
    pipeline, which numbers the boxes flat in geom-id order: the model is accepted, sg_model_nboxes counts all of them, and the
    two-finger plan's reason says why it is not the fast kernels' model."""
    import subprocess
    x = open(os.path.join(ROOT, "tests", "data", "mini_gripper.xml")).read()
    extra = '<geom class="link" name="fL2" size="0.25 0.08 0.2" mass="0.06"/>'
    assert extra in x
    three = x.replace(extra, extra + '\n            <geom class="link" name="fL3" pos="0.3 0 0" size="0.05 0.08 0.2" mass="0.01"/>')
    p = tmp_path / "three.xml"
    p.write_text(three)
    m = sg.compile_mjcf(str(p), composite_neighbors=False)
    nm = native.NativeModel(m)                                   # accepted: the tree pipeline's model
    assert nm.nboxes == 5
    moving = [g for g in range(m.ngeom) if m.body_weldid[m.geom_bodyid[g]] != 0 and m.geom_type[g] == 6]
    assert [m.geom_names[g] for g in moving] == ["fL1", "fL2", "fL3", "fR1", "fR2"]     # bit g of the touch words = box g in this order
    # the two-finger plan on the same blob: refused, with the reason
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    E = C.CDLL(os.path.join(ROOT, "tests", "emu", "libsgemu.so"))
    blob = m.to_blob()
    out = (C.c_int * (4 * 8 * 4096))()
    nelem, nnb, err = C.c_int(), C.c_int(), C.create_string_buffer(256)
    E.emu_plan_schedule.restype = C.c_int
    assert E.emu_plan_schedule(blob, C.c_size_t(len(blob)), out, 8 * 4096, C.byref(nelem), C.byref(nnb), err, C.c_size_t(256)) <= 0
    assert b"box geoms" in err.value, err.value
