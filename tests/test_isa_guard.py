"""The build's static check for the exec-restore miscompilation (soft-grip_amd/isa_check.py; DESIGN.md 4.10, scripts/repro/tree_mono):
the detector on hand-written assembly, on the device assembly the product build kept, and -- the positive control -- on the tree
kernel compiled as ONE function, the layout that produced r04's dropped stores."""
import os
import shutil
import subprocess
import sys

import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, "soft-grip_amd"))
import isa_check  # noqa: E402

BAD = """
_Z6kernelv:
.LBB0_1:
	v_add_f64 v[0:1], v[0:1], v[2:3]
	s_andn2_b64 exec, exec, s[4:5]
	s_cbranch_execz .LBB0_3
	s_branch .LBB0_1
.LBB0_3:
	v_readlane_b32 s0, v255, 3
	v_accvgpr_write_b32 a0, v6
	s_mov_b32 s88, s78
	s_or_b64 exec, exec, s[0:1]
	v_add_f64 v[4:5], v[0:1], v[6:7]
	s_endpgm
"""
GOOD = BAD.replace("\tv_accvgpr_write_b32 a0, v6\n", "").replace("\tv_add_f64 v[4:5], v[0:1], v[6:7]\n", "\tv_accvgpr_write_b32 a0, v6\n\tv_add_f64 v[4:5], v[0:1], v[6:7]\n")
IF_JOIN = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
	v_mov_b32_e32 v1, v2
.LBB0_2:
	v_mov_b32_e32 v3, v1
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
"""


def test_detector_on_handwritten_assembly():
    f = isa_check.check_asm_text(BAD)
    assert len(f) == 1 and f[0][1] == ".LBB0_3" and [t for _, t in f[0][2]] == ["v_accvgpr_write_b32 a0, v6"]
    assert isa_check.check_asm_text(GOOD) == []            # the same copy behind the exec restore; scalar-spill traffic in front of it is fine
    assert isa_check.check_asm_text(IF_JOIN) == []         # an `if` join (lanes of the taken branch are active there): not this pattern


def test_product_device_assembly_is_clean():
    """every .hip translation unit of the product build: no vector instruction in front of a loop exit's exec restore"""
    from softgrip_amd import build_native
    build_native.build()
    files = build_native.device_asm_files()
    if len(files) < 4:      # objects of a build older than the check: rebuild them once
        build_native.build(verbose=True)
        files = build_native.device_asm_files()
    assert sorted(os.path.basename(f) for f in files) == ["sg_api.device.s", "sg_phase.device.s", "sg_rows.device.s", "sg_tree.device.s"]
    for f in files:
        assert isa_check.check_asm(f) == [], isa_check.describe(isa_check.check_asm(f), f)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_one_function_tree_kernel_is_flagged(tmp_path):
    """positive control: sg_tree.hip with -DSGT_X_MONO (one env's whole step as ONE function, r04's layout) under the product's flags
    gets the misplaced copy in sg_tree_kernel<24> -- the build that failed tests/test_gpu_tree.py on the GPU (profiles/r05_tree_mono_*)"""
    from softgrip_amd import build_native
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "mono.s"
    res = subprocess.run([hipcc] + build_native.FLAGS + ["--cuda-device-only", "-DSGT_X_MONO", "-S", "-o", str(out), os.path.join(build_native.CSRC, "sg_tree.hip")],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    f = isa_check.check_asm(str(out))
    assert len(f) == 1 and "sg_tree_kernelILi24" in f[0][0], isa_check.describe(f, "mono")
    assert all(t.startswith("v_accvgpr_write_b32") for _, t in f[0][2])
