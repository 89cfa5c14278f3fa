"""The build's static check for the exec-restore miscompilation (soft-grip_amd/isa_check.py; DESIGN.md 4.10, scripts/repro/tree_mono):
the detector on hand-written assembly, on the device assembly the product build kept, and -- the positive control -- on the site itself,
an excerpt of the assembly of the tree kernel compiled as ONE function, the layout that produced r04's dropped stores."""
import os
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


def test_one_function_tree_kernel_is_flagged():
    """positive control: the site in the build that failed tests/test_gpu_tree.py on the GPU (profiles/r05_tree_mono_*) -- sg_tree.hip with
    -DSGT_X_MONO (one env's whole step as ONE function, r04's layout) under the product's flags at commit d03dd81, sg_tree_kernel<24>: the
    end of the loop and its exit block, as the compiler wrote them (tests/data/tree_mono_d03dd81_site.s).  A committed excerpt and not a
    fresh compile: whether today's source still gets the misplaced copy depends on unrelated edits -- it went away twice during r05 (a
    constant in the limit rows' lookahead, the Newton loop by hand) -- which is the whole point of checking every build.
    scripts/repro/tree_mono/variant.sh compiles the layout afresh and reports what it finds."""
    f = isa_check.check_asm(os.path.join(ROOT, "tests", "data", "tree_mono_d03dd81_site.s"))
    assert len(f) == 1 and "sg_tree_kernelILi24" in f[0][0] and f[0][1] == ".LBB16_1313", isa_check.describe(f, "mono")
    assert [t for _, t in f[0][2]] == ["v_accvgpr_write_b32 a0, v6", "v_accvgpr_write_b32 a1, v7"]


def test_tree_stage_functions_save_no_callee_saved_registers():
    """the tree kernel's called functions (stages, sweeps) used to save the callee-saved registers they use to scratch memory at
    every call -- 10 678 scratch instructions in the translation unit, ~0.6 MB per substep and env, more than half of the tree scenes' fabric
    traffic.  LLVM's no-CSR optimisation removes them as long as no call of these functions carries the IR's `tail` marker
    (build_native.SOURCE_FLAGS: -fno-optimize-sibling-calls for sg_tree.hip).  Held here on the assembly the product build kept: the whole
    unit below 2 500 scratch instructions and every sweep function below 40 (they have 4 - 11), so a flag or compiler change that brings
    the saves back fails the CPU suite instead of showing up as traffic."""
    import re
    from softgrip_amd import build_native
    build_native.build()
    tree = [f for f in build_native.device_asm_files() if os.path.basename(f) == "sg_tree.device.s"]
    assert len(tree) == 1
    total, per, cur = 0, {}, None
    with open(tree[0]) as f:
        for line in f:
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
            elif cur and line.lstrip().startswith("scratch_"):
                total += 1
                per[cur] = per.get(cur, 0) + 1
    sweeps = {k: v for k, v in per.items() if "tree_sweep" in k}
    assert total < 2500, total
    assert all(v < 40 for v in sweeps.values()), sweeps


def test_phase_kernel_lds_blocks_fit_eight_workgroups_per_cu():
    """a launch of the phase kernel is ROUNDS of wavefronts x one wavefront's life, so a workgroup per CU less is a whole
    round more -- the ball's instantiation (four element rounds) at 22.6 KB of LDS ran seven per CU and its 4096 wavefronts in three rounds
    instead of two until its pair list and slot pushes moved to the work space (SG_PHASE_SLIM).  Held here on the assembly the product
    build kept: every instantiation of sg_phase_kernel asks for at most 160 KB / 8 of LDS and for at most 256 registers (two wavefronts per
    SIMD = eight workgroups per CU), so an array added to Smem2 fails the CPU suite instead of showing up as a third round."""
    import re
    from softgrip_amd import build_native
    build_native.build()
    phase = [f for f in build_native.device_asm_files() if os.path.basename(f) == "sg_phase.device.s"]
    assert len(phase) == 1
    text = open(phase[0]).read()
    seen = {}
    for block in text.split("  - .agpr_count:")[1:]:           # one metadata entry per kernel
        name = re.search(r"\.name:\s+(\S+)", block)
        lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", block)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", block)
        if name and lds and vg and "sg_phase_kernel" in name.group(1):
            seen[name.group(1)] = (int(lds.group(1)), int(vg.group(1)))
    assert len(seen) >= 16, sorted(seen)                        # four element rounds x (neighbour rows or not) x (main pass, general pass)
    for k, (lds, vg) in seen.items():
        assert lds <= 160 * 1024 // 8, (k, lds)
        assert vg <= 256, (k, vg)
    assert any("ILi4E" in k for k in seen)                      # the four-round instantiations are among them
