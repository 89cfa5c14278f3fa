"""Builds libsoftgrip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsoftgrip.so")
SOURCES = ["sg_api.hip", "sg_plan.cpp", "sg_mjcf.cpp"]
DEPS = SOURCES + ["sg_kernels.hip", "sg_split.hip", "sg_tree.hip", "sg_tree.h", "sg_tree_plan.h", "sg_general.h", "sg_math.h", "sg_plan.h", "sg_mjcf.h", "../../include/softgrip.h", "../../include/softgrip_model.h"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, prof=False, count=False):
    """prof=True builds libsoftgrip_prof.so with the kernel section stamps (-DSG_SECTION_PROF) for scripts/section_profile.py;
    count=True builds libsoftgrip_count.so, which also counts events inside the contact update (-DSG_SECTION_COUNT: contact
    updates, updates outside the friction cone, QCQP fallback entries and Newton evaluations) -- its cycle stamps are not timings"""
    if count:
        return _compile(os.path.join(_HERE, "libsoftgrip_count.so"), ["-DSG_SECTION_PROF", "-DSG_SECTION_COUNT"], verbose)
    if prof:
        return _compile(os.path.join(_HERE, "libsoftgrip_prof.so"), ["-DSG_SECTION_PROF"], verbose)
    if not force and not needs_build():
        return LIB
    return _compile(LIB, [], verbose)


def _compile(out, extra, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -amdgpu-sched-strategy=iterative-ilp: LLVM's iterative ILP machine scheduler instead of the default max-occupancy one.  The
    # kernels run at one or two wavefronts per SIMD whatever their register count (the solver by design, the phase kernel by its
    # LDS), so trading registers for a shorter dependency-stalled schedule is free: solver -2.5 %, phase kernel -4.9 % per episode
    # (profiles/r02_sched_strategy.txt; max-ilp, max-memory-clause and iterative-minreg are slower, iterative-maxocc gains 2.3 %);
    # results bit-identical
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-o", out] + extra + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    if verbose:
        print(res.stderr)
    return out


if __name__ == "__main__":
    import sys
    if "--ko" in sys.argv:  # knock-out / experiment builds: --ko NAME -DFLAG ... -> libsoftgrip_NAME.so
        i = sys.argv.index("--ko")
        print(_compile(os.path.join(_HERE, "libsoftgrip_%s.so" % sys.argv[i + 1]), sys.argv[i + 2:], False))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, prof="--prof" in sys.argv, count="--count" in sys.argv))
