"""Builds libsoftgrip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc: one translation unit per kernel family, compiled side
by side, objects cached under soft-grip_amd/build/<variant>/ and re-made only when their source or a header is newer."""
import concurrent.futures
import contextlib
import sys
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)      # isa_check.py (this directory is a package with a hyphen in its name: imported by file)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsoftgrip.so")
LEGACY_LIB = os.path.join(_HERE, "libsoftgrip_legacy.so")
SOURCES = ["sg_api.hip", "sg_phase.hip", "sg_rows.hip", "sg_tree.hip", "sg_plan.cpp", "sg_mjcf.cpp"]   # what the product runs
LEGACY_SOURCES = ["sg_legacy.hip"]   # r01's fused / split pipelines: test builds only (-DSG_LEGACY_PIPELINES)
# -amdgpu-sched-strategy=iterative-ilp: LLVM's iterative ILP machine scheduler instead of the default max-occupancy one.  The
# kernels run at one or two wavefronts per SIMD whatever their register count (the solver by design, the phase kernel by its
# LDS), so trading registers for a shorter dependency-stalled schedule is free: solver -2.5 %, phase kernel -4.9 % per episode
# (profiles/r02_sched_strategy.txt; max-ilp, max-memory-clause and iterative-minreg are slower, iterative-maxocc gains 2.3 %);
# results bit-identical
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
# sg_tree.hip -fno-optimize-sibling-calls (r05): the tree kernel is a driver that CALLS its stage functions and the sweep.  A called function
# saves the callee-saved registers it uses -- 200 - 400 of them to scratch memory at every call, although the driver keeps nothing in them:
# 2 300 scratch instructions per substep and env, ~0.6 MB, most of the tree scenes' fabric traffic.  LLVM drops those saves for a local function
# whose callers are all known (TargetFrameLowering::determineCalleeSaves under IPRA, which the AMDGPU target enables) -- unless a call of it
# carries the IR's `tail` marker, which the tail-call pass puts on nearly every call.  With the pass off for this file the saves are gone:
# 10 678 -> 1 554 scratch instructions in the translation unit (scripts/scratch_audit.py, profiles/r05_tree_scratch_audit.txt).
SOURCE_FLAGS = {"sg_tree.hip": ["-fno-optimize-sibling-calls"]}


def _headers():
    # (this file too: the flags are in it)
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h")) + [os.path.join(CSRC, "sg_kernels.hip"), os.path.abspath(__file__)])


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build(lib=LIB, sources=SOURCES):
    return _stale(lib, [os.path.join(CSRC, s) for s in sources] + _headers())


def build(force=False, verbose=False, prof=False, count=False, legacy=False):
    """prof=True builds libsoftgrip_prof.so with the kernel section stamps (-DSG_SECTION_PROF) for scripts/section_profile.py;
    count=True builds libsoftgrip_count.so, which also counts events inside the contact update (-DSG_SECTION_COUNT: contact
    updates, updates outside the friction cone, QCQP fallback entries and Newton evaluations) -- its cycle stamps are not timings;
    legacy=True builds libsoftgrip_legacy.so: the product plus r01's fused and split pipelines (-DSG_LEGACY_PIPELINES), which the
    cross-check tests load by themselves.  force=True also removes the variant libraries THIS function makes (prof, count, legacy) and
    the object cache -- not the hand-made `--ko NAME` experiment libraries (scripts/dev/ab_*.sh keep an A/B library of an earlier round
    there)."""
    if force:
        with _build_lock():
            for name in OWNED_VARIANTS:
                with contextlib.suppress(FileNotFoundError):
                    os.remove(os.path.join(_HERE, "libsoftgrip_%s.so" % name))
            shutil.rmtree(os.path.join(_HERE, "build"), ignore_errors=True)
    if count:
        # (sg_phase.hip under the default machine scheduler: with iterative-ilp the compiler itself crashes in its register allocator on
        #  this variant of sg_phase_kernel<4, ...> -- r05, ROCm 7.2.0 -- while sg_tree.hip under the DEFAULT scheduler ends in "Illegal
        #  instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base"; the counting build's cycle stamps are not timings anyway)
        return _compile(os.path.join(_HERE, "libsoftgrip_count.so"), ["-DSG_SECTION_PROF", "-DSG_SECTION_COUNT"], verbose, default_sched=("sg_phase.hip",))
    if prof:
        return _compile(os.path.join(_HERE, "libsoftgrip_prof.so"), ["-DSG_SECTION_PROF"], verbose, default_sched=("sg_phase.hip",))   # (as for --count, below)
    if legacy:
        if not force and not needs_build(LEGACY_LIB, SOURCES + LEGACY_SOURCES):
            return LEGACY_LIB
        return _compile(LEGACY_LIB, ["-DSG_LEGACY_PIPELINES"], verbose, SOURCES + LEGACY_SOURCES)
    if not force and not verbose and not needs_build():
        return LIB
    return _compile(LIB, [], verbose)


OWNED_VARIANTS = ("prof", "count", "legacy")


def _compiler_crashed(stderr):
    return "PLEASE submit a bug report" in stderr or "clang frontend command failed due to signal" in stderr or "Segmentation fault" in stderr


@contextlib.contextmanager
def _build_lock():
    """one builder at a time per checkout (ranks, pytest-xdist workers and helpers.library_for may all ask for a build at once): an
    exclusive flock on soft-grip_amd/build/.lock; re-entrant inside one process"""
    if getattr(_build_lock, "depth", 0):
        yield
        return
    os.makedirs(os.path.join(_HERE, "build"), exist_ok=True)
    with open(os.path.join(_HERE, "build", ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        _build_lock.depth = 1
        try:
            yield
        finally:
            _build_lock.depth = 0
            fcntl.flock(f, fcntl.LOCK_UN)


def _compile(out, extra, verbose, sources=SOURCES, default_sched=()):
    with _build_lock():
        return _compile_locked(out, extra, verbose, sources, default_sched)


def _compile_locked(out, extra, verbose, sources, default_sched):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags_all = FLAGS + extra + (["-Rpass-analysis=kernel-resource-usage"] if verbose else [])
    objdir = os.path.join(_HERE, "build", _objdir_name(extra, default_sched))
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()

    def one(src):
        flags = [f for f in flags_all if f not in ("-mllvm", "-amdgpu-sched-strategy=iterative-ilp")] if src in default_sched else flags_all
        own = SOURCE_FLAGS.get(src, [])
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if not verbose and not _stale(obj, [path] + hdrs):
            return obj, ""
        # compiled in a private directory (never a half-written object under the name a linker may pick up); a .hip source with
        # -save-temps, whose by-product -- the device assembly -- is kept beside the object and CHECKED (isa_check.py: the
        # miscompilation pattern behind the tree kernel's dropped stores): a build with such a site fails here
        tmpdir = tempfile.mkdtemp(prefix=os.path.splitext(src)[0] + ".", dir=objdir)
        try:
            tmp = os.path.join(tmpdir, os.path.basename(obj))
            hip = src.endswith(".hip")
            res = subprocess.run([hipcc] + flags + own + (["-save-temps=obj"] if hip else []) + ["-c", "-o", tmp, path], capture_output=True, text=True)
            log = ""
            if res.returncode != 0 and flags is flags_all and _compiler_crashed(res.stderr):
                # ROCm 7.2.0's register allocator segfaults on some variants of the big kernels under iterative-ilp (which variant changes with
                # unrelated edits: r05 saw --prof, --count and then --legacy go while the product built).  The compiler crashing is not an error
                # in the source: compile this one file under the default scheduler (same results, a few per cent slower) and SAY so.
                log = "WARNING: hipcc crashed on %s under -amdgpu-sched-strategy=iterative-ilp; compiled it under the default scheduler instead\n" % src
                sys.stderr.write(log)
                for f in os.listdir(tmpdir):
                    os.remove(os.path.join(tmpdir, f))
                flags = [f for f in flags_all if f not in ("-mllvm", "-amdgpu-sched-strategy=iterative-ilp")]
                res = subprocess.run([hipcc] + flags + own + (["-save-temps=obj"] if hip else []) + ["-c", "-o", tmp, path], capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, res.stderr[-4000:]))
            log += res.stderr
            if hip:
                dev = glob.glob(os.path.join(tmpdir, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
                if len(dev) != 1:
                    raise RuntimeError("no device assembly among the by-products of %s: %s" % (src, sorted(os.listdir(tmpdir))))
                from isa_check import check_asm, describe
                bad = check_asm(dev[0])
                if bad and not os.environ.get("SG_ALLOW_ISA_FINDINGS"):     # (experiment builds of the reproducer set it)
                    raise RuntimeError("the compiler produced the exec-restore miscompilation pattern (soft-grip_amd/isa_check.py) in %s:\n%s\n"
                                       "change the source or the scheduling flags until the site is gone; do NOT ship this object" % (src, describe(bad, src)))
                os.replace(dev[0], os.path.splitext(obj)[0] + ".device.s")
                log += "isa_check: %s clean\n" % src if not bad else describe(bad, src) + "\n"
            os.replace(tmp, obj)
        finally:
            shutil.rmtree(tmpdir, ignore_errors=True)
        return obj, log

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 1)) as ex:
        done = list(ex.map(one, sources))
    fd, tmp = tempfile.mkstemp(suffix=".so", dir=_HERE)
    os.close(fd)
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [o for o, _ in done], capture_output=True, text=True)
    if res.returncode != 0:
        os.remove(tmp)
        raise RuntimeError("hipcc (link) failed:\n" + res.stderr[-4000:])
    os.chmod(tmp, 0o755)
    os.replace(tmp, out)      # a process that has the old library mapped keeps it; a new load sees a whole file
    if verbose:
        print("".join(log for _, log in done))
    return out


def _objdir_name(extra=(), default_sched=()):
    key = FLAGS + list(extra) + list(default_sched) + ["%s:%s" % (k, " ".join(v)) for k, v in sorted(SOURCE_FLAGS.items())]
    return hashlib.sha1(" ".join(key).encode()).hexdigest()[:12]


def device_asm_files(extra=()):
    """the device assembly kept by the last build of the given variant (flags beyond FLAGS), one file per .hip source"""
    objdir = os.path.join(_HERE, "build", _objdir_name(extra))
    return sorted(glob.glob(os.path.join(objdir, "*.device.s")))


if __name__ == "__main__":
    if "--ko" in sys.argv:  # knock-out / experiment builds: --ko NAME -DFLAG ... -> libsoftgrip_NAME.so
        i = sys.argv.index("--ko")
        print(_compile(os.path.join(_HERE, "libsoftgrip_%s.so" % sys.argv[i + 1]), sys.argv[i + 2:], False))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, prof="--prof" in sys.argv, count="--count" in sys.argv, legacy="--legacy" in sys.argv))
