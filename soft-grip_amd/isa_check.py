"""Static check of the device assembly the build produces (hipcc -save-temps: <objdir>/<source>.device.s) for ONE miscompilation
pattern -- the one behind the tree kernel's "dropped stores" of r04, root-caused in r05 (scripts/repro/tree_mono):

    .LBB16_1313:                              ; exit block of a divergent loop: every predecessor arrives with exec = 0
        v_accvgpr_write_b32 a0, v6            ; <- a register-allocator copy that re-establishes a value for the lanes that ran the loop
        v_accvgpr_write_b32 a1, v7            ;    ... executed with NO lane active: it never happens
        s_or_b64 exec, exec, s[0:1]           ; the lanes come back here

LLVM's AMDGPU backend allocates scalar registers first, vector registers afterwards.  A live-range split copy of the SCALAR allocation
placed at the head of a control-flow join block (`$sgpr88 = COPY $sgpr78`, legal in front of the exec restore) stops
MachineBasicBlock::SkipPHIsLabelsAndDebug / SIInstrInfo::isBasicBlockPrologue, which the VECTOR allocation uses to find "the first place
in this block where all lanes are back": its own split copy then lands in front of the scalar copy -- and in front of `S_OR_B64 $exec`.
It takes heavy scalar AND vector register pressure around a divergent loop (the one-function tree kernel: 750 scalar spills); the
product's kernels have no such site (this check is part of every build and of the CPU test suite).

What is flagged: a vector instruction (VALU / LDS / memory; v_readlane / v_writelane excepted: scalar-spill traffic, independent of exec)
between the label of a block that a divergent LOOP EXIT enters (`s_andn2_b64 exec, exec, <done>` + `s_cbranch_execz <label>`) and that
block's `s_or_b64 exec, exec, s[..]`.  At such a block exec is 0 until the restore, so whatever the instruction was put there to do
does not happen; a compiler never has a reason to put one there.

usage: python isa_check.py file.s [...]      (exit code 1 when a site is found)"""
import re
import sys

_VEC = re.compile(r"^(v_|ds_|global_|flat_|scratch_|buffer_)")
_OK = re.compile(r"^v_(readlane|writelane|readfirstlane)_b32")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_FUNC = re.compile(r"^(\.?L?_Z\w+|\w+):\s*(;.*)?$")


def check_asm_text(text):
    """-> list of (function, block label, [(line number, instruction), ...]) for every flagged block"""
    lines = text.split("\n")
    loop_exits = set()
    for k, l in enumerate(lines):
        mm = re.match(r"\s*s_cbranch_execz (\.LBB\d+_\d+)", l)
        if not mm:
            continue
        q = k - 1
        while q > 0 and (not lines[q].strip() or lines[q].strip().startswith((";", ".L", ".loc"))):
            q -= 1
        if re.match(r"\s*s_andn2_b64 exec, exec,", lines[q]):
            loop_exits.add(mm.group(1))
    out, func = [], None
    for i, l in enumerate(lines):
        m = _LABEL.match(l)
        if not m:
            f = _FUNC.match(l)
            if f and not l.startswith("."):
                func = f.group(1)
            elif f and l.startswith((".L_Z", "_Z")):
                func = f.group(1)
            continue
        if m.group(1) not in loop_exits:
            continue
        j, pre = i + 1, []
        while j < len(lines):
            t = lines[j].strip()
            if _LABEL.match(lines[j]) or t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                pre = []          # the block has no exec restore of its own: nothing to say about it
                break
            if re.match(r"s_or_b64 exec, exec, s\[", t):
                break
            if t and not t.startswith((";", ".")) and _VEC.match(t) and not _OK.match(t):
                pre.append((j + 1, t))
            j += 1
        if pre:
            out.append((func, m.group(1), pre))
    return out


def check_asm(path):
    with open(path) as f:
        return check_asm_text(f.read())


def describe(findings, path=""):
    msg = []
    for func, label, pre in findings:
        msg.append("%s: function %s, block %s: %d vector instruction(s) in front of the exec restore of a divergent loop's exit block" % (path, func, label, len(pre)))
        msg += ["    line %d: %s" % p for p in pre[:8]]
    return "\n".join(msg)


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        f = check_asm(p)
        if f:
            print(describe(f, p))
            bad += len(f)
        else:
            print("%s: clean" % p)
    sys.exit(1 if bad else 0)
