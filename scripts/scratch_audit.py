"""Where a kernel's scratch memory goes (r05): per function of a device assembly file (the build keeps them:
soft-grip_amd/build/<hash>/<source>.device.s) the private segment size, the scratch instructions in the prologue / epilogue (callee-saved
registers of a CALLED function: the price of the staged layout) and the scratch instructions inside loops (what could cost time).
usage: python scripts/scratch_audit.py file.device.s [name filter]"""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
funcs, cur = [], None
for i, l in enumerate(lines):
    m = re.match(r"^(\.?L?_Z\w+|\w+):\s*(;.*)?$", l)
    if m and not l.startswith(".LBB") and (l.startswith("_Z") or l.startswith(".L_Z")):
        cur = {"name": m.group(1), "start": i}
        funcs.append(cur)
    if l.startswith("; codeLenInByte") and cur is not None and "end" not in cur:
        cur["end"] = i
        for j in range(i, min(i + 12, len(lines))):
            mm = re.match(r";\s*ScratchSize:\s*(\d+)", lines[j])
            if mm:
                cur["scratch"] = int(mm.group(1))
print("%-58s %9s %8s %10s %10s %9s" % ("function", "bytes/lane", "instr", "scratch ops", "in loops", "in entry/exit"))
for f in funcs:
    if "end" not in f or flt not in f["name"]:
        continue
    body = lines[f["start"]:f["end"]]
    nins = nscr = nloop = nedge = 0
    inloop, blk = False, 0
    nblocks = sum(1 for l in body if re.match(r"^\.LBB\d+_\d+:", l))
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blk += 1
            inloop = "Loop" in l
            continue
        t = l.strip()
        if "Loop" in l and t.startswith(";"):
            inloop = True
        if not t or t.startswith((";", ".")):
            continue
        nins += 1
        if t.startswith("scratch_") or ("offen" in t and t.startswith("buffer_")):
            nscr += 1
            if inloop:
                nloop += 1
            if blk == 0 or blk == nblocks:
                nedge += 1
    name = subprocess.run(["c++filt", f["name"].lstrip(".L")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    print("%-58s %9s %8d %10d %10d %9d" % (name[:58], f.get("scratch", "?"), nins, nscr, nloop, nedge))
