"""Per-kernel means of rocprofv3 --pmc counters from a results .db (run on the GPU box; the .db files are too large to copy back).
usage: python scripts/pmc_summary.py <dir> [first_dispatch last_dispatch]  -- dispatch range counted per kernel name"""
import glob
import sqlite3
import sys
import collections

import numpy as np

f = glob.glob(sys.argv[1] + "/*/*.db")[0]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 10 ** 9)
db = sqlite3.connect(f)
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]  # noqa: E731
kd, ks, pc, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
if "--schema" in sys.argv:
    for t in (kd, pc, pi):
        print(t, [r[1] for r in cur.execute("pragma table_info(%s)" % t)])
names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
pmc = {r[0]: r[1] for r in cur.execute("select id, name from %s" % pi)}
disp = list(cur.execute("select id, kernel_id, start, end, event_id from %s order by start" % kd))
vals = collections.defaultdict(dict)
for ev, pid, v in cur.execute("select event_id, pmc_id, value from %s" % pc):
    vals[ev][pmc[pid]] = vals[ev].get(pmc[pid], 0.0) + v
seen = collections.Counter()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for did, kid, s, e, ev in disp:
    n = names[kid].split("(")[0]
    key = "pgs_rows" if "pgs_rows" in n else "phase" if "phase" in n else "chain" if "chain" in n else None
    if key is None:
        continue
    i = seen[key]
    seen[key] += 1
    if lo <= i < hi:
        acc[key]["dur_us"].append((e - s) / 1e3)
        for c, v in vals.get(ev, {}).items():
            acc[key][c].append(v)
for key, d in acc.items():
    print(key, "n=%d" % len(d["dur_us"]), " ".join("%s=%.4g" % (c, np.mean(v)) for c, v in sorted(d.items())))
