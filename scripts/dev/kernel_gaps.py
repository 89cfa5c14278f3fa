"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (the stream of one bench run): how much of the wall time of a
sg_step call is between kernels rather than in them.  usage: kernel_gaps.py <..._kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
gaps = defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g < 50000:            # longer: a host sync between calls
        gaps[(n0, n1)].append(g)
tot = sum(sum(v) for v in gaps.values())
print("kernels %d  busy %.1f ms  short gaps total %.1f ms (%.2f %% of busy)" % (len(rows), busy / 1e6, tot / 1e6, 100.0 * tot / busy))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:10]:
    v.sort()
    print("%-42s -> %-42s n %5d  mean %6.0f ns  median %6.0f  p95 %6.0f  sum %.2f ms" % (k[0], k[1], len(v), sum(v) / len(v), v[len(v) // 2], v[int(0.95 * len(v))], sum(v) / 1e6))
