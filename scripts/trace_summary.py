"""Summarise a rocprofv3 --kernel-trace CSV of a bench.py run (1 reset + 200 sg_step calls): per-kernel time by episode phase -- the
rows pipeline's kernel chain, or the tree pipeline's one kernel per call."""
import csv
import glob
import sys

import numpy as np

d = sys.argv[1]
f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
is_gen = lambda r: 'sg_phase' in r['Kernel_Name'] and r['Kernel_Name'].count(',') >= 3 and r['Kernel_Name'].split('>')[0].rstrip().endswith('true')  # <R, CPL, NB, GEN = true>
gp = [r for r in rows if is_gen(r)]           # the general contact pass (a small grid after every phase launch that begins a substep)
ch = [r for r in rows if 'sg_chain' in r['Kernel_Name']]
ph = [r for r in rows if 'sg_phase' in r['Kernel_Name'] and not is_gen(r)]
pg = [r for r in rows if 'sg_pgs' in r['Kernel_Name']]
tr = [r for r in rows if 'sg_tree_kernel' in r['Kernel_Name']]
if tr and not pg:   # tree pipeline: one launch per sg_step call (the first one is the reset)
    d = np.array([dur(r) for r in tr][-200:])
    print(len(tr), "launches of", tr[0]['Kernel_Name'].split('(')[0], "VGPR", tr[0]['VGPR_Count'], tr[0].get('Accum_VGPR_Count'), "LDS", tr[0]['LDS_Block_Size'], "scratch", tr[0].get('Scratch_Size', tr[0].get('Private_Segment_Size')))
    for a, b in [(0, 40), (40, 50), (50, 80), (80, 120), (120, 140), (140, 200)]:
        print(a, b, "tree ms/call %.2f" % (d[a:b].mean() / 1e3))
    print("sum kernels per episode ms %.1f" % (d.sum() / 1e3))
    print("wall of timed region ms %.1f" % ((int(tr[-1]['End_Timestamp']) - int(tr[-200]['Start_Timestamp'])) / 1e6))
    sys.exit(0)
print(len(ph), len(pg), "phase VGPR", ph[0]['VGPR_Count'], ph[0].get('Accum_VGPR_Count'), "pgs VGPR", pg[0]['VGPR_Count'], pg[0].get('Accum_VGPR_Count'), "LDS", pg[0]['LDS_Block_Size'])
php = np.array([dur(r) for r in ph][-1600:]).reshape(200, 8)
pgp = np.array([dur(r) for r in pg][-1400:]).reshape(200, 7)
for a, b in [(0, 40), (40, 50), (50, 80), (80, 120), (120, 140), (140, 200)]:
    print(a, b, "phase us/kernel %.0f  pgs us/kernel %.0f   step ms %.2f (phase %.2f pgs %.2f)" % (
        php[a:b].mean(), pgp[a:b].mean(), (php[a:b].sum(1) + pgp[a:b].sum(1)).mean() / 1e3, php[a:b].sum(1).mean() / 1e3, pgp[a:b].sum(1).mean() / 1e3))
print("sum kernels per episode ms %.1f  (phase %.1f, pgs %.1f)" % ((php.sum() + pgp.sum()) / 1e3, php.sum() / 1e3, pgp.sum() / 1e3))
print("chain kernel: %d launches, %.1f ms; general contact pass: %d launches, %.2f ms (%.2f us each)" % (
    len(ch), sum(dur(r) for r in ch) / 1e3, len(gp), sum(dur(r) for r in gp) / 1e3, (sum(dur(r) for r in gp) / max(1, len(gp)))))
t0 = int(ph[-1600]['Start_Timestamp']); t1 = int(ph[-1]['End_Timestamp'])
print("wall of timed region ms %.1f" % ((t1 - t0) / 1e6))
