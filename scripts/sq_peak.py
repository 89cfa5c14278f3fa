"""Per-wavefront SQ counter figures of the costliest dispatch of each kernel in a rocprofv3 --pmc counter_collection.csv
(diagnostic).  usage: sq_peak.py <dir>"""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*_counter_collection.csv"))[0]
disp = collections.defaultdict(dict)
name = {}
for r in csv.DictReader(open(f)):
    if "sg_" not in r["Kernel_Name"]:
        continue
    d = int(r["Dispatch_Id"])
    disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    name[d] = r["Kernel_Name"].split("(")[0].replace("void ", "")
best = {}
for d, c in disp.items():
    k = name[d]
    if k not in best or c.get("SQ_WAVE_CYCLES", 0) > disp[best[k]].get("SQ_WAVE_CYCLES", 0):
        best[k] = d
for k, d in best.items():
    c = disp[d]
    w = c.get("SQ_WAVES", 1.0)
    print(k, "dispatch", d, "waves %d" % w)
    for n, v in sorted(c.items()):
        if n != "SQ_WAVES":
            print("   %-22s %12.0f per wavefront" % (n, v / w))
