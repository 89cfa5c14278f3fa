"""Sum the SQ counters of a rocprofv3 --pmc run of `bench.py --steps 200 --warmup 0 --no-cpu-baseline` per kernel and over the
episode -> the JSON bench.py reads for its secondary (instruction-issue) roofline.
usage: sq_totals.py <dir> <n envs> <n sg_step calls> <out.json>"""
import collections
import csv
import glob
import json
import sys

d, nenv, calls, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "sg_" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
allk = collections.defaultdict(float)
for k in tot:
    for c, v in tot[k].items():
        allk[c] += v
res = {
    "note": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS over one 200-step episode "
            "(1 reset + %d sg_step calls, %d envs, default scene); sums over all dispatches; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY count quad-cycles" % (calls, nenv),
    "kernels": {k: dict(v) for k, v in tot.items()},
    "episode": dict(allk),
    "per_env_step": {c: v / (nenv * calls) for c, v in allk.items()},
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["per_env_step"]))
