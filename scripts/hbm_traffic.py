"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench.py command) into the
per-sg_step-call HBM traffic figure bench.py reports as roofline.traffic.

usage: hbm_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <n sg_step calls + 1 reset> <out.json> [bench_line.json]

The bench line of the same command (scripts/profile_round.sh writes it first) gives the scene's own algorithmic bytes per call
(5 816 / 9 752 / 11 000 / 13 880 B per env step x the envs of the batch) and the pipeline that ran.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
    tot, n = collections.OrderedDict(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "sg_" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]) * 1024.0  # counter unit: KiB
        n[k] += 1
    return tot, n


def main():
    rd, nrd = per_kernel(sys.argv[1], "FETCH_SIZE")
    wr, nwr = per_kernel(sys.argv[2], "WRITE_SIZE")
    calls = int(sys.argv[3])
    pipeline, abytes, nenv = "rows", 5816, 4096
    if len(sys.argv) > 5:
        line = json.load(open(sys.argv[5]))
        abytes = line["roofline"]["algorithmic_bytes_per_env_step"]
        nenv = line["config"]["envs_per_gpu"]
        pipeline = "tree" if "sg_tree_kernel" in line["roofline"]["kernel"] else "rows"
    kernels = {}
    total = 0.0
    for k in rd:
        # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of 16-B-per-lane streaming reads.  The rows
        # PGS kernel reads its contact rows as 16-byte pairs (its whole traffic but a few per cent), so its figure is doubled; the
        # phase and chain kernels read 8 B per lane (uncalibrated width): raw.  WRITE_SIZE needs no correction.
        # The tree kernel reads 8 B per lane nearly everywhere: raw.
        corr = 2.0 if "sg_pgs_rows_kernel" in k else 1.0
        kernels[k] = {"FETCH_SIZE_raw_bytes": rd[k], "FETCH_SIZE_correction": corr, "FETCH_SIZE_total_bytes": corr * rd[k],
                      "FETCH_SIZE_dispatches": nrd[k], "WRITE_SIZE_total_bytes": wr.get(k, 0.0), "WRITE_SIZE_dispatches": nwr.get(k, 0)}
        total += corr * rd[k] + wr.get(k, 0.0)
    out = {
        "unit": "bytes",
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps %d --warmup 0 "
                "--no-cpu-baseline` (1 sg_reset + %d sg_step calls, %d envs, %s pipeline); counter unit = KiB (x1024). gfx950 "
                "correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of 16-B/lane streaming reads -- %s  "
                "These are fabric-side counters: Infinity-Cache hits are included, so this is L2<->fabric "
                "traffic, an upper bound of the HBM traffic." % (calls - 1, calls - 1, nenv, pipeline,
                    "none applied: sg_tree_kernel reads 8 B/lane (uncalibrated width): raw." if pipeline == "tree" else
                    "applied (x2) to sg_pgs_rows_kernel, whose contact rows are read as 16-byte pairs; the phase and chain kernels read 8 B/lane (uncalibrated width): raw."),
        "kernels": kernels,
        "episode_total_bytes": total,
        "per_sg_step_call_bytes": total / calls,
        "algorithmic_bytes_per_call": abytes * nenv,
        "traffic_over_algorithmic": total / calls / (abytes * nenv),
    }
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("episode_total_bytes", "per_sg_step_call_bytes", "algorithmic_bytes_per_call")}))


if __name__ == "__main__":
    main()
