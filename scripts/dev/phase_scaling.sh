#!/bin/bash
# r05s: the phase and chain kernels' launch time against the batch size (whole episode, kernel statistics of a trace per size):
# one round of wavefronts or two, and what a round costs
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SCENE=${SCENE:-softbox}
for n in ${@:-512 1024 2048 3072 4096}; do
  rm -rf /tmp/ps_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$n -- python3 $ROOT/scripts/phase_time.py $SCENE $n > /dev/null 2>&1
  f=$(ls /tmp/ps_$n/*/*_kernel_stats.csv | head -1)
  python3 - $f $n <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("true, false>", "sg_chain_kernel", "sg_pgs_rows")):
        print("envs %5s  %-46s calls %5s  avg %7.1f us  min %7.1f us" % (sys.argv[2], r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
