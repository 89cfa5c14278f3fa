#!/bin/bash
# r05, VERDICT r04 item 6 ("find the 1.4 %"): the r03-final tree (commit 1ac8eaf, rebuilt with today's compiler into scratch_r03/, git-ignored) and
# the current tree on ONE box: alternating whole-episode bench runs, then rocprofv3 kernel stats of each -- per-kernel averages side by side.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in $(seq 1 ${1:-3}); do
  for d in scratch_r03 .; do
    v=$(cd $d && timeout -k 10 300 python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --no-event-pass 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$d $v"
  done
done
for d in scratch_r03 .; do
  n=$( [ $d = . ] && echo r05 || echo r03 )
  rm -rf /tmp/prof_ab_$n
  (cd $d && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab_$n -- python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --no-event-pass > /dev/null 2>&1)
  f=$(find /tmp/prof_ab_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n"; head -5 $f | cut -c1-150
done
