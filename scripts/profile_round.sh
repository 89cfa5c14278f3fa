#!/bin/bash
# Profile passes of one round (run on the GPU box through gpurun): kernel trace + stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes), SQ instruction totals -- all over `python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant
# --no-event-pass` (one reset + 200 sg_step calls: the line bench.py prints by default also runs the region a second time with HIP events).
# usage: scripts/profile_round.sh <tag> <scene>      -> gpurun_out/<tag>_<scene>_*  (copy the summaries into profiles/)
set -e
TAG=$1; SCENE=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
CMD="python3 $ROOT/bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --no-event-pass --scene $SCENE"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_* 
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $CMD > $OUT/${TAG}_${SCENE}_bench_line.json 2>/dev/null
cp $(ls /tmp/prof_kt/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_${SCENE}_kernel_stats.csv
python3 $ROOT/scripts/trace_summary.py /tmp/prof_kt > $OUT/${TAG}_${SCENE}_phase_breakdown.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_rd -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_wr -- $CMD > /dev/null 2>&1
python3 $ROOT/scripts/hbm_traffic.py /tmp/prof_rd /tmp/prof_wr 201 $OUT/${TAG}_${SCENE}_hbm_traffic.json $OUT/${TAG}_${SCENE}_bench_line.json
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/prof_sq -- $CMD > /dev/null 2>&1
python3 $ROOT/scripts/sq_totals.py /tmp/prof_sq 4096 200 $OUT/${TAG}_${SCENE}_sq_totals.json
echo "profile_round done: $TAG $SCENE"
