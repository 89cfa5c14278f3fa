#!/bin/bash
# one iteration on the rows pipeline (r05): its parity tests, then the three reference scenes' bench lines (whole episode, no CPU baseline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05r}
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${2:-episode_matches_oracle or full_size_properties or bit_reproducible or ragged or general_contact or envs_per_wavefront}" > gpurun_out/${T}_rows_tests.log 2>&1 || { tail -30 gpurun_out/${T}_rows_tests.log; exit 1; }
tail -2 gpurun_out/${T}_rows_tests.log
for s in ${3:-softbox softball softcylinder}; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --scene $s > gpurun_out/${T}_bench_$s.json 2> gpurun_out/${T}_bench_$s.err || { tail -5 gpurun_out/${T}_bench_$s.err; exit 1; }
  python3 -c "import json;l=json.load(open('gpurun_out/${T}_bench_$s.json'));print('$s',round(l['value']),l['roofline']['avg_kernel_ms'],l['config']['envs_flagged_bad'], l['roofline'].get('dominant_kernel',{}).get('avg_launch_ms'))"
done
