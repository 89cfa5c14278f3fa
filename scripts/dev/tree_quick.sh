#!/bin/bash
# one iteration on the tree pipeline (r05): the quick GPU tree tests, then the tree scenes' bench lines (whole episode, no CPU baseline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05t}
timeout -k 10 700 python3 -m pytest tests/test_gpu_tree.py -m gpu -x -q -k "${2:-four_finger_episode or free_ball_episode or chain_capacities or tree_pipeline_on_two_finger or random_grippers}" > gpurun_out/${T}_tree_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tree_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tree_tests.log
for s in ${3:-fourfinger_softball_fix freeball_fix fourfinger_softball freeball}; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --scene $s > gpurun_out/${T}_bench_$s.json 2> gpurun_out/${T}_bench_$s.err || { tail -5 gpurun_out/${T}_bench_$s.err; exit 1; }
  python3 -c "import json;l=json.load(open('gpurun_out/${T}_bench_$s.json'));print('$s',round(l['value']),l['roofline']['avg_kernel_ms'],l['config']['envs_flagged_bad'])"
done
