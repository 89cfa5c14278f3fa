set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04y}
timeout -k 10 900 python3 -m pytest tests/test_gpu_tree.py -m gpu -x -q > gpurun_out/${T}_tree_tests.log 2>&1 || { tail -40 gpurun_out/${T}_tree_tests.log; exit 1; }
tail -3 gpurun_out/${T}_tree_tests.log
for s in freeball_fix fourfinger_softball_fix fourfinger_softball freeball; do
  python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --scene $s > gpurun_out/${T}_bench_$s.json 2> gpurun_out/${T}_bench_$s.err
  python3 -c "import json;l=json.load(open('gpurun_out/${T}_bench_$s.json'));print('$s',l['value'],l['roofline']['avg_kernel_ms'],l['config']['envs_flagged_bad'])"
done
python3 scripts/tree_section_profile.py freeball_fix > gpurun_out/${T}_tree_sections_freeball.txt 2>&1
grep -A24 "squeeze peak" gpurun_out/${T}_tree_sections_freeball.txt
