set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q  > gpurun_out/r04d_tests.log 2>&1 || { tail -30 gpurun_out/r04d_tests.log; exit 1; }
tail -3 gpurun_out/r04d_tests.log
for s in softball softcylinder softbox; do
  python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --scene $s > gpurun_out/r04d_bench_$s.json 2> gpurun_out/r04d_bench_$s.err
  python3 -c "import json;l=json.load(open('gpurun_out/r04d_bench_$s.json'));print('$s',l['value'],l['roofline']['dominant_kernel']['avg_launch_ms'],l['config']['envs_flagged_bad'])"
done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04d_bench_default.json 2>/dev/null
python3 -c "import json;l=json.load(open('gpurun_out/r04d_bench_default.json'));print('default',l['value'])"
