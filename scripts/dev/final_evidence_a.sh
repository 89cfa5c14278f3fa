# final evidence, part A: the GPU suite, the bench lines without a profiler, the dataset's end-to-end rate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests -m gpu -q > gpurun_out/r04_gputests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_gputests.log
[ $rc -eq 0 ] || exit $rc
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default_command.json 2> gpurun_out/r04_bench_default_command.err || exit 1
python3 bench.py --steps 200 --warmup 0 > gpurun_out/r04_bench_whole_episode.json 2> /dev/null || exit 1
for s in softball softcylinder fourfinger_softball_fix freeball_fix fourfinger_softball freeball; do
  python3 bench.py --steps 200 --warmup 0 --no-fix-variant --scene $s > gpurun_out/r04_bench_$s.json 2> /dev/null || exit 1
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_bench_*.json")):
    l = json.load(open(f)); print(f.split("r04_bench_")[1], l["value"], l["roofline"].get("avg_kernel_ms"), l["config"].get("envs_flagged_bad"), l.get("cpu_baseline", {}).get("value"))
PY
python3 scripts/dataset_e2e.py 4096 3 softbox > gpurun_out/r04_dataset_end_to_end.txt 2>&1 || exit 1
python3 scripts/dataset_e2e.py 4096 3 softbox --mask-contact >> gpurun_out/r04_dataset_end_to_end.txt 2>&1 || true
tail -3 gpurun_out/r04_dataset_end_to_end.txt
