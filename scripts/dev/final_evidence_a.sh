# final evidence, part A: the GPU suite, the bench lines without a profiler, the dataset's end-to-end rate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05}
timeout -k 10 900 python3 -m pytest tests -m gpu -q -s > gpurun_out/${T}_gputests.log 2>&1; rc=$?
tail -4 gpurun_out/${T}_gputests.log
[ $rc -eq 0 ] || exit $rc
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_default_command.json 2> gpurun_out/${T}_bench_default_command.err || exit 1
python3 bench.py --steps 200 --warmup 0 > gpurun_out/${T}_bench_whole_episode.json 2> /dev/null || exit 1
for s in softball softcylinder fourfinger_softball_fix freeball_fix fourfinger_softball freeball; do
  python3 bench.py --steps 200 --warmup 0 --no-fix-variant --scene $s > gpurun_out/${T}_bench_$s.json 2> /dev/null || exit 1
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    l = json.load(open(f)); print(f.split("${T}_bench_")[1], l["value"], l["roofline"].get("avg_kernel_ms"), l["config"].get("envs_flagged_bad"), l.get("cpu_baseline", {}).get("value"))
PY
python3 scripts/dataset_e2e.py 4096 1 softbox,softcylinder,softball > gpurun_out/${T}_dataset_end_to_end.txt 2>&1 || exit 1
python3 scripts/dataset_e2e.py 4096 3 softbox >> gpurun_out/${T}_dataset_end_to_end.txt 2>&1 || exit 1
python3 scripts/dataset_e2e.py 4096 3 softbox --mask-contact >> gpurun_out/${T}_dataset_end_to_end.txt 2>&1 || true
grep "end to end" gpurun_out/${T}_dataset_end_to_end.txt
