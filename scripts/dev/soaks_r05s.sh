# r05s: soak / determinism runs of the scenes whose kernels the last session of r05 changed (scripts/soak.py: twice from scratch, no flag, bit-identical),
# then the 400-gripper fuzz sweep of the tree pipeline (its neighbour-row half runs the pipelined equality rounds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05s}
for a in "softbox 3" "softball 3" "fourfinger_softball_fix 2" "fourfinger_softball 2" "freeball_fix 1"; do
  set -- $a
  python3 scripts/soak.py $1 $2 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_soak_$1.txt || { tail -5 gpurun_out/${T}_soak_$1.txt; exit 1; }
  echo "$1: $(tail -1 gpurun_out/${T}_soak_$1.txt)"
done
SG_FUZZ_SCENES=100 SG_FUZZ_SEED=1000 timeout -k 10 900 python3 -m pytest tests/test_gpu_tree.py -m gpu -q -s -k random_grippers_on_the_gpu > gpurun_out/${T}_gpu_fuzz_sweep.txt 2>&1; rc=$?
tail -6 gpurun_out/${T}_gpu_fuzz_sweep.txt
exit $rc
