# soak / determinism runs of a round (scripts/soak.py): every scene twice from scratch, no flag, bit-identical
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04}
for a in "softbox 5" "softball 3" "softcylinder 3" "fourfinger_softball_fix 2" "freeball_fix 1"; do
  set -- $a
  python3 scripts/soak.py $1 $2 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_soak_$1.txt || { tail -5 gpurun_out/${T}_soak_$1.txt; exit 1; }
  echo "$1: $(tail -1 gpurun_out/${T}_soak_$1.txt)"
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_tree.py -m gpu -q -k workgroups_per_cu 2>&1 | tail -2
