# final evidence, part B: the profile rounds (kernel stats, HBM traffic, SQ totals) of the five scenes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in softbox softball softcylinder fourfinger_softball_fix freeball_fix; do
  bash scripts/profile_round.sh r04 $s || exit 1
done
ls gpurun_out | grep "^r04_" | head -60
