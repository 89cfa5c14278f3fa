# final evidence, part B: the profile rounds (kernel stats, HBM traffic, SQ totals) of the five scenes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05}
for s in ${2:-softbox softball softcylinder fourfinger_softball_fix freeball_fix}; do
  bash scripts/profile_round.sh $T $s || exit 1
done
ls gpurun_out | grep "^${T}_" | head -60
