cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in fourfinger_softball_fix freeball_fix; do
  bash scripts/profile_round.sh r04 $s || exit 1
done
python3 scripts/tree_section_profile.py fourfinger_softball_fix > gpurun_out/r04_tree_sections_fourfinger.txt 2>&1 || exit 1
python3 scripts/tree_section_profile.py freeball_fix > gpurun_out/r04_tree_sections_freeball.txt 2>&1 || exit 1
grep -A24 "squeeze peak" gpurun_out/r04_tree_sections_fourfinger.txt | head -26
