cd $GRAFT_REPO_ROOT
for n in 256 1024 4096; do
  timeout -k 5 300 python3 scripts/tree_section_profile.py fourfinger_softball_fix $n > gpurun_out/r04q_sections_n$n.txt 2>&1
  echo "n=$n"; grep -A24 "squeeze peak" gpurun_out/r04q_sections_n$n.txt | grep "squeeze\|13 sweep: contacts\|19 sweep: chain\|10 contact rows\|21 pairs\|12 sweep"
done
