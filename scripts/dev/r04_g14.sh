cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04x}
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "episode_matches_oracle and softcylinder" > gpurun_out/${T}_cyl.log 2>&1 || { tail -30 gpurun_out/${T}_cyl.log; exit 1; }
tail -2 gpurun_out/${T}_cyl.log
python3 scripts/tree_section_profile.py fourfinger_softball_fix > gpurun_out/${T}_tree_sections.txt 2>&1 || exit 1
python3 scripts/tree_section_profile.py freeball_fix > gpurun_out/${T}_tree_sections_freeball.txt 2>&1 || exit 1
grep -A24 "squeeze peak" gpurun_out/${T}_tree_sections.txt
grep -A24 "squeeze peak" gpurun_out/${T}_tree_sections_freeball.txt
