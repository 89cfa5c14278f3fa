cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04p}
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > gpurun_out/${T}_gputests.log 2>&1
tail -15 gpurun_out/${T}_gputests.log
SOFTGRIP_LIB=soft-grip_amd/libsoftgrip_noqcqp.so timeout -k 5 200 python3 scripts/tree_section_profile.py fourfinger_softball_fix > gpurun_out/${T}_tree_sections_noqcqp.txt 2>&1
grep -A24 "squeeze peak" gpurun_out/${T}_tree_sections_noqcqp.txt | grep "squeeze\|contacts\|chain limit"
