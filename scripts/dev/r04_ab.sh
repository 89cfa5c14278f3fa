set -e
cd $GRAFT_REPO_ROOT
bash scripts/dev/ab_bench.sh soft-grip_amd/libsoftgrip_r03.so soft-grip_amd/libsoftgrip.so 4 > gpurun_out/r04e_ab_box.txt 2>&1
cat gpurun_out/r04e_ab_box.txt
