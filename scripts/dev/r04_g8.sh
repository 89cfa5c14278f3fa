cd $GRAFT_REPO_ROOT
for L in libsoftgrip_r03.so libsoftgrip.so libsoftgrip_r03.so libsoftgrip.so; do bash scripts/kstats.sh $L softbox 2>&1 | tail -5; done > gpurun_out/r04t_kstats_ab.txt 2>&1
cat gpurun_out/r04t_kstats_ab.txt
