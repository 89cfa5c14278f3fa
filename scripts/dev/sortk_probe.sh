#!/bin/bash
# r05s experiment: the same stiffness draws, assigned to the envs in drawn order against sorted order (envs of similar stiffness share a solver wavefront)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in softbox softcylinder softball; do
  for k in "" 1; do
    SG_EXP_SORT_K=$k timeout -k 10 300 python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --scene $s > gpurun_out/r05s_sortk_${s}_$k.json 2> gpurun_out/r05s_sortk_${s}_$k.err || { tail -5 gpurun_out/r05s_sortk_${s}_$k.err; exit 1; }
    python3 -c "import json;l=json.load(open('gpurun_out/r05s_sortk_${s}_$k.json'));print('$s sorted=$k',round(l['value']),l['roofline']['avg_kernel_ms'],l['config']['envs_flagged_bad'], l['roofline'].get('dominant_kernel',{}).get('avg_launch_ms'))"
  done
done
