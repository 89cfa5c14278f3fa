set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04v}
for L in "" _d1ff; do
  for s in fourfinger_softball_fix freeball_fix; do
    SOFTGRIP_LIB=soft-grip_amd/libsoftgrip$L.so python3 bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --scene $s > gpurun_out/${T}_bench_$s$L.json 2> gpurun_out/${T}_bench_$s$L.err
    python3 -c "import json;l=json.load(open('gpurun_out/${T}_bench_$s$L.json'));print('$s$L',l['value'],l['roofline']['avg_kernel_ms'],l['config']['envs_flagged_bad'])"
  done
done
python3 scripts/tree_section_profile.py fourfinger_softball_fix > gpurun_out/${T}_tree_sections.txt 2>&1
grep -A24 "squeeze peak" gpurun_out/${T}_tree_sections.txt
