cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r04 fourfinger_softball_fix > gpurun_out/r04_pr_ff.log 2>&1
tail -3 gpurun_out/r04_pr_ff.log
cat gpurun_out/r04_fourfinger_softball_fix_phase_breakdown.txt
python3 -c "
import json
q=json.load(open('gpurun_out/r04_fourfinger_softball_fix_sq_totals.json'))
for k,v in q['kernels'].items():
    if 'tree' in k: print(k, {c:'%.4g'%x for c,x in v.items()})
t=json.load(open('gpurun_out/r04_fourfinger_softball_fix_hbm_traffic.json')); print({k:t[k] for k in ('per_sg_step_call_bytes','algorithmic_bytes_per_call','traffic_over_algorithmic')})
"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_FLAT --output-format csv -d /tmp/prof_x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --no-event-pass --scene fourfinger_softball_fix > /dev/null 2>/tmp/prof_x.err; python3 $GRAFT_REPO_ROOT/scripts/sq_totals.py /tmp/prof_x 4096 200 $GRAFT_REPO_ROOT/gpurun_out/r04_fourfinger_extra_counters.json 2>&1 | tail -2; tail -3 /tmp/prof_x.err
