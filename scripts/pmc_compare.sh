#!/bin/bash
# PMC comparison of two builds of the library over scripts/phase_time.py (one episode pair): per-kernel means at the squeeze peak.
# usage: scripts/pmc_compare.sh <scene> <libA> <libB>   (run on the GPU box through gpurun; prints to stdout)
SCENE=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for LIB in $2 $3; do
  export SOFTGRIP_LIB=$ROOT/soft-grip_amd/$LIB
  echo "== $LIB"
  python3 $ROOT/scripts/phase_time.py $SCENE 2>/dev/null | tail -2
  rm -rf /tmp/pc1 /tmp/pc2
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d /tmp/pc1 -- python3 $ROOT/scripts/phase_time.py $SCENE > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_summary.py /tmp/pc1 700 760
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d /tmp/pc2 -- python3 $ROOT/scripts/phase_time.py $SCENE > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_summary.py /tmp/pc2 700 760
done
