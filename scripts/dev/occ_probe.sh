#!/bin/bash
# r05s experiment: the phase kernel at three wavefronts per SIMD (launch bounds (64, 3): 168 registers, spills) -- windows of the episode per library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in "" ${@}; do
  if [ -n "$lib" ]; then export SOFTGRIP_LIB=$GRAFT_REPO_ROOT/soft-grip_amd/libsoftgrip_$lib.so; fi
  timeout -k 10 200 python3 scripts/phase_time.py softbox 2>&1 | grep ms/env-step
done
