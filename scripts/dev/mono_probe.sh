#!/bin/bash
# r05: the tree GPU tests against experiment builds of the library (the r04 monolithic layout and variants), one after the other;
# stops at the first run that was killed or timed out (never start another GPU step after one)
mkdir -p gpurun_out
for v in "$@"; do
  SOFTGRIP_LIB=$PWD/soft-grip_amd/libsoftgrip_$v.so timeout -k 10 500 python -m pytest tests/test_gpu_tree.py -q -x -k "four_finger_episode or random_grippers or chain_capacities or free_ball_episode or tree_pipeline_on_two_finger" > gpurun_out/r05_probe_$v.log 2>&1
  rc=$?
  echo "variant $v rc=$rc" | tee -a gpurun_out/r05_probe_$v.log
  tail -4 gpurun_out/r05_probe_$v.log
  if [ $rc -ge 124 ]; then exit $rc; fi
done
exit 0
