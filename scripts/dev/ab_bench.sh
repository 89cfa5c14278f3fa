#!/bin/bash
# A/B of two builds of the library on ONE box: alternating whole-episode runs of the default benchmark (softbox, 4096 envs).
# usage: scripts/dev/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    v=$(SOFTGRIP_LIB=$L timeout -k 10 300 python bench.py --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant --no-event-pass 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$L $v"
  done
done
