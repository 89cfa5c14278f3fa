#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by ~10 % in clocks): alternating bench runs of a tree-pipeline scene.
# usage: scripts/dev/ab_tree.sh <libA.so> <libB.so> [scene] [rounds]
A=$1; B=$2; SCENE=${3:-fourfinger_softball_fix}; R=${4:-2}
for i in $(seq 1 $R); do
  for L in $A $B; do
    v=$(SOFTGRIP_LIB=$L timeout -k 10 300 python bench.py --scene $SCENE --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$L $v"
  done
done
