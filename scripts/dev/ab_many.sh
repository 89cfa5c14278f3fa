#!/bin/bash
# several builds of the library alternating on ONE box: scripts/dev/ab_many.sh <scene> <rounds> <name> [<name> ...]   (libsoftgrip_<name>.so; "product" = libsoftgrip.so)
cd $GRAFT_REPO_ROOT
SCENE=$1; R=$2; shift 2
for i in $(seq 1 $R); do
  for n in "$@"; do
    L=$GRAFT_REPO_ROOT/soft-grip_amd/libsoftgrip_$n.so; [ "$n" = product ] && L=$GRAFT_REPO_ROOT/soft-grip_amd/libsoftgrip.so
    v=$(SOFTGRIP_LIB=$L timeout -k 10 300 python3 bench.py --scene $SCENE --steps 200 --warmup 0 --no-cpu-baseline --no-fix-variant 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(l['value']), l['config']['envs_flagged_bad'])")
    echo "$SCENE $n $v"
  done
done
