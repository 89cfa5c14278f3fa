#!/bin/bash
# builds libsoftgrip_<name>.so = the monodbg library with sg_tree.hip recompiled under extra flags (everything else from monodbg's objects)
# usage: scripts/repro/tree_mono/variant.sh NAME [flags for sg_tree.hip ...]     (default base flags: the product's + -DSGT_X_MONO -DSG_DEBUG_WORK)
set -e
cd "$(dirname "$0")/../../.."
NAME=$1; shift
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-sched-strategy=iterative-ilp -DSGT_X_MONO -DSG_DEBUG_WORK"
if [ "$1" = "--base" ]; then shift; BASE="$1"; shift; fi
OBJDIR=$(python - <<'PY'
import hashlib,sys
sys.path.insert(0,'soft-grip_amd')
import build_native as b
print('soft-grip_amd/build/'+hashlib.sha1(" ".join(b.FLAGS+["-DSGT_X_MONO","-DSG_DEBUG_WORK"]).encode()).hexdigest()[:12])
PY
)
mkdir -p /tmp/variants
hipcc $BASE "$@" -c -o /tmp/variants/sg_tree_$NAME.o soft-grip_amd/csrc/sg_tree.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o soft-grip_amd/libsoftgrip_$NAME.so $OBJDIR/sg_api.o $OBJDIR/sg_phase.o $OBJDIR/sg_rows.o /tmp/variants/sg_tree_$NAME.o $OBJDIR/sg_plan.o $OBJDIR/sg_mjcf.o
echo soft-grip_amd/libsoftgrip_$NAME.so
