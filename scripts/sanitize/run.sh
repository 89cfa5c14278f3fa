#!/bin/bash
# AddressSanitizer + UBSan over the host-side native code (MJCF compiler, plan builder): CPU build only (GPU sanitizers are not
# available on the pool).  usage: scripts/sanitize/run.sh [more.xml ...]   -- compiles every tests/data/*.xml (+ arguments) in both
# composite variants and builds the kernel plan; malformed and missing files must come back as messages.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
OUT=${TMPDIR:-/tmp}/sg_sanitize_host
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -o "$OUT" \
    "$HERE/host_main.cpp" "$ROOT/soft-grip_amd/csrc/sg_mjcf.cpp" "$ROOT/soft-grip_amd/csrc/sg_plan.cpp"
T=${TMPDIR:-/tmp}
printf '<mujoco><worldbody><body></worldbody></mujoco>' > "$T/sg_broken.xml"
# an include cycle (stack overflow before the depth limit) and composite counts beyond any sane size
printf "<mujoco><include file='sg_cycle.xml'/></mujoco>" > "$T/sg_cycle.xml"
for c in "2000 2000 2000" "1e300 4 4" "20 20 20"; do
  printf "<mujoco><compiler angle='radian'/><option solver='PGS' cone='elliptic'/><worldbody><body pos='0 0 1'><composite type='box' count='%s' spacing='.3'><geom type='capsule' size='.02 .05' mass='.01'/></composite></body></worldbody></mujoco>" "$c" > "$T/sg_count_${c%% *}.xml"
done
"$OUT" "$ROOT"/tests/data/*.xml "$T/sg_broken.xml" "$T/sg_missing.xml" "$T/sg_cycle.xml" "$T"/sg_count_*.xml "$@"
echo "sanitize: clean"
