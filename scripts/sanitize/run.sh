#!/bin/bash
# AddressSanitizer + UBSan over the host-side native code (MJCF compiler, plan builder): CPU build only (GPU sanitizers are not
# available on the pool).  usage: scripts/sanitize/run.sh [more.xml ...]   -- compiles every tests/data/*.xml (+ arguments) in both
# composite variants and builds the kernel plan; malformed and missing files must come back as messages.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
OUT=${TMPDIR:-/tmp}/sg_sanitize_host
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -o "$OUT" \
    "$HERE/host_main.cpp" "$ROOT/soft-grip_amd/csrc/sg_mjcf.cpp" "$ROOT/soft-grip_amd/csrc/sg_plan.cpp"
printf '<mujoco><worldbody><body></worldbody></mujoco>' > "${TMPDIR:-/tmp}/sg_broken.xml"
"$OUT" "$ROOT"/tests/data/*.xml "${TMPDIR:-/tmp}/sg_broken.xml" "${TMPDIR:-/tmp}/sg_missing.xml" "$@"
echo "sanitize: clean"
