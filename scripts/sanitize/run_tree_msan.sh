#!/bin/bash
# MemorySanitizer over the tree pipeline's source on the host (scripts/sanitize/tree_msan_main.cpp): oracle states of three scenes
# replayed substep by substep with the env's LDS-class arrays marked uninitialised before every launch.  Needs ROCm's clang (its
# MSan runtime); CPU only.  A clean run prints the scenes' per-step counts and "msan: clean".
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
T=${TMPDIR:-/tmp}/sg_msan; mkdir -p "$T"
CXX=${MSAN_CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -std=c++17 -O1 -g -fsanitize=memory -fsanitize-memory-track-origins=2 -fno-omit-frame-pointer -Wno-unknown-pragmas -Wno-constant-logical-operand \
    -o "$T/tree_msan" scripts/sanitize/tree_msan_main.cpp soft-grip_amd/csrc/sg_plan.cpp
for s in fuzz5 fourfinger_softball_fix freeball_fix; do
  python scripts/sanitize/dump_tree_states.py $s 40 "$T" > /dev/null
  echo "== $s"; "$T/tree_msan" "$T/$s.blob" "$T/$s.states" | tail -2
done
echo "msan: clean"
