#!/bin/bash
# The kernels' per-lane math (csrc/sg_math.h through tests/emu) and the oracle under AddressSanitizer + UBSan on the CPU: builds
# sanitized copies of tests/emu/libsgemu.so, tests/emu/libsgtreeemu.so and oracle/liboracle.so in place, runs the emulation and oracle test files with the
# sanitizer runtimes preloaded into python, and puts the normal builds back.  (GPU sanitizers are not available on the pool.)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
T=${TMPDIR:-/tmp}
cp tests/emu/libsgemu.so "$T/libsgemu.bak"; cp oracle/liboracle.so "$T/liboracle.bak"; cp tests/emu/libsgtreeemu.so "$T/libsgtreeemu.bak"
restore() { cp "$T/libsgemu.bak" tests/emu/libsgemu.so; cp "$T/liboracle.bak" oracle/liboracle.so; cp "$T/libsgtreeemu.bak" tests/emu/libsgtreeemu.so; }
trap restore EXIT
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -std=c++17 -Wno-unknown-pragmas -o tests/emu/libsgemu.so \
    tests/emu/sg_emu.cpp soft-grip_amd/csrc/sg_plan.cpp
# the tree pipeline's source (csrc/sg_tree.h) compiled for the host, in its checking layout (SGT_EMU_SEPARATE: every array of the env's
# LDS block and work space a heap block of its own, exact in size -- an access one element past ANY array is an ASan report, not a
# read of the neighbouring array); the emulation poisons the LDS-class blocks before every launch (sg_tree_emu.cpp)
g++ -O1 -g -DSGT_EMU_SEPARATE -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -std=c++17 -Wno-unknown-pragmas -o tests/emu/libsgtreeemu.so \
    tests/emu/sg_tree_emu.cpp soft-grip_amd/csrc/sg_plan.cpp
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -fopenmp -o oracle/liboracle.so oracle/sg_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_emu_vs_oracle.py tests/test_oracle_kat.py tests/test_tree_emu.py -x -q ${SG_SANITIZE_PYTEST_ARGS:-}
