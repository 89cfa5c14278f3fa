#!/bin/bash
# per-kernel totals (rocprofv3 --kernel-trace --stats) of one episode pair of scripts/phase_time.py for a build of the library
# usage: scripts/kstats.sh <lib.so> [scene]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export SOFTGRIP_LIB=$ROOT/soft-grip_amd/$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$1 -- python3 $ROOT/scripts/phase_time.py ${2:-softbox} > /dev/null 2>&1
echo "== $1"
grep -E "pgs_rows|phase_kernel|chain_kernel" $(ls /tmp/ks_$1/*/*_kernel_stats.csv | head -1) | awk -F'",' '{print $1}' | cut -c2-60 | paste - <(grep -E "pgs_rows|phase_kernel|chain_kernel" $(ls /tmp/ks_$1/*/*_kernel_stats.csv | head -1) | awk -F'",' '{print $2}' | awk -F, '{printf "calls %s total_ms %.1f avg_us %.1f\n", $1, $2/1e6, $3/1e3}')
