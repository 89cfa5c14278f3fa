cd $GRAFT_REPO_ROOT
N=$1; shift
for L in "$@"; do echo "== $L"; for i in $(seq $N); do SOFTGRIP_LIB=soft-grip_amd/libsoftgrip_$L.so timeout -k 10 300 python3 scripts/dev/fuzz_probe.py 2>&1 | grep -a "FAILS\|Error\|fault" || true; done; done
