cd $GRAFT_REPO_ROOT
for L in "$@"; do echo "== $L"; SOFTGRIP_LIB=soft-grip_amd/libsoftgrip_$L.so timeout -k 10 300 python3 scripts/dev/fuzz_probe.py 2>&1 | tail -4 || exit 1; done
