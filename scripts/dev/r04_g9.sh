cd $GRAFT_REPO_ROOT
for L in libsoftgrip_d255.so libsoftgrip_d63.so; do echo "== $L"; SOFTGRIP_LIB=soft-grip_amd/$L timeout -k 10 300 python3 scripts/dev/fuzz_probe.py 2>&1 | tail -9; done
