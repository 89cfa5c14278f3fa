cd $GRAFT_REPO_ROOT
for b in 63 7 56; do SOFTGRIP_LIB=soft-grip_amd/libsoftgrip_diet$b.so timeout -k 5 120 python3 scripts/dev/diet_probe.py 2>/dev/null | tail -1; done
timeout -k 5 120 python3 scripts/dev/diet_probe.py 2>/dev/null | tail -1
