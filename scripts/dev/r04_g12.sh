cd $GRAFT_REPO_ROOT
R=${REPS:-1}
for L in "$@"; do echo "== $L reps $R"; for i in 1 2; do REPS=$R SOFTGRIP_LIB=soft-grip_amd/libsoftgrip_$L.so timeout -k 10 300 python3 scripts/dev/fuzz_probe.py 2>&1 | grep -a "FAILS\|Error\|fault\|^3 " | sed 's/np.int32(\([0-9]*\))/\1/g' | cut -c1-230 || true; done; done
