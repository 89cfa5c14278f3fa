cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04p}
timeout -k 10 1150 python3 -m pytest tests -m gpu -q > gpurun_out/${T}_gputests.log 2>&1
rc=$?
tail -15 gpurun_out/${T}_gputests.log
exit $rc
