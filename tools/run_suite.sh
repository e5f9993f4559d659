# The whole GPU suite with per-test durations -> gpurun_out/<tag>/suite.log (commit as profiles/<tag>_gpu_test_suite.log)
cd "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}"
O=gpurun_out/${1:-r03}; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -q -m gpu --durations=45 > $O/suite.log 2>&1
rc=$?
tail -60 $O/suite.log
exit $rc
