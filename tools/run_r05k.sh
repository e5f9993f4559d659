#!/bin/bash
# r05: the whole GPU suite at HEAD (timings) and the acceptance tests with their printed tables
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05k; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=15 > $O/gpu_test_suite.log 2>&1; echo "suite rc $?"; tail -22 $O/gpu_test_suite.log
timeout -k 10 600 python -m pytest -q -s tests/test_acceptance_gpu.py tests/test_golden_gpu.py -k "acceptance or well_conditioned or pipelined_replay" > $O/acceptance_raw.log 2>&1; echo "acceptance rc $?"
grep -h "MAPPROXY\|WELLCOND\|GRADNORM193\|PIPEREPLAY" $O/acceptance_raw.log | sort > $O/acceptance.txt; wc -l $O/acceptance.txt
