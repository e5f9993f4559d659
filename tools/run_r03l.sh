cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_graph_safety_gpu.py tests/test_cli_gpu.py tests/test_validate_gpu.py "tests/test_acceptance_gpu.py::test_map_after_training_inside_the_reference_arithmetics_own_spread" -q -m gpu -rA --durations=12 > $O/tests.log 2>&1
echo "rc $?"
grep "MAPPROXY\|passed\|failed\|^FAILED\|^E  \| call " $O/tests.log | grep -v "print(" | cut -c1-400
