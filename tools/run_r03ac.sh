cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_heads_train_gpu.py tests/test_acceptance_gpu.py tests/test_fullsize_gpu.py tests/test_graph_safety_gpu.py tests/test_module_surface_gpu.py -q -m gpu --durations=10 > $O/tests.log 2>&1 || { tail -60 $O/tests.log | cut -c1-300; exit 1; }
tail -14 $O/tests.log
