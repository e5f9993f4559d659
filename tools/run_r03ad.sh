cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ad; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_golden_gpu.py tests/test_graph_safety_gpu.py tests/test_validate_gpu.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
CL4_BENCH_LAYER_TABLE=$O/layers.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
CL4_TEST_TIMING=1 timeout -k 10 400 python -m pytest tests/test_ddp_trainer_gpu.py -q -m gpu -s 2>&1 | grep "rank\|passed\|failed" | head -20
