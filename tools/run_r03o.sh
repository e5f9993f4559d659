cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_pool_gpu.py tests/test_label_gpu.py tests/test_insmap_gpu.py tests/test_golden_gpu.py tests/test_model_gpu.py tests/test_validate_gpu.py -q -m gpu --durations=8 > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -12 $O/tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench$i.json 2> $O/bench$i.err || { tail -5 $O/bench$i.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench$i.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
