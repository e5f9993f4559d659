cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ah; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_heads_train_gpu.py tests/test_acceptance_gpu.py tests/test_fullsize_gpu.py tests/test_conv_gpu.py -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log | cut -c1-300; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench$i.json 2> $O/bench$i.err || { tail -5 $O/bench$i.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench$i.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['conv_wgrad']['ms_per_step'])"
done
