cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu > $O/tests_conv.log 2>&1 || { tail -40 $O/tests_conv.log | cut -c1-300; exit 1; }
tail -1 $O/tests_conv.log
timeout -k 10 400 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_heads_train_gpu.py -x -q -m gpu > $O/tests_model.log 2>&1 || { tail -40 $O/tests_model.log | cut -c1-300; exit 1; }
tail -1 $O/tests_model.log
CL4_BENCH_LAYER_TABLE=$O/layers.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('bench x6 planes', d['value'], d['ms_per_step'], d['roofline']['frac'], d['conv_fwd_bf16x6'])"
CL4_X6_PLANES=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench0.json 2> $O/bench0.err
python -c "import json; d=json.load(open('$O/bench0.json')); print('bench x6 role-split', d['value'], d['ms_per_step'], d['roofline']['frac'], d['conv_fwd_bf16x6'])"
grep "bf16x6" $O/layers.txt | head -8
for i in 1 2; do timeout -k 10 300 python -m pytest "tests/test_acceptance_gpu.py::test_map_after_training_inside_the_reference_arithmetics_own_spread" -q -m gpu -rA > $O/proxy$i.log 2>&1; echo "proxy run $i rc $?"; grep "MAPPROXY" $O/proxy$i.log | grep -v "print(" | cut -c1-420; done
