cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_golden_gpu.py tests/test_model_gpu.py tests/test_graph_safety_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for i in 1 2; do
CL4_BENCH_LAYER_TABLE=$O/layers$i.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench$i.json 2> $O/bench$i.err || { tail -5 $O/bench$i.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench$i.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
CL4_CONV_B=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench0.json 2> $O/bench0.err
python -c "import json; d=json.load(open('$O/bench0.json')); print('bench CL4_CONV_B=0', d['value'], d['ms_per_step'], d['roofline']['frac'])"
