cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "three_plane or big_tile" > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for x6 in 1 0 1; do
CL4_X6_PLANES=$x6 CL4_BENCH_LAYER_TABLE=$O/layers_x6_$x6.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench_x6_$x6.json 2> $O/bench_x6_$x6.err || { tail -5 $O/bench_x6_$x6.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_x6_$x6.json')); print('bench CL4_X6_PLANES=$x6', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('conv_fwd_bf16x6'))"
done
grep bf16x6 $O/layers_x6_1.txt | head -8
