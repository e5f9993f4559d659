cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ae; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_wgrad_gemm_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for g in 1 0 1; do
CL4_WGRAD_GEMM=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench_g$g.json 2> $O/bench_g$g.err || { tail -5 $O/bench_g$g.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_g$g.json')); print('bench CL4_WGRAD_GEMM=$g', d['value'], d['ms_per_step'], d['roofline']['frac'], d['conv_wgrad'])"
done
