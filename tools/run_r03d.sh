cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests/test_conv_s_gpu.py tests/test_graph_safety_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for s in 0 auto 8 16 24 32; do CL4_S_STAGGER=$s python tools/probe_s_epi.py 32 16 > $O/epi_stagger_$s.txt 2>&1 || { tail -5 $O/epi_stagger_$s.txt; exit 1; }; done
CL4_S_STAGGER=16 CL4_S_STAGGER_MIN_TILES=600 python tools/probe_s_epi.py 32 16 > $O/epi_stagger_16_min600.txt 2>&1
paste -d'\n' $O/epi_stagger_0.txt $O/epi_stagger_auto.txt | head -40
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
CL4_S_STAGGER=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench_nostagger.json 2> $O/bench_nostagger.err
python -c "import json; d=json.load(open('$O/bench_nostagger.json')); print('bench stagger 0', d['value'], d['ms_per_step'], d['roofline']['frac'])"
bash tools/soak_single_stream.sh r03d 10 30
echo "== expected to fault: hipMemsetAsync nodes, default graph path"
CL4_MEMSET_KERNEL=0 bash tools/soak_single_stream.sh r03d_memset 3 30
exit 0
