cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests/test_conv_s_gpu.py tests/test_stem_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for plan in 2,128,1 2,144,1 2,128,2 2,144,2 2,128,-8 2,144,-8; do PLAN=$plan python tools/probe_s_epi.py 32 16 > $O/epi_plan_$plan.txt 2>&1 || { tail -5 $O/epi_plan_$plan.txt; exit 1; }; done
for f in $O/epi_plan_*.txt; do echo "== $f"; grep -v amdgpu.ids $f; done
CL4_BENCH_LAYER_TABLE=$O/layers.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
CL4_PLANES_TILES=128 CL4_STEM_FUSED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench_old.json 2> $O/bench_old.err
python -c "import json; d=json.load(open('$O/bench_old.json')); print('bench tiles=128, unfused stem', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
