cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "big_tile" > $O/tests_b.log 2>&1 || { tail -40 $O/tests_b.log | cut -c1-300; exit 1; }
tail -1 $O/tests_b.log
for plan in 6,144,1 5,144,1 2,144,1; do
PLAN=$plan timeout -k 10 120 python tools/probe_s_epi.py 16 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
done
PLAN=6,144,1 timeout -k 10 120 python tools/probe_s_epi.py 32 8 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
for i in 1 2; do
CL4_BENCH_LAYER_TABLE=$O/layers$i.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench$i.json 2> $O/bench$i.err || { tail -5 $O/bench$i.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench$i.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
