cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
for v in hip nt2 nt3 nt1 hip; do
CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_$v.json')); print('bench $v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
