cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ai; mkdir -p $O
for e in 0 1 0 1; do
CL4_EXPERIMENT_TRUNK_IN_FORK=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench_e$e.json 2> $O/bench_e$e.err || { tail -5 $O/bench_e$e.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_e$e.json')); print('bench extra trunk in fork=$e', d['value'], d['ms_per_step'])"
done
bash tools/run_suite.sh r03ai
