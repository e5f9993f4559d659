# Per-GPU batches 8 / 4 / 2 (what each rank of an N = 2 / 4 / 8 run executes) on one GPU: step time of the replayed step.
set -e
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r02g}_small
mkdir -p $O
for b in 8 4 2; do
  CL4_BENCH_PER_GPU_BATCH=$b timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 > $O/b$b.json 2> $O/b$b.err || { tail -5 $O/b$b.err; exit 1; }
  python -c "import json; d=json.load(open('$O/b$b.json')); print('per-GPU batch $b: ms_per_step', round(d['ms_per_step'],3), 'images/s', round(d['value'],1))"
done
