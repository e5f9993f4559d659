# The N > 1 code path on ONE GPU: a world-size-1 RCCL group with CL4_FORCE_COLLECTIVES=1 (graph up to the ASPP backward, eager ASPP
# tail with bucketed all-reduces on the communication stream, Adam), at the per-GPU batches an N = 1 / 2 / 4 / 8 run executes.
set -e
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r02g}_solo
mkdir -p $O
for b in 16 8 4 2; do
  CL4_FORCE_COLLECTIVES=1 CL4_BENCH_PER_GPU_BATCH=$b MASTER_PORT=$((29600 + b)) timeout -k 10 200 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 \
    > $O/b$b.json 2> $O/b$b.err || { tail -5 $O/b$b.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/b$b.json').read().strip().splitlines()[-1]); print('collective path, per-GPU batch $b: ms_per_step', round(d['ms_per_step'],3), d.get('allreduce'))"
done
