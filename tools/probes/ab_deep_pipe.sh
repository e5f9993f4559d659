# A/B of the deep-pipelined replay (CL4_DEEP_PIPE) at the per-GPU batches an N = 1 / 2 / 4 / 8 run executes, on the one-process path and on the
# N > 1 code path rehearsed at world size 1 (CL4_FORCE_COLLECTIVES=1). usage: bash tools/probes/ab_deep_pipe.sh TAG
set -e
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r04}_deep
mkdir -p $O
for coll in 0 1; do
  for b in 2 4 8 16; do
    for deep in 0 1; do
      CL4_DEEP_PIPE=$deep CL4_DEEP_PIPE_MAX_BATCH=16 CL4_FORCE_COLLECTIVES=$coll CL4_BENCH_PER_GPU_BATCH=$b MASTER_PORT=$((29700 + 10 * b + deep + 100 * coll)) \
        timeout -k 10 200 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-structured > $O/c${coll}_b${b}_d$deep.json 2> $O/c${coll}_b${b}_d$deep.err \
        || { tail -5 $O/c${coll}_b${b}_d$deep.err; exit 1; }
      python -c "import json; d=json.loads(open('$O/c${coll}_b${b}_d$deep.json').read().strip().splitlines()[-1]); print('collectives $coll, per-GPU batch $b, deep $deep: ms_per_step', round(d['ms_per_step'],3), 'final losses', d.get('final_losses'))"
    done
  done
done
