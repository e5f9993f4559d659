# Experiment: HIP stream priorities for the two side streams of the deep-pipelined replay at small per-GPU batches.
set -e
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r04}_prio
mkdir -p $O
for b in 2 4; do
  for pr in "0 0" "-1 0" "-1 -1" "0 -1"; do
    set -- $pr
    CL4_SIDE_PRIORITY=$1 CL4_SIDE2_PRIORITY=$2 CL4_BENCH_PER_GPU_BATCH=$b timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured > $O/b${b}_$1_$2.json 2> $O/b${b}_$1_$2.err || { tail -5 $O/b${b}_$1_$2.err; exit 1; }
    python -c "import json; d=json.loads(open('$O/b${b}_$1_$2.json').read().strip().splitlines()[-1]); print('per-GPU batch $b, priorities old-model stream $1 / new-model stream $2: ms_per_step', round(d['ms_per_step'],3))"
  done
done
