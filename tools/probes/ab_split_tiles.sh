# Experiment: below how many 128-row tiles a launch may take a whole-launch K split (tuner candidates), per per-GPU batch.
set -e
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r04}_margin
mkdir -p $O
for b in 2 4 8 16; do
  for m in 256 128 96 64; do
    CL4_TUNE_SPLIT_MAX_TILES=$m CL4_BENCH_PER_GPU_BATCH=$b timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-structured > $O/b${b}_m$m.json 2> $O/b${b}_m$m.err || { tail -5 $O/b${b}_m$m.err; exit 1; }
    python -c "import json; d=json.loads(open('$O/b${b}_m$m.json').read().strip().splitlines()[-1]); print('per-GPU batch $b, whole-launch K splits below $m tiles: ms_per_step', round(d['ms_per_step'],3))"
  done
done
