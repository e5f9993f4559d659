# Soak of the N > 1 code path at world size 1 with deep pipelining, small per-GPU batches: N runs each, stop at the first failure / hang.
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r04}_soakc
mkdir -p $O
for b in 2 4 8; do
  for i in $(seq 1 ${2:-6}); do
    CL4_BENCH_VERBOSE=1 CL4_FORCE_COLLECTIVES=1 CL4_BENCH_PER_GPU_BATCH=$b MASTER_PORT=$((29900 + 10 * b + i)) timeout -k 10 120 python bench.py --gpus 1 --steps 40 --warmup 3 --no-cpu-baseline > $O/b${b}_$i.json 2> $O/b${b}_$i.err
    rc=$?
    if [ $rc != 0 ]; then echo "b $b run $i FAILED rc $rc"; tail -5 $O/b${b}_$i.err; exit 1; fi
    python -c "import json; d=json.loads(open('$O/b${b}_$i.json').read().strip().splitlines()[-1]); print('b $b run $i:', round(d['ms_per_step'],3), 'ms/step; structured', round(d['structured_labels']['ms_per_step'],3))"
  done
done
