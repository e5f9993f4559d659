# A/B of the K-split LDS-DMA kernel as a tuner candidate (CL4_CONV_B_SPLIT) at the per-GPU batches of an N = 2 / 4 / 8 run, on the N > 1 code path
# rehearsed at world size 1. usage: bash tools/ab_conv_b_split.sh TAG
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r05}_bsplit
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_conv_s_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
for b in 2 4 8; do
  for sp in 1 0 1 0; do
    CL4_CONV_B_SPLIT=$sp CL4_FORCE_COLLECTIVES=1 CL4_BENCH_PER_GPU_BATCH=$b MASTER_PORT=$((29800 + 10 * b + sp)) timeout -k 10 200 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 > $O/b${b}_sp$sp.json 2> $O/b${b}_sp$sp.err || { tail -5 $O/b${b}_sp$sp.err; exit 1; }
    python -c "import json; d=json.loads(open('$O/b${b}_sp$sp.json').read().strip().splitlines()[-1]); print('per-GPU batch $b, K-split LDS-DMA candidate $sp: ms_per_step', round(d['ms_per_step'],3))" | tee -a $O/summary.txt
  done
done
