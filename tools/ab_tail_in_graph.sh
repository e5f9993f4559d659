# A/B on the N > 1 code path at world size 1: the ASPP backward inside the step graph with the gradient buckets behind it (CL4_TAIL_IN_GRAPH_MAX_BATCH,
# default 4 images per GPU) against the eager tail with bucket-by-bucket overlap (0). usage: bash tools/ab_tail_in_graph.sh TAG
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r05}_tailgraph
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_ddp_trainer_gpu.py tests/test_golden_gpu.py tests/test_bench_two_ranks_gpu.py tests/test_graph_safety_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
for b in 2 4; do
  for mb in 4 0 4 0; do
    CL4_TAIL_IN_GRAPH_MAX_BATCH=$mb CL4_FORCE_COLLECTIVES=1 CL4_BENCH_PER_GPU_BATCH=$b MASTER_PORT=$((29820 + 10 * b + mb)) timeout -k 10 200 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 > $O/b${b}_mb$mb.json 2> $O/b${b}_mb$mb.err || { tail -5 $O/b${b}_mb$mb.err; exit 1; }
    python -c "import json; d=json.loads(open('$O/b${b}_mb$mb.json').read().strip().splitlines()[-1]); print('per-GPU batch $b, tail inside the graph up to $mb images: ms_per_step', round(d['ms_per_step'],3), 'exposed all-reduce', round(d.get('allreduce', {}).get('exposed_ms_per_step', 0), 3))" | tee -a $O/summary.txt
  done
done
CL4_BENCH_PER_GPU_BATCH=2 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one process, per-GPU batch 2:', round(d['ms_per_step'],3))" | tee -a $O/summary.txt
CL4_BENCH_PER_GPU_BATCH=4 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one process, per-GPU batch 4:', round(d['ms_per_step'],3))" | tee -a $O/summary.txt
