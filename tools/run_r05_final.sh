#!/bin/bash
# r05 measurement pass, part 2: small per-GPU batches (one-process and collective path), phase 1 with its layer table, BASELINE configs 1 and 5,
# the chain Gantt, the whole GPU suite with timings. Part 1 = tools/run_final.sh <tag> (default bench, kernel trace, PMC passes, per-kernel table, soak).
TAG=${1:-r05y}
cd "${GRAFT_REPO_ROOT:?run under gpurun}"
O=gpurun_out/${TAG}_extra; mkdir -p $O
export PYTHONUNBUFFERED=1
bash tools/run_small_batches.sh $TAG || exit 1
cd "${GRAFT_REPO_ROOT:?}"; bash tools/run_solo_collectives.sh $TAG || exit 1
cd "${GRAFT_REPO_ROOT:?}"
CL4_BENCH_LAYER_TABLE_P1=$O/p1_layer_table.txt timeout -k 10 300 python bench.py --config p1 --steps 20 > $O/p1_bench.json 2> $O/p1_bench.err; echo "p1 rc $?"
timeout -k 10 300 python tools/bench_s0.py 4 5 kinds > $O/s0_bench_b4.txt 2> $O/s0.err; echo "s0 b4 rc $?"
timeout -k 10 300 python tools/bench_s0.py 16 5 kinds > $O/s0_bench_b16.txt 2>> $O/s0.err; echo "s0 b16 rc $?"
timeout -k 10 300 python tools/bench_cfg5.py 4 10 > $O/cfg5_bench.json 2> $O/cfg5.err; echo "cfg5 rc $?"
CL4_FORCE_COLLECTIVES=1 timeout -k 10 400 python tools/chain_gantt.py 2 4 8 2> $O/gantt.err | grep -v "socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/chain_gantt_collective_path.txt; echo "gantt rc $?"
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=12 > $O/gpu_test_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_test_suite.log
head -c 600 $O/p1_bench.json; echo; cat $O/s0_bench_b4.txt | head -3; head -c 400 $O/cfg5_bench.json
