# Kernel trace of the replayed step at a small per-GPU batch (what a rank of an N = 8 run executes): tools/prof_small_batch.sh <tag> <batch>
TAG=${1:-small}; B=${2:-2}
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
export TMPDIR=/tmp
cd /tmp && CL4_BENCH_PER_GPU_BATCH=$B timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-structured > $R/gpurun_out/$TAG/prof_bench.json 2> $R/gpurun_out/$TAG/prof.err
cd $R
python tools/roofline_leg_stats.py gpurun_out/$TAG/prof 8 4 > gpurun_out/$TAG/step_breakdown_graph_b$B.txt
python tools/step_timeline.py gpurun_out/$TAG/prof 8 4 > gpurun_out/$TAG/timeline_b$B.txt 2>&1
head -45 gpurun_out/$TAG/step_breakdown_graph_b$B.txt | cut -c1-130
head -8 gpurun_out/$TAG/timeline_b$B.txt
