# Per-GPU batch 2 (what each rank of an 8-GPU run executes): step time and kernel breakdown of graph-replayed steps.
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02_b2
export TMPDIR=/tmp CL4_BENCH_PER_GPU_BATCH=${1:-2}
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured > gpurun_out/r02_b2/bench.json 2> gpurun_out/r02_b2/bench.err
python -c "import json; d=json.load(open('gpurun_out/r02_b2/bench.json')); print('ms_per_step', d['ms_per_step'])"
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02_b2/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-structured > $R/gpurun_out/r02_b2/prof_bench.json 2> $R/gpurun_out/r02_b2/prof.err
cd $R
python tools/roofline_leg_stats.py gpurun_out/r02_b2/prof 8 4 > gpurun_out/r02_b2/step_breakdown_graph.txt
head -45 gpurun_out/r02_b2/step_breakdown_graph.txt
