set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01i_b2
export TMPDIR=/tmp
export CL4_BENCH_PER_GPU_BATCH=2
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r01i_b2/bench.json 2> gpurun_out/r01i_b2/bench.err
python -c "import json; d=json.load(open('gpurun_out/r01i_b2/bench.json')); print(d['ms_per_step'])"
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01i_b2/prof -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01i_b2/prof_bench.json 2> $R/gpurun_out/r01i_b2/prof.err
