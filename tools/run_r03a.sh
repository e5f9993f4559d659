set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
CL4_BENCH_LAYER_TABLE=gpurun_out/r03a/layers.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
tail -c 600 gpurun_out/r03a/bench.json
python tools/probe_s.py planes 32 16 > gpurun_out/r03a/probe_s.txt 2>&1
bash tools/soak_single_stream.sh r03a 10 30
