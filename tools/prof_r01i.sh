set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01i
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r01i/bench.json 2> gpurun_out/r01i/bench.err
tail -c 600 gpurun_out/r01i/bench.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01i/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01i/prof_bench.json 2> $R/gpurun_out/r01i/prof.err
ls -R $R/gpurun_out/r01i/prof | head -20
