set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01n
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r01n/bench.json 2> gpurun_out/r01n/bench.err
tail -c 600 gpurun_out/r01n/bench.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01n/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01n/prof_bench.json 2> $R/gpurun_out/r01n/prof.err
ls -R $R/gpurun_out/r01n/prof | head -20
