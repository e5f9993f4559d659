set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01j_p1
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01j_p1/prof -- python3 $R/tools/bench_p1.py 16 5 5 > $R/gpurun_out/r01j_p1/bench.json 2> $R/gpurun_out/r01j_p1/prof.err
tail -c 300 $R/gpurun_out/r01j_p1/bench.json
