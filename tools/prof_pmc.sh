# HBM traffic of the dominant kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over eager, non-overlapped
# bench steps; no tracing options alongside --pmc.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01k_pmc
export TMPDIR=/tmp CL4_GRAPH=0 CL4_OVERLAP=0
cd /tmp
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01k_pmc/fetch -- python3 $R/bench.py --steps 2 --warmup 4 --no-cpu-baseline > $R/gpurun_out/r01k_pmc/fetch.json 2> $R/gpurun_out/r01k_pmc/fetch.err
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01k_pmc/write -- python3 $R/bench.py --steps 2 --warmup 4 --no-cpu-baseline > $R/gpurun_out/r01k_pmc/write.json 2> $R/gpurun_out/r01k_pmc/write.err
python3 $R/tools/pmc_traffic.py $R/gpurun_out/r01k_pmc/fetch $R/gpurun_out/r01k_pmc/write $R/gpurun_out/r01k_pmc/traffic.json
