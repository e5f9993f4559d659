# HBM traffic of the dominant kernel: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over eager, non-overlapped
# bench steps, plus calibration launches of the same kernel with known traffic; no tracing options alongside --pmc.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
O=$R/gpurun_out/${TAG}_pmc
mkdir -p $O
export TMPDIR=/tmp CL4_GRAPH=0 CL4_OVERLAP=0
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/bench_$c -- python3 $R/bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-structured --no-phase1 > $O/bench_$c.json 2> $O/bench_$c.err
  echo "pass bench $c done"
  CL4_CONV_VEC=1 timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/calib1_$c -- python3 $R/tools/pmc_calib.py > $O/calib1_$c.log 2>&1
  CL4_CONV_VEC=0 CL4_CONV_U=0 timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/calib0_$c -- python3 $R/tools/pmc_calib.py > $O/calib0_$c.log 2>&1
  echo "pass calib $c done"
done
# third pass: matrix-pipe busy cycles of every launch of the same eager step (SQ + GRBM counters fit one pass)
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/bench_MFMA -- python3 $R/bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-structured --no-phase1 > $O/bench_MFMA.json 2> $O/bench_MFMA.err
echo "pass bench MFMA done"
python3 $R/tools/pmc_traffic.py $O $O/traffic.json
# per-kernel table (needs the kernel trace of tools/prof_bench_trace.sh for the durations): python3 tools/pmc_kernels.py $O <trace dir> 13 <out.json>
