# Round-2 measurement pass: default bench (all legs), then the same command under rocprofv3 --kernel-trace --stats.
set -e
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 1500 gpurun_out/$TAG/bench.json
export TMPDIR=/tmp
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/$TAG/prof_bench.json 2> $R/gpurun_out/$TAG/prof.err
cd $R
python tools/roofline_leg_stats.py gpurun_out/$TAG/prof 13 2 > gpurun_out/$TAG/roofline_leg.txt
python tools/step_breakdown.py gpurun_out/$TAG/prof_graphsteps 3 40 > gpurun_out/$TAG/step_breakdown.txt 2>/dev/null || true
head -5 gpurun_out/$TAG/roofline_leg.txt
rm -f gpurun_out/$TAG/prof/*/*_kernel_trace.csv.bak
