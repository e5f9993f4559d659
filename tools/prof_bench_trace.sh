# Measurement pass of a round: (1) the default bench command (all legs) and (2) the same workload under rocprofv3 --kernel-trace
# --stats, from which the per-step tables are cut: graph-replayed steps (the timed mode) and the two eager overlap-off steps of
# the roofline leg (the mode roofline.frac is measured in).
set -e
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -5 gpurun_out/$TAG/bench.err; exit 1; }
tail -c 1200 gpurun_out/$TAG/bench.json
export TMPDIR=/tmp
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase1 > $R/gpurun_out/$TAG/prof_bench.json 2> $R/gpurun_out/$TAG/prof.err
cd $R
# steps of the profiled process: 0-2 warm-up, 3-12 timed graph replays, 13-14 eager roofline leg, 15-16 + 17-26 structured-labels leg
python tools/roofline_leg_stats.py gpurun_out/$TAG/prof 13 2 > gpurun_out/$TAG/roofline_leg.txt
python tools/roofline_leg_stats.py gpurun_out/$TAG/prof 8 4 > gpurun_out/$TAG/step_breakdown_graph.txt
python tools/roofline_leg_stats.py gpurun_out/$TAG/prof 21 4 > gpurun_out/$TAG/step_breakdown_structured.txt
cp gpurun_out/$TAG/prof/*/*_kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv 2>/dev/null || true
head -4 gpurun_out/$TAG/roofline_leg.txt; head -3 gpurun_out/$TAG/step_breakdown_graph.txt
python - <<PY
import json
d = json.load(open("gpurun_out/$TAG/prof_bench.json"))
print("profiled run roofline:", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "kernel_ms_per_step", "launches_per_step")})
PY
