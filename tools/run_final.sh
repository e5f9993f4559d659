# Final measurement pass of a round: default bench + rocprof kernel trace (tools/prof_bench_trace.sh), timeline, PMC passes, per-kernel PMC table, soak.
TAG=${1:-r03y}
cd "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}"
bash tools/prof_bench_trace.sh $TAG || exit 1
python tools/step_timeline.py gpurun_out/$TAG/prof 8 4 > gpurun_out/$TAG/timeline.txt 2>&1
echo "trace done"
bash tools/prof_pmc.sh $TAG || exit 1
cd "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}"
python tools/pmc_kernels.py gpurun_out/${TAG}_pmc gpurun_out/$TAG/prof 13 gpurun_out/$TAG/pmc_kernels.json 0.15 > gpurun_out/$TAG/pmc_kernels.log 2>&1 || tail -5 gpurun_out/$TAG/pmc_kernels.log
echo "pmc done"
bash tools/soak_default.sh $TAG 6
