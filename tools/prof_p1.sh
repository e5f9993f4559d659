# Phase-1 step (BASELINE config 2) under rocprofv3 --kernel-trace: per-kernel table of four graph-replayed steps. usage: bash tools/prof_p1.sh TAG
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=$R/gpurun_out/${1:-r05}_p1_prof
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config p1 --steps 20 > $O/prof_bench.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
cd $R
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv
python tools/step_kernels.py $O/prof sgd_kernel 5 12 4 > $O/step_kernels_graph.txt 2>&1
rm -rf $O/prof
head -70 $O/step_kernels_graph.txt | cut -c1-170
