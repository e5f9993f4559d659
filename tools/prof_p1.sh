# Phase-1 step under rocprofv3 --kernel-trace --stats with the ABN-backward gradient planes on / off: per-kernel totals of the whole run
# (warm-up + eager roofline leg included; compare the two tables). usage: bash tools/prof_p1.sh TAG
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=$R/gpurun_out/${1:-r05}_p1_prof
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for gp in 1 0; do
  export CL4_GRAD_PLANES=$gp
  cd $R && timeout -k 10 300 python bench.py --config p1 --steps 20 > $O/p1_gp$gp.json 2> $O/p1_gp$gp.err || { tail -5 $O/p1_gp$gp.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/p1_gp$gp.json').read().strip().splitlines()[-1]); print('gradient planes $gp: ms_per_step', round(d['ms_per_step'],3))"
  cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gp$gp -- python3 $R/bench.py --config p1 --steps 20 > $O/prof_gp$gp.json 2> $O/prof_gp$gp.err || { tail -5 $O/prof_gp$gp.err; exit 1; }
  cp $O/prof_gp$gp/*/*_kernel_stats.csv $O/kernel_stats_gp$gp.csv
  rm -rf $O/prof_gp$gp
  head -30 $O/kernel_stats_gp$gp.csv | cut -c1-150
done
