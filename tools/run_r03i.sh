cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
python -m pytest tests/test_stem_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('bench inplace', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
CL4_INPLACE_TRUNK=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench0.json 2> $O/bench0.err
python -c "import json; d=json.load(open('$O/bench0.json')); print('bench out-of-place', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof.err
cd $R
python tools/step_timeline.py $O/prof 8 4 > $O/timeline_graph.txt; cat $O/timeline_graph.txt
python tools/roofline_leg_stats.py $O/prof 13 2 > $O/roofline_leg.txt; head -45 $O/roofline_leg.txt
rm -rf $O/prof/*/*.db 2>/dev/null; du -sh $O/prof
