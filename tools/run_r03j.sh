cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "persistent" > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for plan in 2,128,1 3,128,1; do echo "== plan $plan"; PLAN=$plan timeout -k 10 200 python tools/probe_s_epi.py 32 16 2>&1 | grep "conv3 \|res5 conv1\|res4 conv1" | grep -v "fp32 out"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('bench conv_p on', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
CL4_CONV_P=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench0.json 2> $O/bench0.err
python -c "import json; d=json.load(open('$O/bench0.json')); print('bench conv_p off', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
