cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu > $O/tests_conv.log 2>&1 || { tail -40 $O/tests_conv.log; exit 1; }
tail -1 $O/tests_conv.log
python tools/dbg_pool_grad.py > $O/pool_grad.txt 2>&1; grep -v amdgpu.ids $O/pool_grad.txt | tail -8
timeout -k 10 500 python -m pytest tests/test_acceptance_gpu.py::test_map_after_training_inside_the_reference_arithmetics_own_spread tests/test_ddp_trainer_gpu.py -q -m gpu -rA --durations=5 > $O/acceptance.log 2>&1
echo "rc $?"
grep "MAPPROXY\|DDP-EMU\|passed\|failed\|^FAILED\|^E  \| call " $O/acceptance.log | cut -c1-600
PLAN=2,144,1 python tools/probe_s_epi.py 32 16 2>&1 | grep "aspp\|res5 conv2"
CL4_BENCH_LAYER_TABLE=$O/layers.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
