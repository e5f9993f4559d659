cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_wgrad_gemm_gpu.py tests/test_conv_s_gpu.py -x -q -m gpu -k "wgrad or big_tile or three_plane" > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for st in 4 3; do echo "CL4_GEMM_STAGES=$st"; CL4_GEMM_STAGES=$st timeout -k 10 120 python tools/probe_wgrad_gemm.py 6 2>&1 | grep "gemm "; done
PLAN=6,144,1 timeout -k 10 120 python tools/probe_s_epi.py 16 2>&1 | grep -v amdgpu.ids | grep "conv1\|conv2\|aspp"
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench$i.json 2> $O/bench$i.err || { tail -5 $O/bench$i.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench$i.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['conv_wgrad']['ms_per_step'])"
done
