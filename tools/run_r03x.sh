cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "plain or fused or split_plans or channel_slices" > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for plan in 2,64,1 2,144,1; do
PLAN=$plan timeout -k 10 120 python tools/probe_s_epi.py 32 16 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
done
