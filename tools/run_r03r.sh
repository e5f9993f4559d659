cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "big_tile" > $O/tests_b.log 2>&1 || { tail -40 $O/tests_b.log | cut -c1-300; exit 1; }
tail -1 $O/tests_b.log
for plan in 2,144,1 4,144,1; do
  PLAN=$plan timeout -k 10 120 python tools/probe_s_epi.py 32 16 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
done
timeout -k 10 400 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "not big_tile" > $O/tests_s.log 2>&1 || { tail -40 $O/tests_s.log | cut -c1-300; exit 1; }
tail -1 $O/tests_s.log
