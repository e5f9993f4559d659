cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "big_tile" > $O/tests_b.log 2>&1 || { tail -40 $O/tests_b.log | cut -c1-300; exit 1; }
tail -1 $O/tests_b.log
for v in hip bs0; do
  echo "== library $v" | tee -a $O/probe.txt
  CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_$v.so PLAN=4,144,1 timeout -k 10 120 python tools/probe_s_epi.py 32 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
  CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_$v.so PLAN=5,144,1 timeout -k 10 120 python tools/probe_s_epi.py 16 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
done
