cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu -k "big_tile" > $O/tests_b.log 2>&1 || { tail -40 $O/tests_b.log | cut -c1-300; exit 1; }
tail -1 $O/tests_b.log
PLAN=5,144,1 timeout -k 10 120 python tools/probe_s_epi.py 16 32 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
for v in bd1 bd4; do
  echo "== variant $v (CL4_B_DBG bits: 1 no LDS-DMA in the K loop, 4 no MFMA), plan 4,144,1" | tee -a $O/probe.txt
  CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_$v.so PLAN=4,144,1 timeout -k 10 120 python tools/probe_s_epi.py 32 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
done
