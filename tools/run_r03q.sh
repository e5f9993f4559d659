cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
for v in hip sd1 sd2 sd3 sd4 sd6; do
  echo "== variant $v (CL4_S_DBG bits: 1 no loads in the K loop, 2 no LDS stores, 4 no MFMA)" | tee -a $O/attrib.txt
  CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_$v.so PLAN=2,144,1 timeout -k 10 120 python tools/probe_s_epi.py 32 2>&1 | grep -v amdgpu.ids | tee -a $O/attrib.txt
done
