cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
for v in hip nt1 nt2 nt3; do
  echo "== library $v (CL4_EPI_NT bits: 1 nt residual loads, 2 nt planes stores)" | tee -a $O/probe.txt
  CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_$v.so PLAN=2,144,1 timeout -k 10 120 python tools/probe_s_epi.py 32 2>&1 | grep -v amdgpu.ids | grep "conv3\|plan" | tee -a $O/probe.txt
done
