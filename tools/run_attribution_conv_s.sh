# K-loop attribution of conv_s (profiles/r03_conv_s_attribution.txt): build the variants first —
#   for d in 1 2 3 4 6; do python -m cl4wsis_amd.build --variant sd$d --src conv_s.hip CL4_S_DBG=$d; done
# (conv_b: --variant bd1 / bd4 --src conv_b.hip CL4_B_DBG=1 / 4, probe with PLAN=4,144,1)
cd "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}"
O=gpurun_out/attrib_s; mkdir -p $O
for v in hip sd1 sd2 sd3 sd4 sd6; do
  echo "== variant $v (CL4_S_DBG bits: 1 no loads in the K loop, 2 no LDS stores, 4 no MFMA)" | tee -a $O/attrib.txt
  CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_$v.so PLAN=2,144,1 timeout -k 10 120 python tools/probe_s_epi.py 32 2>&1 | grep -v amdgpu.ids | tee -a $O/attrib.txt
done
