cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests/test_wgrad_gemm_gpu.py -x -q -m gpu 2>&1 | tail -1
for m in 1 0; do echo "CL4_GEMM_XCD_2D=$m"; CL4_GEMM_XCD_2D=$m timeout -k 10 120 python tools/probe_wgrad_gemm.py 6 2>&1 | grep "gemm "; done
