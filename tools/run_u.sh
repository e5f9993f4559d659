set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/u
timeout -k 10 300 python -m pytest tests/test_conv_u_gpu.py -x -q -m gpu > gpurun_out/u/test_u.log 2>&1 || { tail -30 gpurun_out/u/test_u.log; exit 1; }
tail -3 gpurun_out/u/test_u.log
CL4_CONV_U=1 timeout -k 10 200 python tools/probe_u.py > gpurun_out/u/probe_u1.txt 2>&1
CL4_CONV_U=0 timeout -k 10 200 python tools/probe_u.py > gpurun_out/u/probe_u0.txt 2>&1
paste gpurun_out/u/probe_u1.txt gpurun_out/u/probe_u0.txt | cut -c1-130
