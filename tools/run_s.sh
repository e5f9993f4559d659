set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s
timeout -k 10 300 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu > gpurun_out/s/test_s.log 2>&1 || { tail -40 gpurun_out/s/test_s.log; exit 1; }
tail -3 gpurun_out/s/test_s.log
CL4_CONV_S_MFMA=16 timeout -k 10 300 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu > gpurun_out/s/test_s16.log 2>&1 || { tail -40 gpurun_out/s/test_s16.log; exit 1; }
tail -3 gpurun_out/s/test_s16.log
timeout -k 10 200 python tools/probe_s.py planes 30 15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s/probe_s32.txt
CL4_CONV_S_MFMA=16 timeout -k 10 200 python tools/probe_s.py planes 30 15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s/probe_s16.txt
timeout -k 10 200 python tools/probe_s.py fp32 30 15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s/probe_u.txt
