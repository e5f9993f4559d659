set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s2
for mf in 32 16; do
  CL4_CONV_S_MFMA=$mf timeout -k 10 300 python -m pytest tests/test_conv_s_gpu.py -x -q -m gpu > gpurun_out/s2/test_s$mf.log 2>&1 || { tail -40 gpurun_out/s2/test_s$mf.log; exit 1; }
  tail -1 gpurun_out/s2/test_s$mf.log
  CL4_CONV_S_MFMA=$mf timeout -k 10 200 python tools/probe_s.py planes 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2/probe_s$mf.txt
done
for mf in 32 16 32 16; do
  CL4_CONV_S_MFMA=$mf timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured > gpurun_out/s2/bench_mf$mf.json 2> gpurun_out/s2/bench_mf$mf.err || tail -5 gpurun_out/s2/bench_mf$mf.err
  python -c "
import json
d=json.load(open('gpurun_out/s2/bench_mf$mf.json')); print('mf$mf', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
