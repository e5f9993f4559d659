# Phase probes of the bf16x3 kernel over the instrumented builds (tools/probe_phases.py).
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/phases
for v in tl2 tl2d1 tl2d2 tl2d4 tl2d6; do
  CL4_LIB=$R/cl4wsis_amd/libcl4wsis_$v.so timeout -k 10 120 python tools/probe_phases.py $v > gpurun_out/phases/$v.txt 2>&1 || echo "variant $v failed"
  echo "done $v"
done
timeout -k 10 300 python -m pytest tests/test_backward_blocks_gpu.py -x -q -m gpu > gpurun_out/phases/test_bwd.log 2>&1 || echo "bwd test failed"
tail -3 gpurun_out/phases/test_bwd.log
