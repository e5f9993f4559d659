#!/bin/bash
# Experiments: conv_f.hip with parts of it compiled out (CL4_F_DBG bits), timed by tools/probes/probe_f.py. Run under gpurun.
cd "${GRAFT_REPO_ROOT:?run under gpurun}"
mkdir -p gpurun_out
for v in 0 1 2 4 6 7 8 16 15 23 31; do
  lib=$(python -m cl4wsis_amd.build --variant fdbg$v --src conv_f.hip CL4_F_DBG=$v 2>/dev/null | tail -1)
  echo "CL4_F_DBG=$v"
  CL4_LIB=$lib timeout -k 10 120 python tools/probes/probe_f.py ${PROBE_N:-32} 2>&1 | grep "fused"
done
