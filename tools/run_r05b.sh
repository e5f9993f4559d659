#!/bin/bash
# r05 call B: fp16x3 feasibility, cross-CU stagger (cycles this time), two workgroups per CU on the 2-stage conv_b tile
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05b; mkdir -p $O
export PYTHONUNBUFFERED=1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/f16x3_probe tools/f16x3_probe.hip && timeout -k 10 120 /tmp/f16x3_probe > $O/f16x3_probe.txt 2>&1 || exit 1
echo f16 probe done
PROBE_B_PLANS=4 timeout -k 10 200 python tools/probe_b.py 0 10000 20000 30000 45000 > $O/stagger_cross_cu.txt 2>&1 || exit 1
CL4_CONV_B_STAGES=2 PROBE_B_PLANS=6 PROBE_B_SHAPES=closing timeout -k 10 200 python tools/probe_b.py 0 -5000 -10000 -15000 -20000 -30000 > $O/two_wg_per_cu.txt 2>&1 || exit 1
echo probes done
timeout -k 10 600 python -m pytest -x -q -s tests/test_ddp_phase1_gpu.py > $O/pytest_ddp_p1.log 2>&1
echo pytest rc $?
tail -5 $O/pytest_ddp_p1.log
