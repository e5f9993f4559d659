#!/bin/bash
# r05 call C: fp16x3 kernels (accuracy, phase-1 / phase-2 conditioning, speed), restored tolerance tests, chain Gantt
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -m pytest -x -q -s tests/test_f16x3_gpu.py > $O/pytest_f16x3.log 2>&1; echo "f16x3 rc $?"; tail -3 $O/pytest_f16x3.log
timeout -k 10 600 python -m pytest -q -s "tests/test_acceptance_gpu.py::test_phase1_gradients_well_conditioned" "tests/test_acceptance_gpu.py::test_phase2_gradients_well_conditioned" > $O/pytest_wellcond.log 2>&1; echo "wellcond rc $?"; grep -h "WELLCOND phase\|bars\|passed\|failed" $O/pytest_wellcond.log
CL4_TRAIN_PRECISION=3 timeout -k 10 300 python bench.py --config p1 --steps 10 > $O/p1_prec3.json 2> $O/p1_prec3.err; echo "p1 rc $?"
CL4_TRAIN_PRECISION=3 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-structured --no-phase1 > $O/p2_prec3.json 2> $O/p2_prec3.err; echo "p2/3 rc $?"
CL4_TRAIN_PRECISION=2 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-structured --no-phase1 > $O/p2_prec2.json 2> $O/p2_prec2.err; echo "p2/2 rc $?"
timeout -k 10 600 python -m pytest -x -q -s tests/test_golden_gpu.py -k "well_conditioned or pipelined_replay or networks_match" > $O/pytest_golden.log 2>&1; echo "golden rc $?"; grep -h "GRADNORM193\|PIPEREPLAY\|passed\|failed" $O/pytest_golden.log
CL4_FORCE_COLLECTIVES=1 timeout -k 10 400 python tools/chain_gantt.py 2 4 8 > $O/chain_gantt_collective.txt 2>&1; echo "gantt rc $?"
