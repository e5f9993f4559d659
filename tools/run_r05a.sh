#!/bin/bash
# r05 call A: conv_b knock-outs + stagger probe, phase-1 precision A/B, new tests
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 240 python tools/probe_b.py 0 300 600 900 1200 1800 > $O/probe_b_default.txt 2>&1 || exit 1
for v in 8 16 24; do
  CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_bko$v.so timeout -k 10 120 python tools/probe_b.py 0 > $O/probe_b_ko$v.txt 2>&1 || exit 1
done
echo probes done
CL4_TRAIN_PRECISION=2 timeout -k 10 300 python bench.py --config p1 --steps 10 > $O/p1_prec2.json 2> $O/p1_prec2.err || exit 1
CL4_TRAIN_PRECISION=1 timeout -k 10 300 python bench.py --config p1 --steps 10 > $O/p1_prec1.json 2> $O/p1_prec1.err || exit 1
echo p1 benches done
timeout -k 10 900 python -m pytest -x -q -s tests/test_phase1_gpu.py::test_phase1_adam_step_count_survives_graph_replays tests/test_bench_two_ranks_gpu.py "tests/test_acceptance_gpu.py::test_phase1_gradients_well_conditioned" > $O/pytest.log 2>&1
echo pytest rc $?
tail -5 $O/pytest.log
