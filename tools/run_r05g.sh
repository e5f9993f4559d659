#!/bin/bash
# r05 call G: the two-plane format of the whole library is fp16x3 — whole suite (all failures), conditioning, bench A/B against the bf16 build
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05g; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests -q -m gpu --deselect tests/test_acceptance_gpu.py::test_map_after_training_is_an_unremarkable_draw_from_the_reference_distribution > $O/pytest_suite.log 2>&1; echo "suite rc $?"; grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_suite.log | head -60
timeout -k 10 300 python -m pytest -q -s tests/test_golden_gpu.py -k "well_conditioned" "tests/test_acceptance_gpu.py::test_phase2_gradients_well_conditioned" "tests/test_acceptance_gpu.py::test_phase1_gradients_well_conditioned" > $O/pytest_cond.log 2>&1; echo "cond rc $?"; grep -h "GRADNORM193 frozen\|WELLCOND phase\|bars" $O/pytest_cond.log
for v in f16 bf16 f16 bf16; do
  L=""; [ $v = bf16 ] && L=$PWD/cl4wsis_amd/libcl4wsis_bf16planes.so
  CL4_LIB=$L timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-structured --no-phase1 2> $O/p2_$v.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('p2 $v planes', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'img/s frac', round(d['roofline']['frac'],3))"
done
