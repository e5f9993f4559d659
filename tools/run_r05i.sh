#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05i; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests -q -m gpu -x --deselect tests/test_acceptance_gpu.py::test_map_after_training_is_an_unremarkable_draw_from_the_reference_distribution > $O/pytest_suite.log 2>&1; echo "suite rc $?"; grep -h "^FAILED\|^ERROR" $O/pytest_suite.log | sed 's/ - .*//' | head; grep -h "passed\|failed" $O/pytest_suite.log | tail -1
for v in f16 bf16 f16 bf16 f16 bf16; do
  L=""; [ $v = bf16 ] && L=$PWD/cl4wsis_amd/libcl4wsis_bf16planes.so
  CL4_LIB=$L timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-structured --no-phase1 2> $O/p2_$v.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('p2 $v planes', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'img/s frac', round(d['roofline']['frac'],3))"
done
timeout -k 10 300 python bench.py --config p1 --steps 10 2> $O/p1.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('p1', round(d['ms_per_step'],2), 'ms', {k: (round(v['ms_per_step'], 2), round(v['achieved'], 1)) for k, v in d['kinds'].items()})"
