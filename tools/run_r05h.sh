#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05h; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests -q -m gpu --deselect tests/test_acceptance_gpu.py::test_map_after_training_is_an_unremarkable_draw_from_the_reference_distribution > $O/pytest_suite.log 2>&1; echo "suite rc $?"; grep -h "^FAILED\|^ERROR" $O/pytest_suite.log | sed 's/ - .*//' | head -40; grep -h "passed\|failed" $O/pytest_suite.log | tail -1
CL4_LIB=$PWD/cl4wsis_amd/libcl4wsis_bf16planes.so timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "conv_s or conv_f or stem or golden or model_gpu" > $O/pytest_bf16_build.log 2>&1; echo "bf16-build subset rc $?"; tail -3 $O/pytest_bf16_build.log
for v in f16 bf16 f16 bf16; do
  L=""; [ $v = bf16 ] && L=$PWD/cl4wsis_amd/libcl4wsis_bf16planes.so
  CL4_LIB=$L timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-structured --no-phase1 2> $O/p2_$v.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('p2 $v planes', round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'img/s frac', round(d['roofline']['frac'],3), 'losses', d['final_losses'])"
done
timeout -k 10 600 python -m pytest -q -s tests/test_acceptance_gpu.py::test_map_after_training_is_an_unremarkable_draw_from_the_reference_distribution > $O/pytest_map.log 2>&1; echo "map rc $?"; grep -h "MAPPROXY BIAS\|MAPPROXY VERDICT\|MAPPROXY reference\|passed\|failed" $O/pytest_map.log
