#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05l; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest -q -x tests/test_phase1_gpu.py tests/test_body_train_gpu.py tests/test_heads_train_gpu.py tests/test_step0_gpu.py tests/test_wrn_train_gpu.py tests/test_abn_gpu.py tests/test_module_surface_gpu.py "tests/test_acceptance_gpu.py::test_phase1_gradients_well_conditioned" tests/test_ddp_phase1_gpu.py tests/test_cli_gpu.py > $O/pytest.log 2>&1; echo "tests rc $?"; tail -4 $O/pytest.log
for ft in 1 0 1 0; do
  CL4_FUSE_BLOCK_TAIL=$ft timeout -k 10 300 python bench.py --config p1 --steps 10 2> $O/p1_$ft.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('p1 fused block tail $ft', round(d['ms_per_step'],2), 'ms', {k: (round(v['ms_per_step'], 2), round(v['achieved'], 1)) for k, v in d['kinds'].items()})"
done
