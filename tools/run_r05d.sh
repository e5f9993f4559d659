#!/bin/bash
# r05 call D: fp16x3 after the split fix
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05d; mkdir -p $O
export PYTHONUNBUFFERED=1
python tests/tools/dbg_f16x3.py 2>&1 | grep -v amdgpu.ids | head -8
timeout -k 10 300 python -m pytest -x -q -s tests/test_f16x3_gpu.py > $O/pytest_f16x3.log 2>&1; echo "f16x3 rc $?"; grep -h "F16X3" $O/pytest_f16x3.log | head -30; tail -3 $O/pytest_f16x3.log
timeout -k 10 600 python -m pytest -q -s "tests/test_acceptance_gpu.py::test_phase1_gradients_well_conditioned" "tests/test_acceptance_gpu.py::test_phase2_gradients_well_conditioned" > $O/pytest_wellcond.log 2>&1; echo "wellcond rc $?"; grep -h "WELLCOND phase\|bars\|passed\|failed" $O/pytest_wellcond.log
CL4_TRAIN_PRECISION=3 timeout -k 10 300 python bench.py --config p1 --steps 10 > $O/p1_prec3.json 2> $O/p1_prec3.err; echo "p1 rc $?"
CL4_TRAIN_PRECISION=3 timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-structured --no-phase1 > $O/p2_prec3.json 2> $O/p2_prec3.err; echo "p2/3 rc $?"
python - <<PY
import json
for f in ("p1_prec3", "p2_prec3"):
    try:
        d = json.loads([l for l in open("gpurun_out/r05d/%s.json" % f) if l.startswith("{")][0])
        print(f, round(d["ms_per_step"], 2), "ms", round(d["value"], 1), "img/s", {k: (round(v["ms_per_step"], 2), round(v["achieved"], 1)) for k, v in d.get("kinds", {}).items()}, d.get("final_losses"))
    except Exception as e:
        print(f, "failed", e)
PY
