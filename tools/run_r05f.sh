#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05f; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -m pytest -q -s tests/test_f16x3_gpu.py > $O/pytest_f16x3.log 2>&1; echo "f16x3 rc $?"; tail -2 $O/pytest_f16x3.log
timeout -k 10 300 python -m pytest -q -s tests/test_golden_gpu.py -k "well_conditioned" > $O/pytest_g193.log 2>&1; echo "g193 rc $?"; grep -h "GRADNORM193" $O/pytest_g193.log
for p in 2 3; do
  CL4_TRAIN_PRECISION=$p timeout -k 10 300 python bench.py --config p1 --steps 10 > $O/p1_prec$p.json 2> $O/p1_prec$p.err; echo "p1/$p rc $?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r05f/p*_prec*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms", round(d["value"], 1), "img/s", {k: (round(v["ms_per_step"], 2), round(v["achieved"], 1)) for k, v in d.get("kinds", {}).items()})
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_acceptance_gpu.py::test_map_after_training_is_an_unremarkable_draw_from_the_reference_distribution > $O/pytest_suite.log 2>&1; echo "suite rc $?"; tail -12 $O/pytest_suite.log
for b in 2 4 8; do
  CL4_FORCE_COLLECTIVES=1 CL4_BENCH_PER_GPU_BATCH=$b timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 2> $O/b$b.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('b=$b collective path', round(d['ms_per_step'],2), 'ms')"
  CL4_DEEP_ORDER=old CL4_FORCE_COLLECTIVES=1 CL4_BENCH_PER_GPU_BATCH=$b timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured --no-phase1 2>> $O/b$b.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('b=$b collective path, r04 chain order', round(d['ms_per_step'],2), 'ms')"
done
