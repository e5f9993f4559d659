#!/bin/bash
# r05 call E: same-box A/B of fp16x3 against bf16x6 (phase 1, phase 2) with per-shape tables, then the whole GPU suite
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05e; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -m pytest -q -s tests/test_f16x3_gpu.py > $O/pytest_f16x3.log 2>&1; echo "f16x3 rc $?"; grep -h "F16X3 d" $O/pytest_f16x3.log; tail -2 $O/pytest_f16x3.log
for p in 2 3; do
  CL4_TRAIN_PRECISION=$p CL4_BENCH_LAYER_TABLE_P1=$O/p1_layers_prec$p.txt timeout -k 10 300 python bench.py --config p1 --steps 10 > $O/p1_prec$p.json 2> $O/p1_prec$p.err; echo "p1/$p rc $?"
done
for p in 2 3 2 3; do
  CL4_TRAIN_PRECISION=$p CL4_BENCH_LAYER_TABLE=$O/p2_layers_prec$p.txt timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-structured --no-phase1 > $O/p2_prec${p}_$RANDOM.json 2> $O/p2_prec$p.err; echo "p2/$p rc $?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r05e/p*_prec*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f.split("/")[-1], round(d["ms_per_step"], 2), "ms", round(d["value"], 1), "img/s", {k: (round(v["ms_per_step"], 2), round(v["achieved"], 1)) for k, v in d.get("kinds", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest_suite.log 2>&1; echo "suite rc $?"; tail -25 $O/pytest_suite.log
