# r05 working check: parity subsets, phase-1 and default bench lines. usage: bash tools/run_r05_check.sh TAG
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r05}_check
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_abn_gpu.py tests/test_backward_blocks_gpu.py tests/test_f16x3_gpu.py tests/test_phase1_gpu.py tests/test_step0_gpu.py tests/test_golden_gpu.py tests/test_ddp_phase1_gpu.py tests/test_heads_train_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --config p1 --steps 20 > $O/p1.json 2> $O/p1.err; echo "p1 rc $?"
python -c "import json; d=json.loads(open('$O/p1.json').read().strip().splitlines()[-1]); print('p1 ms_per_step', round(d['ms_per_step'],3))"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-phase1 --no-structured > $O/p2.json 2> $O/p2.err; echo "p2 rc $?"
python -c "import json; d=json.loads(open('$O/p2.json').read().strip().splitlines()[-1]); print('p2 ms_per_step', round(d['ms_per_step'],3), 'images/s', round(d['value'],1))"
for f in 1 0 1 0; do
  CL4_BN_SPLIT_FIXED=$f CL4_MAXPOOL_BWD_TILE=$((1 - f)) timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-phase1 --no-structured > $O/p2_fixed$f.json 2> $O/p2_fixed$f.err
  python -c "import json; d=json.loads(open('$O/p2_fixed$f.json').read().strip().splitlines()[-1]); print('fixed reduction grid $f: p2 ms_per_step', round(d['ms_per_step'],3), 'images/s', round(d['value'],1))"
  CL4_BN_SPLIT_FIXED=$f CL4_MAXPOOL_BWD_TILE=$((1 - f)) timeout -k 10 300 python bench.py --config p1 --steps 20 > $O/p1_fixed$f.json 2> $O/p1_fixed$f.err
  python -c "import json; d=json.loads(open('$O/p1_fixed$f.json').read().strip().splitlines()[-1]); print('fixed reduction grid $f, old maxpool backward $f: p1 ms_per_step', round(d['ms_per_step'],3))"
done
