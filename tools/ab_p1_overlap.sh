# A/B of the r05 phase-1 / step-0 additions: one-launch weight packs and InPlaceABN scales (CL4_PACK_BATCH), the ABN backward handing its gradient
# on as fp16 planes (CL4_GRAD_PLANES), train-mode ABN without an exchange step at one rank (CL4_ABN_ONE_RANK), and the two second-stream
# experiments (CL4_P1_OVERLAP: the frozen old model beside the new model's forward; CL4_WGRAD_SIDE: weight gradients beside the input-gradient
# chain). usage: bash tools/ab_p1_overlap.sh TAG
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
O=gpurun_out/${1:-r05}_p1_ab
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_f16x3_gpu.py tests/test_abn_gpu.py tests/test_phase1_gpu.py tests/test_step0_gpu.py tests/test_ddp_phase1_gpu.py tests/test_wrn_train_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
for cfg in "1 1 1 0 0" "1 1 0 0 0" "0 1 1 0 0" "1 0 1 0 0" "0 0 0 0 0" "1 1 1 1 0" "1 1 1 0 1" "1 1 1 0 0"; do
  set -- $cfg
  CL4_PACK_BATCH=$1 CL4_GRAD_PLANES=$2 CL4_ABN_ONE_RANK=$3 CL4_P1_OVERLAP=$4 CL4_WGRAD_SIDE=$5 timeout -k 10 300 python bench.py --config p1 --steps 20 > $O/p1_$1$2$3$4$5.json 2> $O/p1_$1$2$3$4$5.err || { tail -5 $O/p1_$1$2$3$4$5.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/p1_$1$2$3$4$5.json').read().strip().splitlines()[-1]); print('batched packs $1, gradient planes from the ABN backward $2, one-rank ABN $3, old model on the side stream $4, weight gradients on the side stream $5: ms_per_step', round(d['ms_per_step'],3), 'images/s', round(d['value'],1))" | tee -a $O/summary.txt
done
for w in 1 0; do
  CL4_PACK_BATCH=$w CL4_ABN_ONE_RANK=$w CL4_GRAD_PLANES=$w timeout -k 10 300 python tools/bench_s0.py 16 5 > $O/s0_r05_$w.txt 2> $O/s0_r05_$w.err; echo "s0 with the three $w rc $?"; head -c 300 $O/s0_r05_$w.txt; echo
done
