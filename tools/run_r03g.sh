cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_acceptance_gpu.py tests/test_ddp_trainer_gpu.py -q -m gpu -rA --durations=10 > $O/acceptance.log 2>&1
echo "rc $?"
grep "MAPPROXY\|WELLCOND\|DDP-EMU\|passed\|failed\|^FAILED\|^E  " $O/acceptance.log | cut -c1-400
