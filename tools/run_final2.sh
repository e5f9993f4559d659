TAG=${1:-r03z}
cd "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}"
bash tools/run_final.sh $TAG || exit 1
cd "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}"
bash tools/run_small_batches.sh $TAG || exit 1
bash tools/run_solo_collectives.sh $TAG || exit 1
cd "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}"
timeout -k 10 400 python -m pytest tests/test_ddp_trainer_gpu.py -q -m gpu --durations=2 2>&1 | tail -5
