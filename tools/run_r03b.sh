set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
python -m pytest tests/test_label_gpu.py -x -q -m gpu > gpurun_out/r03b/label_tests.log 2>&1 || { tail -20 gpurun_out/r03b/label_tests.log; exit 1; }
tail -2 gpurun_out/r03b/label_tests.log
bash tools/soak_single_stream.sh r03b 10 30
