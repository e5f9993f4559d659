cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
CL4_BENCH_LAYER_TABLE=$O/layers.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structured > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
bash tools/run_small_batches.sh r03z || exit 1
CL4_SMALL_TILE_BELOW=0 CL4_CONV_B=0 CL4_BENCH_PER_GPU_BATCH=2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-structured > $O/b2_old.json 2> $O/b2_old.err
python -c "import json; d=json.load(open('$O/b2_old.json')); print('per-GPU batch 2 without conv_b / 64-pixel tiles: ms_per_step', round(d['ms_per_step'],3))"
timeout -k 10 300 python -m pytest tests/test_cli_gpu.py -q -m gpu --durations=3 2>&1 | tail -6
