"""The command the driver runs at N > 1, executed end to end once: `python bench.py --gpus 2 --steps 2 --warmup 1
--no-cpu-baseline` with no launcher. Two ranks share the one GPU of the box and talk over gloo (RCCL refuses two ranks on one device;
CL4_DIST_BACKEND selects the transport, everything else is the N > 1 code path): self-launch -> 8 images per rank -> deep-pipelined graph
replays + the eager ASPP tail with the bucketed gradient all-reduce -> Adam -> roofline leg -> structured-labels leg -> `allreduce`
record -> ONE JSON line from rank 0, exit status 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_over_gloo_on_one_gpu(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CL4_DIST_BACKEND="gloo", CL4_BENCH_DEADLINE_S="500", OMP_NUM_THREADS="8")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["per_gpu_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "strong"
    assert d["value"] > 0 and abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    ar = d["allreduce"]                                    # the gradient exchange really ran: buckets, bytes, exposed time
    assert ar["buckets"] >= 3 and sum(ar["bucket_mb"]) > 60 and ar["steps"] >= 2 and ar["path"] == "torch.distributed.all_reduce", ar
    assert "phase1" not in d and "cpu_baseline" not in d and "stock_rocm" not in d
    assert d["roofline"]["frac"] > 0 and d["structured_labels"]["value"] > 0
    assert d["label_counters"]["overflow"] == 0
