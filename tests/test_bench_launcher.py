"""CPU: `python bench.py --gpus N` needs no external launcher — the parent spawns N ranks (gloo rehearsal here: no GPU),
rank 0 prints the single JSON line, every rank is seen by the all-reduce, and a failing rank makes the command fail."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_self_launch_two_ranks_prints_one_json_line():
    r = _run(["--gpus", "2", "--rehearse-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]     # gloo itself prints a connection notice
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["rehearsal"] is True


def test_self_launch_propagates_a_rank_failure():
    # without --rehearse-launcher the workers need a GPU: here (no GPU) every rank exits non-zero, and so must the command
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("build-container check (no GPU)")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"CL4_DIST_BACKEND": "gloo"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_external_launcher_environment_is_respected():
    r = _run(["--gpus", "1", "--rehearse-launcher"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_ranks_seen"] == 1


def test_self_launch_deadline_terminates_stuck_ranks_and_names_them():
    """a rank that never reaches a collective leaves the others blocked in it; the parent's deadline
    (CL4_BENCH_DEADLINE_S) terminates the children, prints which ranks were still alive and exits non-zero — no JSON line."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--rehearse-launcher"], env={"CL4_BENCH_REHEARSE_HANG_RANK": "1", "CL4_BENCH_DEADLINE_S": "20"})
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 120
    assert "deadline" in r.stderr and "[0, 1]" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_multi_rank_processes_get_the_dmabuf_ipc_setting(monkeypatch):
    """run.py and bench.py start RCCL identically: init_distributed() itself puts HSA_ENABLE_IPC_MODE_LEGACY=0 into the environment of
    a multi-rank process before anything touches the GPU (an explicit setting of the caller wins)."""
    import torch.distributed as dist
    from cl4wsis_amd import parallel
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(dist, "is_initialized", lambda: True)       # stop before the rendezvous: only the environment matters here
    parallel.init_distributed("gloo")
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    parallel.init_distributed("gloo")
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
