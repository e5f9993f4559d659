#!/usr/bin/env python
"""Benchmark of the CL4WSIS phase-2 training step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 needs no launcher: when WORLD_SIZE is not set the command starts N worker processes itself (one per GPU, RCCL over
xGMI; the parent never touches a GPU), rank 0 prints the JSON line and the exit status is non-zero if any rank failed.
Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` the launcher's environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) is used as is.

One "step" = one `Trainer.step` (cl4wsis_amd.train) on a synthetic VOC-shaped batch: global batch 16 split over the N
ranks (run.py:66 semantics, strong scaling), 513x513, VOC 15-5 step 1 phase 2, fp32 tensors, inputs resident in HBM.
Every timed step announces the next batch to `Trainer.step` (one batch of loader lookahead, as `Trainer.train` does), so that the
frozen old model's forward of batch t+1 overlaps step t; every warm-up and timed step does (steady state), i.e. exactly K old-model
forwards are enqueued and finish inside the timed region of K steps.
Rank 0 prints ONE JSON line with
  * the throughput of the timed region (e2e-random mode: the label stage consumes the random-init networks' own outputs),
  * `structured_labels`: the same step with the label stage fed by the structured synthetic maps of SURVEY.md §8(d),
  * `roofline` of the dominant kernel (implicit-GEMM conv forward, timed live with HIP events on its launch stream),
  * at N = 1: `cpu_baseline` (the oracle restatement of the same step on the host cores) and `stock_rocm` (the same restatement
    with stock PyTorch-ROCm ops on the GPU, the comparator of BASELINE.md §4) — both outside the timed region.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GLOBAL_BATCH = 16
CROP = 513
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense MFMA peak (no sparsity)
ALGO_GFLOP_PER_IMAGE = 545.7    # SURVEY.md §8(d) / BASELINE.md §3
METRIC = "images/sec at batch 16, VOC 15-5 step-1 phase-2, 1/2/4/8 MI355X"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host-side legs (cpu_baseline, stock_rocm)")
    ap.add_argument("--cpu-baseline-iters", type=int, default=3,
                    help="timed CPU iterations at batch 16 after the batch-2 warm-up (default 3 = BASELINE.md §4's protocol, ~80 s each on a "
                         "16-thread box share; the stock-ops GPU comparator runs 3 timed iterations after its MIOpen warm-up)")
    ap.add_argument("--no-structured", action="store_true", help="skip the structured-labels sub-record")
    ap.add_argument("--no-phase1", action="store_true", help="skip the `phase1` sub-record (BASELINE config 2, 10 timed steps; N = 1 only)")
    ap.add_argument("--config", choices=("p2", "p1"), default="p2",
                    help="p2 (default): BASELINE config 3, the metric; p1: BASELINE config 2 (VOC 15-5 step 1 phase 1, batch 16) through the same "
                         "Trainer.step, with its own roofline sub-record (no CPU / stock legs)")
    ap.add_argument("--rehearse-launcher", action="store_true",
                    help="CPU-only check of the self-launch path: rendezvous over gloo, all-reduce of ones, no GPU work")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# self-launch (N > 1 without an external launcher)
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_workers(a, argv):
    """Parent of an N-rank run. It never imports torch.cuda state or touches a GPU: it only starts N fresh interpreters
    (never re-execs itself) with the torch.distributed environment, forwards rank 0's stdout, and reaps the others.
    A deadline (CL4_BENCH_DEADLINE_S, default 540 s) bounds the whole run: ranks stuck in a mismatched collective are terminated, the
    ranks that were still alive are named on stderr and the exit status is non-zero (124) — the driver sees a diagnosis, not its own
    timeout. HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, what RCCL needs on this host driver) is what
    cl4wsis_amd.parallel.init_distributed sets for every multi-rank process, run.py's included; it is passed here as well so that a
    rank has it before its interpreter imports anything."""
    port = _free_port()
    deadline = time.time() + float(os.environ.get("CL4_BENCH_DEADLINE_S", "540"))
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if r == 0 else subprocess.DEVNULL           # rank 0 inherits stdout: it prints the one JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    try:
        pending = set(range(a.gpus))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr)
                    for q in pending:                       # a dead rank leaves the others stuck in a collective
                        procs[q].terminate()
            if pending and time.time() > deadline:
                print(f"bench.py: deadline of {os.environ.get('CL4_BENCH_DEADLINE_S', '540')} s (CL4_BENCH_DEADLINE_S) passed with ranks "
                      f"{sorted(pending)} of {a.gpus} still running (finished: {sorted(set(range(a.gpus)) - pending)}): terminating them",
                      file=sys.stderr, flush=True)
                for q in pending:
                    procs[q].terminate()
                t_kill = time.time() + 10
                while time.time() < t_kill and any(procs[q].poll() is None for q in pending):
                    time.sleep(0.2)
                rc = rc or 124
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# workload
# ----------------------------------------------------------------------------------------------------------------------
def p2_opts(batch):
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "2", "--alpha", "0.5",
            "--loss_de", "1", "--lr_policy", "warmup", "--affinity", "--optim", "adam", "--weight_decay", "0",
            "--lr", "5e-5", "--crop_size", str(CROP), "--batch_size", str(batch), "--no_pretrained", "--epochs", "30"]
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))


def _progress(msg):
    """Progress lines on stderr (stdout carries exactly one JSON line): the host-side legs take minutes."""
    print(f"[bench.py +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.time()


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources, in name order: identifies the code a counter profile was taken on. The PMC
    summaries under profiles/ carry the value they were measured at (tools/pmc_traffic.py, tools/pmc_kernels.py); bench.py reports their
    figures only when it matches the sources this run was built from (there is no .git on the GPU box to ask)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cl4wsis_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def padded_tap_fraction(hin, k, stride, dil):
    """Share of a convolution's (tap, output pixel) pairs whose input pixel lies in the zero padding ('same'-style padding dil * (k - 1) / 2,
    square maps): those products are algorithmic FLOPs of 2 * cout * cin * k^2 * pixels that a kernel need not execute."""
    if k == 1:
        return 0.0
    if k == 0:       # a fused bottleneck span (ops.bottleneck_fused): only its 3x3 layer has padding taps, 9 / 17 of the block's FLOPs
        return padded_tap_fraction(hin, 3, stride, dil) * 9.0 / 17.0
    pad = dil * (k - 1) // 2
    hout = (hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    valid_1d = sum(1 for o in range(hout) for t in range(k) if 0 <= o * stride - pad + t * dil < hin)
    return 1.0 - (valid_1d / float(hout * k)) ** 2


def _host_cores():
    """Host threads for the CPU legs. A 1-GPU box is a 16-thread share of a 256-thread host whose affinity mask still lists all
    256: oversubscribing them made a batch-2 warm-up take 270 s (r02b). 16 threads per visible GPU, capped by the mask."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    import torch
    share = 16 * max(1, torch.cuda.device_count())
    return max(1, min(avail, int(os.environ.get("CL4_BENCH_CPU_THREADS", share))))


def _oracle_state(opts, device="cpu"):
    import torch
    from cl4wsis_amd import segmentation_module as sm
    from cl4wsis_amd.wss import PeakGenerator, PseudoLabeler
    torch.manual_seed(42)
    sd = lambda m: {k: v.detach().clone().to(device) for k, v in m.state_dict().items()}
    return (sd(sm.make_model(opts, classes=[16, 5])), sd(sm.make_model(opts, classes=[16])),
            sd(PseudoLabeler(2048, 21, sm.get_norm(opts))), sd(PeakGenerator(21, 20, old_classes=15)))


def cpu_baseline(opts, iters):
    """the oracle (CPU restatement of the reference step, stock torch.nn ops, fp32) at batch 16, one
    process, every host core this process may use, 1 warm-up + `iters` timed iterations (Adam state carried)."""
    import torch
    from cl4wsis_amd.synthetic import synthetic_batch
    from oracle import step as ostep
    cores = _host_cores()
    torch.set_num_threads(cores)
    new, old, pl, pg = _oracle_state(opts)
    wi, wl = synthetic_batch(2, 41, torch.device("cpu"), CROP)     # untimed warm-up (thread pool, oneDNN primitives)
    out = ostep.p2_step(new, old, pl, pg, wi, wl, 16, pooling=opts.pooling, lr=opts.lr * opts.lr_head)
    b = GLOBAL_BATCH
    images, l1h = synthetic_batch(b, 42, torch.device("cpu"), CROP)
    times, state = [], out["adam_state"]
    _progress(f"cpu_baseline: warm-up done, {iters} x batch {b} on {cores} threads")
    for i in range(iters):
        t0 = time.time()
        out = ostep.p2_step(new, old, pl, pg, images, l1h, 16, pooling=opts.pooling, lr=opts.lr * opts.lr_head,
                            adam_state=state, step=i + 2)
        state = out["adam_state"]
        times.append(time.time() - t0)
        _progress(f"cpu_baseline: iteration {i + 1}/{iters} {times[-1]:.1f} s")
    dt = sum(times) / len(times)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": b / dt, "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": model,
            "os_cpu_count": os.cpu_count(), "iterations": iters, "seconds_per_iteration": [round(t, 2) for t in times],
            "sample": f"oracle p2_step (PyTorch-CPU fp32 restatement of train.py:241-541), batch {b} at {CROP}x{CROP}, "
                      f"{iters} timed iterations after a batch-2 warm-up, torch threads = {cores}"}


def stock_rocm(opts, device, iters=3, rank=0, world=1):
    """BASELINE.md §4 second comparator: the same restatement with stock PyTorch-ROCm ops (MIOpen convolutions, ATen
    elementwise kernels) on this GPU, batch 16 — like the reference on a GPU it keeps the label stage on the host
    (OpenCV/numpy loops behind .cpu() copies, modules/utils.py:179-385) — under the reference's own autocast (train.py:252).
    At N > 1 it is the stock DATA-PARALLEL comparator: every rank runs the step on its 16 / N images and the gradients are averaged
    with torch.distributed all-reduces between backward and Adam (what the DDP wrappers of train.py:187-196 do); the value is the
    global batch over the slowest rank's iteration time. A collective leg: every rank must call it."""
    import torch
    import torch.distributed as dist
    from cl4wsis_amd.synthetic import synthetic_batch
    from oracle import step as ostep
    torch.set_num_threads(max(1, _host_cores() // max(1, world)))
    b = GLOBAL_BATCH // world
    images, l1h = synthetic_batch(b, 42 + rank, device, CROP)
    sync = None
    if world > 1:
        def sync(grads):
            for t in grads.values():
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                t.div_(world)
    res = {"unit": "images/s", "batch": GLOBAL_BATCH, "per_gpu_batch": b, "iterations": iters}
    modes = (("fp32", False), ("autocast_fp16", True)) if os.environ.get("CL4_BENCH_STOCK_FP32") == "1" else (("autocast_fp16", True),)
    for name, amp in modes:
        new, old, pl, pg = _oracle_state(opts, device)
        state, times = None, []
        for i in range(iters + 1):                              # first iteration = warm-up (MIOpen find, allocator)
            torch.cuda.synchronize()
            t0 = time.time()
            out = ostep.p2_step(new, old, pl, pg, images, l1h, 16, pooling=opts.pooling, lr=opts.lr * opts.lr_head,
                                adam_state=state, step=i + 1, autocast=amp, grad_sync=sync)
            state = out["adam_state"]
            torch.cuda.synchronize()
            dt = time.time() - t0
            if world > 1:                                       # the step ends when the slowest rank has finished
                tt = torch.tensor([dt], device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            if rank == 0:
                _progress(f"stock_rocm[{name}]: iteration {i} {dt:.2f} s")
            if i:
                times.append(dt)
        del out, state
        torch.cuda.empty_cache()
        med = sorted(times)[len(times) // 2]
        res[name] = {"value": GLOBAL_BATCH / med, "ms_per_step": 1e3 * med, "best_ms_per_step": 1e3 * min(times),
                     "seconds_per_iteration": [round(t, 2) for t in times]}
    if world > 1:
        res["parallelism"] = f"dp{world}: per-rank step + torch.distributed all-reduce (SUM / N) of the 61 trainable gradient tensors before Adam"
    res["note"] = ("oracle p2_step on cuda: stock torch ops (MIOpen convolutions), the label stage as the reference runs it on a GPU "
                   "(per-contour torch ops on the device, masks / centroids via .cpu(), modules/utils.py:257-385), under the reference's "
                   f"own amp.autocast(float16) (train.py:252); 1 warm-up (MIOpen's kernel search) + {iters} timed iterations, value = batch / their "
                   "median; r02b measured fp32 at 9.3 s per iteration (CL4_BENCH_STOCK_FP32=1 adds that mode)")
    res["value"] = max(v["value"] for k, v in res.items() if isinstance(v, dict))
    return res


def rehearse(a):
    """--rehearse-launcher: rendezvous + one all-reduce over gloo on the CPU (what the build container can run)."""
    import torch
    import torch.distributed as dist
    from cl4wsis_amd.parallel import init_distributed
    rank, world, _ = init_distributed("gloo")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    seen = torch.ones(1)
    if os.environ.get("CL4_BENCH_REHEARSE_HANG_RANK") == str(rank):       # test hook: a rank that never reaches the collective
        time.sleep(3600)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "rehearsal": True, "n_gpus": world, "n_ranks_seen": int(seen.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def p1_opts(batch):
    """BASELINE config 2: scripts/run.sh:31-33 + --crop_size."""
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "1", "--alpha", "0.5", "--loss_de", "1",
            "--lr_policy", "warmup", "--affinity", "--optim", "sgd", "--lr", "0.001", "--epochs", "40", "--crop_size", str(CROP),
            "--batch_size", str(batch), "--no_pretrained"]
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))


ALGO_GFLOP_PER_IMAGE_P1 = 684.0   # old fwd + new fwd / bwd + PseudoLabeler on 3b (FLAC) fwd / bwd + eval pass


def run_p1(a, device, rank, world, b, steps, warmup):
    """The phase-1 step (BASELINE config 2) at global batch 16, epoch >= pseudo_ep (PAMR, l_seg, l_cls, RandDrop active): `steps` timed
    `Trainer.step` calls under the metric run's timing contract; HIP-graph replayed on one GPU (three graphs, one per FLAC angle).
    Returns the record (rank 0) or None."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cl4wsis_amd import body_train, ops
    from cl4wsis_amd.synthetic import synthetic_batch
    from cl4wsis_amd.train import Trainer
    opts = p1_opts(GLOBAL_BATCH)
    opts.batch_size = b
    opts.max_iters, opts.start_decay = 40 * 661, 5 * 661
    torch.manual_seed(42)

    class _Log:
        def info(self, *a, **k): pass
        debug = info

    trainer = Trainer(_Log(), device, opts)
    trainer._cur_epoch = opts.pseudo_ep                     # the full loss stack (train.py:370-416)
    images, l1h = synthetic_batch(b, 42 + rank, device, CROP)
    batch = (images, l1h)
    np.random.seed(7)
    for _ in range(max(warmup, 3) + 6):                     # eager + tuner, then every FLAC angle's graph gets captured before the timed region
        trainer.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = trainer.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    graphs = len(trainer._p1_static["graphs"]) if trainer._p1_static else 0
    # roofline leg: one more step, eager, HIP events around every convolution launch
    ops.TIMER = ops.KernelTimer()
    trainer._use_graph = False
    trainer.step(batch)
    torch.cuda.synchronize()
    timer_p1 = ops.TIMER
    summ, ops.TIMER = ops.TIMER.summary(), None
    if rank == 0 and os.environ.get("CL4_BENCH_LAYER_TABLE_P1"):
        rows = sorted(timer_p1.by_shape().items(), key=lambda kv: -kv[1][2])
        tot = sum(v[2] for _, v in rows)
        with open(os.environ["CL4_BENCH_LAYER_TABLE_P1"], "w") as f:
            f.write("kind (n,cin,hin,cout,k,stride,dil) launches/step ms/step TFLOP/s share\n")
            for (kind, shp), (n_, fl_, ms_) in rows:
                f.write(f"{kind:16s} {str(shp):36s} {n_:5d} {ms_:8.3f} {fl_ / ms_ / 1e9:8.1f} {100 * ms_ / tot:5.1f}%\n")
    res = None
    if rank == 0:
        peaks = {"conv_fwd_bf16x6": PEAK_BF16_MFMA_TFLOPS / 6.0, "conv_fwd_bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0, "conv_fwd_f32": PEAK_FP32_MFMA_TFLOPS,
                 "conv_fwd_f16x3": PEAK_BF16_MFMA_TFLOPS / 3.0,      # fp16 MFMA = the bf16 rate (tools/probes/f16x3_probe.hip), three products per MAC
                 "conv_wgrad": PEAK_BF16_MFMA_TFLOPS / 3.0 if body_train.WGRAD_PRECISION == 1 else PEAK_FP32_MFMA_TFLOPS}
        kinds = {k: {"achieved": fl / (ms * 1e-3) / 1e12, "peak": peaks.get(k), "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / peaks[k] if k in peaks else None,
                     "ms_per_step": ms, "launches_per_step": n} for k, (n, fl, ms) in summ.items()}
        dom = max(kinds, key=lambda k: kinds[k]["ms_per_step"])
        arith = {0: "exact fp32 MFMA", 1: "bf16x3 (three bf16 products per MAC, fp32 accumulation)", 2: "bf16x6 (fp32-level)",
                 3: "fp16x3 (three fp16 products per MAC on split-fp16 operands with power-of-two scales: fp32-level, like bf16x6; bf16x6 where the shape is not routed)"}[body_train.PRECISION]
        res = {"metric": "images/sec at batch 16, VOC 15-5 step-1 phase-1 (BASELINE config 2), 1/2/4/8 MI355X", "value": GLOBAL_BATCH * steps / dt,
               "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "VOC 15-5 step 1 phase 1 (BASELINE config 2), Trainer.step, global batch 16, 513x513, ResNet101 + DeepLabV3 + "
                                      "PseudoLabeler + PeakGenerator trained with SGD-Nesterov, epoch >= pseudo_ep (PAMR, l_seg, l_cls, RandDrop on), "
                                      "random-init weights", "global_batch": GLOBAL_BATCH, "per_gpu_batch": b, "crop": CROP, "parallelism": f"dp{world}",
                          "arithmetic": f"fp32 tensors; differentiated convolutions {arith}, weight gradients "
                                        f"{'bf16x3' if body_train.WGRAD_PRECISION == 1 else 'fp32 MFMA'}, frozen old model bf16x3",
                          "hip_graphs": graphs},
               "roofline": dict(kinds[dom], bound="mfma", kernel=dom + ": every launch of one eager step after the timed region, HIP events on the launch stream",
                                traffic=None),
               "kinds": kinds,
               "algorithmic_tflops_whole_step": ALGO_GFLOP_PER_IMAGE_P1 * GLOBAL_BATCH * steps / dt / 1e3,
               "final_losses": {k: float(v) for k, v in out["losses"].items()}}
    del trainer, out
    from cl4wsis_amd.nn import CACHE
    CACHE.clear()
    torch.cuda.empty_cache()
    return res


def worker_p1(a):
    """--config p1: BASELINE config 2 through the same Trainer.step (run_p1), as its own JSON line."""
    import torch
    import torch.distributed as dist
    from cl4wsis_amd.parallel import init_distributed
    rank, world, local = init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    b = int(os.environ.get("CL4_BENCH_PER_GPU_BATCH", GLOBAL_BATCH // world))
    res = run_p1(a, device, rank, world, b, a.steps, a.warmup)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def worker(a):
    import torch
    import torch.distributed as dist
    from cl4wsis_amd import ops
    from cl4wsis_amd.parallel import init_distributed
    from cl4wsis_amd.synthetic import structured_label_inputs, synthetic_batch
    from cl4wsis_amd.train import Trainer

    rank, world, local = init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    local = local % torch.cuda.device_count()     # identity on an N-GPU node; lets N ranks rehearse on one GPU (gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    b = int(os.environ.get("CL4_BENCH_PER_GPU_BATCH", GLOBAL_BATCH // world))   # override = experiments only
    opts = p2_opts(GLOBAL_BATCH)
    opts.batch_size = b
    opts.max_iters, opts.start_decay = 30 * 661, 5 * 661   # 10582 train images / 16
    torch.manual_seed(42)                      # identical replicas on every rank

    class _Log:
        def info(self, *a, **k): pass
        debug = info

    trainer = Trainer(_Log(), device, opts)
    trainer.buckets.timing = trainer.buckets.distributed         # HIP events around the join with the communication stream (exposed all-reduce)
    images, l1h = synthetic_batch(b, 42 + rank, device, CROP)
    seen = torch.ones(1, device=device)
    if world > 1:
        dist.all_reduce(seen)                  # every rank really took part (RCCL)

    def timed(steps, warmup):
        # `next_batch` announces the following iteration's batch (as Trainer.train does with one batch of loader lookahead), so
        # that its frozen old-model forward overlaps the current step. Steady state: every replayed step contains exactly one
        # old-model forward (the next batch's), so the K timed steps enqueue — and finish — K old-model forwards inside the timed
        # region: the one the first timed step consumes was made by the last warm-up step, the one the last timed step makes is for
        # the step after it. (An unannounced step would add an eager old-model forward to the following call: Trainer.step does that
        # at the start of an epoch.)
        batch = (images, l1h)
        for i in range(warmup):
            trainer.step(batch, next_batch=batch)
            if os.environ.get("CL4_BENCH_VERBOSE"):       # debugging aid: where a run stops
                torch.cuda.synchronize()
                _progress(f"rank {rank}: warm-up step {i} done")
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = trainer.step(batch, next_batch=batch)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, out

    elapsed, out = timed(a.steps, a.warmup)
    if rank == 0:
        _progress(f"timed region done: {1e3 * elapsed / a.steps:.2f} ms/step")
    losses = out["losses"].tolist()
    counters_random = trainer.check_label_overflow()          # raises if a device-side label table overflowed

    # Roofline leg: the timed steps are HIP-graph replays that overlap the old-model and new-model forwards on two
    # streams, so a launch cannot be bracketed by events there (and would share the device anyway). Two more steps are run
    # right here eagerly with the overlap switched off (same process, same tuned plans, same inputs) and every
    # cl4_conv2d_fwd launch is timed with HIP events on its launch stream. A rocprofv3 --kernel-trace of this command
    # contains exactly these two steps after the last graph replay; tools/roofline_leg_stats.py extracts them
    # (profiles/r02*_roofline_leg.txt) and must agree with `avg_launch_ms`.
    side, trainer._side_stream = trainer._side_stream, None
    ops.TIMER = ops.KernelTimer()
    for _ in range(2):
        trainer.step((images, l1h))
    torch.cuda.synchronize()
    timer_excl, ops.TIMER = ops.TIMER, None
    trainer._side_stream = side

    structured = None
    if not a.no_structured:
        # SURVEY.md §8(d) second mode: identical step, but the label stage reads structured maps (blobs, CAM bumps, consistent
        # centre / offset fields) instead of the random-init networks' noise, so CC / peak / refine kernels do representative work.
        inp = structured_label_inputs(1042 + rank, b=b, H=CROP, W=CROP)
        trainer.peakgenerator.load_state_dict({"extra_conv4.weight": inp["pg_w"], "extra_conv4.bias": inp["pg_b"]})
        trainer.set_label_override(inp)
        dt_s, out_s = timed(a.steps, 2)
        structured = {"value": GLOBAL_BATCH * a.steps / dt_s, "unit": "images/s", "ms_per_step": 1e3 * dt_s / a.steps,
                      "label_counters": trainer.check_label_overflow(),
                      "final_losses": dict(zip(("center_old", "offset_old", "center_new", "offset_new"), out_s["losses"].tolist())),
                      "note": "label stage fed by cl4wsis_amd.synthetic.structured_label_inputs (1-3 blobs per labelled class, "
                              "CAM bumps, consistent centre/offset maps); networks, losses, backward and Adam unchanged"}
        trainer.set_label_override(None)

    if rank == 0 and os.environ.get("CL4_BENCH_LAYER_TABLE"):
        # per-shape table of the conv launches (experiments / profiles): ms per step, TFLOP/s, share of conv time
        rows = sorted(timer_excl.by_shape().items(), key=lambda kv: -kv[1][2])
        tot = sum(v[2] for _, v in rows)
        # the launch plan the tuner pinned for the shape: impl (1 register-staged / role-split, 2 conv_u or conv_s, 3 conv_p, 4 / 5 conv_b
        # with 256 / 128 rows), pixel tile, K splits, @ precision code (1 bf16x3 fp32 input, 2 bf16x6, 3 planes input, 4 three planes)
        from cl4wsis_amd import ops as _ops
        plans = {}
        for (n_, cin_, hin_, _win, cout_, k_, st_, _pad, dil_, prec_, _dev), (best, _ms) in _ops._tuned.items():
            plans.setdefault((n_, cin_, hin_, cout_, k_, st_, dil_), []).append("%d/%d/%d@%d" % (best + (prec_,)))
        variants = {}
        with open(os.environ["CL4_BENCH_LAYER_TABLE"], "w") as f:
            f.write("kind (n,cin,hin,cout,k,stride,dil) launches/step ms/step TFLOP/s share plan(impl/tile/splits@precision)\n")
            for (kind, shp), (n_, fl_, ms_) in rows:
                codes = {"conv_fwd_bf16x3": ("@1", "@3"), "conv_fwd_bf16x6": ("@2", "@4"), "conv_fwd_f32": ("@0",)}.get(kind, ())
                pl = "|".join(sorted(q for q in plans.get(shp, []) if q.endswith(codes))) or "-"
                f.write(f"{kind:16s} {str(shp):34s} {n_ / 2:6.1f} {ms_ / 2:8.3f} {fl_ / ms_ / 1e9:8.1f} {100 * ms_ / tot:5.1f}%  {pl}\n")
                v = variants.setdefault((kind, pl.split("/")[0] + "/" + pl.split("/")[1] if "/" in pl and "|" not in pl else pl), [0.0, 0.0, 0.0])
                v[0] += n_ / 2; v[1] += fl_ / 2; v[2] += ms_ / 2
            f.write("\n# per kernel variant (kind, impl/tile): launches/step GFLOP/step ms/step TFLOP/s\n")
            for (kind, var), (n_, fl_, ms_) in sorted(variants.items(), key=lambda kv: -kv[1][2]):
                f.write(f"{kind:16s} {var:22s} {n_:7.1f} {fl_ / 1e9:10.1f} {ms_:8.3f} {fl_ / ms_ / 1e9:8.1f}\n")
    if rank == 0:
        summ = timer_excl.summary()
        n, fl, ms = summ["conv_fwd_bf16x3"]
        ach = fl / (ms * 1e-3) / 1e12
        # The dominant kernel multiplies fp32 operands as three bf16 MFMA products (hi*hi + hi*lo + lo*hi), so one
        # algorithmic (fp32-equivalent) FLOP costs three bf16 MFMA FLOPs: its ceiling is the dense bf16 peak / 3.
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0
        # HBM-side bytes per launch of the same kernel: PMC counters cannot be read from inside the process, so this is the
        # committed result of the two separate rocprofv3 --pmc passes over the last eager step of this same command
        # (tools/prof_pmc.sh + tools/pmc_traffic.py; newest profiles/r*_pmc_traffic.json); null when absent.
        # Counter-derived figures are NOT measured by this run: each comes with its source file and the hash of the kernel sources it was
        # measured on, and is reported as null (the stale value under `pmc.stale_*`) when that hash is not this build's.
        sha = csrc_sha16()
        pmc = {"csrc_sha16_of_this_build": sha}
        traffic, traffic_note = None, "no PMC file"
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        if pmcs and b == GLOBAL_BATCH:
            with open(pmcs[-1]) as f:
                tj = json.load(f)
            t = tj["conv_igemm_bf16x3"]
            pmc["traffic_source"] = os.path.relpath(pmcs[-1], ROOT)
            pmc["traffic_source_csrc_sha16"] = tj.get("csrc_sha16")
            if tj.get("csrc_sha16") == sha:
                traffic = t["hbm_bytes_per_launch"]
                traffic_note = ("bytes per launch, average over the %d bf16x3 forward launches of one step; rocprofv3 --pmc "
                                "FETCH_SIZE + WRITE_SIZE in separate passes, corrected as calibrated in %s (same kernel sources as this run)"
                                % (t["launches_per_step"], pmc["traffic_source"]))
            else:
                pmc["stale_traffic"] = t["hbm_bytes_per_launch"]
                traffic_note = ("null: %s was measured on other kernel sources (csrc_sha16 %s, this build %s); its figure is under pmc.stale_traffic"
                                % (pmc["traffic_source"], tj.get("csrc_sha16"), sha))
                _progress("WARNING: " + traffic_note)
        # matrix-pipe busy share of the same kernel family, from the committed counter pass of the same command (tools/prof_pmc.sh third
        # pass + tools/pmc_kernels.py: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)); null when absent or stale
        mfma_busy, hbm_gbs = None, None
        pk = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_kernels.json")))
        if pk and b == GLOBAL_BATCH:
            with open(pk[-1]) as f:
                kj = json.load(f)
            fam = kj["bf16x3_forward_family"]
            pmc["kernels_source"] = os.path.relpath(pk[-1], ROOT)
            pmc["kernels_source_csrc_sha16"] = kj.get("csrc_sha16")
            if kj.get("csrc_sha16") == sha:
                mfma_busy, hbm_gbs = fam["mfma_busy"], fam["hbm_gbs"]
            else:
                pmc["stale_mfma_busy"], pmc["stale_hbm_gbs"] = fam["mfma_busy"], fam["hbm_gbs"]
                _progress("WARNING: %s was measured on other kernel sources; roofline.mfma_busy / hbm_gbs are null" % pmc["kernels_source"])
        # algorithmic operand bytes of the same launches: input + output activations (fp32) + split weights (4 B / weight);
        # epilogue operands (residual, scale / shift) not counted
        alg_bytes, alg_n, exec_fl = 0.0, 0, 0.0
        for (kind, shp), (cnt, fl_, _) in timer_excl.by_shape().items():
            if kind == "conv_fwd_bf16x3" and shp is not None:
                n_, cin_, hin_, cout_, k_, st_, dil_ = shp
                ho = -(-hin_ // st_)
                alg_bytes += cnt * 4.0 * (n_ * cin_ * hin_ * hin_ + n_ * cout_ * ho * ho + cin_ * cout_ * k_ * k_)
                alg_n += cnt
                exec_fl += fl_ * (1.0 - padded_tap_fraction(hin_, k_, st_, dil_))
            elif kind == "conv_fwd_bf16x3":
                exec_fl += fl_
        res = {
            "metric": METRIC,
            "value": GLOBAL_BATCH * a.steps / elapsed, "unit": "images/s", "n_gpus": world, "n_ranks_seen": int(seen.item()),
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VOC 15-5 step 1 phase 2 (BASELINE config 3), Trainer.step, global batch 16, "
                                   "513x513, ResNet101+DeepLabV3+Panoptic-DeepLab, run_refine on, random-init weights, "
                                   "label stage fed by the networks' own outputs (e2e-random mode)",
                       "global_batch": GLOBAL_BATCH, "per_gpu_batch": b, "crop": CROP, "parallelism": f"dp{world}",
                       "arithmetic": "fp32 tensors; convolutions as split 16-bit MFMA products with fp32 accumulation: frozen networks x3 (three products per MAC: "
                                     + ("fp16 hi / lo planes, 22-bit operands" if __import__("cl4wsis_amd.ops", fromlist=["x"]).planes_f16() else "bf16 hi / lo planes, 16-17-bit operands")
                                     + "; bf16x3 on the fp32-input kernels of the thin layers); differentiated "
                                     "convs " + {0: "exact fp32 MFMA", 1: "bf16x3", 2: "bf16x6 (six bf16 products, fp32-level)",
                                                 3: "fp16x3 (three fp16 products on split-fp16 operands with power-of-two scales, fp32-level; bf16x6 where "
                                                    "the shape is not routed)"}[__import__("cl4wsis_amd.heads", fromlist=["x"]).TRAIN_PRECISION]
                                     + "; bf16x3 weight gradients"},
            "label_counters": counters_random,
            "roofline": {"bound": "mfma", "kernel": "bf16x3 forward convolution (cl4_conv2d_fwd precision 1 + cl4_conv2d_fwd_planes + cl4_stem_fused: kernels "
                                                    "conv_igemm_bf16x3 / conv_u_bf16x3 / conv_s_bf16x3 / conv_b_bf16x3 / stem_fused): all its launches of 2 "
                                                    "eager steps run right after the timed region (graph replay and two-stream "
                                                    "overlap off), HIP events on the launch stream around every launch",
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         # the same with the products of zero-padding taps removed from the numerator (a dilated 3x3 layer on a 33 x 33 map
                         # multiplies padding for 12-60 % of its tap-pixel pairs; the kernels skip most of them, FINDINGS.md finding 27)
                         "executed_frac": exec_fl / (ms * 1e-3) / 1e12 / peak,
                         "traffic": traffic, "pmc": pmc,
                         "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": alg_bytes / max(alg_n, 1),
                         # ... and with the operands the fused epilogues add (residual of every bottleneck's last conv: as large as its
                         # output) — the bytes a launch cannot avoid moving; traffic / this = re-read factor
                         "algorithmic_bytes_per_launch_incl_epilogue_operands":
                             (alg_bytes + timer_excl.epilogue_bytes.get("conv_fwd_bf16x3", 0)) / max(alg_n, 1),
                         "traffic_over_algorithmic": (traffic / ((alg_bytes + timer_excl.epilogue_bytes.get("conv_fwd_bf16x3", 0)) / max(alg_n, 1))
                                                      if traffic else None),
                         "peak_note": "2500 TFLOP/s dense bf16 MFMA / 3 MFMA products per fp32-equivalent MAC",
                         "mfma_busy": mfma_busy, "hbm_gbs": hbm_gbs,
                         "mfma_issued_tflops": 3 * ach, "launches_per_step": n / 2,
                         "algorithmic_gflop_per_step": fl / 2 / 1e9, "avg_launch_ms": ms / n,
                         "kernel_ms_per_step": ms / 2},
            "algorithmic_tflops_whole_step": ALGO_GFLOP_PER_IMAGE * GLOBAL_BATCH * a.steps / elapsed / 1e3,
            "whole_step_frac_of_peak": ALGO_GFLOP_PER_IMAGE * GLOBAL_BATCH * a.steps / elapsed / 1e3 / peak,
            "final_losses": dict(zip(("center_old", "offset_old", "center_new", "offset_new"), losses)),
        }
        if structured is not None:
            res["structured_labels"] = structured
        if trainer.buckets.distributed:
            res["allreduce"] = trainer.comm_stats()
        wgrad_peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if ops.WGRAD_PRECISION == 1 else PEAK_FP32_MFMA_TFLOPS
        for kind, pk in (("conv_fwd_bf16x6", PEAK_BF16_MFMA_TFLOPS / 6.0), ("conv_fwd_f16x3", PEAK_BF16_MFMA_TFLOPS / 3.0), ("conv_fwd_f32", PEAK_FP32_MFMA_TFLOPS),
                         ("conv_wgrad", wgrad_peak)):
            if kind in summ:
                kn, kf, kms = summ[kind]
                res[kind] = {"achieved": kf / (kms * 1e-3) / 1e12, "peak": pk, "unit": "TFLOP/s",
                             "frac": kf / (kms * 1e-3) / 1e12 / pk, "ms_per_step": kms / 2,
                             "launches_per_step": kn / 2}
    # Comparator legs, outside the timed region. stock_rocm is a COLLECTIVE leg at N > 1 (every rank takes part: stock PyTorch-ROCm
    # ops + gradient all-reduce = the DDP comparator BASELINE.md section 4 names); cpu_baseline runs on rank 0 at N = 1 only.
    stock, phase1 = None, None
    del trainer, out
    from cl4wsis_amd.nn import CACHE
    CACHE.clear()
    torch.cuda.empty_cache()
    if world == 1 and not a.no_phase1 and b == GLOBAL_BATCH:
        # BASELINE config 2 in the driver's record: the phase-1 step through the same Trainer.step, 10 timed graph
        # replays after its own warm-up (tuner + one capture per FLAC angle), before the host-side legs
        _progress("phase1 sub-record: warm-up + 10 timed steps")
        phase1 = run_p1(a, device, rank, world, b, 10, 3)
        if phase1 is not None:
            for k in ("metric", "n_gpus", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "final_losses"):
                phase1.pop(k, None)
            _progress(f"phase1 sub-record: {phase1['ms_per_step']:.2f} ms/step")
    if not a.no_cpu_baseline:
        stock = stock_rocm(opts, device, rank=rank, world=world)
    if rank == 0:
        if phase1 is not None:
            res["phase1"] = phase1
        if stock is not None:
            # BASELINE.md publishes no number for this metric; its §4 names stock PyTorch-ROCm as the comparator to beat (>= 4x),
            # measured here on the same GPUs, same global batch, same synthetic inputs.
            res["stock_rocm"] = stock
            res["vs_stock_rocm"] = res["value"] / stock["value"]
            res["vs_baseline_note"] = ("vs_baseline is null: BASELINE.md holds no published number for this metric (§1); vs_stock_rocm = value / "
                                       "stock_rocm.value is the comparator its §4 names")
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(opts, a.cpu_baseline_iters)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(a, argv))
    if a.rehearse_launcher:
        rehearse(a)
    elif a.config == "p1":
        worker_p1(a)
    else:
        worker(a)


if __name__ == "__main__":
    main()
