"""Phase-1 step (BASELINE config 2: VOC 15-5 step 1 phase 1, batch 16, 513x513) timing on one GPU.
usage: bench_p1.py [per_gpu_batch] [steps] [epoch]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import p1_opts
from cl4wsis_amd.train import Trainer
from cl4wsis_amd.synthetic import synthetic_batch

class _Log:
    def info(self, *a, **k): pass
    debug = info

b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
epoch = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
opts = p1_opts(513)
opts.batch_size = b
opts.max_iters, opts.start_decay = 40 * 661, 5 * 661
torch.manual_seed(42)
tr = Trainer(_Log(), dev, opts)
images, l1h = synthetic_batch(b, 42, dev)
for _ in range(2):
    tr._step_phase1((images, l1h), cur_epoch=epoch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = tr._step_phase1((images, l1h), cur_epoch=epoch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": f"VOC 15-5 step 1 phase 1, epoch {epoch} (>= pseudo_ep: full loss stack), batch {b}, 513x513, Trainer.step eager",
                  "ms_per_step": 1e3 * dt, "images_per_s": b / dt, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30,
                  "losses": {k: float(v) for k, v in out["losses"].items()}}))
if len(sys.argv) > 4:      # per-kind time of the conv launches of one more step (HIP events on the launch stream)
    from cl4wsis_amd import ops
    ops.TIMER = ops.KernelTimer()
    tr._step_phase1((images, l1h), cur_epoch=epoch)
    torch.cuda.synchronize()
    for kind, (n, fl, ms) in sorted(ops.TIMER.summary().items(), key=lambda kv: -kv[1][2]):
        print(f"{kind:18s} launches {n:4d}  {ms:8.2f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
