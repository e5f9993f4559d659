"""Step-0 step (BASELINE config 1: VOC 15-5 step 0, supervised --bce, Adam; the reference runs it at batch 4 on the CPU as plumbing) on one
MI355X: the whole network in training mode, instance branch back-propagating into the backbone, BCE-with-ignore + weighted MSE / L1.
usage: bench_s0.py [per_gpu_batch] [steps]       (with a third argument: per-kind conv table of one more step)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import s0_opts
from cl4wsis_amd import body_train
from cl4wsis_amd.train import Trainer

class _Log:
    def info(self, *a, **k): pass
    debug = info

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
H = 513
opts = s0_opts(H)
opts.batch_size = b
opts.max_iters, opts.start_decay = 100 * 661, 5 * 661
torch.manual_seed(42)
tr = Trainer(_Log(), dev, opts)
g = torch.Generator().manual_seed(42)
images = torch.randn(b, 3, H, H, generator=g).to(dev)
seg = torch.randint(0, 16, (b, H, H), generator=g).to(dev)
center = torch.rand(b, 15, H, H, generator=g).to(dev)
offset = torch.randn(b, 2, H, H, generator=g).to(dev)
weights = (torch.rand(b, 1, H, H, generator=g) > 0.5).float().to(dev)
batch = (images, seg, center, offset, weights)
for _ in range(4):          # iteration 0 times the launch plans, 1 builds the pack table, 2 captures the step graph (r05)
    tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = tr.step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
arith = {0: "exact fp32 MFMA", 1: "bf16x3", 2: "bf16x6", 3: "fp16x3 (bf16x6 where not routed)"}[body_train.PRECISION]
print(json.dumps({"workload": f"VOC 15-5 step 0 (BASELINE config 1: supervised, --bce, Adam), batch {b}, 513x513, Trainer.step (HIP-graph replay from its third call on unless CL4_S0_GRAPH=0), differentiated convolutions {arith}",
                  "ms_per_step": 1e3 * dt, "images_per_s": b / dt, "algorithmic_tflops": 531.0 * b / dt / 1e3, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30,
                  "losses": {k: float(v) for k, v in out["losses"].items()}}))
if len(sys.argv) > 3:
    from cl4wsis_amd import ops
    ops.TIMER = ops.KernelTimer()
    tr.step(batch)
    torch.cuda.synchronize()
    for kind, (n, fl, ms) in sorted(ops.TIMER.summary().items(), key=lambda kv: -kv[1][2]):
        print(f"{kind:18s} launches {n:4d}  {ms:8.2f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
