"""BASELINE config 5 (coco-to-voc phase 2: WiderResNet-38 A2, output stride 8, 61+20 classes, crop 448, batch 32 over 8 GPUs =
4 per GPU): per-GPU step time on one MI355X. usage: bench_cfg5.py [per_gpu_batch] [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_oracle_golden import _wrn_opts
from cl4wsis_amd.train import Trainer

class _Log:
    def info(self, *a, **k): pass
    debug = info

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
opts = _wrn_opts(448)
opts.batch_size = b
opts.max_iters, opts.start_decay = 30 * 1000, 5 * 1000
torch.manual_seed(42)
tr = Trainer(_Log(), dev, opts)
g = torch.Generator().manual_seed(42)
images = torch.randn(b, 3, 448, 448, generator=g).to(dev)
l1h = torch.zeros(b, 80)
for i in range(b):
    l1h[i, 60 + int(torch.randint(0, 20, (1,), generator=g))] = 1
l1h = l1h.to(dev)
for _ in range(3):
    tr.step((images, l1h))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = tr.step((images, l1h))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": f"coco-voc phase 2 (config 5), WiderResNet-38 A2 OS8, crop 448, per-GPU batch {b}", "ms_per_step": 1e3 * dt,
                  "images_per_s_per_gpu": b / dt, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30,
                  "losses": out["losses"].tolist()}))
