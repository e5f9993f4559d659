"""5x5 depthwise conv timing on the decoder shapes (b=16, 129x129)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops
dev = torch.device("cuda:0")
for c in (128, 256, 32):
    x = torch.randn(16, c, 129, 129, device=dev)
    w = torch.randn(c, 1, 5, 5, device=dev)
    for _ in range(3):
        ops.dwconv5(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.dwconv5(x, w)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"c={c:4d}  {ms * 1e3:7.1f} us  {2 * x.numel() * 4 / ms / 1e9:6.2f} TB/s (read + write)", flush=True)
