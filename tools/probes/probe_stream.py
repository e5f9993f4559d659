"""Practical streaming rates on this box for the res4 conv3 output tensor (143 MB): elementwise read+write, write-only fill."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cl4wsis_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(32, 1024, 33, 33, device=dev)
sc = torch.ones(1024, device=dev); sh = torch.zeros(1024, device=dev)
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
mb = x.numel() * 4 / 1e6
us = t(lambda: ops.channel_affine(x, sc, sh)); print(f"channel_affine r+w {us:7.1f} us  {2 * mb / us:6.2f} TB/s (read + write)")
y = torch.empty_like(x)
us = t(lambda: y.copy_(x)); print(f"torch copy     r+w {us:7.1f} us  {2 * mb / us:6.2f} TB/s")
us = t(lambda: y.zero_()); print(f"torch zero_      w {us:7.1f} us  {mb / us:6.2f} TB/s (write only)")
us = t(lambda: ops.add_act(x, y, ops.ACT_RELU, 0.0) if hasattr(ops, 'add_act') else None); print(f"add_act      2r+w {us:7.1f} us  {3 * mb / us:6.2f} TB/s")
