"""Micro-benchmark of cl4_conv2d_wgrad on the trainable-branch shapes (b=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops
dev = torch.device("cuda:0")
for name, n, cin, h, cout, k, pad, dil in [("aspp 3x3 2048->256 d6", 16, 2048, 33, 256, 3, 6, 6), ("aspp 1x1 2048->256", 16, 2048, 33, 256, 1, 0, 1),
                                           ("proj 1x1 1280->256", 16, 1280, 33, 256, 1, 0, 1), ("fuse 1x1 144->128 @129", 16, 144, 129, 128, 1, 0, 1),
                                           ("head 1x1 128->32 @129", 16, 128, 129, 32, 1, 0, 1)]:
    x = torch.randn(n, cin, h, h, device=dev)
    dy = torch.randn(n, cout, h, h, device=dev)
    dw = torch.zeros(cout, cin, k, k, device=dev)
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, dw, k, pad=pad, dil=dil)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d_wgrad(x, dy, dw, k, pad=pad, dil=dil)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:28s} {ms:8.3f} ms {2.0 * n * h * h * cout * cin * k * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
