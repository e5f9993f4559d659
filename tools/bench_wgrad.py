"""Micro-benchmark of cl4_conv2d_wgrad on the trainable-branch shapes (b=16) for each arithmetic (0 fp32, 1 bf16x3, 2 bf16x6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops
dev = torch.device("cuda:0")
for name, n, cin, h, cout, k, pad, dil in [("aspp 3x3 2048->256 d6", 16, 2048, 33, 256, 3, 6, 6), ("aspp 1x1 2048->256", 16, 2048, 33, 256, 1, 0, 1),
                                           ("proj 1x1 1280->256", 16, 1280, 33, 256, 1, 0, 1), ("fuse 1x1 144->128 @129", 16, 144, 129, 128, 1, 0, 1),
                                           ("head 1x1 128->32 @129", 16, 128, 129, 32, 1, 0, 1), ("res4 3x3 256->256", 16, 256, 33, 256, 3, 1, 1),
                                           ("res4 1x1 1024->256", 16, 1024, 33, 256, 1, 0, 1), ("res2 1x1 64->256 @129", 16, 64, 129, 256, 1, 0, 1)]:
    res = []
    for prec in (0, 1, 2):
        x = torch.randn(n, cin, h, h, device=dev)
        dy = torch.randn(n, cout, h, h, device=dev)
        dw = torch.zeros(cout, cin, k, k, device=dev)
        for _ in range(2):
            ops.conv2d_wgrad(x, dy, dw, k, pad=pad, dil=dil, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d_wgrad(x, dy, dw, k, pad=pad, dil=dil, precision=prec)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res.append(f"p{prec} {ms:7.3f} ms {2.0 * n * h * h * cout * cin * k * k / ms / 1e9:6.1f} TF")
    print(f"{name:26s} " + " | ".join(res), flush=True)
