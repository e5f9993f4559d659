"""Times cl4_conv2d_wgrad on the weight gradients of the phase-2 step (b = 16): the three ASPP 3x3 and the 1x1 layers of the decoder."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops

dev = torch.device("cuda", 0)
for n, cin, hw, cout, k, d in [(16, 2048, 33, 256, 3, 3), (16, 2048, 33, 256, 3, 9), (16, 2048, 33, 256, 1, 1), (16, 1280, 33, 256, 1, 1), (16, 320, 65, 256, 1, 1), (16, 288, 129, 128, 1, 1)]:
    x = torch.randn(n, cin, hw, hw, device=dev)
    dy = torch.randn(n, cout, hw, hw, device=dev)
    dw = torch.zeros(cout, cin, k, k, device=dev)
    for prec in (1, 2):
        ops.conv2d_wgrad(x, dy, dw, k, pad=d * (k // 2), dil=d, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d_wgrad(x, dy, dw, k, pad=d * (k // 2), dil=d, precision=prec)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * n * hw * hw * cout * cin * k * k
        print(f"n {n} cin {cin} {hw}x{hw} cout {cout} k {k} d {d} precision {prec}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
