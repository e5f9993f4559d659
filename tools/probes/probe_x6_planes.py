"""Experiments: bf16x6 convolutions of the phase-1 step (body / head / PseudoLabeler, forward and input-gradient shapes) on the role-split
kernel (fp32 input) against cl4_to_planes3 + the three-plane LDS-DMA kernel (conv_b). usage: python tools/probes/probe_x6_planes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cl4wsis_amd import ops  # noqa: E402

SHAPES = [  # (n, cin, h, cout, k, dil)
    (16, 256, 33, 256, 3, 1), (16, 256, 33, 1024, 1, 1), (16, 1024, 33, 256, 1, 1), (16, 512, 33, 512, 3, 2), (48, 256, 33, 2048, 3, 1),
    (48, 2048, 33, 256, 3, 1), (16, 512, 33, 2048, 1, 1), (16, 2048, 33, 512, 1, 1), (16, 256, 33, 2048, 3, 6), (16, 2048, 33, 256, 3, 6),
    (16, 128, 65, 128, 3, 1), (16, 512, 65, 128, 1, 1), (48, 256, 33, 256, 3, 1), (16, 1024, 33, 2048, 1, 1), (16, 2048, 33, 1024, 1, 1),
]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    print(f"{'shape (n, cin, h, cout, k, dil)':36s} {'role-split us':>13s} {'to_planes3 us':>13s} {'conv_b x6 us':>12s} {'sum':>8s} {'gain':>6s}  max rel diff")
    for n, cin, h, cout, k, dil in SHAPES:
        x = torch.randn(n, cin, h, h, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        wp = ops.pack_conv_weight(w, 2)
        pad = dil * (k - 1) // 2
        ops.AUTOTUNE = True
        y0 = ops.conv2d(x, wp, k, pad=pad, dil=dil)
        x3 = ops.to_planes(x, planes=3)
        y1 = ops.conv2d_planes(x3, wp, k, pad=pad, dil=dil, want_fp32=True)[0]
        ops.AUTOTUNE = False
        t0 = timeit(lambda: ops.conv2d(x, wp, k, pad=pad, dil=dil))
        t1 = timeit(lambda: ops.to_planes(x, planes=3, out=x3))
        t2 = timeit(lambda: ops.conv2d_planes(x3, wp, k, pad=pad, dil=dil, want_fp32=True))
        rel = ((y0 - y1).abs().max() / y0.abs().max()).item()
        print(f"{str((n, cin, h, cout, k, dil)):36s} {t0:13.1f} {t1:13.1f} {t2:12.1f} {t1 + t2:8.1f} {t0 / (t1 + t2):6.2f}  {rel:.1e}")
        del x, x3, y0, y1


if __name__ == "__main__":
    main()
