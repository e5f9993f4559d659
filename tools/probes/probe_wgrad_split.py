"""Experiment: the role-split weight-gradient kernel's K-split target (CL4_WGRAD_TARGET_WGS workgroups per launch) on the shapes of the
phase-1 / phase-2 steps. usage: CL4_WGRAD_TARGET_WGS=512 python tools/probes/probe_wgrad_split.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cl4wsis_amd import ops  # noqa: E402

SHAPES = [  # (n, cin, h, cout, k, dil)
    (16, 256, 33, 256, 3, 1), (16, 256, 33, 1024, 1, 1), (16, 1024, 33, 256, 1, 1), (16, 512, 33, 512, 3, 2), (16, 512, 33, 2048, 1, 1),
    (16, 2048, 33, 512, 1, 1), (16, 128, 65, 128, 3, 1), (16, 128, 65, 512, 1, 1), (16, 512, 65, 128, 1, 1), (16, 64, 129, 64, 3, 1),
    (16, 64, 129, 256, 1, 1), (16, 256, 129, 64, 1, 1), (16, 256, 33, 256, 1, 1), (16, 1280, 33, 256, 1, 1), (16, 2048, 33, 256, 1, 1),
]


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    ops.WGRAD_GEMM = False
    tot = 0.0
    for n, cin, h, cout, k, dil in SHAPES:
        x = torch.randn(n, cin, h, h, generator=g).to(dev)
        dy = torch.randn(n, cout, h, h, generator=g).to(dev)
        dw = torch.zeros(cout, cin, k, k, device=dev)
        pad = dil * (k - 1) // 2
        f = lambda: ops.conv2d_wgrad(x, dy, dw, k, pad=pad, dil=dil, precision=1)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us
        print(f"{str((n, cin, h, cout, k, dil)):32s} {us:8.1f} us  {2.0 * n * h * h * cin * cout * k * k / us / 1e6:7.1f} TFLOP/s", flush=True)
    print(f"sum {tot:.1f} us (target {os.environ.get('CL4_WGRAD_TARGET_WGS', '1024')})")


if __name__ == "__main__":
    main()
