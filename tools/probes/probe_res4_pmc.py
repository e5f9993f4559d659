"""Experiment: HBM-side traffic of res4's three planes convolutions as the frozen networks launch them (tuned plan, in-place trunk), one shape
per kernel-name / grid pair so that a rocprofv3 --pmc pass can be read per shape. usage: python tools/probes/probe_res4_pmc.py [n]
(tools/probes/prof_pmc_res4.sh runs it under FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes and prints bytes per launch against the algorithmic bytes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cl4wsis_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    h = 33
    trunk = ops.to_planes((torch.randn(n, 1024, h, h, generator=g) * 0.5).to(dev))
    w1 = ops.pack_conv_weight((torch.randn(256, 1024, 1, 1, generator=g) / 32).to(dev), 1)
    w2 = ops.pack_conv_weight((torch.randn(256, 256, 3, 3, generator=g) / 48).to(dev), 1)
    w3 = ops.pack_conv_weight((torch.randn(1024, 256, 1, 1, generator=g) / 16).to(dev), 1)
    sc = [torch.rand(c, generator=g).to(dev) + 0.5 for c in (256, 256, 1024)]
    sh = [torch.randn(c, generator=g).to(dev) * 0.1 for c in (256, 256, 1024)]
    kw = dict(act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)

    def block():
        y1 = ops.conv2d_planes(trunk, w1, 1, scale=sc[0], shift=sh[0], **kw)[1]
        y2 = ops.conv2d_planes(y1, w2, 3, pad=1, scale=sc[1], shift=sh[1], **kw)[1]
        ops.conv2d_planes(y2, w3, 1, scale=sc[2], shift=sh[2], post=trunk, act1=ops.ACT_NONE, act2=ops.ACT_LEAKY, slope2=0.01, want_fp32=False,
                          out_planes=trunk)

    ops.AUTOTUNE = True
    block()
    ops.AUTOTUNE = False
    torch.cuda.synchronize()
    for _ in range(3):
        block()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        block()
    e1.record()
    e1.synchronize()
    px = n * h * h
    print(f"n = {n}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per block; algorithmic MB read / written: conv1 {px * 1024 * 4 / 1e6 + 1:.1f} / {px * 256 * 4 / 1e6:.1f}, "
          f"conv2 {px * 256 * 4 / 1e6 + 2.4:.1f} / {px * 256 * 4 / 1e6:.1f}, conv3 {px * (256 + 1024) * 4 / 1e6 + 1:.1f} / {px * 1024 * 4 / 1e6:.1f}")
    print("plans:", {k[:7]: v for k, v in ops._tuned.items()})


if __name__ == "__main__":
    main()
