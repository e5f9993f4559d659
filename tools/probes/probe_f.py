"""Experiments: the fused identity bottleneck (csrc/conv_f.hip) against the same block as three tuned planes launches, res2 shape.
usage: python tools/probes/probe_f.py [n ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cl4wsis_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
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
    cm, h, w = int(os.environ.get("CM", 64)), int(os.environ.get("HW", 129)), int(os.environ.get("HW", 129))
    c4 = 4 * cm
    g = torch.Generator().manual_seed(1)
    ws = [torch.randn(cm, c4, 1, 1, generator=g) / c4 ** 0.5, torch.randn(cm, cm, 3, 3, generator=g) / (9 * cm) ** 0.5,
          torch.randn(c4, cm, 1, 1, generator=g) / cm ** 0.5]
    ps = [ops.pack_conv_weight(t.to(dev), 1) for t in ws]
    aff = [torch.rand(c, generator=g) + 0.5 if i % 2 == 0 else torch.randn(c, generator=g) * 0.1 for i, c in enumerate((cm, cm, cm, cm, c4, c4))]
    aff = [t.to(dev) for t in aff]
    affc = torch.cat(aff).contiguous()
    for n in [int(v) for v in sys.argv[1:]] or [32, 16]:
        x = torch.randn(n, c4, h, w, generator=g).to(dev)
        xp = ops.to_planes(x)
        del x
        out = ops.Planes(n, c4, h, w, dev)
        y1 = ops.Planes(n, cm, h, w, dev)
        y2 = ops.Planes(n, cm, h, w, dev)

        def three():
            ops.conv2d_planes(xp, ps[0], 1, scale=aff[0], shift=aff[1], act1=ops.ACT_LEAKY, want_fp32=False, out_planes=y1)
            ops.conv2d_planes(y1, ps[1], 3, pad=1, scale=aff[2], shift=aff[3], act1=ops.ACT_LEAKY, want_fp32=False, out_planes=y2)
            ops.conv2d_planes(y2, ps[2], 1, scale=aff[4], shift=aff[5], post=xp, act2=ops.ACT_LEAKY, want_fp32=False, out_planes=out)

        def fused():
            ops.bottleneck_fused(xp, ps[0], ps[1], ps[2], affc, 0.01, 0.01, 0.01, out_planes=out)

        def fused32():
            ops.bottleneck_fused(xp, ps[0], ps[1], ps[2], affc, 0.01, 0.01, 0.01, out_planes=out, want_fp32=True)

        ops.AUTOTUNE = True
        three()
        ops.AUTOTUNE = False
        t3, tf, tf32 = timeit(three), timeit(fused), timeit(fused32)
        gflop = 2.0 * n * h * w * (2 * c4 * cm + 9 * cm * cm) / 1e9
        gb = 2.0 * n * c4 * h * w * 4 / 1e9
        print(f"n={n} {c4}ch {h}x{w}: three launches {t3:8.1f} us | fused {tf:8.1f} us = {gflop / tf * 1e3:6.1f} TFLOP/s, "
              f"{gb / tf * 1e3:5.2f} TB/s of trunk read + write ({gb:.2f} GB) | with fp32 copy {tf32:8.1f} us")


if __name__ == "__main__":
    main()
