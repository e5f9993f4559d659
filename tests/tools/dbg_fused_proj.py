"""Debug: fused first block (projection) vs four planes launches: mismatch pattern."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cl4wsis_amd import ops  # noqa: E402

gpu = torch.device("cuda", 0)
cm = 64
for n, h, w in ((1, 13, 13), (2, 33, 33)):
    g = torch.Generator().manual_seed(5)
    w1 = (torch.randn(cm, cm, 1, 1, generator=g) / cm ** 0.5).to(gpu)
    w2 = (torch.randn(cm, cm, 3, 3, generator=g) / (9 * cm) ** 0.5).to(gpu)
    w3 = (torch.randn(4 * cm, cm, 1, 1, generator=g) / cm ** 0.5).to(gpu)
    wp = (torch.randn(4 * cm, cm, 1, 1, generator=g) / cm ** 0.5).to(gpu)
    aff = [(torch.rand(c, generator=g) * 1.5 + 0.25 if i % 2 == 0 else torch.randn(c, generator=g) * 0.3).to(gpu)
           for i, c in enumerate((cm, cm, cm, cm, 4 * cm, 4 * cm, 4 * cm, 4 * cm))]
    s1, b1, s2, b2, s3, b3, sp, bp = aff
    x = torch.randn(n, cm, h, w, generator=g).to(gpu)
    xp = ops.to_planes(x)
    p1, p2, p3, pp = (ops.pack_conv_weight(t, 1) for t in (w1, w2, w3, wp))
    for cin, cout, k, pad in ((cm, cm, 1, 0), (cm, cm, 3, 1), (cm, 4 * cm, 1, 0)):
        d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, 1, cout, 0, ops.ACT_LEAKY, 0.01, ops.ACT_NONE, 0.01, 3)
        ops.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    _, rp = ops.conv2d_planes(xp, pp, 1, scale=sp, shift=bp, want_fp32=False, want_planes=True)
    _, y1 = ops.conv2d_planes(xp, p1, 1, scale=s1, shift=b1, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    _, y2 = ops.conv2d_planes(y1, p2, 3, pad=1, scale=s2, shift=b2, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    _, ref_p = ops.conv2d_planes(y2, p3, 1, scale=s3, shift=b3, post=rp, act2=ops.ACT_LEAKY, slope2=0.01, want_fp32=False, want_planes=True)
    z = lambda t: torch.zeros_like(t)
    for name, (a1, a2, a3, ap) in (("full", (w1, w2, w3, wp)), ("w2 = 0", (w1, z(w2), w3, wp)), ("w1 = 0", (z(w1), w2, w3, wp)),
                                   ("w3 = 0", (w1, w2, z(w3), wp)), ("wp = 0", (w1, w2, w3, z(wp)))):
        q1, q2, q3, qp = (ops.pack_conv_weight(t, 1) for t in (a1, a2, a3, ap))
        _, rp = ops.conv2d_planes(xp, qp, 1, scale=sp, shift=bp, want_fp32=False, want_planes=True)
        _, y1 = ops.conv2d_planes(xp, q1, 1, scale=s1, shift=b1, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
        _, y2 = ops.conv2d_planes(y1, q2, 3, pad=1, scale=s2, shift=b2, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
        _, ref = ops.conv2d_planes(y2, q3, 1, scale=s3, shift=b3, post=rp, act2=ops.ACT_LEAKY, slope2=0.01, want_fp32=False, want_planes=True)
        _, out_p = ops.bottleneck_fused(xp, q1, q2, q3, torch.cat(aff).contiguous(), 0.01, 0.01, 0.01, wp=qp)
        torch.cuda.synchronize()
        a, b = ops.from_planes(out_p), ops.from_planes(ref)
        bad = a != b
        print(n, h, w, name, "mismatching", int(bad.sum()), "of", bad.numel(), "max|d| %.3e" % (a - b).abs().max().item(), "max|ref| %.2e" % b.abs().max().item())
        if name == "w2 = 0" and n == 1:
            print("   got  ", a[0, :12, 5, 5].tolist())
            print("   want ", b[0, :12, 5, 5].tolist())
            print("   got px", a[0, 3, 0, :13].tolist())
            print("   want px", b[0, 3, 0, :13].tolist())
            bad_c = bad[0].flatten(1).float().mean(1)
            print("   frac bad per channel (first 40)", [round(float(v), 2) for v in bad_c[:40]])
