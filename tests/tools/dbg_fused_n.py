"""Debug: fused bottleneck vs three planes launches at a given n / size (bitwise)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cl4wsis_amd import ops  # noqa: E402
from test_conv_f_gpu import _block  # noqa: E402

gpu = torch.device("cuda", 0)
for n, h, w in ((3, 129, 129), (5, 129, 129), (8, 129, 129), (16, 129, 129), (16, 65, 65), (40, 33, 33)):
    cm = 64
    ws, aff = _block(cm, 1, gpu)
    x = torch.randn(n, 4 * cm, h, w, generator=torch.Generator().manual_seed(2)).to(gpu)
    xp = ops.to_planes(x)
    p1, p2, p3 = (ops.pack_conv_weight(t, 1) for t in ws)
    s1, b1, s2, b2, s3, b3 = aff
    for cin, cout, k, pad in ((4 * cm, cm, 1, 0), (cm, cm, 3, 1), (cm, 4 * cm, 1, 0)):
        d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, 1, cout, 0, ops.ACT_LEAKY, 0.01, ops.ACT_NONE, 0.01, 3)
        ops.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    _, y1 = ops.conv2d_planes(xp, p1, 1, scale=s1, shift=b1, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    _, y2 = ops.conv2d_planes(y1, p2, 3, pad=1, scale=s2, shift=b2, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    _, ref_p = ops.conv2d_planes(y2, p3, 1, scale=s3, shift=b3, post=xp, act2=ops.ACT_LEAKY, slope2=0.01, want_fp32=False, want_planes=True)
    _, out_p = ops.bottleneck_fused(xp, p1, p2, p3, torch.cat(aff).contiguous(), 0.01, 0.01, 0.01)
    torch.cuda.synchronize()
    a, b = ops.from_planes(out_p), ops.from_planes(ref_p)
    bad = (a != b)
    per_img = bad.flatten(1).any(1).nonzero().flatten().tolist()
    print(n, h, w, "equal" if not bad.any() else f"DIFF in images {per_img} max|d| {(a - b).abs().max().item():.3e}")
    if bad.any():
        nb = bad.sum().item()
        im = per_img[0]
        bi = bad[im]
        ch = bi.flatten(1).any(1).nonzero().flatten().tolist()
        ys = bi.any(0).any(1).nonzero().flatten().tolist()
        xs_ = bi.any(0).any(0).nonzero().flatten().tolist()
        print("   wrong elements", nb, "of", bad.numel(), "| image", im, "channels", ch[:40], "... rows", ys[:40], "cols", xs_[:40])
        c0 = ch[0]
        print("   sample got/want", a[im, c0][bi[c0]][:6].tolist(), b[im, c0][bi[c0]][:6].tolist())
