"""Debug (CL4_F_DBG=2048 build): the PROJ kernel emits leaky(shift3 + y2[8q + r]) for every 32-channel slab; compare with the real y2."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cl4wsis_amd import ops
gpu = torch.device("cuda", 0)
cm, n, h, w = 64, 1, 13, 13
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
for cin, cout, k, pad in ((cm, cm, 1, 0), (cm, cm, 3, 1)):
    d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, 1, cout, 0, ops.ACT_LEAKY, 0.01, ops.ACT_NONE, 0.01, 3)
    ops.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
_, y1 = ops.conv2d_planes(xp, p1, 1, scale=s1, shift=b1, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
_, y2 = ops.conv2d_planes(y1, p2, 3, pad=1, scale=s2, shift=b2, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
y2f = ops.from_planes(y2)            # [1, 64, 13, 13]
_, out_p = ops.bottleneck_fused(xp, p1, p2, p3, torch.cat(aff).contiguous(), 0.01, 0.01, 0.01, wp=pp)
torch.cuda.synchronize()
out = ops.from_planes(out_p)
for m in (0, 1, 6, 7):
    want = torch.nn.functional.leaky_relu(b3[32 * m:32 * m + 32][None, :, None, None] + y2f[:, 32 * (m & 1):32 * (m & 1) + 32], 0.01)
    got = out[:, 32 * m:32 * m + 32]
    d = (got - want).abs()
    print("   per-channel max diff", [round(float(v), 3) for v in d[0].flatten(1).max(1).values], "\n   per-row max diff", [round(float(v), 3) for v in d[0].max(0).values.max(1).values])
    print("slab", m, "max |got - want| %.3e" % (got - want).abs().max().item(), "got sample", got[0, :4, 5, 5].tolist(), "want", want[0, :4, 5, 5].tolist())
