"""Where does the persistent kernel (plan impl 3) differ from the plain planes kernel? Error pattern by channel / pixel / tile."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import _hip, ops
gpu = torch.device("cuda:0")
n, cin, h, w, cout = 9, 256, 33, 33, 1024
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, h, w, generator=g)
wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
post = torch.randn(n, cout, h, w, generator=g)
xs, wp = ops.to_planes(x.to(gpu)), ops.pack_conv_weight(wt.to(gpu), 1)
d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, 1, 1, 0, 1, cout, 0, 0, 0.0, 0, 0.0, 3)
outs = {}
for impl in (2, 3):
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), impl, 128, 1)
    outs[impl] = ops.from_planes(ops.conv2d_planes(xs, wp, 1, post=ops.to_planes(post.to(gpu)), act2=ops.ACT_NONE, want_fp32=False, want_planes=True)[1]).cpu()
    torch.cuda.synchronize()
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
a, b = outs[2], outs[3]
bad = (a - b).abs() > 1e-4 * a.abs().max()
print("bad fraction", float(bad.float().mean()))
flat = bad.permute(1, 0, 2, 3).reshape(cout, -1)            # [channel][pixel index p = n*HW + hw]
npix = flat.shape[1]
print("by channel % 16:", [round(float(flat[c::16].float().mean()), 3) for c in range(16)])
print("by channel block (c // 16) % 8:", [round(float(flat.reshape(cout // 16, 16, npix)[k::8].float().mean()), 3) for k in range(8)])
print("by M tile (c // 128):", [round(float(flat[k * 128:(k + 1) * 128].float().mean()), 3) for k in range(cout // 128)])
pcol = torch.arange(npix)
print("by pixel % 32 // 16 (j):", [round(float(flat[:, (pcol % 32) // 16 == j].float().mean()), 3) for j in range(2)])
print("by pixel quarter (p % 128) // 32 (wn):", [round(float(flat[:, (pcol % 128) // 32 == k].float().mean()), 3) for k in range(4)])
nt = -(-npix // 128)
per_tile = [round(float(flat[:, t * 128:(t + 1) * 128].float().mean()), 2) for t in range(nt)]
print("by pixel tile:", per_tile)
# is the wrong value = conv without residual, or residual of another place?
nores = ops.from_planes(ops.conv2d_planes(xs, wp, 1, want_fp32=False, want_planes=True)[1]).cpu()
d0 = (b - nores)
print("where bad: |b - conv_without_residual| median", float(d0[bad].abs().median()), " |post| median", float(post[bad].abs().median()))
r_used = b - nores                                            # the residual the kernel actually added
idx = bad.nonzero()[:5]
for i in idx:
    nn, cc, yy, xx = [int(v) for v in i]
    val = float(r_used[nn, cc, yy, xx])
    m = ((post - val).abs() < 2e-2 * max(abs(val), 1e-3)).nonzero()
    print("bad at (n,c,y,x)", (nn, cc, yy, xx), "added", round(val, 4), "expected", round(float(post[nn, cc, yy, xx]), 4), "candidates holding that value:", m[:4].tolist())
