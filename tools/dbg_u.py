"""Where does the unified-wave kernel differ from a float64 convolution? usage: dbg_u.py n cin h w cout k dil"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from cl4wsis_amd import _hip, ops
n, cin, h, w, cout, k, dil = [int(v) for v in sys.argv[1:8]]
gpu = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, h, w, generator=g)
wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
pad = dil * (k // 2)
ref = F.conv2d(x.double(), wt.double(), padding=pad, dilation=dil).float()
xd = x.to(gpu)
wp = ops.pack_conv_weight(wt.to(gpu), 1)
d = ops._conv_desc(xd, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
_hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
y = ops.conv2d(xd, wp, k, pad=pad, dil=dil).cpu()
print("u launches", _hip.load().cl4_conv_debug_u_launches())
err = (y - ref).abs()
bad = (err > 1e-4 * ref.abs().max()).nonzero()
print("max err", err.max().item(), "bad elements", len(bad), "of", y.numel())
import collections
print("bad by (y,x):", collections.Counter((int(b[2]), int(b[3])) for b in bad).most_common(40))
print("bad by channel (first 10):", collections.Counter(int(b[1]) for b in bad).most_common(10))
# which single (tap) term explains the error at a bad position? try removing / adding each tap's contribution
if len(bad):
    b = bad[0]; nn, co, yy, xx = [int(v) for v in b]
    print("example", (nn, co, yy, xx), "got", y[nn, co, yy, xx].item(), "ref", ref[nn, co, yy, xx].item())
    for ky in range(k):
        for kx in range(k):
            iy, ix = yy + (ky - k // 2) * dil, xx + (kx - k // 2) * dil
            inside = 0 <= iy < h and 0 <= ix < w
            term = float((x[nn, :, iy, ix] * wt[co, :, ky, kx]).sum()) if inside else 0.0
            # wrapped read: flat index
            flat = yy * w + xx + (ky - k // 2) * dil * w + (kx - k // 2) * dil
            wrapped = float((x[nn].reshape(cin, -1)[:, flat] * wt[co, :, ky, kx]).sum()) if 0 <= flat < h * w else 0.0
            print(f"  tap ({ky},{kx}) inside={inside} term={term:+.4f} wrapped-read term={wrapped:+.4f}")
    print("  diff", (y[nn, co, yy, xx] - ref[nn, co, yy, xx]).item())
