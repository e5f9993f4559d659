import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from cl4wsis_amd import _hip, ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (n, cin, h, w, cout) in ((4, 64, 32, 32, 256), (4, 64, 33, 33, 256), (8, 64, 65, 67, 256), (2, 64, 16, 20, 128), (1, 32, 8, 16, 128), (3, 64, 36, 36, 256)):
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    ref = F.conv2d(x.double(), wt.double()).float()
    xd = x.to(dev)
    wp = ops.pack_conv_weight(wt.to(dev), 1)
    d = ops._conv_desc(xd, cin, 0, cout, 1, 1, 0, 1, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    y = ops.conv2d(xd, wp, 1).cpu()
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    # which pixels are wrong?
    bad = ((y - ref).abs() > 1e-3 * ref.abs().max()).any(1)      # [n, h, w]
    print(f"n={n} cin={cin} {h}x{w} HW%4={(h * w) % 4} cout={cout}: rel err {err:.2e}; bad pixels {int(bad.sum())} of {bad.numel()}; "
          f"first bad flat idx {bad.flatten().nonzero()[:6].flatten().tolist()}")
