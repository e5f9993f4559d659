"""Debug aid (r05): take the fp16x3 chain apart — decoded planes, decoded packed weights, output / reference ratio."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from cl4wsis_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n, cin, h, w, cout, k = 1, 64, 5, 7, 128, 1
x = torch.randn(n, cin, h, w, generator=g)
wt = torch.randn(cout, cin, k, k, generator=g) / 8
xs = ops.to_planes_f16(x.to(dev))
raw = xs.data.cpu().numpy().view(np.float16).reshape(n, cin // 8, 2, h * w, 8).astype(np.float64)
xd = (raw[:, :, 0] + raw[:, :, 1]).transpose(0, 1, 3, 2).reshape(n, cin, h * w).reshape(n, cin, h, w) / float(xs.slot[0])
print("scale", float(xs.slot[0]), "planes decode max err", np.abs(xd - x.numpy()).max())
wp = ops.pack_conv_weight(wt.to(dev), 3)
wr = wp.data.cpu().numpy().view(np.float16).reshape(cin * k * k // 8, 2, cout, 8).astype(np.float64)
wd = (wr[:, 0] + wr[:, 1]).transpose(1, 0, 2).reshape(cout, cin) / 4096.0
print("weights decode max err (k = 1: K order = ci)", np.abs(wd - wt.numpy().reshape(cout, cin)).max())
y = ops.conv2d_planes_f16(xs, wp, k).cpu().double()
ref = F.conv2d(x.double(), wt.double())
print("rel err", ((y - ref).norm() / ref.norm()).item())
r = (y / ref).flatten()
print("ratio y/ref: median", r.median().item(), "min", r.min().item(), "max", r.max().item())
print("y[0, :4, 0, :4]", y[0, :4, 0, :4], "\nref", ref[0, :4, 0, :4])
# the products separately: hi x hi only etc. via zeroed planes
for name, (zh, zl) in {"x lo zeroed": (False, True), "x hi zeroed": (True, False)}.items():
    d = xs.data.clone().view(torch.int32)
    v = d.view(n, cin // 8, 2, h * w, 4)
    if zh: v[:, :, 0] = 0
    if zl: v[:, :, 1] = 0
    x2 = ops.Planes(n, cin, h, w, dev, data=d, planes=2, slot=xs.slot)
    y2 = ops.conv2d_planes_f16(x2, wp, k).cpu().double()
    print(name, "rel err vs full ref", ((y2 - ref).norm() / ref.norm()).item())
