import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from cl4wsis_amd import ops
dev = torch.device("cuda", 0)
n, cin, h, w, cout, k = 1, 64, 8, 8, 128, 1
# single-pixel probes: x = 1 at (ci, pixel p), dy = 1 at (co, pixel p) -> dw[co, ci] = 1, everything else 0
for (ci, co, p) in [(0, 0, 0), (3, 5, 0), (3, 5, 1), (3, 5, 2), (3, 5, 3), (3, 5, 4), (3, 5, 9), (40, 70, 33), (63, 127, 63)]:
    x = torch.zeros(n, cin, h, w, device=dev); dy = torch.zeros(n, cout, h, w, device=dev)
    x.view(n, cin, -1)[0, ci, p] = 1.0
    dy.view(n, cout, -1)[0, co, p] = 1.0
    dw = torch.zeros(cout, cin, k, k, device=dev)
    ops.conv2d_wgrad(x, dy, dw, k, precision=1)
    nz = dw.flatten().nonzero().flatten().tolist()
    print((ci, co, p), "->", [(i // cin, i % cin, round(dw.flatten()[i].item(), 4)) for i in nz][:8])
# full random: where are the errors
g = torch.Generator().manual_seed(0)
x = torch.randn(n, cin, h, w, generator=g).to(dev); dy = torch.randn(n, cout, h, w, generator=g).to(dev)
dw = torch.zeros(cout, cin, k, k, device=dev)
ops.conv2d_wgrad(x, dy, dw, k, precision=1)
ref = torch.einsum("nchw,ndhw->dc", x.double(), dy.double()).float()
err = (dw.view(cout, cin) - ref).abs()
print("max err", err.max().item(), "ref scale", ref.abs().max().item())
print("rows with error:", (err.max(1).values > 1e-3).nonzero().flatten().tolist()[:40])
print("cols with error:", (err.max(0).values > 1e-3).nonzero().flatten().tolist()[:40])
