"""Relative L2 error of conv2d at precision 0/1/2 against float64 for the small-crop body shapes (debug aid)."""
import sys
import torch
import torch.nn.functional as F
from cl4wsis_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = []
for n, hin in ((3, 4), (3, 7), (3, 13), (3, 25), (3, 49)):
    for cin, cout in ((64, 64), (64, 256), (256, 64), (512, 128), (128, 512), (2048, 512), (512, 2048), (3, 64)):
        for k, s, p, d in ((1, 1, 0, 1), (3, 1, 1, 1), (3, 2, 1, 1), (3, 1, 2, 2), (3, 1, 4, 4), (1, 2, 0, 1), (7, 2, 3, 1)):
            if cin * cout * k * k > 2048 * 512 * 9 or (k == 7) != (cin == 3):
                continue
            if (hin + 2 * p - d * (k - 1) - 1) // s + 1 < 1:
                continue
            shapes.append((n, hin, cin, cout, k, s, p, d))
bad = 0
for (n, hin, cin, cout, k, s, p, d) in shapes:
    x = torch.randn(n, cin, hin, hin, generator=g)
    if len(sys.argv) > 1:      # sparse, wide-dynamic-range input like a zero-inserted gradient
        x = x * torch.exp(3 * torch.randn(n, cin, hin, hin, generator=g)) * (torch.rand(n, cin, hin, hin, generator=g) < 0.25)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), stride=s, padding=p, dilation=d)
    errs = []
    for prec in (0, 1, 2):
        wp = ops.pack_conv_weight(w.to(dev), prec)
        y = ops.conv2d(x.to(dev), wp, k, stride=s, pad=p, dil=d)
        errs.append(((y.double().cpu() - ref).norm() / ref.norm()).item())
    flag = "" if errs[2] < 3 * errs[0] + 1e-7 else "   <<<<"
    bad += bool(flag)
    if flag or "-v" in sys.argv:
        print((n, hin, cin, cout, k, s, p, d), " ".join(f"{e:.2e}" for e in errs), flag)
print("shapes", len(shapes), "bad", bad)
