"""Debug: ResNet101 body, fused res2 blocks on / off, per-stage difference and time. usage: dbg_fused_body.py [n] [crop]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cl4wsis_amd import backbone, nn as cnn, ops  # noqa: E402
from oracle import weights  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
crop = int(sys.argv[2]) if len(sys.argv) > 2 else 513
gpu = torch.device("cuda", 0)
torch.manual_seed(0)
body = backbone.net_resnet101(norm_act=cnn.InPlaceABNSync, output_stride=16, keep_outputs=True)
body.load_state_dict(weights.fill_(body.state_dict(), seed=5))
body.to(gpu).eval()
for p in body.parameters():
    p.requires_grad_(False)
x = torch.randn(n, 3, crop, crop, generator=torch.Generator().manual_seed(3)).to(gpu)
outs = {}
for fused in (False, True):
    ops.FUSED_BLOCK = fused
    with torch.no_grad():
        body(x)
        torch.cuda.synchronize()
        t0 = time.time()
        o = body(x)
        torch.cuda.synchronize()
    print("fused", fused, "%.2f ms" % ((time.time() - t0) * 1e3), [bool(torch.isfinite(t).all()) for t in o])
    outs[fused] = [t.clone() for t in o]
for i, (a, b) in enumerate(zip(outs[False], outs[True])):
    print("res%d" % (i + 1), tuple(a.shape), "rel err %.3e" % ((a - b).abs().max() / a.abs().max()).item(), "max |a| %.3e" % a.abs().max().item())
