"""where do exact-fp32 products (CL4_CONV_PRECISION=0) land further from float64 than PyTorch-CPU fp32? Per layer:
relative L2 error against a float64 convolution of (a) oneDNN fp32 on the host (the reference arithmetic: the CPU oracle), (b) the exact-fp32
MFMA kernel (one fp32 accumulator per output walks the whole K = cin k^2 range, 2 products per step), (c) bf16x6, (d) fp16x3, (e) bf16x3 —
and an emulation on the host of (f) ONE sequential fp32 accumulator over K and (g) 16 interleaved accumulators summed at the end (what a
16-lane FMA loop does), on a sample of outputs. usage: python tests/tools/probe_fp32_gap.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from cl4wsis_amd import ops

dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
print(f"{'layer':34s} {'K':>6s} {'oneDNN fp32':>12s} {'HIP fp32 MFMA':>14s} {'bf16x6':>9s} {'fp16x3':>9s} {'bf16x3':>9s} {'1 accumulator':>14s} {'16 accumulators':>16s}")
for name, n, cin, h, cout, k in (("res4 conv2 3x3 256->256", 2, 256, 33, 256, 3), ("res4 conv1 1x1 1024->256", 2, 1024, 33, 256, 1),
                                 ("res4 conv3 1x1 256->1024", 2, 256, 33, 1024, 1), ("ASPP 3x3 2048->256", 2, 2048, 33, 256, 3),
                                 ("res2 conv2 3x3 64->64 @129", 1, 64, 129, 64, 3)):
    g = torch.Generator().manual_seed(cin + k)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    e_cpu = rel(F.conv2d(x, w, padding=k // 2), ref)
    xd, wd = x.to(dev), w.to(dev)
    errs = []
    for prec in (0, 2, 3, 1):
        wp = ops.pack_conv_weight(wd, prec, force_f16=prec == 3)
        errs.append(rel(ops.conv2d(xd, wp, k, pad=k // 2), ref))
    # host emulation on 4096 sampled outputs: im2col rows in float32, summed (f) sequentially, (g) in 16 interleaved partial sums
    cols = F.unfold(x[:1], k, padding=k // 2)[0].t()[:64]                      # [64 pixels, K]
    prod = (cols[:, None, :] * w.reshape(cout, -1)[None, :64, :]).reshape(-1, cin * k * k)     # [4096, K] fp32 products
    exact = prod.double().sum(1)
    seq = torch.zeros(prod.shape[0])
    for j in range(prod.shape[1]):
        seq = seq + prod[:, j]
    K = prod.shape[1]
    pad = (-K) % 16
    p16 = F.pad(prod, (0, pad)).reshape(prod.shape[0], -1, 16)
    acc = torch.zeros(prod.shape[0], 16)
    for j in range(p16.shape[1]):
        acc = acc + p16[:, j]
    lanes = acc[:, :8] + acc[:, 8:]
    lanes = lanes[:, :4] + lanes[:, 4:]
    lanes = lanes[:, :2] + lanes[:, 2:]
    par = lanes[:, 0] + lanes[:, 1]
    e_seq, e_par = float((seq.double() - exact).norm() / exact.norm()), float((par.double() - exact).norm() / exact.norm())
    print(f"{name:34s} {cin * k * k:6d} {e_cpu:12.2e} {errs[0]:14.2e} {errs[1]:9.2e} {errs[2]:9.2e} {errs[3]:9.2e} {e_seq:14.2e} {e_par:16.2e}")
