import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (n, cin, h, cout, k, pad, dil) in [(2, 2048, 7, 256, 3, 6, 6), (2, 128, 25, 32, 1, 0, 1), (2, 1280, 7, 256, 1, 0, 1)]:
    x = torch.randn(n, cin, h, h, generator=g).to(dev)
    x = torch.relu(x)                       # many exact zeros like post-ReLU activations
    dy = torch.randn(n, cout, h, h, generator=g).to(dev) * 1e-3
    outs = []
    for rep in range(4):
        dw = torch.zeros(cout, cin, k, k, device=dev)
        ops.conv2d_wgrad(x, dy, dw, k, pad=pad, dil=dil)
        torch.cuda.synchronize()
        outs.append(dw.clone())
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), padding=pad, dilation=dil).float()
    d = max((outs[0] - o).abs().max().item() for o in outs[1:])
    print((n, cin, h, cout, k), "run-to-run max diff", d, "scale", ref.abs().max().item(), "err vs fp64", (outs[0] - ref).abs().max().item())
