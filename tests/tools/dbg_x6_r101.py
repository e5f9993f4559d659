"""Train-mode ResNet-101 body at crop 65, b=2: forward / gradient distance to a float64 evaluation for each conv precision
and for the CPU fp32 oracle, over a few seeds (debug aid for the conditioning of the phase-1 gradient test)."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))

import sys
import torch
from cl4wsis_amd import body_train as BT, ops
from cl4wsis_amd.backbone import ResNet
from oracle import model as omodel, weights

gpu = torch.device("cuda:0")
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    torch.manual_seed(0)
    body = ResNet([3, 4, 23, 3], keep_outputs=True)
    sd = weights.fill_({k: v.clone() for k, v in body.state_dict().items()}, seed=11 + seed)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(2, 3, 65, 65, generator=g)
    res = {}
    for dt in (torch.float64, torch.float32):
        osd = {"body." + k: v.to(dt).clone() if v.is_floating_point() else v.clone() for k, v in sd.items()}
        names = [k for k in osd if k.endswith("weight") or k.endswith("bias")]
        for k in names:
            osd[k].requires_grad_(True)
        outs = omodel.resnet_body(x.to(dt), osd, structure=(3, 4, 23, 3), training=True)
        if dt == torch.float64:
            dout = torch.randn(outs[4].shape, generator=g, dtype=torch.float64) / outs[4].numel() ** 0.5
        (outs[4] * dout.to(dt)).sum().backward()
        res[str(dt)] = (outs[4].detach().double(), {k[5:]: osd[k].grad.double() for k in names})
    ref_o, ref_g = res["torch.float64"]
    body.to(gpu).train()
    for prec in (0, 2, 1):
        BT.PRECISION, BT.WGRAD_PRECISION = prec, 0
        body.load_state_dict(sd)
        outs, tape = BT.body_forward(body, x.to(gpu))
        grads = {}

        def grad_of(p):
            if id(p) not in grads:
                grads[id(p)] = torch.zeros_like(p, dtype=torch.float32)
            return grads[id(p)]
        BT.body_backward(body, tape, {4: dout.float().to(gpu)}, grad_of)
        res[f"gpu{prec}"] = (outs[4].double().cpu(), {n: grads[id(p)].double().cpu() for n, p in body.named_parameters()})
        ops._hip.call("cl4_conv2d_fwd_clear_plans")
    line = [f"seed {seed}"]
    for key in ("torch.float32", "gpu0", "gpu2", "gpu1"):
        o, gr = res[key]
        fe = ((o - ref_o).norm() / ref_o.norm()).item()
        ge = torch.tensor([((gr[n] - ref_g[n]).norm() / (ref_g[n].norm() + 1e-30)).item() for n in ref_g])
        line.append(f"{key[-7:]}: fwd {fe:.1e} grad med {ge.median():.1e} max {ge.max():.1e}")
    print(" | ".join(line), flush=True)
