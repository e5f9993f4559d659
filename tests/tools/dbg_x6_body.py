"""Per-parameter gradient difference of the train-mode body between conv precisions (debug aid)."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))

import torch
from cl4wsis_amd import body_train as BT
from cl4wsis_amd.backbone import ResNet
from oracle import weights

gpu = torch.device("cuda:0")
torch.manual_seed(0)
body = ResNet([2, 1, 2, 1], keep_outputs=True)
sd = weights.fill_({k: v.clone() for k, v in body.state_dict().items()}, seed=11)
body.load_state_dict(sd)
body.to(gpu).train()
g = torch.Generator().manual_seed(2)
x = torch.randn(3, 3, 49, 49, generator=g).to(gpu)
res = {}
for prec in (0, 2, 1):
    BT.PRECISION = prec
    BT.WGRAD_PRECISION = 0
    body.load_state_dict(sd)
    outs, tape = BT.body_forward(body, x)
    if prec == 0:
        douts = [torch.randn(o.shape, generator=g).to(gpu) / o.numel() ** 0.5 for o in outs]
    grads = {}

    def grad_of(p):
        if id(p) not in grads:
            grads[id(p)] = torch.zeros_like(p, dtype=torch.float32)
        return grads[id(p)]
    BT.body_backward(body, tape, {i: d for i, d in enumerate(douts)}, grad_of)
    res[prec] = ({n: grads[id(p)].double() for n, p in body.named_parameters()}, [o.double() for o in outs])
for i in range(len(res[0][1])):
    print("out", i, " ".join(f"{((res[p][1][i] - res[0][1][i]).norm() / res[0][1][i].norm()).item():.2e}" for p in (2, 1)))
for n in res[0][0]:
    r = res[0][0][n]
    print(f"{n:40s}", " ".join(f"{((res[p][0][n] - r).norm() / r.norm()).item():.2e}" for p in (2, 1)))
