"""GPU parity of the train-mode DeepLabV3 head + incremental classifier and of the PseudoLabeler (heads_train.py) against
float64 autograd of the oracle restatement (oracle/model.py). Same criterion as test_body_train_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

from oracle import model as omodel
from oracle import weights

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _grad_store():
    grads = {}
    return grads, (lambda p: grads.setdefault(id(p), torch.zeros_like(p, dtype=torch.float32)))


def test_deeplab_head_and_classifier(gpu):
    from cl4wsis_amd import heads_train
    from cl4wsis_amd.heads import DeeplabV3, IncrementalClassifier
    torch.manual_seed(0)
    cin = 96
    head = DeeplabV3(cin, 256, 256, out_stride=16, pooling_size=32)
    cls = IncrementalClassifier([torch.nn.Conv2d(256, c, 1) for c in (16, 5)])
    sd = weights.fill_({**{"head." + k: v.clone() for k, v in head.state_dict().items()},
                        **{"cls." + k: v.clone() for k, v in cls.state_dict().items()}}, seed=3)
    head.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("head.")})
    cls.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("cls.")})
    head.to(gpu).train(); cls.to(gpu).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, cin, 9, 9, generator=g)
    osd = {k: v.double().clone() for k, v in sd.items()}
    for k in osd:
        if k.endswith("weight") or k.endswith("bias"):
            osd[k].requires_grad_(True)
    xr = x.double().requires_grad_(True)
    ref = omodel.incremental_classifier(omodel.deeplab_head(xr, osd, training=True), osd)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * dy).sum().backward()

    xd = x.to(gpu)
    y, tape = heads_train.deeplab_forward(head, xd)
    out, cx = heads_train.cls_forward(cls, y)
    assert _rel(out, ref.detach()) < 1e-4
    grads, grad_of = _grad_store()
    dyh = heads_train.cls_backward(cls, cx, dy.float().to(gpu), grad_of)
    dx = heads_train.deeplab_backward(head, tape, dyh, grad_of)
    assert _rel(dx, xr.grad) < 1e-3
    errs = []
    for prefix, mod in (("head.", head), ("cls.", cls)):
        for n, p in mod.named_parameters():
            r = osd[prefix + n].grad
            errs.append(((grads[id(p)].double().cpu() - r).norm() / (r.norm() + 1e-30)).item())
    errs = torch.tensor(errs)
    assert errs.max() < 3e-2 and errs.median() < 1e-3, (errs.max(), errs.median())


def test_pseudolabeler_train(gpu):
    from cl4wsis_amd import heads_train
    from cl4wsis_amd.nn import ABN
    from cl4wsis_amd.wss import PseudoLabeler
    torch.manual_seed(0)
    pl = PseudoLabeler(64, 21, ABN)
    sd = weights.fill_({k: v.clone() for k, v in pl.state_dict().items()}, seed=8)
    pl.load_state_dict(sd)
    pl.to(gpu).train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(6, 64, 9, 9, generator=g)
    osd = {k: v.double().clone() for k, v in sd.items()}
    for k in osd:
        if k.endswith("weight") or k.endswith("bias"):
            osd[k].requires_grad_(True)
    xr = x.double().requires_grad_(True)
    ref = omodel.pseudolabeler(xr, osd, training=True)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * dy).sum().backward()
    y, tape = heads_train.pseudolabeler_forward(pl, x.to(gpu))
    assert _rel(y, ref.detach()) < 1e-4
    grads, grad_of = _grad_store()
    dx = heads_train.pseudolabeler_backward(pl, tape, dy.float().to(gpu), grad_of)
    assert _rel(dx, xr.grad) < 1e-3
    errs = torch.tensor([((grads[id(p)].double().cpu() - osd[n].grad).norm() / (osd[n].grad.norm() + 1e-30)).item()
                         for n, p in pl.named_parameters()])
    assert errs.max() < 3e-2 and errs.median() < 1e-3, (errs.max(), errs.median())


def test_pooling_branch_gradient_commuted_sum_equals_literal_order(gpu, monkeypatch):
    """CL4_POOL_GRAD=1: sum over the plane first (cl4_plane_sum, fp64 partials), 1x1 input-gradient product on the sums — against the
    literal order (product, then the plane sum) and against float64, on a well-conditioned input."""
    import torch.nn as nn
    from cl4wsis_amd import heads, ops
    g = torch.Generator().manual_seed(3)
    n, cout, h = 3, 64, 9
    conv = nn.Conv2d(5 * cout, cout, 1, bias=False).to(gpu)
    dpz = torch.randn(n, cout, h, h, generator=g) + 0.3
    dcat5 = heads._dx_1x1(dpz.to(gpu), conv)
    want = torch.einsum("nkhw,kc->nc", dpz.double(), conv.weight.detach().cpu().double()[:, 4 * cout:, 0, 0])
    s = ops.plane_sum(dpz.to(gpu))
    assert torch.allclose(s.cpu().double().flatten(), dpz.double().sum((2, 3)).flatten(), rtol=1e-6, atol=1e-5)
    for mode in (False, True):
        monkeypatch.setattr(heads, "POOL_GRAD_COMMUTED", mode)
        got = heads.InstanceBranch._pool_branch_grad(dpz.to(gpu), dcat5, conv, cout)
        assert tuple(got.shape) == (n, cout, 1, 1)
        err = (got.cpu().double().flatten() - want.flatten()).abs().max() / want.abs().max()
        assert err < 2e-5, (mode, float(err))
