"""GPU parity of the train-mode WiderResNet-38 A2 body (cl4wsis_amd/body_train.py: wrn_body_forward / wrn_body_backward — coco-voc
step 0 / phase 1, models/wide_resnet.py:82-207, modules/residual.py:101-181) against the oracle restatement differentiated by
PyTorch autograd in float64 on the CPU. Dropout2d masks (mod6 / mod7) are injected on both sides: the reference draws them from
torch's generator, which no other implementation can reproduce. Outputs and running statistics: 1e-4 relative; gradients: relative
L2 per tensor, max < 3e-2 and median < 5e-3 (the bar of tests/test_body_train_gpu.py, for the reason given there)."""
import pytest
import torch

from oracle import model as omodel
from oracle import weights

pytestmark = pytest.mark.gpu


def _iabn():
    from functools import partial
    from cl4wsis_amd.nn import InPlaceABNSync
    return partial(InPlaceABNSync, activation="leaky_relu", activation_param=0.01)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("structure,crop,partial_out", [((1, 1, 1, 1, 1, 1), 41, False), ((2, 1, 2, 1, 1, 1), 33, False),
                                                         ((1, 1, 1, 1, 1, 1), 33, True)])
def test_wrn_body_forward_backward(gpu, structure, crop, partial_out):
    from cl4wsis_amd import body_train
    from cl4wsis_amd.backbone import WiderResNetA2
    torch.manual_seed(0)
    body = WiderResNetA2(list(structure), norm_act=_iabn(), keep_outputs=True)
    sd = weights.fill_({k: v.clone() for k, v in body.state_dict().items()}, seed=13)
    body.load_state_dict(sd)
    body.to(gpu).train()
    g = torch.Generator().manual_seed(3)
    n = 3
    x = torch.randn(n, 3, crop, crop, generator=g)
    keeps = {"mod6.block1": (torch.rand(n, 1024, generator=g) > 0.3).to(torch.uint8),
             "mod7.block1": (torch.rand(n, 2048, generator=g) > 0.5).to(torch.uint8)}

    osd = {"body." + k: v.double().clone() for k, v in sd.items()}
    names = [k for k in osd if k.endswith("weight") or k.endswith("bias")]
    for k in names:
        osd[k].requires_grad_(True)
    outs_ref = omodel.wider_resnet_a2_body(x.double(), osd, structure=structure, training=True, keeps=keeps)
    douts = [torch.randn(o.shape, generator=g, dtype=torch.float64) / o.numel() ** 0.5 for o in outs_ref]
    used = [4] if partial_out else list(range(5))           # phase 1 reaches the body through the last output alone
    sum((outs_ref[i] * douts[i]).sum() for i in used).backward()

    outs, tape = body_train.wrn_body_forward(body, x.to(gpu), keeps={k: v.to(gpu) for k, v in keeps.items()})
    assert len(outs) == 5
    for o, r in zip(outs, outs_ref):
        assert o.shape == r.shape and _rel(o, r.detach()) < 1e-4
    grads = {}

    def grad_of(p):
        if id(p) not in grads:
            grads[id(p)] = torch.zeros_like(p, dtype=torch.float32)
        return grads[id(p)]

    body_train.wrn_body_backward(body, tape, {i: douts[i].float().to(gpu) for i in used}, grad_of)
    torch.cuda.synchronize()
    errs = []
    for name, p in body.named_parameters():
        ref = osd["body." + name].grad
        got = grad_of(p).double().cpu()
        errs.append(((got - ref).norm() / (ref.norm() + 1e-30)).item())
    errs = torch.tensor(errs)
    assert len(errs) == len(names)
    assert errs.max() < 3e-2 and errs.median() < 5e-3, (errs.max(), errs.median())
    for name, b in body.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            assert _rel(b, osd["body." + name]) < 1e-4, name


def test_dropout2d_matches_torch(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 10, 7, 5, generator=g)
    keep = (torch.rand(4, 10, generator=g) > 0.3).to(torch.uint8)
    y = ops.dropout2d(x.to(gpu), keep.to(gpu), 0.3).cpu()
    assert torch.allclose(y, torch.where(keep.bool()[:, :, None, None], x * (1.0 / (1.0 - 0.3)), torch.zeros(())), rtol=1e-6, atol=0)
    drawn = ops.dropout_mask((64, 512), 0.5, 1234, gpu)
    assert 0.45 < drawn.float().mean().item() < 0.55
