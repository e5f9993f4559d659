"""GPU parity of the fused optimisers against torch.optim on CPU (same parameters, same gradients, several steps with a
changing learning rate): FusedSGD = torch.optim.SGD(momentum=0.9, nesterov=True) with per-group lr / weight decay
(train.py:145-180), FusedAdam = torch.optim.Adam (train.py:181-182). Tolerance 1e-6 relative (fp32 update arithmetic)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(gen, device):
    shapes = [(7, 3, 3, 3), (7,), (5, 7, 1, 1), (5,), (130, 20)]
    return [torch.nn.Parameter(torch.randn(s, generator=gen).to(device)) for s in shapes]


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_fused_optimizer_matches_torch(gpu, kind):
    from cl4wsis_amd.optim import FusedAdam, FusedSGD
    g = torch.Generator().manual_seed(3)
    ref_p = _params(g, "cpu")
    dev_p = [torch.nn.Parameter(p.detach().clone().to(gpu)) for p in ref_p]
    groups = lambda ps: [{"params": ps[:2], "weight_decay": 1e-4}, {"params": ps[2:4], "weight_decay": 1e-4, "lr": 0.1},
                         {"params": ps[4:], "weight_decay": 0.0, "lr": 0.01}]
    if kind == "sgd":
        ref = torch.optim.SGD(groups(ref_p), lr=0.01, momentum=0.9, nesterov=True)
        opt = FusedSGD(groups(dev_p), lr=0.01, momentum=0.9, nesterov=True)
    else:
        ref = torch.optim.Adam(groups(ref_p), lr=0.01, weight_decay=1e-4)
        opt = FusedAdam(groups(dev_p), lr=0.01, weight_decay=1e-4)
    for it in range(5):
        for gr, go in zip(ref.param_groups, opt.param_groups):
            gr["lr"] = go["lr"] = gr["lr"] * 0.9          # a schedule
        opt.zero_grad()
        for pr, pd in zip(ref_p, dev_p):
            grad = torch.randn(pr.shape, generator=g)
            pr.grad = grad.clone()
            opt.grad_view(pd).copy_(grad.to(gpu))
        ref.step()
        opt.step()
    torch.cuda.synchronize()
    for pr, pd in zip(ref_p, dev_p):
        assert ((pr.detach() - pd.detach().cpu()).abs().max() / pr.detach().abs().max()).item() < 2e-6
    if kind == "sgd":
        sd = opt.state_dict()["state"]
        rd = ref.state_dict()["state"]
        for k in rd:
            assert torch.allclose(rd[k]["momentum_buffer"], sd[k]["momentum_buffer"].cpu(), rtol=1e-5, atol=1e-7)
