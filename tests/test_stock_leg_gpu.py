"""GPU: bench.py's `stock_rocm` comparator is the oracle's p2_step with its tensors on the GPU (stock PyTorch-ROCm ops; label
stage on the host like the reference). It must compute the same step as the CPU oracle, or the ratio it feeds is meaningless."""
import pytest
import torch

from util import p2_opts

pytestmark = pytest.mark.gpu


def test_oracle_step_on_cuda_matches_cpu(gpu):
    from cl4wsis_amd import segmentation_module as sm
    from cl4wsis_amd.wss import PeakGenerator, PseudoLabeler
    from oracle import step as ostep, weights
    crop = 97
    opts = p2_opts(crop)
    torch.manual_seed(0)
    mods = (sm.make_model(opts, classes=[16, 5]), sm.make_model(opts, classes=[16]), PseudoLabeler(2048, 21, sm.get_norm(opts)),
            PeakGenerator(21, 20, old_classes=15))
    sds = [weights.fill_({k: v.clone() for k, v in m.state_dict().items()}, seed=11 + i) for i, m in enumerate(mods)]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, crop, crop, generator=g)
    l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
    keep = (torch.rand(2, 256, 7, 7, generator=g) >= 0.5).float()
    names = ("center_old", "offset_old", "center_new", "offset_new")

    def run(dev, autocast=False):
        new, old, pl, pg = ({k: v.clone().to(dev) for k, v in d.items()} for d in sds)
        out = ostep.p2_step(new, old, pl, pg, x.to(dev), l1h.to(dev), 16, pooling=opts.pooling, dropout_mask=keep.to(dev),
                            lr=opts.lr * opts.lr_head, autocast=autocast)
        return [float(out[k].detach()) for k in names], out["seg_gt"].cpu()

    ref, seg_ref = run(torch.device("cpu"))
    got, seg = run(gpu)
    assert (seg != seg_ref).float().mean() < 1e-4
    for a, b, k in zip(got, ref, names):
        assert abs(a - b) <= 2e-3 * max(abs(b), 1e-3), (k, a, b)
    amp, _ = run(gpu, autocast=True)                      # the reference's own arithmetic (train.py:252): only sanity here
    assert all(v == v and abs(v) < 1e6 for v in amp), amp
