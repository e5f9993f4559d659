"""GPU parity of the network forward passes (HIP kernels) against the CPU oracle restatement on the same
name-keyed deterministic weights. Tolerance: 1e-3 relative (north star); observed ~1e-5."""
import pytest
import torch

from util import p2_opts, rel_err

pytestmark = pytest.mark.gpu


def _models(gpu, crop):
    from cl4wsis_amd import segmentation_module as sm
    from oracle import weights
    opts = p2_opts(crop)
    torch.manual_seed(0)
    model = sm.make_model(opts, classes=[16, 5])
    sd = weights.fill_(model.state_dict(), seed=1)
    model.load_state_dict(sd)
    model.to(gpu)
    return opts, model, {k: v.clone() for k, v in sd.items()}


@pytest.mark.parametrize("crop", [65, 97])
def test_forward_seg_and_full_eval(gpu, crop):
    from oracle import model as om
    opts, model, sd = _models(gpu, crop)
    model.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, crop, crop, generator=g)
    with torch.no_grad():
        ref, aux = om.forward_full(x, sd, pooling=opts.pooling, training=False)
    pred, feats = model(x.to(gpu), interpolate=False)
    torch.cuda.synchronize()
    for k in ("res1", "res2", "res3", "res4", "res5"):
        assert rel_err(feats["features"][k].cpu(), aux["features"][k]) < 1e-4, k
    for k in ("seg", "center", "offset"):
        assert pred[k].shape == ref[k].shape
        assert rel_err(pred[k].cpu(), ref[k]) < 1e-4, k
    pred2, _ = model.forward_seg(x.to(gpu), interpolate=False)
    assert rel_err(pred2["seg"].cpu(), ref["seg"]) < 1e-4
