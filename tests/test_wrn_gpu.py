"""GPU parity of the coco-voc configuration's network (WiderResNet-38 A2 body at output stride 8, DeepLabV3 with dilations
12/24/32, 61+20 classes; argparser.py:13-17) in eval mode — the frozen forward passes of its phase-2 step — against the
reference fixture (tests/golden/model_wrn38_c65.npz) and the oracle, in both conv arithmetics. Tolerance 1e-3 (north star)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("precision", [0, 1])
def test_wrn38_eval_forward(gpu, precision, monkeypatch):
    from cl4wsis_amd import ops, segmentation_module as sm
    from oracle import weights
    from test_oracle_golden import _wrn_opts
    monkeypatch.setattr(ops, "CONV_PRECISION", precision)
    gold = np.load(os.path.join(GOLD, "model_wrn38_c65.npz"))
    opts = _wrn_opts()
    torch.manual_seed(0)
    model = sm.make_model(opts, classes=[61, 20])
    model.load_state_dict(weights.fill_(model.state_dict(), seed=21))
    model.to(gpu).eval()
    g = torch.Generator().manual_seed(25)
    x = torch.randn(2, 3, 65, 65, generator=g)
    out, aux = model(x.to(gpu), interpolate=False)
    torch.cuda.synchronize()
    tol = 1e-4 if precision == 0 else 1e-3
    for k in ("seg", "center", "offset"):
        assert _rel(out[k], gold[k]) < tol, k
    for k, v in aux["features"].items():
        assert tuple(v.shape) == tuple(gold[f"feat_{k}_shape"])
        assert _rel(v[:, ::17, ::3, ::3], gold[f"feat_{k}_slice"]) < tol


def test_coco_voc_phase2_step_matches_oracle(gpu):
    """BASELINE config 5 (coco-to-voc, 61 old + 20 new classes, WiderResNet-38 at output stride 8): one phase-2 iteration at
    crop 129 against the CPU oracle on identical weights, inputs and Dropout mask (losses, semantic pseudo GT)."""
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep, weights
    from test_oracle_golden import _wrn_opts

    class _Log:
        def info(self, *a, **k):
            pass
        debug = info

    crop = 129
    opts = _wrn_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    assert tr.old_classes == 61 and tr.tot_classes == 81
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=51 + i)
        m.load_state_dict(sd)
        sds.append({k: v.clone() for k, v in sd.items()})
    tr._realias()
    new, old, pl, pg = sds
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, crop, crop, generator=g)
    l1h = torch.zeros(2, 80); l1h[0, 62] = 1; l1h[1, 60] = 1; l1h[1, 79] = 1
    h5 = (crop + 7) // 8
    keep = torch.rand(2, 256, h5, h5, generator=g) >= 0.5
    ref = ostep.p2_step(new, old, pl, pg, x, l1h, 61, pooling=opts.pooling, dropout_mask=keep.float(), lr=opts.lr * opts.lr_head)
    out = tr.step((x, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
    torch.cuda.synchronize()
    mism = out["seg_gt"].cpu().long() != ref["seg_gt"]
    if mism.any():
        top2 = ref["soft"].topk(2, dim=1).values
        assert (top2[:, 0] - top2[:, 1])[mism].max() < 1e-5
    assert _rel(out["center"], ref["pred_center"]) < 1e-3 and _rel(out["offset"], ref["pred_offset"]) < 1e-3
    for v, k in zip(out["losses"].cpu().tolist(), ("center_old", "offset_old", "center_new", "offset_new")):
        r = float(ref[k].detach())
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)


def test_cam_smooth_large_maps_match_lds_path(gpu):
    """20 new classes at 56x56 (coco-voc) exceed LDS: the four-launch path must equal the oracle and, on a case that fits,
    the single-launch LDS path bit for bit."""
    from cl4wsis_amd import _hip, ops
    from oracle import labels as L, model as M
    g = torch.Generator().manual_seed(12)
    b, tot, nc, h, w = 2, 81, 20, 56, 56
    int_masks = torch.randn(b, tot, h, w, generator=g) * 2
    pg = {"extra_conv4.weight": torch.randn(nc, nc, 1, 1, generator=g) * 0.3, "extra_conv4.bias": torch.randn(nc, generator=g) * 0.1}
    l1h = torch.zeros(b, tot - 1); l1h[0, 62] = 1; l1h[1, 60] = 1; l1h[1, 79] = 1
    assert _hip.load().cl4_cam_smooth_workspace_bytes(b, nc, h, w) > 0
    _, cam = M.peak_generator_eval(int_masks.clone(), pg, l1h, 60)
    ref = L.smoothing(cam)[:, 60:]
    got = ops.cam_smooth(int_masks.to(gpu), pg["extra_conv4.weight"].to(gpu), pg["extra_conv4.bias"].to(gpu), l1h[:, 60:].contiguous().to(gpu))
    assert _rel(got, ref) < 1e-5
    # same kernels' arithmetic on a small case: LDS path vs global path are not both reachable for one shape, so compare each
    # with the oracle at its own size
    b, tot, nc, h, w = 2, 21, 5, 33, 33
    assert _hip.load().cl4_cam_smooth_workspace_bytes(b, nc, h, w) == 0
