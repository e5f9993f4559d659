"""GPU parity of the phase-1 / step-0 loss-stack kernels (csrc/wss.hip, through the C ABI) against the outputs of the
reference's own functions (tests/golden/wss_sXX.npz) and, at other shapes, against the oracle restatement.
0/1 masks are compared exactly; floats within 1e-5 of the tensor scale (north-star tolerance: 1e-3 relative)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import synth, wss

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "wss_s*.npz")))
TOL = 1e-5


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_wss_stack_matches_reference(gpu, path):
    from cl4wsis_amd import ops
    g = np.load(path)
    seed, b, c, h, w = (int(v) for v in g["case"])
    d = synth.wss_inputs(seed, b, c, h, w)
    soft = d["logits"].softmax(1)
    logits, labels = d["logits"].to(gpu), d["labels"].to(gpu)

    out = ops.pamr(d["img"].to(gpu), soft.to(gpu))
    assert _rel(out, g["pamr"]) < TOL

    nl = labels.shape[1] // 2 + 1
    for red in ("mean", "sum"):
        loss, ngwp, dx = ops.ngwp_focal_bce(logits, labels[:, -nl:].contiguous(), reduction=red)
        assert _rel(loss[0], g[f"bce_{red}"]) < TOL
        assert _rel(ngwp, g["ngwp"]) < TOL
        assert _rel(dx, g[f"bce_{red}_grad"]) < 2e-5

    softd = soft.to(gpu)
    pg = ops.pseudo_gtmask(softd, labels)
    assert np.array_equal(pg.cpu().numpy().astype(np.uint8), g["pseudo_gt"])
    assert np.array_equal(ops.binarize(softd, labels).cpu().numpy().astype(np.uint8), g["binarize"])

    loss, dmask = ops.balanced_mask_loss_ce(logits, pg, labels)
    assert _rel(loss[0], g["bal_ce"]) < TOL
    assert _rel(dmask, g["bal_ce_grad"]) < 2e-5

    loss, dx = ops.bce_ignore_mean(d["big"].to(gpu), d["seg_lab"].to(gpu))
    torch.cuda.synchronize()
    assert _rel(loss[0], g["bce_ignore"]) < TOL
    assert _rel(dx[:, :, ::7, ::5], g["bce_ignore_grad_slice"]) < TOL
    assert _rel(torch.stack([dx.double().abs().sum(), dx.double().pow(2).sum()]), g["bce_ignore_grad_sum"]) < TOL


@pytest.mark.parametrize("shape", [(1, 3, 97, 97), (2, 2, 65, 65), (1, 5, 9, 7), (2, 7, 33, 33)])
def test_pamr_paths_vs_oracle(gpu, shape):
    """LDS kernel with 4 / 1 planes per workgroup, the global ping-pong path (97x97 planes do not fit LDS), class
    counts that are not a multiple of the plane group, a plane smaller than the largest dilation."""
    from cl4wsis_amd import ops
    b, c, h, w = shape
    g = torch.Generator().manual_seed(h * 100 + c)
    img = torch.rand(b, 3, h, w, generator=g)
    img[:, :, : h // 2] *= 0.3
    mask = torch.randn(b, c, h, w, generator=g).softmax(1)
    ref = wss.pamr(img, mask)
    out = ops.pamr(img.to(gpu), mask.to(gpu))
    assert _rel(out, ref) < TOL
    # odd iteration count exercises the other ping-pong parity; a low-resolution mask goes through the resize
    small = torch.randn(b, c, max(h // 2, 2), max(w // 2, 2), generator=g).softmax(1)
    ref = wss.pamr(img, small, num_iter=3, dilations=(1, 3))
    out = ops.pamr(img.to(gpu), small.to(gpu), num_iter=3, dilations=(1, 3))
    assert _rel(out, ref) < TOL


def test_balanced_ce_gating_and_empty(gpu):
    """Images whose pseudo-label class set differs from the image labels get zero weight (bw), an all-zero pseudo GT
    gives a zero loss and zero gradient."""
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(5)
    b, c, h, w = 3, 6, 12, 10
    x = torch.randn(b, c, h, w, generator=g)
    labels = torch.zeros(b, c - 1)
    labels[0, 1] = labels[1, 2] = labels[2, 0] = 1
    pg = torch.zeros(b, c, h, w)
    pg[0, 0, :6] = 1; pg[0, 2, 6:9] = 1            # bg + class 2 present, label says class 2 -> counted
    pg[1, 0, :6] = 1                                # labelled class missing -> bw = 0
    pg[2, 0, :3] = 1; pg[2, 1, 3:5] = 1; pg[2, 4, 8:] = 1   # an unlabelled class present: ignored by the count
    xr = x.clone().requires_grad_(True)
    ref = wss.balanced_mask_loss_ce(xr, pg, labels)
    ref.backward()
    loss, dmask = ops.balanced_mask_loss_ce(x.to(gpu), pg.to(gpu), labels.to(gpu))
    assert _rel(loss[0], ref.detach()) < TOL
    assert _rel(dmask, xr.grad) < 2e-5
    loss, dmask = ops.balanced_mask_loss_ce(x.to(gpu), torch.zeros_like(pg).to(gpu), labels.to(gpu))
    assert loss.item() == 0.0 and dmask.abs().max().item() == 0.0


def test_bce_ignore_all_ignored_and_large(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 129, 129, generator=g)
    t = torch.full((2, 129, 129), 255, dtype=torch.int64)
    loss, dx = ops.bce_ignore_mean(x.to(gpu), t.to(gpu))
    assert loss.item() == 0.0 and dx.abs().max().item() == 0.0
    t = torch.randint(0, 16, (2, 129, 129), generator=g)
    t[:, :40] = 255
    xr = x.clone().requires_grad_(True)
    ref = wss.bce_ignore_mean(xr, t)
    ref.backward()
    loss, dx = ops.bce_ignore_mean(x.to(gpu), t.to(gpu))
    assert _rel(loss[0], ref.detach()) < TOL and _rel(dx, xr.grad) < TOL


def test_wss_bad_arguments_raise(gpu):
    from cl4wsis_amd import ops
    x = torch.zeros(1, 4, 8, 8, device=gpu)
    with pytest.raises(RuntimeError):
        ops.pamr(torch.zeros(1, 3, 8, 8, device=gpu), x, dilations=())
    with pytest.raises(RuntimeError):
        ops.ngwp_focal_bce(x, torch.zeros(1, 5, device=gpu))
    with pytest.raises(RuntimeError):
        ops.bce_ignore_mean(x, torch.zeros(1, 8, 8, device=gpu))
