"""GPU parity of the network forward passes (HIP kernels) against the CPU oracle restatement on the same
name-keyed deterministic weights. Tolerance: 1e-3 relative (north star); observed ~1e-5."""
import pytest
import torch

from util import p2_opts, rel_err

pytestmark = pytest.mark.gpu

# Network outputs: the north-star tolerance (1e-3 relative, fp32). Measured: ~2e-5 with the exact fp32 MFMA kernels
# (CL4_CONV_PRECISION=0) and ~1.2e-4 with the default bf16x3 split kernels, through ~110 conv layers.
NET_TOL = 1e-3


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
        assert rel_err(feats["features"][k].cpu(), aux["features"][k]) < NET_TOL, k
    for k in ("seg", "center", "offset"):
        assert pred[k].shape == ref[k].shape
        assert rel_err(pred[k].cpu(), ref[k]) < NET_TOL, k
    pred2, _ = model.forward_seg(x.to(gpu), interpolate=False)
    assert rel_err(pred2["seg"].cpu(), ref["seg"]) < NET_TOL


def test_instance_branch_train_fwd_bwd(gpu):
    """Trainable phase-2 sub-net: batch-stat BN, injected Dropout mask, hand-written backward vs autograd."""
    from oracle import model as om
    crop = 65
    opts, model, sd = _models(gpu, crop)
    model.train(); model.body.eval(); model.head.eval(); model.cls.eval()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 3, crop, crop, generator=g)
    # oracle ------------------------------------------------------------------------------------
    # the oracle runs in float64 here: gradients through batch-stat BN are scale-invariant directions with
    # heavy cancellation, so an fp32 reference is itself only good to ~1e-3 on some tensors.
    osd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    x64 = x.double()
    names = [k for k in osd if (k.startswith("decoder.") or k.startswith("instance_head.")) and osd[k].is_floating_point()
             and not k.endswith(("running_mean", "running_var"))]
    for k in names:
        osd[k].requires_grad_(True)
    with torch.no_grad():
        _, aux = om.forward_seg(x64, osd, pooling=opts.pooling)
    feats = {k: v.detach() for k, v in aux["features"].items()}
    keep = (torch.rand(3, 256, feats["res5"].shape[2], feats["res5"].shape[3], generator=g) >= 0.5)
    ins = om.instance_decoder(feats, osd, training=True, dropout_mask=keep.double())
    ref = om.instance_head(ins, osd, training=True)
    gc = torch.randn(ref["center"].shape, generator=g)
    go = torch.randn(ref["offset"].shape, generator=g)
    ((ref["center"] * gc.double()).sum() + (ref["offset"] * go.double()).sum()).backward()
    # HIP ---------------------------------------------------------------------------------------
    # The unit under test is the trainable branch (exact fp32 MFMA kernels): feed it the ORACLE's features, so that the
    # ~1e-4 deviation of the bf16x3 body (checked on its own above) is not amplified ~100x by batch-stat BN over 75
    # samples into this gradient comparison.
    f = {k: v.float().to(gpu).contiguous() for k, v in feats.items()}
    br = model.instance_branch()
    pred, tape = br.forward_train(f, dropout_keep=keep.to(torch.uint8).to(gpu))
    grads = br.backward(tape, {"center": gc.to(gpu), "offset": go.to(gpu)})
    torch.cuda.synchronize()
    for k in ("center", "offset"):
        assert rel_err(pred[k].cpu().double(), ref[k].detach()) < NET_TOL, k
    params = dict(model.named_parameters())
    bufs = dict(model.named_buffers())
    # Gradients through batch-stat BN + ReLU are not a smooth function at fp32 resolution: a pre-activation
    # within ~1e-7 of zero flips its ReLU mask between two fp32 evaluation orders and moves a per-channel sum by
    # one element (measured: the CPU fp32 oracle itself is 1e-3..1e-2 max-norm away from this float64 truth).
    # So the per-tensor check is a relative L2 norm, with the 1e-3 bar applied to the median over tensors.
    errs = []
    for k in names:
        gg = grads.get(id(params[k]))
        assert gg is not None, f"no gradient for {k}"
        ref_g = osd[k].grad
        e = ((gg.cpu().double() - ref_g).norm() / (ref_g.norm() + 1e-30)).item()
        errs.append(e)
        assert e < 3e-2, (k, e)
    errs.sort()
    print("grad rel-L2 error: median %.2e  max %.2e" % (errs[len(errs) // 2], errs[-1]))
    assert errs[len(errs) // 2] < 1e-3
    for k in osd:
        if k.endswith(("running_mean", "running_var")) and k.startswith(("decoder.", "instance_head.")):
            assert rel_err(bufs[k].cpu().double(), osd[k]) < NET_TOL, k
