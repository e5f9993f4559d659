"""GPU: the module-level plugin surface in TRAINING mode — the reference's own call sequence (train.py:253-274 phase 2,
train.py:274,304-337 phase 1, train.py:425-428 eval CAM, train.py:537 `loss.backward()`) replayed on cl4wsis_amd's modules with
torch glue in between, compared with what `Trainer.step` computes for the same iteration (gradients in its flat buffer)."""
import pytest
import torch
import torch.nn.functional as F

from util import QuietLogger, p1_opts, p2_opts, rel_err

pytestmark = pytest.mark.gpu


def _fill(tr, seed0):
    from oracle import weights
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=seed0 + i))
    tr._realias()


def _rel_l2(a, b):
    return ((a - b).double().norm() / (b.double().norm() + 1e-30)).item()


def test_phase2_reference_call_sequence_with_autograd(gpu):
    from cl4wsis_amd.train import Trainer
    from oracle import labels as L
    crop = 97
    opts = p2_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    g = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, crop, crop, generator=g).to(gpu)
    l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
    l1h = l1h.to(gpu)
    keep = (torch.rand(2, 256, 7, 7, generator=g) >= 0.5).to(torch.uint8).to(gpu)

    torch.manual_seed(0)
    ta = Trainer(QuietLogger(), gpu, opts)
    _fill(ta, 11)
    out = ta.step((images, l1h), dropout_keep=keep)                 # the product path: gradients stay in the flat buffer
    torch.cuda.synchronize()
    want = {n: ta._grad_map[id(p)].clone() for n, p in ta.model.named_parameters() if id(p) in ta._grad_map}

    torch.manual_seed(0)
    tb = Trainer(QuietLogger(), gpu, opts)
    _fill(tb, 11)
    model, model_old, bs, oc = tb.model, tb.model_old, 2, tb.old_classes
    tb._set_phase2_modes()                                          # train.py:234-240
    # ---- train.py:253-272, verbatim in structure -----------------------------------------------------------------
    with torch.no_grad():
        outputs_old, features_old = model_old(images, interpolate=False)
    with torch.no_grad():
        outputs, features = model.forward_seg(torch.cat([images, images.flip(-1)], dim=0), interpolate=False)
    features_ = {key: features["features"][key][:bs].detach().clone() for key in features["features"]}
    model.decoder._dropout_keep = keep                              # nn.Dropout's draw, recorded (as in the golden fixtures)
    _, instance_l = model.decoder(features_)
    instance = model.instance_head(instance_l)
    for key in instance.keys():
        outputs[key] = instance[key]
    assert instance["center"].requires_grad and instance["offset"].requires_grad
    assert rel_err(instance["center"].detach(), out["center"]) < 1e-6
    # ---- train.py:425-428: eval CAM through the modules + smoothing ----------------------------------------------------
    from cl4wsis_amd import ops
    tb.pseudolabeler.eval(); tb.peakgenerator.eval()
    int_masks = tb.pseudolabeler(features["body"][:bs])
    logit, cam = tb.peakgenerator(int_masks, l1h)
    assert tuple(cam.shape) == (2, 20, 7, 7) and tuple(logit.shape) == (2, 20)
    assert float(cam[:, :oc - 1].abs().max()) == 0 and float(logit[:, :oc - 1].abs().max()) == 0
    cam_s = ops.smoothing3x3(cam)
    fused = tb.peakgenerator.cam(int_masks, l1h[:, oc - 1:].contiguous())
    assert torch.allclose(cam_s[:, oc - 1:], fused, rtol=1e-6, atol=1e-7)
    # ---- train.py:431-522: upsample, blend with the labels of the same iteration, four losses (torch ops) --------------
    size = images.shape[-2:]
    center = F.interpolate(outputs["center"], size=size, mode="bilinear", align_corners=False)
    offset = F.interpolate(outputs["offset"], size=size, mode="bilinear", align_corners=False)
    old_center = F.interpolate(outputs_old["center"], size=size, mode="bilinear", align_corners=False)
    old_offset = F.interpolate(outputs_old["offset"], size=size, mode="bilinear", align_corners=False)

    def full(d):                                                   # the label tensors carry new-class channels only
        c = torch.zeros(bs, 20, *size, device=gpu)
        c[:, oc - 1:] = d["center"]
        return {"center": c, "offset": d["offset"], "weight": d["weight"]}

    res = L.blend_and_losses(center, offset, old_center, old_offset, out["old_w"], full(out["pseudo"]), full(out["refined"]), oc)
    loss = res["l_center"] + res["l_offset"]
    got_losses = torch.stack([res[k].detach() for k in ("center_old", "offset_old", "center_new", "offset_new")])
    assert torch.allclose(got_losses, out["losses"], rtol=1e-4, atol=1e-7), (got_losses, out["losses"])
    # ---- train.py:537 ---------------------------------------------------------------------------------------------------
    loss.backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    worst = 0.0
    for n, gw in want.items():
        assert params[n].grad is not None, n
        worst = max(worst, _rel_l2(params[n].grad, gw))
    print("MODULE-SURFACE phase 2: worst relative L2 gradient difference", worst)
    assert worst < 1e-3
    assert all(p.grad is None for n, p in params.items() if n.startswith(("body.", "head.", "cls.")))


def test_phase1_reference_call_sequence_with_autograd(gpu):
    from cl4wsis_amd.train import Trainer
    from oracle import wss as W
    crop = 65
    opts = p1_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    g = torch.Generator().manual_seed(5)
    images = torch.randn(3, 3, crop, crop, generator=g).to(gpu)
    l1h = torch.zeros(3, 20); l1h[0, 16] = 1; l1h[1, 15] = 1; l1h[1, 18] = 1; l1h[2, 19] = 1
    l1h = l1h.to(gpu)
    angle = 180

    torch.manual_seed(0)
    ta = Trainer(QuietLogger(), gpu, opts)
    _fill(ta, 41)
    la = ta._step_phase1((images, l1h), cur_epoch=0, flac_angle=angle)["losses"]
    torch.cuda.synchronize()
    named = list(ta.model.named_parameters()) + [("pl." + n, p) for n, p in ta.pseudolabeler.named_parameters()] + \
        [("pg." + n, p) for n, p in ta.peakgenerator.named_parameters()]
    want = {n: ta._grad_map[id(p)].clone() for n, p in named if id(p) in ta._grad_map}

    torch.manual_seed(0)
    tb = Trainer(QuietLogger(), gpu, opts)
    _fill(tb, 41)
    model, model_old, pl, pg = tb.model, tb.model_old, tb.pseudolabeler, tb.peakgenerator
    bs, old = 3, tb.old_classes
    model.train()
    # ---- train.py:253-255, 274 -------------------------------------------------------------------------------------------
    with torch.no_grad():
        outputs_old, features_old = model_old(images, interpolate=False)
    outputs, features = model(images, interpolate=False)
    assert outputs["seg"].requires_grad
    # ---- train.py:304-337 --------------------------------------------------------------------------------------------------
    pl.eval()
    feat_body = features["body"]
    int_masks = pl(feat_body).detach()
    pl.train()
    feat3 = torch.cat([feat_body, feat_body.flip(-1), W.rot90k(feat_body, angle)], dim=0)     # train.py:314-317
    int_masks_raw = pl(feat3)
    flac = W.flac_loss(int_masks_raw, bs, old, angle)
    int_masks_raw = int_masks_raw[:bs]
    pg.train()
    peak_logits, _ = pg(int_masks_raw)
    l_cam_new = W.bce_loss(int_masks_raw, l1h[:, old - 1:], reduction="mean")
    l_peak = F.binary_cross_entropy_with_logits(peak_logits[:, old - 1:], l1h[:, old - 1:])
    l_loc = W.l_loc(int_masks_raw, outputs_old["seg"], old)
    lde = tb.lde * W.lde(features["body"], features_old["body"])
    loss = l_cam_new + l_loc + flac + l_peak + lde
    for k, v in (("l_cam_new", l_cam_new), ("l_peak", l_peak), ("l_loc", l_loc), ("flac", flac), ("lde", lde)):
        ref = float(la[k])
        assert abs(float(v) - ref) <= 1e-3 * max(abs(ref), 1e-3), (k, float(v), ref)
    loss.backward()                                                 # train.py:537
    torch.cuda.synchronize()
    got = dict(model.named_parameters())
    got.update({"pl." + n: p for n, p in pl.named_parameters()})
    got.update({"pg." + n: p for n, p in pg.named_parameters()})
    errs = []
    for n, gw in want.items():
        if float(gw.abs().max()) == 0:
            continue                                                # cls / head receive no gradient before pseudo_ep
        assert got[n].grad is not None, n
        errs.append(_rel_l2(got[n].grad, gw))
    errs = torch.tensor(errs)
    print("MODULE-SURFACE phase 1: gradient relative L2 median %.2e max %.2e over %d tensors" % (errs.median(), errs.max(), len(errs)))
    # same kernels, same order on both routes except the accumulation order into res5's gradient (autograd sums the three
    # consumers in its own order) -> agreement far below the 1e-3 bar, with the BN+ReLU amplification of the tiny crop
    assert errs.median() < 1e-4 and errs.max() < 2e-2
