"""ORACLE (test infrastructure, not product code) — the phase-2 per-batch step restated on the CPU.

`label_and_loss` follows upstream train.py:418-522 (centre / offset label generation and the four
weighted losses); `p2_step` chains the network restatement (oracle.model) in front of it the way
train.py:253-272 does and back-propagates with PyTorch-CPU autograd.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import labels as L
from . import model as M


def label_and_loss(center_lo, offset_lo, old_center_lo, old_offset_lo, seg2b, l1h, int_masks, pg_sd, size, old_classes,
                   pseudo_thresh=0.7, refine_thresh=0.3, kernel=41, beta=3.0, sigma=6, run_refine=True, no_mask=False,
                   K=25):
    """center_lo/offset_lo may require grad. Returns a dict with every intermediate named in SURVEY.md §8(c)."""
    H, W = size
    bs = l1h.shape[0]
    tot = seg2b.shape[1]
    out = {}
    # train.py:425-431 PeakGenerator (eval) -> smoothing -> upsample
    with torch.no_grad():
        _, cam = M.peak_generator_eval(int_masks, pg_sd, l1h, old_classes - 1)
        cam = L.smoothing(cam)
        out["cam_lo"] = cam
        cam = F.interpolate(cam, size=size, mode="bilinear", align_corners=False)
    center = F.interpolate(center_lo, size=size, mode="bilinear", align_corners=False)
    offset = F.interpolate(offset_lo, size=size, mode="bilinear", align_corners=False)
    dev = center_lo.device      # cpu for the oracle proper; "cuda" only in bench.py's stock-ops comparator leg, where — like
    # the reference on a GPU — everything below that is numpy / OpenCV-style host code works on .cpu() copies
    peak_conf, peak_y, peak_x = L.peak_extract(cam.detach().float().cpu(), kernel=15, K=K)
    out.update(peak_conf=peak_conf, peak_y=peak_y, peak_x=peak_x)
    with torch.no_grad():
        seg_gt, old_w, soft, seg_up = L.seg_pseudo_gt(seg2b, l1h, old_classes, size, no_mask=no_mask)
    out.update(seg_gt=seg_gt, old_w=old_w, soft=soft)
    cls_label = l1h.cpu().numpy().copy()
    cls_label[:, :old_classes - 1] = 0
    g = L.gaussian(sigma)
    pc, po, pw = [], [], []
    for b in range(bs):
        pts = L.select_points(peak_conf, peak_y, peak_x, cls_label[b], b, pseudo_thresh)
        c, o, w, _ = L.pseudo_label_generation(seg_gt[b].cpu().numpy(), pts, cls_label[b], tot - 1, sigma, g)
        pc.append(torch.from_numpy(c)); po.append(torch.from_numpy(o)); pw.append(torch.from_numpy(w))
    pseudo = {"center": torch.stack(pc).to(dev), "offset": torch.stack(po).to(dev), "weight": torch.stack(pw).to(dev)}
    out["pseudo"] = {k: v.clone() for k, v in pseudo.items()}
    refined = None
    if run_refine:
        l1h_new = l1h.clone()
        l1h_new[:, :old_classes - 1] = 0
        own = lambda t: t.detach().float().clone()
        refined = L.refine_label_generation(own(seg_up), own(center), own(offset), l1h_new, seg_gt.clone(),
                                            refine_thresh, kernel, beta, sigma)
        out["refined"] = refined
    old_center = F.interpolate(old_center_lo, size=size, mode="bilinear", align_corners=False)
    old_offset = F.interpolate(old_offset_lo, size=size, mode="bilinear", align_corners=False)
    out.update(L.blend_and_losses(center, offset, old_center, old_offset, old_w, pseudo, refined, old_classes))
    out["loss"] = out["l_center"] + out["l_offset"]
    return out


TRAINABLE_PREFIXES = ("decoder.instance_decoder.", "instance_head.")


def trainable_names(sd):
    """Phase-2 trainable tensors (train.py:150-161 with --freeze --freeze_seg): instance decoder + instance head."""
    return [k for k in sd if k.startswith(TRAINABLE_PREFIXES) and sd[k].is_floating_point()
            and not k.endswith(("running_mean", "running_var"))]


def p2_step(sd_new, sd_old, sd_pl, sd_pg, images, l1h, old_classes, pooling=32, dropout_mask=None, lr=5e-4,
            adam_state=None, step=1, betas=(0.9, 0.999), eps=1e-8, autocast=False, grad_sync=None, **label_kw):
    """One phase-2 iteration on the CPU (train.py:241-541): returns losses, intermediates, gradients and applies Adam
    to `sd_new` in place (torch.optim.Adam arithmetic). `grad_sync(grads)` (optional) is called between backward and Adam.
    `autocast` (timing legs only): run the networks under
    torch.autocast(float16) like the reference's amp.autocast block (train.py:252); outputs are widened to fp32 before the
    label stage."""
    bs = images.shape[0]
    size = tuple(images.shape[-2:])
    names = trainable_names(sd_new)
    with torch.autocast(device_type=images.device.type, dtype=torch.float16, enabled=autocast):
        with torch.no_grad():
            out_old, _ = M.forward_full(images, sd_old, pooling=pooling, training=False)
            out2, aux = M.forward_seg(torch.cat([images, images.flip(-1)], 0), sd_new, pooling=pooling, training=False)
        feats = {k: v[:bs].detach().clone() for k, v in aux["features"].items()}
        for k in names:
            sd_new[k].requires_grad_(True)
            sd_new[k].grad = None
        ins = M.instance_decoder(feats, sd_new, training=True, dropout_mask=dropout_mask)
        pred = M.instance_head(ins, sd_new, training=True)
        with torch.no_grad():
            int_masks = M.pseudolabeler(aux["body"][:bs], sd_pl)
    if autocast:
        out_old = {k: v.float() for k, v in out_old.items()}
        out2 = {k: v.float() for k, v in out2.items()}
        pred = {k: v.float() for k, v in pred.items()}
        int_masks = int_masks.float()
    out = label_and_loss(pred["center"], pred["offset"], out_old["center"], out_old["offset"], out2["seg"], l1h, int_masks,
                         sd_pg, size, old_classes, **label_kw)
    out["loss"].backward()
    grads = {k: sd_new[k].grad.clone() for k in names}
    if grad_sync is not None:       # data-parallel legs (bench.py stock_rocm at N > 1): what DDP's bucketed all-reduce does between
        grad_sync(grads)            # backward and optimizer.step() — averages the gradients across ranks, in place
    if adam_state is None:
        adam_state = {k: (torch.zeros_like(sd_new[k]), torch.zeros_like(sd_new[k])) for k in names}
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    with torch.no_grad():
        for k in names:
            m, v = adam_state[k]
            g = grads[k]
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
            sd_new[k].addcdiv_(m, denom, value=-lr / bc1)
    for k in names:
        sd_new[k].requires_grad_(False)
    out.update(pred_center=pred["center"].detach(), pred_offset=pred["offset"].detach(), grads=grads,
               adam_state=adam_state, seg2b=out2["seg"], int_masks=int_masks,
               old_center=out_old["center"], old_offset=out_old["offset"])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Phase 1 (train.py:241-566 with --phase 1: body / head / classifier / PseudoLabeler / PeakGenerator train with SGD)
# ---------------------------------------------------------------------------------------------------------------------
DENORM_MEAN = (0.485, 0.456, 0.4069)      # utils/utils.py:23 (sic)
DENORM_STD = (0.229, 0.224, 0.225)


def _leaf(sd, keys):
    for k in keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None


def p1_param_groups(sd_new, sd_pl, sd_pg, lr, lr_head, lr_pseudo, weight_decay):
    """The parameter groups of train.py:144-177 for --phase 1 (branch 'none'): body, head, cls, PseudoLabeler,
    PeakGenerator — (dict, key, lr, weight_decay) per trainable tensor."""
    is_par = lambda k: k.endswith(".weight") or k.endswith(".bias")
    out = []
    for k in sd_new:
        if not is_par(k):
            continue
        if k.startswith("body."):
            out.append((sd_new, k, lr, weight_decay))
        elif k.startswith("head.") or k.startswith("cls."):
            out.append((sd_new, k, lr * lr_head, weight_decay))
    out += [(sd_pl, k, lr_pseudo, weight_decay) for k in sd_pl if is_par(k)]
    out += [(sd_pg, k, lr_pseudo, weight_decay) for k in sd_pg if is_par(k)]
    return out


def p1_step(sd_new, sd_old, sd_pl, sd_pg, images, l1h, old_classes, cur_epoch=0, pseudo_epoch=5, flac_angle=90,
            randrop_neg=None, lr=1e-3, lr_head=10.0, lr_pseudo=0.01, weight_decay=1e-4, loss_de=1.0, alpha=0.5,
            l_seg_weight=1.0, icarl_bkg=-1.0, momentum_state=None, pooling=32, structure=None, wrn_keeps=None):
    """One phase-1 iteration on the CPU. Applies torch.optim.SGD(momentum 0.9, nesterov) arithmetic in place to the three
    state dicts and returns the loss terms, the gradients and the momentum buffers."""
    from . import wss as W
    bs = images.shape[0]
    tot = sd_pl["cls.weight"].shape[0]
    groups = p1_param_groups(sd_new, sd_pl, sd_pg, lr, lr_head, lr_pseudo, weight_decay)
    for d, k, _, _ in groups:
        d[k].requires_grad_(True)
        d[k].grad = None
    with torch.no_grad():
        outs_old, os_ = M.body_any(images, sd_old, structure=structure)      # ResNet-101, or WiderResNet-38 for the coco-voc task
        old_seg = M.incremental_classifier(M.deeplab_head(outs_old[-1], sd_old, pooling=pooling, output_stride=os_), sd_old)
        res5_old = outs_old[-1]
    outs, os_ = M.body_any(images, sd_new, structure=structure, training=True, keeps=wrn_keeps)
    res5 = outs[-1]
    seg = M.incremental_classifier(M.deeplab_head(res5, sd_new, pooling=pooling, training=True, output_stride=os_), sd_new)
    with torch.no_grad():
        int_masks = M.pseudolabeler(res5, sd_pl)                              # eval mode, detached (train.py:304-311)
    feat3 = torch.cat([res5, res5.flip(-1), W.rot90k(res5, flac_angle)], 0)   # train.py:314-317
    A = M.pseudolabeler(feat3, sd_pl, training=True)
    flac = W.flac_loss(A, bs, old_classes, flac_angle)
    raw = A[:bs]
    lab_new = l1h[:, old_classes - 1:]
    l_peak, _ = W.peakgen_train_loss(raw, sd_pg["extra_conv4.weight"], sd_pg["extra_conv4.bias"], lab_new)
    l_cam_new = W.bce_loss(raw, lab_new, reduction="mean")
    l_loc = W.l_loc(raw, old_seg, old_classes)
    l_cam_int = l_cam_new + l_loc + flac + l_peak
    lde = loss_de * W.lde(res5, res5_old)
    l_seg = torch.zeros((), dtype=images.dtype)
    l_cls = torch.zeros((), dtype=images.dtype)
    l_rand = torch.zeros((), dtype=images.dtype)
    if cur_epoch >= pseudo_epoch:
        soft = int_masks.softmax(dim=1)
        img_raw = images * torch.tensor(DENORM_STD, dtype=images.dtype)[None, :, None, None] + \
            torch.tensor(DENORM_MEAN, dtype=images.dtype)[None, :, None, None]
        im = F.interpolate(img_raw, int_masks.shape[-2:], mode="bilinear", align_corners=True)
        soft = W.pamr(im, soft.detach())
        soft[:, 1:] *= l1h[:, :, None, None]
        pseudo_gt = W.pseudo_gtmask(soft).detach()
        l_seg = W.l_seg(seg, old_seg, int_masks.clone(), l1h, old_classes, alpha=alpha, icarl_bkg=icarl_bkg, weight=l_seg_weight)
        l_cls = W.balanced_mask_loss_ce(raw, pseudo_gt, l1h)
        if randrop_neg is not None:
            ref = torch.sigmoid(int_masks).clone()
            ref[:, 1:] *= l1h[:, :, None, None]
            live = ref[:, old_classes:].max(dim=1)[0] > 0.5
            if live.sum() > 0:
                t = randrop_neg.long().clone()
                t[~live] = 255
                l_rand = W.bce_ignore_mean_selected(raw, t)
    loss = l_seg + l_cls + l_cam_int + l_rand + lde
    loss.backward()
    if momentum_state is None:
        momentum_state = {}
    grads = {}
    with torch.no_grad():
        for d, k, g_lr, wd in groups:
            gr = d[k].grad if d[k].grad is not None else torch.zeros_like(d[k])   # `0 * outputs['seg']...` (train.py:365)
            g = gr + wd * d[k]
            key = (id(d), k)
            buf = momentum_state.get(key)
            buf = g.clone() if buf is None else buf.mul_(0.9).add_(g)
            momentum_state[key] = buf
            grads[key] = gr.clone()
            d[k].sub_(g_lr * (g + 0.9 * buf))
    for d, k, _, _ in groups:
        d[k].requires_grad_(False)
    return dict(loss=loss.detach(), l_cam_new=l_cam_new.detach(), l_loc=l_loc.detach(), flac=flac.detach(), l_peak=l_peak.detach(),
                lde=lde.detach(), l_seg=l_seg.detach(), l_cls=l_cls.detach(), l_rand=l_rand.detach(), grads=grads,
                momentum_state=momentum_state, seg=seg.detach(), int_masks=int_masks, A=A.detach(), res5=res5.detach())


# ---------------------------------------------------------------------------------------------------------------------
# Step 0 (supervised: train.py:241-300 with --step 0 --bce, Adam; BASELINE config 1)
# ---------------------------------------------------------------------------------------------------------------------
def s0_step(sd, images, seg_maps, center_maps, offset_maps, weights, dropout_mask=None, lr=5e-5, pooling=32,
            structure=None, adam_state=None, step=1, betas=(0.9, 0.999), eps=1e-8, wrn_keeps=None):
    """One supervised iteration: the whole network (body, DeepLabV3, classifier, instance decoder + heads) in training mode,
    loss = BCE-with-ignore(seg) + 200 * Weighted_MSE(center) + 0.01 * Weighted_L1(offset) on the upsampled outputs
    (train.py:279-300, utils/loss.py:62-85,228-275), torch.optim.Adam arithmetic (lr_head = 1 at step 0, argparser.py:32)."""
    from . import wss as W
    size = tuple(images.shape[-2:])
    names = [k for k in sd if (k.endswith(".weight") or k.endswith(".bias")) and sd[k].is_floating_point()]
    for k in names:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    outs, os_ = M.body_any(images, sd, structure=structure, training=True, keeps=wrn_keeps)
    feats = {f"res{i + 1}": o for i, o in enumerate(outs)}
    seg = M.incremental_classifier(M.deeplab_head(outs[-1], sd, pooling=pooling, training=True, output_stride=os_), sd)
    ins = M.instance_decoder(feats, sd, training=True, dropout_mask=dropout_mask)
    pred = M.instance_head(ins, sd, training=True)
    up = lambda t: F.interpolate(t, size=size, mode="bilinear", align_corners=False)
    seg_loss = W.bce_ignore_mean(up(seg), seg_maps)
    center_loss = L.weighted_mse(up(pred["center"]), center_maps, weights) * 200
    offset_loss = L.weighted_l1(up(pred["offset"]), offset_maps, weights) * 0.01
    loss = seg_loss + center_loss + offset_loss
    loss.backward()
    if adam_state is None:
        adam_state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    grads = {}
    with torch.no_grad():
        for k in names:
            g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
            grads[k] = g.clone()
            m, v = adam_state[k]
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            sd[k].addcdiv_(m, (v.sqrt() / (bc2 ** 0.5)).add_(eps), value=-lr / bc1)
    for k in names:
        sd[k].requires_grad_(False)
    return dict(loss=loss.detach(), seg_loss=seg_loss.detach(), center_loss=center_loss.detach(), offset_loss=offset_loss.detach(),
                grads=grads, adam_state=adam_state, seg=seg.detach(), center=pred["center"].detach(), offset=pred["offset"].detach())
