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
    peak_conf, peak_y, peak_x = L.peak_extract(cam.detach(), kernel=15, K=K)
    out.update(peak_conf=peak_conf, peak_y=peak_y, peak_x=peak_x)
    with torch.no_grad():
        seg_gt, old_w, soft, seg_up = L.seg_pseudo_gt(seg2b, l1h, old_classes, size, no_mask=no_mask)
    out.update(seg_gt=seg_gt, old_w=old_w, soft=soft)
    cls_label = l1h.numpy().copy()
    cls_label[:, :old_classes - 1] = 0
    g = L.gaussian(sigma)
    pc, po, pw = [], [], []
    for b in range(bs):
        pts = L.select_points(peak_conf, peak_y, peak_x, cls_label[b], b, pseudo_thresh)
        c, o, w, _ = L.pseudo_label_generation(seg_gt[b].numpy(), pts, cls_label[b], tot - 1, sigma, g)
        pc.append(torch.from_numpy(c)); po.append(torch.from_numpy(o)); pw.append(torch.from_numpy(w))
    pseudo = {"center": torch.stack(pc), "offset": torch.stack(po), "weight": torch.stack(pw)}
    out["pseudo"] = {k: v.clone() for k, v in pseudo.items()}
    refined = None
    if run_refine:
        l1h_new = l1h.clone()
        l1h_new[:, :old_classes - 1] = 0
        refined = L.refine_label_generation(seg_up.clone(), center.detach().clone(), offset.detach().clone(), l1h_new,
                                            seg_gt.clone(), refine_thresh, kernel, beta, sigma)
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
            adam_state=None, step=1, betas=(0.9, 0.999), eps=1e-8, **label_kw):
    """One phase-2 iteration on the CPU (train.py:241-541): returns losses, intermediates, gradients and applies Adam
    to `sd_new` in place (torch.optim.Adam arithmetic)."""
    bs = images.shape[0]
    size = tuple(images.shape[-2:])
    with torch.no_grad():
        out_old, _ = M.forward_full(images, sd_old, pooling=pooling, training=False)
        out2, aux = M.forward_seg(torch.cat([images, images.flip(-1)], 0), sd_new, pooling=pooling, training=False)
    feats = {k: v[:bs].detach().clone() for k, v in aux["features"].items()}
    names = trainable_names(sd_new)
    for k in names:
        sd_new[k].requires_grad_(True)
        sd_new[k].grad = None
    ins = M.instance_decoder(feats, sd_new, training=True, dropout_mask=dropout_mask)
    pred = M.instance_head(ins, sd_new, training=True)
    with torch.no_grad():
        int_masks = M.pseudolabeler(aux["body"][:bs], sd_pl)
    out = label_and_loss(pred["center"], pred["offset"], out_old["center"], out_old["offset"], out2["seg"], l1h, int_masks,
                         sd_pg, size, old_classes, **label_kw)
    out["loss"].backward()
    grads = {k: sd_new[k].grad.clone() for k in names}
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
