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
