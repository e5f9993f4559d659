"""ORACLE (test infrastructure, not product code) — CPU restatement of the weakly-supervised loss stack of phase 1 /
step 0: PAMR affinity refinement, nGWP + focal CAM classification loss, pseudo ground-truth masks, class-balanced CE,
one-hot BCE with ignore index. PyTorch-CPU fp32; every function cites the upstream file:line it follows. Pinned by
tests/golden/wss_sXX.npz (outputs of the reference's own functions, tests/golden/make_golden.py)."""
import torch
import torch.nn.functional as F

_TAPS8 = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]      # wss/modules.py:31-40
_TAPS9 = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1)]  # wss/modules.py:93-103


def _shifted(x, taps, dilations):
    """LocalAffinityCopy / LocalStDev gather (wss/modules.py:47-62,65-112): replicate-pad by d, take the value at
    (y + dy*d, x + dx*d) for every tap of every dilation. x: [B,K,H,W] -> [B,K,len(taps)*len(dilations),H,W]."""
    B, K, H, W = x.shape
    out = []
    for d in dilations:
        xp = F.pad(x, [d] * 4, mode="replicate")
        for dy, dx in taps:
            out.append(xp[:, :, d + dy * d:d + dy * d + H, d + dx * d:d + dx * d + W])
    return torch.stack(out, 2)


def pamr(x, mask, num_iter=10, dilations=(1, 2, 4, 8, 12)):
    """PAMR.forward — wss/modules.py:133-152. x: image [B,3,H,W]; mask [B,C,h,w] -> [B,C,H,W]."""
    mask = F.interpolate(mask, size=x.shape[-2:], mode="bilinear", align_corners=True)
    x_std = _shifted(x, _TAPS9, dilations).std(2, keepdim=True)                  # LocalStDev, unbiased
    x_aff = torch.abs(x[:, :, None] - _shifted(x, _TAPS8, dilations))            # LocalAffinityAbs: |x_c - x_n|
    a = (-x_aff / (1e-8 + 0.1 * x_std)).mean(1, keepdim=True)
    a = F.softmax(a, 2)
    for _ in range(num_iter):
        m = _shifted(mask, _TAPS8, dilations)
        mask = (m * a).sum(2)
    return mask


def ngwp_focal(outputs, lam=1e-2):
    """utils/wss_loss.py:7-22."""
    bs, c, h, w = outputs.shape
    m = F.softmax(outputs, dim=1).reshape(bs, c, -1)
    x = outputs.reshape(bs, c, -1)
    y = (x * m).sum(-1) / (1.0 + m.sum(-1))
    return y + torch.pow(1 - m.mean(-1), 3) * torch.log(lam + m.mean(-1))


def bce_loss(outputs, labels, reduction="mean"):
    """utils/wss_loss.py:36-53 (mode='ngwp')."""
    y = ngwp_focal(outputs)[:, -labels.shape[1]:]
    if reduction == "sum":
        return F.binary_cross_entropy_with_logits(y, labels, reduction="none").sum(dim=1).mean()
    return F.binary_cross_entropy_with_logits(y, labels)


def binarize(x):
    """utils/wss_loss.py:188-190."""
    return (x >= x.max(dim=1, keepdim=True)[0]).type_as(x)


def pseudo_gtmask(mask, cutoff_top=0.6, cutoff_bkg=0.7, cutoff_low=0.2):
    """wss/single_stage.py:18-40 (ambiguous=True; called with 0.6 / 0.7 / 0.2 at train.py:384-385)."""
    bs, c, h, w = mask.shape
    m = mask.reshape(bs, c, -1)
    mx = m.max(-1, keepdim=True)[0].clone()
    mx[:, :1] *= cutoff_bkg
    mx[:, 1:] *= cutoff_top
    mx = mx.max(torch.tensor([cutoff_low]).type_as(mx))   # float32 constant widened like the reference does
    gt = (m > mx).type_as(m)
    amb = (gt.sum(1, keepdim=True) > 1).type_as(m)
    return ((1 - amb) * gt).reshape(bs, c, h, w)


def balanced_mask_loss_ce(mask, pseudo_gt, gt_labels, ignore_index=255):
    """wss/single_stage.py:43-80."""
    mask = F.interpolate(mask, size=pseudo_gt.shape[-2:], mode="bilinear", align_corners=True)
    mask_gt = torch.argmax(pseudo_gt, 1)
    mask_gt[pseudo_gt.sum(1) < 1.] = ignore_index
    bs, c, h, w = pseudo_gt.shape
    n = pseudo_gt.reshape(bs, c, -1).sum(-1)
    tot = n.sum(-1, keepdim=True)
    cw = (tot - n) / (1 + tot)
    pw = (pseudo_gt * cw[:, :, None, None]).sum(1).reshape(bs, -1)
    loss = F.cross_entropy(mask, mask_gt, ignore_index=ignore_index, reduction="none").reshape(bs, -1)
    gt_num = gt_labels.sum(-1).type_as(loss) + 1
    n = n.clone()
    n[:, 1:] *= gt_labels
    ps_num = (n > 0).type_as(loss).sum(-1)
    bw = (gt_num == ps_num).type_as(loss)
    return (bw * (pw * loss).mean(-1)).mean()


def bce_ignore_mean(inputs, targets, ignore_index=255):
    """BCEWithLogitsLossWithIgnoreIndex(reduction='none')(...).mean() — utils/loss.py:62-85 with train.py:295-299."""
    n_cl = inputs.shape[1]
    lab = torch.where(targets != ignore_index, targets, torch.tensor(n_cl))
    t = F.one_hot(lab, n_cl + 1).float().permute(0, 3, 1, 2)[:, :n_cl]
    loss = F.binary_cross_entropy_with_logits(inputs, t, reduction="none").sum(dim=1)
    return (loss * t.sum(dim=1)).mean()


# ---------------------------------------------------------------------------------------------------------------------
# Phase-1 training-path pieces
# ---------------------------------------------------------------------------------------------------------------------
def pam(x, alpha=0.7):
    """PAM.forward — wss/modules.py:354-368 (out of place)."""
    x = F.relu(x)
    peak = F.adaptive_max_pool2d(x, 1)
    return torch.where(x < peak * alpha, torch.zeros_like(x), x)


def peakgen_train_loss(int_masks_raw, weight, bias, labels_new, alpha=0.7):
    """PeakGenerator.forward (training) + the phase-1 peak loss: wss/modules.py:390-423 with train.py:334-336,344-345.
    int_masks_raw [b,C,h,w]; the generator sees the last `nc` channels; weight [nc,nc,1,1], bias [nc]; labels_new [b,nc].
    Returns (loss, logits [b,nc])."""
    nc = weight.shape[0]
    x = pam(int_masks_raw[:, -nc:], alpha)
    x = F.conv2d(x, weight, bias)
    m = F.softmax(x, dim=1).flatten(2)
    logit = (x.flatten(2) * m).sum(-1) / (1.0 + m.sum(-1))
    return F.binary_cross_entropy_with_logits(logit, labels_new), logit


def rot90k(t, angle):
    """torchvision.transforms.functional.rotate for multiples of 90 degrees on square maps = torch.rot90."""
    return torch.rot90(t, int(round(angle / 90)) % 4, dims=(-2, -1))


def flac_loss(A, bs, old_classes, angle):
    """FLAC consistency loss — train.py:320-331. A [3*bs, C, h, h]: PseudoLabeler logits on [x, flip(x), rot(x, angle)]."""
    a_rot = torch.sigmoid(torch.mean(A[bs * 2: bs * 3, old_classes:], dim=1))
    a_rot_back = rot90k(a_rot.detach().clone(), -angle)
    a_flip = torch.sigmoid(torch.mean(A[bs: bs * 2, old_classes:].flip(-1), dim=1))
    a_ori = torch.sigmoid(torch.mean(A[:bs, old_classes:], dim=1))
    tgt = torch.maximum(torch.maximum(a_ori, a_flip).detach(), a_rot_back).detach()
    tgt_rot = rot90k(tgt.clone(), angle).detach()
    mse = torch.nn.MSELoss()
    return (mse(a_ori, tgt) + mse(a_flip, tgt) + mse(a_rot, tgt_rot)) / 3


def l_loc(int_masks_raw, old_seg, old_classes, labels=None):
    """Localisation prior — train.py:346-354: BCE(int_masks_raw[:, :old], sigmoid(old seg)), optional --no_mask masking."""
    target = torch.sigmoid(old_seg.detach())
    if labels is not None:
        target = target.clone()
        target[:, 1:] *= labels[:, :old_classes - 1, None, None]
    return F.binary_cross_entropy_with_logits(int_masks_raw[:, :old_classes], target, reduction="mean")


def lde(feat, feat_old):
    """Feature distillation — train.py:362-363 (nn.MSELoss on res5)."""
    return F.mse_loss(feat, feat_old)


def l_seg(outputs_seg, old_seg, int_masks, labels, old_classes, alpha=0.5, icarl_bkg=-1.0, weight=1.0):
    """Segmentation loss against the blended pseudo map — train.py:374-409 (without the PAMR branch, which only feeds
    `pseudo_gt_seg` / l_cls). outputs_seg [b,tot,h,w], old_seg [b,old,h,w], int_masks [b,tot,h,w] (detached logits)."""
    bs, tot = outputs_seg.shape[:2]
    orig = int_masks.softmax(dim=1)
    orig[:, 1:] *= labels[:, :, None, None]
    lx = alpha * binarize(orig) + (1 - alpha) * orig
    px = lx.view(bs, tot, -1).sum(dim=-1)
    bw = torch.eq(px[:, old_classes:] > 0, labels[:, old_classes - 1:].bool())
    bw = (bw.sum(dim=1) == (tot - old_classes)).float()
    target_old = torch.sigmoid(old_seg.detach())
    pseudo = torch.cat((target_old, lx[:, old_classes:]), dim=1)
    if icarl_bkg == -1:
        pseudo[:, 0] = torch.min(pseudo[:, 0], lx[:, 0])
    else:
        pseudo[:, 0] = (1 - icarl_bkg) * pseudo[:, 0] + icarl_bkg * lx[:, 0]
    l = F.binary_cross_entropy_with_logits(outputs_seg, pseudo, reduction="none").sum(dim=1)
    l = l.view(bs, -1).mean(dim=-1)
    return weight * (bw * l).sum() / (bw.sum() + 1e-5)


def bce_ignore_mean_selected(inputs, targets, ignore_index=255):
    """BCEWithLogitsLossWithIgnoreIndex(reduction='mean') — utils/wss_loss.py:250-271 as used by RandropLoss :246-248: the
    one-hot weighted BCE (only the target class contributes), averaged over the non-ignored pixels."""
    n_cl = inputs.shape[1]
    lab = torch.where(targets != ignore_index, targets, torch.tensor(n_cl))
    t = F.one_hot(lab, n_cl + 1).float().permute(0, 3, 1, 2)[:, :n_cl]
    loss = (F.binary_cross_entropy_with_logits(inputs, t, reduction="none") * (t == 1).float()).sum(dim=1)
    return torch.masked_select(loss, t.sum(dim=1) != 0).mean()
