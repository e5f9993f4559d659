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
    mx = mx.max(torch.tensor([cutoff_low]).type_as(mx))
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
