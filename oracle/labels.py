"""ORACLE (test infrastructure, not product code) — CPU restatement of the phase-2 label stage.

numpy / scipy / PyTorch-CPU restatement of the weakly-supervised label generation that
`Trainer.train` runs per batch in phase 2 (upstream train.py:418-522): CAM smoothing and peak
extraction, the flip-TTA semantic pseudo ground truth, BESTIE pseudo-label generation and
refined-label generation, and the weighted centre / offset losses.

Third-party arithmetic absent from the reference tree and restated here from its documented
behaviour (parity unpinned at that boundary): OpenCV
``connectedComponentsWithStats`` -> scipy.ndimage.label + exact integer statistics; components are
numbered in raster order of their first pixel, centroids are float64 means (x, y).

Each function cites the upstream file:line it follows.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage

MINIMUM_MASK_SIZE = 20   # modules/utils.py:14
MAXIMUM_NUM_INST = 5     # modules/utils.py:15

_S8 = np.ones((3, 3), dtype=np.int32)
_S4 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.int32)


def connected_components(mask, connectivity):
    """cv2.connectedComponentsWithStats(mask, connectivity) -> (n, labels, area[n], centroids[n,2] (x,y) f64).
    Label 0 is the background (all zero pixels), as in OpenCV."""
    lab, n = ndimage.label(mask != 0, structure=_S8 if connectivity == 8 else _S4)
    lab = lab.astype(np.int32)
    n += 1
    area = np.bincount(lab.ravel(), minlength=n).astype(np.int64)
    ys, xs = np.indices(mask.shape)
    sx = np.bincount(lab.ravel(), weights=xs.ravel().astype(np.float64), minlength=n)
    sy = np.bincount(lab.ravel(), weights=ys.ravel().astype(np.float64), minlength=n)
    with np.errstate(invalid="ignore", divide="ignore"):
        cent = np.stack([sx / area, sy / area], 1)
    return n, lab, area, cent


def gaussian(sigma=6):
    """modules/utils.py:49-59."""
    size = 6 * sigma + 3
    x = np.arange(0, size, 1, float)
    y = x[:, np.newaxis]
    x0 = y0 = 3 * sigma + 1
    return np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))


def center_map_gen(center_map, x, y, label, sigma, g):
    """modules/utils.py:84-119 — np.maximum paste of the clipped gaussian window (in place)."""
    _, height, width = center_map.shape
    if x < 0 or y < 0 or x >= width or y >= height:
        return center_map
    ul = int(np.round(x - 3 * sigma - 1)), int(np.round(y - 3 * sigma - 1))
    br = int(np.round(x + 3 * sigma + 2)), int(np.round(y + 3 * sigma + 2))
    c, d = max(0, -ul[0]), min(br[0], width) - ul[0]
    a, b = max(0, -ul[1]), min(br[1], height) - ul[1]
    cc, dd = max(0, ul[0]), min(br[0], width)
    aa, bb = max(0, ul[1]), min(br[1], height)
    center_map[label, aa:bb, cc:dd] = np.maximum(center_map[label, aa:bb, cc:dd], g[a:b, c:d])
    return center_map


def smoothing(heat, kernel=3):
    """wss/utils.py:28-32."""
    return F.avg_pool2d(heat, (kernel, kernel), stride=1, padding=(kernel - 1) // 2)


def peak_extract(heat, kernel=15, K=25):
    """wss/utils.py:3-25. Returns numpy (scores f32, ys i32, xs i32), each [B, C, K]."""
    B, C, H, W = heat.shape
    hmax = F.max_pool2d(heat, (kernel, kernel), stride=1, padding=(kernel - 1) // 2)
    peak = heat * (hmax == heat).float()
    scores, inds = torch.topk(peak.view(B, C, -1), K)
    inds = inds % (H * W)
    ys = (inds / W).int()
    xs = (inds % W).int()
    return scores.float().numpy(), ys.numpy(), xs.numpy()


def seg_pseudo_gt(seg2b, l1h, old_classes, out_size, no_mask=False):
    """train.py:263,438-449. seg2b: [2B, C, h, w] logits of [images ; flipped images]; l1h [B, C-1].
    Returns seg_gt i64 [B,H,W], old_pseudo_weight f32 [B,1,H,W], soft_seg_gt [B,C,H,W], seg_max_up [B,C,H,W]."""
    bs = seg2b.shape[0] // 2
    seg_max = (seg2b[:bs] + seg2b[bs:].flip(-1)) / 2
    up = F.interpolate(seg_max, size=out_size, mode="bilinear", align_corners=False)
    soft = up.softmax(dim=1)
    if no_mask:
        soft[:, 1:] *= l1h[:, :, None, None]
    else:
        soft[:, old_classes:] *= l1h[:, old_classes - 1:, None, None]
    seg_gt = torch.argmax(soft, 1)
    old_w = ((seg_gt < old_classes) & (seg_gt != 0))[:, None].float()
    seg_gt = seg_gt.clone()
    seg_gt[seg_gt < old_classes] = 0
    return seg_gt, old_w, soft, up


def select_points(peak_conf, peak_y, peak_x, cls_label_b, b, thresh):
    """train.py:452-458 — per labelled class the prefix of peaks with conf >= pseudo_thresh."""
    points = []
    for l in np.nonzero(cls_label_b)[0]:
        for conf, x, y in zip(peak_conf[b, l], peak_x[b, l], peak_y[b, l]):
            if conf < thresh:
                break
            points.append([x, y, l, conf])
    return points


def pseudo_label_generation(seg_map, point, cls_label, num_classes, sigma, g):
    """modules/utils.py:179-253 (list-of-points branch). seg_map [H,W] int; returns f32 center [C,H,W],
    offset [2,H,W] (channel 0 = y), weight [1,H,W], n_matched."""
    points = [[] for _ in range(num_classes)]
    for px, py, cls, conf in point:
        points[int(cls)].append((int(px), int(py)))
    H, W = seg_map.shape
    offset_map = np.zeros((2, H, W), dtype=np.float32)
    weight_map = np.zeros((1, H, W), dtype=np.float32)
    center_map = np.zeros((num_classes, H, W), dtype=np.float32)
    y_coord, x_coord = np.indices((H, W), dtype=np.float32)
    total = 0
    for cls in np.nonzero(cls_label)[0]:
        n, labels, area, cent = connected_components((seg_map == (cls + 1)).astype(np.uint8), 8)
        for k in range(1, n):
            if area[k] < MINIMUM_MASK_SIZE:
                continue
            cx, cy = int(cent[k][0]), int(cent[k][1])
            match = sum(1 for x, y in points[cls] if labels[y, x] == k)
            if match == 1:
                total += 1
                center_map = center_map_gen(center_map, cx, cy, cls, sigma, g)
                idx = np.where(labels == k)
                weight_map[0][idx] = 1
                offset_map[0][idx] = cy - y_coord[idx]
                offset_map[1][idx] = cx - x_coord[idx]
    return center_map, offset_map, weight_map, total


def find_instance_center(ctr_hmp, threshold, nms_kernel):
    """modules/utils.py:463-502 with top_k=None. ctr_hmp: [H,W] tensor. Returns LongTensor [K,2] (y,x), row-major."""
    t = F.threshold(ctr_hmp[None, None], threshold, -1)
    pooled = F.max_pool2d(t, kernel_size=nms_kernel, stride=1, padding=(nms_kernel - 1) // 2)
    t = torch.where(t != pooled, torch.full_like(t, -1), t)
    return torch.nonzero(t[0, 0] > 0, as_tuple=False)


def cluster_peaks(offset_map, fg, thresh=2.5, beta=5):
    """modules/utils.py:608-632 — 4-connected components of (|offset| < thresh) & fg whose area is within
    (21-beta, 21+beta); k runs from 0 so the background component is a candidate too. Returns int32 [n,2] (y,x)."""
    magnitude = np.sqrt(offset_map[1] ** 2 + offset_map[0] ** 2)
    weak = (magnitude < thresh) * fg
    n, labels, area, cent = connected_components(weak.astype(np.uint8), 4)
    peaks = [cent[k][::-1] for k in range(n) if 21 - beta < area[k] < 21 + beta]
    return np.int32(peaks)


def group_pixels(ctr, offsets):
    """modules/utils.py:505-542. ctr [K,2] (y,x) long; offsets [2,H,W]. Returns [H,W] ids starting at 1."""
    _, H, W = offsets.shape
    y = torch.arange(H, dtype=offsets.dtype, device=offsets.device).repeat(1, W, 1).transpose(1, 2)
    x = torch.arange(W, dtype=offsets.dtype, device=offsets.device).repeat(1, H, 1)
    loc = (torch.cat((y, x), 0) + offsets).reshape(2, H * W).transpose(1, 0)
    dist = torch.norm(ctr.unsqueeze(1) - loc.unsqueeze(0), dim=-1)
    return torch.argmin(dist, dim=0).reshape(H, W) + 1


def get_instance_segmentation(fg, c_center_map, offsets, threshold, nms_kernel, beta, ignore=True):
    """modules/utils.py:545-606 (top_k=None; ignore=True in training, --val_ignore (False) in validation). fg [H,W] bool; c_center_map [H,W] is modified in place
    (cluster centres are marked with 1.0, :583,592). Returns [H,W] long instance ids."""
    ctr = find_instance_center(c_center_map, threshold, nms_kernel)
    new_ctr = ctr
    if beta > 0:
        cl = cluster_peaks(offsets.cpu().numpy(), fg.cpu().numpy(), beta=beta)       # the reference's own .cpu() copies (:565)
        cl = np.int32([[cy, cx] for cy, cx in cl if c_center_map[cy, cx] > 0.05])
        cl = torch.from_numpy(cl).to(ctr.device).long()
        new_ctr = ctr.clone()
        if cl.size(0) > 0:
            if ctr.size(0) == 0:
                new_ctr = cl
                for cy, cx in cl:
                    c_center_map[cy, cx] = 1.0
            else:
                for c in cl:
                    if torch.norm(ctr.float() - c.float(), dim=-1).min() > 100:
                        new_ctr = torch.cat([new_ctr, c.unsqueeze(0)], 0)
                        c_center_map[c[0], c[1]] = 1.0
    if new_ctr.size(0) == 0:
        return torch.zeros_like(fg).long() if ignore else fg.long()
    return (fg * group_pixels(new_ctr, offsets)).long()


def refine_label_generation(seg_logits, center_map, offset_map, label, gt_seg_map, refine_thresh=0.3, kernel=41,
                            beta=3.0, sigma=6):
    """modules/utils.py:257-385 (top_k=None). seg_logits [B,C+1,H,W]; center_map [B,C,H,W]; offset_map [B,2,H,W];
    label [B,C]; gt_seg_map [B,H,W] long. Returns dict(center [B,C,H,W], offset [B,2,H,W], weight [B,1,H,W]) f32."""
    B, C, H, W = center_map.shape
    dev = center_map.device      # cpu for the oracle proper; on a GPU (bench.py's stock comparator) the tensor work stays on the
    # device and only masks / centroids cross to the host — exactly the reference's own split (modules/utils.py:277-325)
    prob = F.softmax(seg_logits, dim=1)
    prob[:, 1:] *= label[:, :, None, None]
    r_off = torch.zeros((B, 2, H, W), device=dev)
    r_w = torch.zeros((B, 1, H, W), device=dev)
    r_ctr = np.zeros((B, C, H, W), dtype=np.float32)
    yc, xc = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev),
                            indexing="ij")
    lab = label.cpu().numpy()
    g = gaussian(sigma)
    for b in range(B):
        for cls in np.nonzero(lab[b])[0]:
            fgm = (gt_seg_map[b] == (cls + 1))
            n, contours, area, cent = connected_components(fgm.cpu().numpy().astype(np.uint8), 8)
            for k in range(1, n):
                if area[k] < MINIMUM_MASK_SIZE:
                    continue
                cx, cy = int(cent[k][0]), int(cent[k][1])
                cmask = torch.from_numpy(contours == k).to(dev)
                c_ctr = center_map[b, cls] * cmask
                ins = get_instance_segmentation(cmask, c_ctr, offset_map[b], refine_thresh, kernel, beta)
                n_ins = int(ins.max())
                if n_ins > MAXIMUM_NUM_INST:
                    continue
                for i in range(1, n_ins + 1):
                    m = ins == i
                    if m.sum() > 0:
                        idx = torch.where(m)
                        pmax = c_ctr[idx].argmax()
                        seg_score = prob[b, cls + 1][idx].mean().item()
                        py, px = idx[0][pmax].item(), idx[1][pmax].item()
                        center_score = c_ctr[py, px].item()
                        if center_score < refine_thresh:
                            py, px = cy, cx
                            conf = seg_score
                        else:
                            conf = center_score * seg_score
                        conf = max(0, min(conf, 1))
                        r_ctr[b] = center_map_gen(r_ctr[b], px, py, cls, sigma, g)
                        r_w[b, 0][idx] = conf
                        r_off[b, 0][idx] = py - yc[idx]
                        r_off[b, 1][idx] = px - xc[idx]
    return {"center": torch.from_numpy(r_ctr).to(dev), "offset": r_off, "weight": r_w}


def weighted_mse(out, target, weight):
    """utils/loss.py:259-275."""
    loss = ((out - target) ** 2) * weight
    return loss.sum() / (weight > 0).float().sum() if weight.sum() > 0 else loss.sum() * 0


def weighted_l1(out, target, weight):
    """utils/loss.py:228-243."""
    loss = (out - target).abs() * weight
    return loss.sum() / (weight > 0).float().sum() if weight.sum() > 0 else loss.sum() * 0


def blend_and_losses(center, offset, old_center, old_offset, old_w, pseudo, refined, old_classes,
                     center_weight=200.0, offset_weight=0.01):
    """train.py:503-522. center/offset are the new model's full-resolution maps (autograd-connected);
    pseudo / refined are dicts(center, offset, weight); old_classes includes background."""
    oc = old_classes - 1
    p_center, p_offset, pw = pseudo["center"].clone(), pseudo["offset"], pseudo["weight"]
    pw_sum = torch.maximum(old_w, pw)
    if refined is not None:
        p_center[:, oc:] = pw * p_center[:, oc:] + (1 - pw) * refined["center"][:, oc:]
        p_offset = pw_sum * p_offset + (1 - pw_sum) * refined["offset"]
        pw = torch.maximum(pw, refined["weight"])
    c1 = 0.5 * weighted_mse(center[:, :oc], old_center, old_w) * center_weight
    o1 = 0.5 * weighted_l1(offset, old_offset, old_w) * offset_weight
    c2 = 0.5 * weighted_mse(center[:, oc:], p_center[:, oc:], pw) * center_weight
    o2 = 0.5 * weighted_l1(offset, p_offset, pw) * offset_weight
    return {"center_old": c1, "offset_old": o1, "center_new": c2, "offset_new": o2,
            "l_center": c1 + c2, "l_offset": o1 + o2,
            "target_center": p_center, "target_offset": p_offset, "target_weight": pw}


VAL_MINIMUM_MASK_SIZE = 50     # dataset/utils.py:147 (the validation copy; training uses 20, modules/utils.py:14)


def get_ins_map(seg, center, offset, target_size=None, val_flip=False, val_thresh=0.1, val_kernel=41, beta=3.0, val_ignore=False,
                cls_label=None):
    """dataset/utils.py:795-892 — network outputs (already at `target_size`) -> semantic map and instance predictions.
    seg [B,C+1,H,W] logits, center [B,C,H,W], offset [B,2,H,W] with B = 2 (image, flipped image) when val_flip else 1.
    Returns (seg_map [H,W] long, pred_label [R] int, pred_mask [R,H,W] bool, pred_score [R] float); one all-zero prediction
    when nothing is found (:882-885)."""
    seg_prob = torch.softmax(seg, 1)
    offset_map = offset[0].clone()
    if val_flip:
        seg_prob = (seg_prob[0] + seg_prob[1].flip(-1)) / 2.
        center_map = (center[0] + center[1].flip(-1)) / 2.
    else:
        seg_prob, center_map = seg_prob[0], center[0].clone()
    if cls_label is not None:                                   # --val_clean
        seg_prob = seg_prob.clone()
        seg_prob[1:] *= cls_label[0, :, None, None]
    seg_map = torch.argmax(seg_prob, 0)
    labels, masks, scores = [], [], []
    for cls in (torch.unique(seg_map).numpy() - 1):
        if cls < 0:
            continue
        n, contours, area, _ = connected_components((seg_map == cls + 1).numpy().astype(np.uint8), 8)
        for k in range(1, n):
            if area[k] < VAL_MINIMUM_MASK_SIZE:
                continue
            cmask = torch.from_numpy(contours == k)
            roi = center_map[cls] * cmask
            ins = get_instance_segmentation(cmask, roi, offset_map, val_thresh, val_kernel, beta, ignore=val_ignore)
            for i in range(1, int(ins.max()) + 1):
                m = ins == i
                if m.sum() > 0:
                    idx = torch.where(m)
                    seg_score = seg_prob[cls + 1][idx].mean().item()
                    center_score = roi[idx].max().item()
                    if center_score >= 1:
                        center_score = seg_score
                    labels.append(int(cls))
                    masks.append(m.numpy())
                    scores.append(center_score * seg_score)
    if not labels:
        H, W = seg_map.shape
        return seg_map, np.zeros(1, dtype=np.int64), np.zeros((1, H, W), dtype=bool), np.zeros(1)
    return seg_map, np.array(labels), np.stack(masks, 0), np.array(scores)


def mask_iou(a, b):
    """chainercv.utils.mask.mask_iou (absent third-party, restated): pairwise IoU of bool masks [N,H,W] x [K,H,W]."""
    a, b = a.reshape(len(a), -1).astype(np.float64), b.reshape(len(b), -1).astype(np.float64)
    inter = a @ b.T
    union = a.sum(1)[:, None] + b.sum(1)[None, :] - inter
    return inter / np.maximum(union, 1e-12) * (union > 0)


def calc_detection_voc_ap(prec, rec, use_07_metric=False):
    """chainercv.evaluations.calc_detection_voc_ap (absent third-party, restated from its documented algorithm)."""
    ap = np.empty(len(prec))
    for l in range(len(prec)):
        if prec[l] is None or rec[l] is None:
            ap[l] = np.nan
            continue
        if use_07_metric:
            ap[l] = 0
            for t in np.arange(0., 1.1, 0.1):
                p = 0 if np.sum(rec[l] >= t) == 0 else np.max(np.nan_to_num(prec[l])[rec[l] >= t])
                ap[l] += p / 11
        else:
            mpre = np.concatenate(([0], np.nan_to_num(prec[l]), [0]))
            mrec = np.concatenate(([0], rec[l], [1]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            i = np.where(mrec[1:] != mrec[:-1])[0]
            ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return ap
