"""Synthetic inputs for benchmarks and smoke runs (there is no dataset on the GPU box).

`synthetic_batch` is the e2e-random batch of SURVEY.md §8(d) (images ~ N(0,1), 1-2 new-class labels per image);
`structured_label_inputs` builds the *structured-labels* inputs of the same section: per image 1-3 blobs per labelled new
class in the semantic logits, a CAM bump at each blob, centre / offset maps consistent with the blobs (plus noise), so that
connected components, peaks, NMS centres, offset clusters and the refine branches do representative work. The label-stage
tests and the golden generator draw from the same generator (oracle/synth.py re-exports it as `label_stage_inputs`).
"""
import numpy as np
import torch


def synthetic_batch(b, seed, device, crop=513):
    """images ~ N(0,1) [b,3,crop,crop]; l1h [b,20] with columns 0-14 zero, 1 new class w.p. 0.6 / 2 w.p. 0.4."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(b, 3, crop, crop, generator=g)
    l1h = torch.zeros(b, 20)
    for i in range(b):
        k = 1 if torch.rand(1, generator=g).item() < 0.6 else 2
        for c in torch.randperm(5, generator=g)[:k].tolist():
            l1h[i, 15 + c] = 1
    return images.to(device), l1h.to(device)


def structured_label_inputs(seed, b=2, H=129, W=129, stride=16, tot_classes=21, old_classes=16, K=25, noise=0.3,
                       blobs=(1, 3), two_label_prob=0.4):
    """Returns a dict of CPU tensors:
      seg2b [2b,C,h,w], l1h [b,C-1], int_masks [b,C,h,w], pg_w [cnew,cnew,1,1], pg_b [cnew],
      center_lo [b,C-1,hc,wc], offset_lo [b,2,hc,wc], old_center_lo [b,old-1,hc,wc], old_offset_lo [b,2,hc,wc]
    with h = (H-1)//stride + 1 and hc = (H-1)//4 + 1 (the network's output strides 16 and 4)."""
    rng = np.random.RandomState(seed)
    C = tot_classes
    cnew = C - old_classes
    h, w = (H - 1) // stride + 1, (W - 1) // stride + 1
    hc, wc = (H - 1) // 4 + 1, (W - 1) // 4 + 1
    seg = rng.randn(b, C, h, w).astype(np.float32) * noise
    seg[:, 0] += 2.0                                   # background prior
    segf = seg + rng.randn(b, C, h, w).astype(np.float32) * 0.05
    int_masks = rng.randn(b, C, h, w).astype(np.float32) * noise
    l1h = np.zeros((b, C - 1), dtype=np.float32)
    center = rng.randn(b, C - 1, hc, wc).astype(np.float32) * 0.05
    offset = rng.randn(b, 2, hc, wc).astype(np.float32) * 1.5
    yy, xx = np.mgrid[0:hc, 0:wc].astype(np.float32)
    for i in range(b):
        nlab = 2 if rng.rand() < two_label_prob else 1
        for cls in rng.choice(cnew, nlab, replace=False):
            l1h[i, old_classes - 1 + cls] = 1
            ch = old_classes + cls
            for _ in range(rng.randint(blobs[0], blobs[1] + 1)):
                bh, bw = rng.randint(1, max(2, h // 3) + 1), rng.randint(1, max(2, w // 3) + 1)
                y0, x0 = rng.randint(0, h - bh + 1), rng.randint(0, w - bw + 1)
                seg[i, ch, y0:y0 + bh, x0:x0 + bw] += 4.0 + rng.rand()
                segf[i, ch, y0:y0 + bh, x0:x0 + bw] += 4.0 + rng.rand()
                cy, cx = y0 + (bh - 1) / 2.0, x0 + (bw - 1) / 2.0
                iy, ix = int(round(cy)), int(round(cx))     # a 3x3 CAM plateau survives PAM (0.7*max) and the 3x3 smoothing
                ya, yb, xa, xb = max(iy - 1, 0), min(iy + 2, h), max(ix - 1, 0), min(ix + 2, w)
                int_masks[i, ch, ya:yb, xa:xb] = 3.0 + rng.rand(yb - ya, xb - xa).astype(np.float32) * 0.3
                # centre bump(s) and offsets pointing at the blob centre, at stride 4
                sc = stride / 4.0
                ninst = rng.randint(1, 3)
                for k in range(ninst):
                    py = (cy + (rng.rand() - 0.5) * bh * 0.6) * sc + (sc - 1) / 2
                    px = (cx + (rng.rand() - 0.5) * bw * 0.6) * sc + (sc - 1) / 2
                    bump = np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * 2.0 ** 2)) * (0.5 + 0.5 * rng.rand())
                    center[i, old_classes - 1 + cls] = np.maximum(center[i, old_classes - 1 + cls], bump)
                    reg = (abs(yy - py) < bh * sc * 0.6) & (abs(xx - px) < bw * sc * 0.6)
                    offset[i, 0][reg] = ((py - yy) * 4.0)[reg] + rng.randn(int(reg.sum())).astype(np.float32) * 0.3
                    offset[i, 1][reg] = ((px - xx) * 4.0)[reg] + rng.randn(int(reg.sum())).astype(np.float32) * 0.3
    # some old-class foreground for the distillation terms
    for i in range(b):
        oc = rng.randint(1, old_classes)
        y0, x0 = rng.randint(0, h - 1), rng.randint(0, w - 1)
        seg[i, oc, y0:y0 + 2, x0:x0 + 2] += 3.5
        segf[i, oc, y0:y0 + 2, x0:x0 + 2] += 3.5
    seg2b = np.concatenate([seg, segf[..., ::-1]], 0).copy()   # second half is the network's view of the flipped image
    t = torch.from_numpy
    return {
        "seg2b": t(seg2b), "l1h": t(l1h), "int_masks": t(int_masks),
        "pg_w": t((np.eye(cnew) + rng.randn(cnew, cnew) * 0.05).astype(np.float32).reshape(cnew, cnew, 1, 1)),
        "pg_b": t((rng.randn(cnew) * 0.01).astype(np.float32)),
        "center_lo": t(center), "offset_lo": t(offset),
        "old_center_lo": t(rng.randn(b, old_classes - 1, hc, wc).astype(np.float32) * 0.1),
        "old_offset_lo": t(rng.randn(b, 2, hc, wc).astype(np.float32)),
        "H": H, "W": W, "old_classes": old_classes, "tot_classes": C, "K": K,
    }
