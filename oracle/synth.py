"""ORACLE helper (test infrastructure): deterministic structured inputs for the label stage.

SURVEY.md §8(d) "structured-labels" mode: per image 1-3 blobs per labelled new class in the semantic
logits, a CAM bump at each blob, centre / offset maps consistent with the blobs (plus noise), so that
connected components, peaks, NMS centres, offset clusters and the refine branches are all exercised.
Used by tests/ and the golden generator; the generator itself lives in cl4wsis_amd/synthetic.py (bench.py's structured-labels mode).
"""
import numpy as np
import torch
import torch.nn.functional as F


from cl4wsis_amd.synthetic import structured_label_inputs as label_stage_inputs  # noqa: E402,F401  (one generator for tests and bench)


def wss_inputs(seed, b, c, h, w):
    """Seeded inputs of the phase-1 loss-stack fixtures (regenerated identically by the tests)."""
    g = torch.Generator().manual_seed(1000 + seed)
    img = torch.rand(b, 3, h, w, generator=g)                                   # denormalised image in [0, 1]
    img[:, :, h // 3:2 * h // 3, w // 4:w // 2] += 0.5                        # an edge for the affinity to respect
    logits = torch.randn(b, c, h, w, generator=g) * 2.0
    logits[:, 0] += 1.0
    labels = (torch.rand(b, c - 1, generator=g) < 0.3).float()
    labels[:, -1] = 1.0
    for i in range(b):                                                          # make labelled classes win somewhere
        for k in torch.nonzero(labels[i]).flatten().tolist():
            y0, x0 = int(torch.randint(0, h - 6, (1,), generator=g)), int(torch.randint(0, w - 6, (1,), generator=g))
            logits[i, k + 1, y0:y0 + 6, x0:x0 + 6] += 6.0
    seg_lab = torch.randint(0, c, (b, 2 * h + 1, 2 * w + 1), generator=g)
    seg_lab[torch.rand(seg_lab.shape, generator=g) < 0.1] = 255
    big = torch.randn(b, c, 2 * h + 1, 2 * w + 1, generator=g)
    return dict(img=img, logits=logits, labels=labels, seg_lab=seg_lab, big=big)


def insmap_inputs(seed, H, W, flip):
    """Full-resolution network outputs for the validation post-processing fixtures (get_ins_map): the structured
    low-resolution maps of label_stage_inputs upsampled to (H, W); batch 2 = (image, flipped image) when flip."""
    inp = label_stage_inputs(seed, b=2 if flip else 1, H=H, W=W)
    tot, old = inp["tot_classes"], inp["old_classes"]
    nb = 2 if flip else 1
    up = lambda t: F.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)
    seg = up(inp["seg2b"][:nb]) * 3
    center = torch.zeros(nb, tot - 1, H, W)
    center[:, old - 1:] = up(inp["center_lo"][:nb, old - 1:])
    center = center.clamp(0, 0.95)
    offset = up(inp["offset_lo"][:nb])
    return seg, center, offset


def dataset_sample(seed, H=120, W=160, n_inst=4, classes=(3, 16, 17, 20), void=True):
    """One synthetic annotated photo for the input-pipeline tests: a smooth uint8 RGB image, a semantic map (VOC class ids,
    255 = void rim), an instance-id map and the 20-way image-level label vector (what VOCInstanceSegmentation.__getitem__
    hands to the dataset wrapper, dataset/voc.py:280-324)."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.zeros((H, W, 3))
    for c in range(3):
        for _ in range(4):
            fy, fx, ph = rs.uniform(0.01, 0.3), rs.uniform(0.01, 0.3), rs.uniform(0, 6.28)
            img[:, :, c] += rs.uniform(20, 60) * np.sin(fy * yy + fx * xx + ph)
    img = np.clip(img + 128 + rs.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    seg = np.zeros((H, W), dtype=np.uint8)
    mask = np.zeros((H, W), dtype=np.uint8)
    l1h = np.zeros(20)
    for k in range(n_inst):
        cy, cx = rs.uniform(0.15, 0.85) * H, rs.uniform(0.15, 0.85) * W
        ry, rx = rs.uniform(0.08, 0.25) * H, rs.uniform(0.08, 0.25) * W
        blob = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        cls = classes[rs.randint(len(classes))]
        if void:
            rim = (((yy - cy) / (ry + 2)) ** 2 + ((xx - cx) / (rx + 2)) ** 2 < 1) & ~blob
            seg[rim & (seg == 0)] = 255
        seg[blob] = cls
        mask[blob] = k + 1
        l1h[cls - 1] = 1
    return img, seg, mask, l1h


# ----------------------------------------------------------------------------------------------------------------------
# Edge scenes of the label stage: one image each, hand-placed so that a specific branch of the reference's
# modules/utils.py executes (statement numbers in tests/golden/label_coverage.txt). All values are literal constants
# (no data-dependent search), so the inputs are identical on every machine.
# ----------------------------------------------------------------------------------------------------------------------
EDGE_SCENES = ("tiny", "many", "lowscore", "cluster_only", "cluster_far", "cluster_near", "nothing")
EDGE_FIXTURES = {                       # fixture name -> (H, W, scenes = one image each)
    "a": (129, 129, ("tiny", "many", "lowscore")),
    "b": (129, 129, ("cluster_only", "nothing")),
    "c": (193, 193, ("cluster_far", "cluster_near")),
}


def label_stage_edge_inputs(name, stride=16, tot_classes=21, old_classes=16, K=25):
    """Same dictionary as label_stage_inputs for the edge fixture `name` of EDGE_FIXTURES.

    Scenes (class = first new class unless noted; coordinates in low-resolution cells):
      tiny          a one-cell semantic bump that wins the arg-max on fewer than 20 full-resolution pixels (plus a normal
                    blob of a second class)                     -> `size < MINIMUM_MASK_SIZE: continue` (:231, :314)
      many          one large blob holding six centre bumps more than 20 px apart, offsets pointing at the nearest bump,
                    two CAM peaks inside the blob               -> `n_ins > MAXIMUM_NUM_INST: continue` (:336), match_count 2
      lowscore      two bumps A, B; every pixel votes for B except a far strip that votes for A, whose instance therefore
                    holds no centre value >= 0.3                -> centre <- contour centroid, conf = seg score (:354-355)
      cluster_only  no centre value above 0.3, a ~21 px region of near-zero offsets over a 0.2 bump
                                                                -> cluster centres adopted and marked 1.0 (:579-583)
      cluster_far   an NMS centre at one end of a 176 px blob and a near-zero-offset cluster > 100 px away
                                                                -> cluster appended and marked (:591-592)
      cluster_near  the same with the cluster < 100 px from the NMS centre -> not appended (:590 false)
      nothing       a blob with neither a centre above 0.3 nor a cluster -> zeros (:599-600)
    """
    H, W, scenes = EDGE_FIXTURES[name]
    b = len(scenes)
    rng = np.random.RandomState(7000 + sum(ord(ch) for ch in name))
    C = tot_classes
    cnew = C - old_classes
    h, w = (H - 1) // stride + 1, (W - 1) // stride + 1
    hc, wc = (H - 1) // 4 + 1, (W - 1) // 4 + 1
    seg = rng.randn(b, C, h, w).astype(np.float32) * 0.02
    seg[:, 0] += 2.0
    int_masks = rng.randn(b, C, h, w).astype(np.float32) * 0.05
    l1h = np.zeros((b, C - 1), dtype=np.float32)
    center = np.abs(rng.randn(b, C - 1, hc, wc).astype(np.float32)) * 0.01
    offset = np.zeros((b, 2, hc, wc), dtype=np.float32)
    offset[:, 0] = 9.0                                     # default: a long vote, |offset| far above the cluster threshold
    offset[:, 1] = -7.0
    offset += rng.randn(b, 2, hc, wc).astype(np.float32) * 0.05
    yy, xx = np.mgrid[0:hc, 0:wc].astype(np.float32)       # stride-4 cell coordinates
    sc = stride // 4

    def blob(i, cls, y0, y1, x0, x1, amp=5.0):
        l1h[i, old_classes - 1 + cls] = 1
        seg[i, old_classes + cls, y0:y1, x0:x1] += amp

    def cam_peak(i, cls, y, x, amp=3.0):
        ya, yb, xa, xb = max(y - 1, 0), min(y + 2, h), max(x - 1, 0), min(x + 2, w)
        int_masks[i, old_classes + cls, ya:yb, xa:xb] = amp + rng.rand(yb - ya, xb - xa).astype(np.float32) * 0.3

    def bump(i, cls, py, px, amp, sigma=2.0):
        g = np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * sigma ** 2)) * amp
        center[i, old_classes - 1 + cls] = np.maximum(center[i, old_classes - 1 + cls], g)

    def vote(i, region, py, px):
        """offsets of the stride-4 cells in `region` point at cell (py, px)."""
        offset[i, 0][region] = ((py - yy) * 4.0)[region]
        offset[i, 1][region] = ((px - xx) * 4.0)[region]

    def cells(y0, y1, x0, x1):
        """stride-4 cells covered by the stride-16 cell rectangle [y0,y1) x [x0,x1), one cell of margin"""
        return (yy >= y0 * sc - 2) & (yy < y1 * sc + 2) & (xx >= x0 * sc - 2) & (xx < x1 * sc + 2)

    def quiet_cluster(i, py, px, ring=3.4):
        """cell (py, px) votes zero, its 5x5 neighbourhood votes `ring` px sideways: after the x4 bilinear upsampling the
        |offset| < 2.5 region is a ~21 px island around the cell."""
        nb = (abs(yy - py) <= 2) & (abs(xx - px) <= 2)
        offset[i, 0][nb] = ring
        offset[i, 1][nb] = 0.0
        offset[i, :, py, px] = 0.0

    for i, scene in enumerate(scenes):
        if scene == "tiny":
            l1h[i, old_classes - 1 + 0] = 1
            seg[i, old_classes + 0, 2, 5] += 2.28             # a marginal one-cell win: a handful of pixels
            blob(i, 2, 4, 7, 1, 5)
            cam_peak(i, 2, 5, 3)
            bump(i, 2, 5 * sc + 1.5, 3 * sc + 1.5, 0.8)
            vote(i, cells(4, 7, 1, 5), 5 * sc + 1.5, 3 * sc + 1.5)
        elif scene == "many":
            blob(i, 1, 1, 7, 1, 8)
            cam_peak(i, 1, 2, 2)
            cam_peak(i, 1, 5, 6, amp=2.9)
            pts = [(8, 8), (8, 17), (8, 26), (20, 8), (20, 17), (20, 26)]
            amps = [0.9, 0.55, 0.8, 0.6, 0.95, 0.7]
            reg = cells(1, 7, 1, 8)
            d = np.stack([(yy - py) ** 2 + (xx - px) ** 2 for py, px in pts], 0).argmin(0)
            for k, ((py, px), a) in enumerate(zip(pts, amps)):
                bump(i, 1, py, px, a, sigma=1.5)
                vote(i, reg & (d == k), py, px)
        elif scene == "lowscore":
            blob(i, 3, 1, 7, 1, 7)
            cam_peak(i, 3, 4, 4)
            A, B = (18, 9), (18, 21)
            bump(i, 3, A[0], A[1], 0.9, sigma=1.5)
            bump(i, 3, B[0], B[1], 0.6, sigma=1.5)
            reg = cells(1, 7, 1, 7)
            vote(i, reg, B[0], B[1])
            vote(i, reg & (yy < 9), A[0], A[1])              # the far strip (>= 9 cells = 36 px above A) votes for A
        elif scene == "cluster_only":
            blob(i, 4, 2, 7, 2, 7)
            cam_peak(i, 4, 4, 4)
            bump(i, 4, 17, 18, 0.2, sigma=3.0)
            quiet_cluster(i, 17, 18)
        elif scene in ("cluster_far", "cluster_near"):
            blob(i, 0, 4, 8, 1, 12)
            cam_peak(i, 0, 5, 3)
            P = (23, 8)
            Q = (23, 40) if scene == "cluster_far" else (23, 26)     # 128 px / 72 px to the right of P
            bump(i, 0, P[0], P[1], 0.85, sigma=1.5)
            bump(i, 0, Q[0], Q[1], 0.2, sigma=3.0)
            reg = cells(4, 8, 1, 12)
            mid = (P[1] + Q[1]) / 2.0
            vote(i, reg & (xx < mid), P[0], P[1])
            vote(i, reg & (xx >= mid), Q[0], Q[1])
            quiet_cluster(i, Q[0], Q[1])
        elif scene == "nothing":
            blob(i, 2, 2, 6, 2, 6)
            bump(i, 2, 15, 15, 0.25, sigma=3.0)
        else:
            raise KeyError(scene)
    segf = seg + rng.randn(b, C, h, w).astype(np.float32) * 0.01
    seg2b = np.concatenate([seg, segf[..., ::-1]], 0).copy()
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
