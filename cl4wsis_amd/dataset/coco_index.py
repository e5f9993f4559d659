"""Minimal index over a COCO-format annotation file — the subset of pycocotools.coco.COCO that the reference's
VOCInstanceSegmentation uses (dataset/voc.py:218-324 via torchvision.datasets.CocoDetection): sorted image ids,
getAnnIds(imgIds, iscrowd), loadAnns, loadImgs and annToMask.

pycocotools is a third-party dependency that is NOT vendored in the reference and not installed here (version unpinned): the
mask decoding below restates its published algorithms (maskApi.c: rleFrPoly / rleFrString / rleMerge / rleDecode). There are
no fixtures in the reference to pin them against — PARITY UNPINNED at this boundary; the self-consistency
tests are in tests/test_dataset_cpu.py."""
import json
from collections import defaultdict

import numpy as np


class CocoIndex:
    def __init__(self, annotation_file):
        with open(annotation_file) as f:
            d = json.load(f)
        self.imgs = {im["id"]: im for im in d.get("images", [])}
        self.anns = {a["id"]: a for a in d.get("annotations", [])}
        self.img_to_anns = defaultdict(list)
        for a in d.get("annotations", []):
            self.img_to_anns[a["image_id"]].append(a)
        self.cats = {c["id"]: c for c in d.get("categories", [])}

    def getAnnIds(self, imgIds=(), iscrowd=None):
        imgIds = imgIds if isinstance(imgIds, (list, tuple)) else [imgIds]
        anns = [a for i in imgIds for a in self.img_to_anns.get(i, [])] if len(imgIds) else list(self.anns.values())
        if iscrowd is not None:
            anns = [a for a in anns if a.get("iscrowd", 0) == iscrowd]
        return [a["id"] for a in anns]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadImgs(self, ids):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.imgs[ids]]

    def annToMask(self, ann):
        im = self.imgs[ann["image_id"]]
        return ann_to_mask(ann["segmentation"], im["height"], im["width"])


def ann_to_mask(segm, h, w):
    """uint8 [h, w] binary mask of a polygon list, an uncompressed RLE ({'counts': [...]}) or a compressed RLE string."""
    if isinstance(segm, list):
        runs = rle_merge([rle_from_polygon(np.asarray(p, dtype=np.float64), h, w) for p in segm], h * w)
    elif isinstance(segm["counts"], list):
        runs = np.asarray(segm["counts"], dtype=np.int64)
    else:
        runs = rle_from_string(segm["counts"])
    return rle_decode(runs, h, w)


def rle_decode(counts, h, w):
    """Runs alternate 0/1 starting with 0, in column-major order."""
    vals = np.zeros(len(counts), dtype=np.uint8)
    vals[1::2] = 1
    flat = np.repeat(vals, np.asarray(counts, dtype=np.int64))
    if flat.size != h * w:
        raise ValueError(f"RLE covers {flat.size} pixels, image has {h * w}")
    return np.ascontiguousarray(flat.reshape(w, h).T)


def rle_from_string(s):
    if isinstance(s, bytes):
        s = s.decode("ascii")
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return np.asarray(cnts, dtype=np.int64)


def rle_merge(rles, n):
    """Union of run-length masks (the annotation's polygons)."""
    if len(rles) == 1:
        return rles[0]
    m = np.zeros(n, dtype=bool)
    for r in rles:
        ends = np.cumsum(r)
        for a, b in zip(ends[0::2], ends[1::2]):
            m[a:b] = True
    edges = np.flatnonzero(np.diff(np.concatenate([[False], m, [False]]).astype(np.int8)))
    runs = np.diff(np.concatenate([[0], edges, [n]]))
    return runs.astype(np.int64)


def rle_from_polygon(xy, h, w):
    """maskApi.c rleFrPoly: up-sample the boundary x5, walk it densely, keep the x-crossings, down-sample, sort, run-length."""
    scale = 5.0
    k = len(xy) // 2
    x = np.empty(k + 1, dtype=np.int64)
    y = np.empty(k + 1, dtype=np.int64)
    x[:k] = (scale * xy[0::2] + .5).astype(np.int64)
    y[:k] = (scale * xy[1::2] + .5).astype(np.int64)
    x[k], y[k] = x[0], y[0]
    us, vs = [], []
    for j in range(k):
        xs, xe, ys, ye = int(x[j]), int(x[j + 1]), int(y[j]), int(y[j + 1])
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else 0.0
            d = np.arange(dx + 1)
            t = dx - d if flip else d
            us.append(t + xs)
            vs.append((ys + s * t + .5).astype(np.int64))
        else:
            s = (xe - xs) / dy
            d = np.arange(dy + 1)
            t = dy - d if flip else d
            vs.append(t + ys)
            us.append((xs + s * t + .5).astype(np.int64))
    u = np.concatenate(us)
    v = np.concatenate(vs)
    ch = np.flatnonzero(u[1:] != u[:-1]) + 1
    xd = np.where(u[ch] < u[ch - 1], u[ch], u[ch] - 1).astype(np.float64)
    xd = (xd + .5) / scale - .5
    keep = (np.floor(xd) == xd) & (xd >= 0) & (xd <= w - 1)
    yd = np.minimum(v[ch], v[ch - 1]).astype(np.float64)
    yd = np.ceil(np.clip((yd + .5) / scale - .5, 0, h))
    a = (xd[keep].astype(np.int64) * h + yd[keep].astype(np.int64))
    a = np.sort(np.concatenate([a, [h * w]]))
    a = np.diff(np.concatenate([[0], a]))
    b = [int(a[0])]
    j = 1
    while j < len(a):
        if a[j] > 0:
            b.append(int(a[j]))
            j += 1
        else:
            j += 1
            if j < len(a):
                b[-1] += int(a[j])
                j += 1
    return np.asarray(b, dtype=np.int64)
