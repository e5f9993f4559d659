"""GPU parity of the validation post-processing (cl4_val_seg + cl4_instance_map through the C ABI) against the outputs of
the reference's own get_ins_map (tests/golden/insmap_sXX.npz) — semantic map, instance masks and labels exactly, scores to
1e-5 — and against the oracle with --val_ignore on / a larger minimum mask / --val_clean."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import labels as L, synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "insmap_s*.npz")))


def canon(lab, msk, sc):
    order = np.lexsort((msk.reshape(len(msk), -1).argmax(1), lab))
    return lab[order], msk[order], sc[order]


def run_device(gpu, seg, center, offset, flip, ignore=False, cls_label=None, **kw):
    from cl4wsis_amd import ops
    seg_map, prob = ops.val_seg(seg.to(gpu), flip=flip, cls_label=None if cls_label is None else cls_label.to(gpu))
    c = center.to(gpu)
    if flip:
        c = ops.scale_(ops.add_act(c[:1].contiguous(), ops.flip_w(c[1:].contiguous()), ops.ACT_NONE, 0.0), 0.5)
    ins, table = ops.instance_map(seg_map, prob, c[0].contiguous(), offset[0].contiguous().to(gpu), ignore=ignore, **kw)
    ins, table = ins.cpu().numpy(), table.cpu().numpy()
    keep = [i for i in range(len(table)) if table[i, 2] > 0]
    if not keep:
        H, W = ins.shape
        return seg_map.cpu().numpy(), np.zeros(1, dtype=np.int64), np.zeros((1, H, W), dtype=bool), np.zeros(1)
    lab = np.array([int(table[i, 0]) for i in keep])
    msk = np.stack([ins == i for i in keep], 0)
    assert all(msk[j].sum() == int(table[i, 2]) for j, i in enumerate(keep))
    return seg_map.cpu().numpy(), lab, msk, np.array([table[i, 1] for i in keep], dtype=np.float64)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_instance_map_matches_reference(gpu, path):
    g = np.load(path)
    seed, H, W, flip = (int(v) for v in g["case"])
    seg, center, offset = synth.insmap_inputs(seed, H, W, bool(flip))
    seg_map, lab, msk, sc = run_device(gpu, seg, center, offset, bool(flip))
    assert np.array_equal(seg_map, g["seg_map"])
    lab, msk, sc = canon(lab, msk, sc)
    n = int(g["n"])
    assert len(lab) == n and np.array_equal(lab, g["labels"])
    assert np.array_equal(msk, np.unpackbits(g["masks"])[:n * H * W].reshape(n, H, W).astype(bool))
    np.testing.assert_allclose(sc, g["scores"], rtol=1e-5)


@pytest.mark.parametrize("variant", ["ignore", "clean", "thresh"])
def test_instance_map_variants_vs_oracle(gpu, variant):
    seed, H, W, flip = 5, 145, 129, True
    seg, center, offset = synth.insmap_inputs(seed, H, W, flip)
    kw, okw, cls_label = {}, {}, None
    if variant == "ignore":
        center = center * 0.05                     # no centre passes the threshold: contours survive only without ignore
        kw, okw = dict(ignore=True), dict(val_ignore=True)
    elif variant == "clean":
        cls_label = torch.zeros(1, 20); cls_label[0, 16:] = 1
        okw = dict(cls_label=cls_label)
    else:
        kw, okw = dict(thresh=0.5, nms_kernel=15, beta=5.0), dict(val_thresh=0.5, val_kernel=15, beta=5.0)
    rseg, rlab, rmsk, rsc = L.get_ins_map(seg, center, offset, val_flip=flip, **okw)
    dseg, lab, msk, sc = run_device(gpu, seg, center, offset, flip, cls_label=None if cls_label is None else cls_label[0], **kw)
    assert np.array_equal(dseg, rseg.numpy().astype(np.uint8))
    rlab, rmsk, rsc = canon(rlab, rmsk, rsc)
    lab, msk, sc = canon(lab, msk, sc)
    assert np.array_equal(lab, rlab) and np.array_equal(msk, rmsk)
    np.testing.assert_allclose(sc, rsc, rtol=1e-5, atol=1e-8)
    if variant == "ignore":                        # and without ignore each contour is one instance
        _, rlab2, rmsk2, _ = L.get_ins_map(seg, center, offset, val_flip=flip)
        _, lab2, msk2, _ = run_device(gpu, seg, center, offset, flip)
        assert len(lab2) == len(rlab2) > len(lab) or len(rlab2) == len(lab2)
        assert np.array_equal(canon(lab2, msk2, np.zeros(len(lab2)))[1], canon(rlab2, rmsk2, np.zeros(len(rlab2)))[1])
