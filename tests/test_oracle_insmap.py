"""CPU: pins oracle.labels.get_ins_map (validation post-processing, dataset/utils.py:795-892) against the outputs of the
reference's own function (tests/golden/insmap_sXX.npz) — exactly: semantic map, instance labels, masks, scores — and the
restated chainercv helpers (mask_iou, calc_detection_voc_ap) against hand-computed cases."""
import glob
import os

import numpy as np
import pytest

from oracle import labels as L, synth

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "insmap_s*.npz")))


def canon(lab, msk, sc):
    order = np.lexsort((msk.reshape(len(msk), -1).argmax(1), lab))
    return lab[order], msk[order], sc[order]


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_get_ins_map_matches_reference(path):
    g = np.load(path)
    seed, H, W, flip = (int(v) for v in g["case"])
    seg, center, offset = synth.insmap_inputs(seed, H, W, bool(flip))
    seg_map, lab, msk, sc = L.get_ins_map(seg, center, offset, val_flip=bool(flip))
    assert np.array_equal(seg_map.numpy().astype(np.uint8), g["seg_map"])
    lab, msk, sc = canon(lab, msk, sc)
    n = int(g["n"])
    assert len(lab) == n and np.array_equal(lab, g["labels"])
    ref_masks = np.unpackbits(g["masks"])[:n * H * W].reshape(n, H, W).astype(bool)
    assert np.array_equal(msk, ref_masks)
    np.testing.assert_allclose(sc, g["scores"], rtol=1e-6)


def test_mask_iou_and_voc_ap():
    a = np.zeros((2, 4, 4), bool); b = np.zeros((2, 4, 4), bool)
    a[0, :2] = True; a[1, 2:] = True
    b[0, :2, :2] = True; b[1] = True
    iou = L.mask_iou(a, b)
    np.testing.assert_allclose(iou, [[4 / 8, 8 / 16], [0, 8 / 16]])
    # precision/recall of 4 detections (tp fp tp fp) for 2 positives: AP = 0.5 * 1 + 0.5 * (2/3)
    match = np.array([1, 0, 1, 0])
    tp, fp = np.cumsum(match == 1), np.cumsum(match == 0)
    ap = L.calc_detection_voc_ap([tp / (tp + fp), None], [tp / 2, None])
    assert abs(ap[0] - (0.5 + 0.5 * 2 / 3)) < 1e-12 and np.isnan(ap[1])
    ap07 = L.calc_detection_voc_ap([tp / (tp + fp)], [tp / 2], use_07_metric=True)
    assert abs(ap07[0] - (6 * 1.0 + 5 * 2 / 3) / 11) < 1e-12
