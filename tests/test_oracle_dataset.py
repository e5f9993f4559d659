"""The input-pipeline oracle (oracle/dataset.py) against the fixtures produced by the reference's own
IncrementalInstanceSegmentationDataset.__getitem__ (tests/golden/make_golden.py: dataset_fixture): bit-exact."""
import glob
import os
import random

import numpy as np
import pytest

from cl4wsis_amd import tasks
from oracle import dataset as ods

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dataset_s*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_train_item_matches_reference(path):
    g = np.load(path)
    step, masking, size, seed = int(g["step"]), bool(g["masking"]), int(g["size"]), int(g["seed"])
    step_dict = tasks.get_task_dict("voc", "15-5", step)
    for k in range(3):
        random.seed(1000 * seed + k)
        img, seg_map, center, offset, weight, l1h, _ = ods.train_item(g[f"in_img{k}"], g[f"in_seg{k}"], g[f"in_mask{k}"],
                                                                      g[f"in_l1h{k}"], step_dict, step, size, masking)
        assert np.array_equal(img.numpy(), g[f"img{k}"])
        assert np.array_equal(seg_map, g[f"seg_map{k}"])
        assert np.array_equal(center, g[f"center{k}"]) and center.max() == 1.0
        assert np.array_equal(offset, g[f"offset{k}"])
        assert np.array_equal(weight, g[f"weight{k}"]) and weight.sum() > 0
        assert np.array_equal(l1h, g[f"l1h{k}"])


def test_crop_params_consume_the_random_stream_like_the_reference():
    random.seed(7)
    a = [ods.random_resized_crop_params(500, 375) for _ in range(50)]
    assert all(0 <= i and 0 <= j and i + h <= 375 and j + w <= 500 for (i, j, h, w) in a)
    assert len(set(a)) > 25
    # a target area above the image area can never fit: ten failed draws, then the whole image (ratio inside [3/4, 4/3])
    assert ods.random_resized_crop_params(500, 375, scale=(1.5, 2.0)) == (0, 0, 375, 500)
    assert ods.random_resized_crop_params(500, 200, scale=(1.5, 2.0)) == (0, 116, 200, 267)


def test_resize_output_size():
    assert ods.resize_output_size(500, 375, 513) == (513, 684)
    assert ods.resize_output_size(375, 500, 513) == (684, 513)
