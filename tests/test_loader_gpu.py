"""End-to-end device loader (cl4wsis_amd/dataset: get_dataset + DeviceLoader) on a tiny Pascal-VOC-shaped tree: every batch
tuple equals, bit for bit, what the oracle restatement of the reference loader produces from the same decoded photos and the
same `random` stream (oracle/dataset.train_item / val_transform, pinned by the reference fixtures)."""
import random

import numpy as np
import pytest
import torch

from cl4wsis_amd import argparser
from oracle import dataset as ods
from util import make_mini_voc

pytestmark = pytest.mark.gpu


def _opts(root, step, extra=()):
    p = argparser.get_argparser()
    o = p.parse_args(["--dataset", "voc", "--task", "15-5", "--step", str(step), "--data_root", str(root), "--crop_size", "97",
                      "--crop_size_val", "80", *extra])
    return argparser.modify_command_options(o)


@pytest.mark.parametrize("step,supervised", [(0, True), (1, False)])
def test_train_batches_match_oracle(gpu, tmp_path, step, supervised):
    from cl4wsis_amd.dataset import DeviceLoader, get_dataset
    from cl4wsis_amd import tasks
    make_mini_voc(tmp_path, n_train=7)
    train, _, _, _, _ = get_dataset(_opts(tmp_path, step))
    step_dict = tasks.get_task_dict("voc", "15-5", step)
    loader = DeviceLoader(train, batch_size=2, device=gpu, shuffle=False, drop_last=True, supervised=supervised)
    assert len(loader) == len(train) // 2
    random.seed(5)
    batches = list(loader)
    random.seed(5)
    for bi, (img, seg, center, offset, weight, l1h) in enumerate(batches):
        assert img.is_cuda and tuple(img.shape) == (2, 3, 97, 97) and seg.dtype == torch.int64
        for k in range(2):
            src = train.dataset[bi * 2 + k]                              # decoded photo + annotation maps (host arrays)
            r = ods.train_item(src[0], src[1], src[2], src[3], step_dict, step, 97, masking=True)
            assert np.array_equal(img[k].cpu().numpy(), r[0].numpy())
            assert np.array_equal(seg[k].cpu().numpy(), r[1].astype(np.int64))
            assert np.array_equal(l1h[k].cpu().numpy(), r[5])
            if supervised:
                assert np.array_equal(center[k].cpu().numpy(), r[2])
                assert np.array_equal(offset[k].cpu().numpy(), r[3])
                assert np.array_equal(weight[k].cpu().numpy(), r[4]) and r[4].sum() > 0
    if not supervised:
        assert batches[0][2].numel() == 1                                # dense targets skipped in the weakly supervised phases


def test_val_batches_match_oracle(gpu, tmp_path):
    from cl4wsis_amd.dataset import DeviceLoader, get_dataset
    truth = make_mini_voc(tmp_path)
    _, val, _, _, _ = get_dataset(_opts(tmp_path, 1))
    loader = DeviceLoader(val, batch_size=1, device=gpu)
    n = 0
    for (img, seg, masks, mask_label, paths), (src_img, src_seg, src_inst, _) in zip(loader, truth["val"]):
        ref_img, ref_seg = ods.val_transform(src_img, src_seg, 80)
        assert np.array_equal(img[0].cpu().numpy(), ref_img.numpy())
        assert np.array_equal(seg[0].cpu().numpy(), ref_seg.astype(np.int64))      # validation mapping is the identity on 0..20
        assert masks.shape[1] == mask_label.shape[1] == len(np.unique(src_inst)) - 1
        assert tuple(masks.shape[-2:]) == src_img.shape[:2]
        n += 1
    assert n == 3


def test_trainer_consumes_device_loader(gpu, tmp_path):
    """One weakly supervised phase-2 step straight from the device loader (no host round trip of the batch)."""
    from cl4wsis_amd.dataset import DeviceLoader, get_dataset
    from cl4wsis_amd.train import Trainer
    from util import p2_opts, QuietLogger
    make_mini_voc(tmp_path, n_train=7)
    opts = p2_opts(65)
    opts.data_root = str(tmp_path)
    opts.batch_size = 2
    train, _, _, _, _ = get_dataset(opts)
    loader = DeviceLoader(train, batch_size=2, device=gpu, shuffle=True, drop_last=True)
    tr = Trainer(QuietLogger(), gpu, opts)
    out = tr.step(next(iter(loader)))
    l = out["losses"]
    assert all(np.isfinite(float(v)) for v in (l.values() if isinstance(l, dict) else l.reshape(-1)))
