"""GPU parity of the device input pipeline (cl4_input_batch / cl4_target_generation, through the C ABI) against the oracle
(oracle/dataset.py: Pillow + numpy restatement of the reference loader) and the fixtures produced by the reference's own
dataset __getitem__ (tests/golden/dataset_s*.npz). Everything here is byte / index / exactly-rounded float work: BIT-EXACT."""
import glob
import os
import random

import numpy as np
import pytest
import torch

from cl4wsis_amd import tasks
from oracle import dataset as ods
from oracle import synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dataset_s*.npz")))


def _pack(arrs, dev):
    offs, pos = [], 0
    for a in arrs:
        offs.append(pos)
        pos += a.size
    buf = torch.from_numpy(np.concatenate([np.ascontiguousarray(a).reshape(-1) for a in arrs])).to(dev)
    return buf, offs


def _lut(mapping, dev):
    return torch.from_numpy(mapping.astype(np.uint8)).to(dev)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_train_batch_matches_reference_fixture(gpu, path):
    from cl4wsis_amd import ops
    g = np.load(path)
    step, masking, size, seed = int(g["step"]), bool(g["masking"]), int(g["size"]), int(g["seed"])
    step_dict = tasks.get_task_dict("voc", "15-5", step)
    mapping, order, labels = ods.label_mapping(step_dict, step, True, masking)
    total = len(order) - 1
    imgs = [g[f"in_img{k}"] for k in range(3)]
    lbls = [np.stack([g[f"in_seg{k}"], g[f"in_mask{k}"]], axis=-1) for k in range(3)]
    ibuf, ioff = _pack(imgs, gpu)
    lbuf, loff = _pack(lbls, gpu)
    descs = []
    for k in range(3):
        random.seed(1000 * seed + k)                     # the reference's draws: crop box, then the flip coin
        H, W = imgs[k].shape[:2]
        i, j, h, w = ods.random_resized_crop_params(W, H)
        descs.append((ioff[k], loff[k], H, W, i, j, h, w, int(random.random() < 0.5)))
    img, seg, mask = ops.input_batch(ibuf, descs, (size, size), labels=lbuf, lut=_lut(mapping, gpu))
    center, offset, weight = ops.target_generation(seg, mask, total, sigma=8)
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(img[k].cpu().numpy(), g[f"img{k}"])
        assert np.array_equal(seg[k].cpu().numpy(), g[f"seg_map{k}"].astype(np.int64))
        assert np.array_equal(center[k].cpu().numpy(), g[f"center{k}"])
        assert np.array_equal(offset[k].cpu().numpy(), g[f"offset{k}"])
        assert np.array_equal(weight[k].cpu().numpy(), g[f"weight{k}"])


@pytest.mark.parametrize("seed", range(6))
def test_resample_matches_pillow_up_and_down(gpu, seed):
    """Random photo sizes / crop boxes / output sizes, including strong down-sampling (long filter windows) and flips."""
    from cl4wsis_amd import ops
    rs = np.random.RandomState(seed)
    b = 5
    oh, ow = int(rs.randint(8, 200)), int(rs.randint(8, 200))
    imgs, lbls, descs_in = [], [], []
    for k in range(b):
        H, W = int(rs.randint(4, 400)), int(rs.randint(4, 400))
        imgs.append(rs.randint(0, 256, (H, W, 3)).astype(np.uint8))
        lbls.append(rs.randint(0, 256, (H, W, 2)).astype(np.uint8))
        h, w = int(rs.randint(1, H + 1)), int(rs.randint(1, W + 1))
        descs_in.append((int(rs.randint(0, H - h + 1)), int(rs.randint(0, W - w + 1)), h, w, int(rs.randint(2))))
    ibuf, ioff = _pack(imgs, gpu)
    lbuf, loff = _pack(lbls, gpu)
    descs = [(ioff[k], loff[k]) + imgs[k].shape[:2] + d for k, d in enumerate(descs_in)]
    img, seg, mask = ops.input_batch(ibuf, descs, (oh, ow), labels=lbuf, lut=None)
    torch.cuda.synchronize()
    for k, (i, j, h, w, flip) in enumerate(descs_in):
        ri = ods.resized_crop(imgs[k], i, j, h, w, (oh, ow))
        rl = ods.resized_crop(lbls[k], i, j, h, w, (oh, ow), nearest=True)
        if flip:
            ri, rl = ods.hflip(ri), ods.hflip(rl)
        assert np.array_equal(img[k].cpu().numpy(), ods.to_tensor_normalize(ri).numpy()), (k, imgs[k].shape, (i, j, h, w), (oh, ow))
        assert np.array_equal(seg[k].cpu().numpy(), rl[:, :, 0].astype(np.int64))
        assert np.array_equal(mask[k].cpu().numpy(), rl[:, :, 1] * (rl[:, :, 0] > 0))


def test_to_tensor_normalize_all_byte_values(gpu):
    """Every uint8 value through ToTensor + Normalize in each channel (identity resample): exactly torch's float32 result."""
    from cl4wsis_amd import ops
    a = np.zeros((16, 16, 3), dtype=np.uint8)
    a[:, :, 0] = np.arange(256).reshape(16, 16)
    a[:, :, 1] = np.arange(256)[::-1].reshape(16, 16)
    a[:, :, 2] = (np.arange(256) * 7 % 256).reshape(16, 16)
    img, _, _ = ops.input_batch(torch.from_numpy(a.reshape(-1)).to(gpu), [(0, 0, 16, 16, 0, 0, 16, 16, 0)], (16, 16))
    assert np.array_equal(img[0].cpu().numpy(), ods.to_tensor_normalize(a).numpy())


def test_val_transform_full_size(gpu):
    """get_dataset's val stack on a VOC-sized photo: Resize(513) of 375x500 -> 513x684, image bilinear + semantic map nearest."""
    from cl4wsis_amd import ops
    img_u8, seg_u8, mask_u8, _ = synth.dataset_sample(5, H=375, W=500, n_inst=5)
    ref_img, ref_seg = ods.val_transform(img_u8, seg_u8, 513)
    oh, ow = ods.resize_output_size(500, 375, 513)
    lbl = np.stack([seg_u8, mask_u8], axis=-1)
    img, seg, _ = ops.input_batch(torch.from_numpy(img_u8.reshape(-1)).to(gpu), [(0, 0, 375, 500, 0, 0, 375, 500, 0)], (oh, ow),
                                  labels=torch.from_numpy(lbl.reshape(-1)).to(gpu), want_mask=False)
    assert tuple(img.shape) == (1, 3, 513, 684)
    assert np.array_equal(img[0].cpu().numpy(), ref_img.numpy())
    assert np.array_equal(seg[0].cpu().numpy(), ref_seg.astype(np.int64))


def test_target_generation_full_size_and_edge_cases(gpu):
    """513x513 with many instances (centres near the border clip the gaussian), an id-255 region and an empty image."""
    from cl4wsis_amd import ops
    rs = np.random.RandomState(3)
    b, S, C = 3, 513, 20
    seg = np.zeros((b, S, S), dtype=np.int64)
    mask = np.zeros((b, S, S), dtype=np.uint8)
    yy, xx = np.mgrid[0:S, 0:S]
    for k in range(40):
        cy, cx, r = rs.randint(0, S), rs.randint(0, S), rs.randint(3, 60)
        blob = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        seg[0][blob] = rs.randint(1, C + 1)
        mask[0][blob] = k + 1
    mask[0][seg[0] == 0] = 0
    seg[1][100:200, 50:300] = 7
    mask[1][100:200, 50:300] = 255                       # ignored id
    seg[1][300:310, 300:500] = 3
    mask[1][300:310, 300:500] = 9
    center, offset, weight = ops.target_generation(torch.from_numpy(seg).to(gpu), torch.from_numpy(mask).to(gpu), C, sigma=8)
    g = ods.gaussian(8)
    for k in range(b):
        rc, ro, rw = ods.label_generation(seg[k], mask[k], C, 8, g)
        assert np.array_equal(center[k].cpu().numpy(), rc)
        assert np.array_equal(offset[k].cpu().numpy(), ro)
        assert np.array_equal(weight[k].cpu().numpy(), rw)
    assert weight[2].abs().sum().item() == 0 and center[2].abs().sum().item() == 0


def test_input_batch_rejects_bad_descriptors(gpu):
    from cl4wsis_amd import ops
    buf = torch.zeros(10 * 10 * 3, dtype=torch.uint8, device=gpu)
    with pytest.raises(RuntimeError, match="crop box"):
        ops.input_batch(buf, [(0, 0, 10, 10, 5, 5, 8, 8, 0)], (8, 8))
    with pytest.raises(RuntimeError, match="exceeds"):
        ops.input_batch(buf, [(0, 0, 20, 20, 0, 0, 8, 8, 0)], (8, 8))
