"""CPU restatement of the reference's input pipeline (TEST INFRASTRUCTURE ONLY — never imported by the product path).

Covers SURVEY.md §8(f) row 2: `dataset/__init__.py:9-73` (get_dataset and its transform stack),
`dataset/dataset.py:110-257` (IncrementalInstanceSegmentationDataset.__getitem__, LabelTransform, LabelSelection),
`dataset/utils.py:182-192,217-252,299-350` (gaussian, center_map_gen, label_generation) and the transforms of
`dataset/transform.py` that get_dataset uses (RandomResizedCrop :582-660, RandomHorizontalFlip :317-340, ToTensor :419-445,
Normalize :452-478, Resize :61-93).

Third-party arithmetic: the reference's transforms call `torchvision.transforms.functional` (version unpinned, not
installed here) on PIL images; for PIL inputs those functions are thin wrappers over Pillow:
  F.resized_crop(img, i, j, h, w, size, interp) = img.crop((j, i, j + w, i + h)).resize(size[::-1], interp)
  F.hflip(img) = img.transpose(FLIP_LEFT_RIGHT);   F.resize(img, int) = smaller edge -> size, other edge int(size * long / short)
  F.to_tensor(u8 HWC) = CHW float32 / 255;   F.normalize(t, mean, std) = (t - mean) / std in float32.
Pillow IS installed here (12.2.0), so the pixel arithmetic is pinned against Pillow itself; `label_generation` and the
dataset's __getitem__ are pinned against the imported reference (tests/golden/make_golden.py: dataset_fixture)."""
import math
import random

import numpy as np
import torch
from PIL import Image

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


# ---- transforms -------------------------------------------------------------------------------------------------------
def random_resized_crop_params(width, height, scale=(0.5, 2.0), ratio=(3. / 4., 4. / 3.), rng=random):
    """transform.py:604-640 — (i, j, h, w); consumes the `random` stream exactly like the reference."""
    area = width * height
    for _ in range(10):
        target_area = rng.uniform(*scale) * area
        log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
        aspect_ratio = math.exp(rng.uniform(*log_ratio))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if w <= width and h <= height:
            i = rng.randint(0, height - h)
            j = rng.randint(0, width - w)
            return i, j, h, w
    in_ratio = width / height
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def resize_output_size(width, height, size):
    """torchvision F.resize with an int size: the smaller edge becomes `size` (transform.py:61-93) -> (out_h, out_w)."""
    short, long = (width, height) if width <= height else (height, width)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if width <= height else (new_short, new_long)


def resized_crop(img_u8, i, j, h, w, out_hw, nearest=False):
    """Pillow crop + resize of an HxWxC uint8 array (C = 3: RGB image, bilinear; C = 2: the (seg, mask) 'LA' label, nearest)."""
    im = Image.fromarray(img_u8)
    im = im.crop((j, i, j + w, i + h)).resize((out_hw[1], out_hw[0]), Image.NEAREST if nearest else Image.BILINEAR)
    return np.array(im, dtype=np.uint8)


def hflip(a):
    return np.array(Image.fromarray(a).transpose(Image.FLIP_LEFT_RIGHT), dtype=np.uint8)


def to_tensor_normalize(img_u8, mean=MEAN, std=STD):
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    return t.sub_(m).div_(s)


def train_transform(img_u8, lbl_u8, size, rng=random):
    """get_dataset's train Compose (dataset/__init__.py:14-20) -> (float32 [3,S,S], uint8 [S,S,2], (i,j,h,w,flip))."""
    H, W = img_u8.shape[:2]
    i, j, h, w = random_resized_crop_params(W, H, rng=rng)
    img = resized_crop(img_u8, i, j, h, w, (size, size))
    lbl = resized_crop(lbl_u8, i, j, h, w, (size, size), nearest=True)
    flip = rng.random() < 0.5
    if flip:
        img, lbl = hflip(img), hflip(lbl)
    return to_tensor_normalize(img), lbl, (i, j, h, w, int(flip))


def val_transform(img_u8, seg_u8, size):
    """get_dataset's val/test Compose (dataset/__init__.py:22-27): Resize(int) -> ToTensor -> Normalize."""
    H, W = img_u8.shape[:2]
    oh, ow = resize_output_size(W, H, size)
    img = resized_crop(img_u8, 0, 0, H, W, (oh, ow))
    seg = np.array(Image.fromarray(seg_u8).resize((ow, oh), Image.NEAREST), dtype=np.uint8)
    return to_tensor_normalize(img), seg


# ---- targets ----------------------------------------------------------------------------------------------------------
def gaussian(sigma=6):
    """dataset/utils.py:182-192."""
    size = 6 * sigma + 3
    x = np.arange(0, size, 1, float)
    y = x[:, np.newaxis]
    x0 = y0 = 3 * sigma + 1
    return np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))


def center_map_gen(center_map, x, y, label, sigma, g):
    """dataset/utils.py:217-252."""
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


def label_generation(seg_map, mask, num_classes, sigma, g):
    """dataset/utils.py:299-350 — centre / offset / weight targets of the fully supervised step from the instance mask."""
    H, W = mask.shape
    offset_map = np.zeros((2, H, W), dtype=np.float32)
    weight_map = np.zeros((1, H, W), dtype=np.float32)
    center_map = np.zeros((num_classes, H, W), dtype=np.float32)
    for id in np.unique(mask):
        if id in (0, 255):
            continue
        ys, xs = np.where(mask == id)
        weight_map[0, ys, xs] = 1
        center_y, center_x = np.mean(ys), np.mean(xs)
        cls = int(seg_map[ys[0], xs[0]]) - 1
        assert cls < num_classes
        center_map = center_map_gen(center_map, int(center_x), int(center_y), cls, sigma, g)
        offset_map[0, ys, xs] = center_y - ys.astype(np.float32)
        offset_map[1, ys, xs] = center_x - xs.astype(np.float32)
    return center_map, offset_map, weight_map


# ---- dataset wrapper --------------------------------------------------------------------------------------------------
def label_mapping(step_dict, step, train=True, masking=True, masking_value=0):
    """dataset/dataset.py:143-183 — (256-entry label LUT, order, labels)."""
    order = [c for s in sorted(step_dict) for c in step_dict[s]]
    labels = ([order[0]] + list(step_dict[step])) if step > 0 else list(step_dict[step])
    inverted = {lb: order.index(lb) for lb in order}
    inverted[255] = masking_value if train else 255
    keys = (labels + [255]) if masking else list(inverted)
    mapping = np.zeros((256,))
    for k in keys:
        mapping[k] = inverted[k]
    return mapping, order, labels


def label_selection(lbl_1h, order, labels, masking):
    """dataset/dataset.py:266-280 (LabelSelection)."""
    o = np.array(order)
    o = o[o != 0] - 1
    masker = np.zeros(len(o)) if masking else np.ones(len(o))
    if masking:
        masker[-len(labels) + 1:] = 1
    return lbl_1h[o] * masker


def train_item(img_u8, seg_u8, mask_u8, lbl_1h, step_dict, step, size, masking=True, sigma=8, rng=random):
    """IncrementalInstanceSegmentationDataset.__getitem__ (train branch, dataset/dataset.py:202-221) with get_dataset's
    train transform -> (img f32 [3,S,S], seg_map f64 [S,S], center [C,S,S], offset [2,S,S], weight [1,S,S], l1h, params)."""
    mapping, order, labels = label_mapping(step_dict, step, True, masking)
    labels_old = [l for s in range(step) for l in step_dict[s]]
    total_classes = len(labels_old) + len(list(step_dict[step])) - 1
    lbl = np.concatenate([seg_u8[:, :, None], mask_u8[:, :, None]], axis=-1).astype(np.uint8)
    img, lbl, params = train_transform(img_u8, lbl, size, rng)
    seg_map = mapping[lbl[:, :, 0]]
    mask = lbl[:, :, 1] * (seg_map > 0)
    center, offset, weight = label_generation(seg_map, mask, total_classes, sigma, gaussian(sigma))
    return img, seg_map, center, offset, weight, label_selection(lbl_1h, order, labels, masking), params
