"""Incremental dataset wrapper — reference dataset/dataset.py:110-280 (IncrementalInstanceSegmentationDataset,
LabelTransform, LabelSelection). Host logic only: class bookkeeping of the incremental protocol, the 256-entry label
mapping and the draw of the augmentation plan. `__getitem__` returns a RawSample (decoded uint8 arrays + the plan); the
pixel work and `label_generation` run batched on the device in loader.DeviceLoader."""
import os

import numpy as np
import torch.utils.data as data


class RawSample:
    """One decoded item on its way to the device."""
    __slots__ = ("image", "label", "plan", "l1h", "masks", "mask_label", "path")

    def __init__(self, image, label, plan, l1h=None, masks=None, mask_label=None, path=None):
        self.image, self.label, self.plan, self.l1h = image, label, plan, l1h
        self.masks, self.mask_label, self.path = masks, mask_label, path


class LabelSelection:
    """dataset/dataset.py:266-280."""

    def __init__(self, order, labels, masking):
        order = np.array(order)
        order = order[order != 0]
        order -= 1
        self.order = order
        if masking:
            self.masker = np.zeros((len(order)))
            self.masker[-len(labels) + 1:] = 1
        else:
            self.masker = np.ones((len(order)))

    def __call__(self, x):
        return x[self.order] * self.masker


class IncrementalInstanceSegmentationDataset(data.Dataset):
    def __init__(self, root, step_dict, train=True, val_on_trainset=False, transform=None, idxs_path=None, masking=True,
                 overlap=True, masking_value=0, step=0, sup="cls", sigma=8, point_thresh=0.5, weakly=False, pseudo=None,
                 ann_file=None):
        self.transform, self.weakly, self.train = transform, weakly, train
        self.step_dict, self.step, self.sup, self.point_thresh, self.sigma = step_dict, step, sup, point_thresh, sigma
        self.pseudo, self.val_on_trainset = pseudo, val_on_trainset
        self.order = [c for s in sorted(step_dict) for c in step_dict[s]]
        self.labels = ([self.order[0]] + list(step_dict[step])) if step > 0 else list(step_dict[step])
        self.labels_old = [lbl for s in range(step) for lbl in step_dict[s]]
        self.total_classes = len(self.labels_old) + len(list(step_dict[step])) - 1
        idxs = None
        if train and idxs_path is not None and "coco" in idxs_path:
            if not os.path.exists(idxs_path):
                raise FileNotFoundError(f"Please, add the traning spilt in {idxs_path}.")
            idxs = np.load(idxs_path)
        self.dataset = self.make_dataset(root, ann_file, old_classes=self.labels_old[1:], new_classes=self.labels[1:],
                                         is_train=train, val_on_trainset=val_on_trainset, pseudo=pseudo, overlap=overlap,
                                         indices=idxs, masking=masking)
        self.masking_value, self.masking = masking_value, masking
        self.inverted_order = {lb: self.order.index(lb) for lb in self.order}
        self.inverted_order[255] = masking_value if train else 255
        mapping_dict = {x: self.inverted_order[x] for x in self.labels + [255]} if masking else self.inverted_order
        mapping = np.zeros((256,))
        for k, v in mapping_dict.items():
            mapping[k] = v
        self.mapping = mapping
        self.lut = mapping.astype(np.uint8)            # what the device applies (cl4_input_batch `lut`)
        self.transform_1h = LabelSelection(self.order, self.labels, self.masking)

    def __getitem__(self, index):
        if index < 0:
            if -index > len(self):
                raise ValueError("absolute value of index should not exceed dataset length")
            index = len(self) + index
        if index >= len(self):
            raise ValueError("absolute value of index should not exceed dataset length")
        item = self.dataset[index]
        img = item[0]
        plan = self.transform.plan(img.shape[1], img.shape[0])
        if not self.train or self.val_on_trainset:
            img, seg, masks, mask_label, path = item
            label = np.stack([seg, np.zeros_like(seg)], axis=-1)
            return RawSample(img, label, plan, masks=masks, mask_label=np.array([self.mapping[i] for i in mask_label]),
                             path=path)
        img, seg, inst, lbl_1h = item
        return RawSample(img, np.stack([seg, inst], axis=-1).astype(np.uint8), plan, l1h=self.transform_1h(lbl_1h))

    def __len__(self):
        return len(self.dataset)

    def make_dataset(self, root, ann_file, **kw):
        raise NotImplementedError
