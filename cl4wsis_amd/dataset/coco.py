"""COCO-2017 instance segmentation source of the coco-to-voc protocol — reference dataset/coco.py:1-124 (COCO, COCOIncremental).

Host side only (file lists, annotation index, JPEG + mask decoding into uint8 arrays); resampling, normalisation and target
generation run on the device (loader.DeviceLoader). Items have the same layout as voc.VOCInstanceSegmentation's."""
import os.path as osp

import numpy as np
from PIL import Image

from .coco_index import CocoIndex
from .dataset import IncrementalInstanceSegmentationDataset
from .voc import has_valid_annotation

ignore_labels = [12, 26, 29, 30, 45, 66, 68, 69, 71, 83, 91]  # ids COCO never uses (1 = person)


class COCO:
    """dataset/coco.py:27-118: images listed in coco/split/{train,val}.txt that carry at least one valid annotation.
    Train items: (image u8 [H,W,3], seg u8 [H,W], instance-id u8 [H,W], 91-way image-level labels); val items:
    (image, seg, masks u8 [R,H,W], mask labels [R], file name)."""

    def __init__(self, root, train=True, transform=None, indices=None):
        ds_root = osp.join(osp.expanduser(root), "coco")
        self.ds_root, self.is_train = ds_root, train
        self.image_set = "train" if train else "val"
        self.folder = f"{self.image_set}2017"
        with open(osp.join(ds_root, "split", f"{self.image_set}.txt")) as f:
            files = f.readlines()
        self.coco = CocoIndex(osp.join(ds_root, "annotations", f"instances_{self.image_set}2017.json"))
        self.ids = sorted(self.coco.imgs.keys())
        self.images = [x[:-1] + ".jpg" for x in files]
        if indices is not None:
            self.images = [self.images[idx] for idx in indices]
        wanted = set(self.images)
        self.indices = []
        for img_id in self.ids:
            anno = self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id, iscrowd=False))
            if has_valid_annotation(anno) and self.coco.imgs[img_id]["file_name"] in wanted:
                self.indices.append(img_id)
        self.transform = transform

    def _load_image(self, id):
        path = self.coco.loadImgs(id)[0]["file_name"]
        with Image.open(osp.join(self.ds_root, "images", self.folder, path)) as im:
            return np.array(im.convert("RGB"), dtype=np.uint8), path

    def __getitem__(self, index):
        id = self.indices[index]
        img, path = self._load_image(id)
        anno = self.coco.loadAnns(self.coco.getAnnIds(id))
        masks = np.stack([self.coco.annToMask(a) for a in anno])
        cats = np.array([a["category_id"] for a in anno])
        seg = np.max(masks.astype(np.int64) * cats[:, None, None], axis=0).astype(np.uint8)
        if not self.is_train:
            return img, seg, masks, cats, path
        inst = np.max(masks.astype(np.int64) * (np.arange(len(anno)) + 1)[:, None, None], axis=0).astype(np.uint8)
        lbls = np.zeros((91,))
        lbls[np.unique(cats) - 1] = 1
        return img, seg, inst, lbls

    def __len__(self):
        return len(self.indices)


class COCOIncremental(IncrementalInstanceSegmentationDataset):
    """dataset/coco.py:121-124."""

    def make_dataset(self, root, ann_file, old_classes, new_classes, is_train, val_on_trainset, pseudo=None, overlap=True,
                     indices=None, masking=True):
        return COCO(root, is_train, transform=None, indices=indices)
