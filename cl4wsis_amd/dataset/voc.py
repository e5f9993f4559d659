"""Pascal-VOC (SBD, COCO-format annotations) instance segmentation source — reference dataset/voc.py:174-337.

Decodes on the host (JPEG via Pillow, masks via coco_index) and returns uint8 arrays; nothing here touches pixels beyond
decoding — resampling / normalisation / target generation happen on the device (loader.py)."""
import os

import numpy as np
from PIL import Image

from .coco_index import CocoIndex
from .dataset import IncrementalInstanceSegmentationDataset


def _has_only_empty_bbox(anno):
    return all(any(o <= 1 for o in obj["bbox"][2:]) for obj in anno)


def has_valid_annotation(anno):
    """dataset/voc.py:177-188."""
    if len(anno) == 0 or _has_only_empty_bbox(anno):
        return False
    return "keypoints" not in anno[0]


def image_annotation(anno, classes):
    """Only the given categories' annotations (dataset/voc.py:190-198)."""
    return [a for a in anno if a["category_id"] in classes]


def check_if_insert(anno, overlap, seen_classes, new_classes, is_train=True):
    """Image filter of the incremental protocol (dataset/voc.py:200-216): overlap = any new class present; disjoint = a new
    class present and no future class."""
    if not is_train:
        return True
    if overlap:
        return any(a["category_id"] in new_classes for a in anno)
    is_new = False
    for a in anno:
        if a["category_id"] in new_classes:
            is_new = True
        if a["category_id"] not in seen_classes:
            return False
    return is_new


class VOCInstanceSegmentation:
    """dataset/voc.py:218-327. Train items: (image u8 [H,W,3], seg u8 [H,W], instance-id u8 [H,W], image-level labels);
    val/test items: (image, seg, masks u8 [R,H,W], mask labels [R], file name)."""

    def __init__(self, data_dir, ann_file, old_classes, new_classes, is_train=True, val_on_trainset=False, pseudo=None,
                 overlap=True, as_coco=False, masking=True):
        self.root = os.path.join(data_dir, "voc", "JPEGImages")
        self.coco = CocoIndex(ann_file)
        self.ids = sorted(self.coco.imgs.keys())
        self.is_train, self.val_on_trainset = is_train, val_on_trainset
        self.old_classes, self.new_classes = list(old_classes), list(new_classes)
        self.pseudo, self.masking, self.as_coco = pseudo, masking, as_coco
        training = is_train or val_on_trainset
        self.indices = []
        for img_id in self.ids:
            anno = self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id, iscrowd=False))
            if has_valid_annotation(anno) and check_if_insert(anno, overlap, self.new_classes + self.old_classes,
                                                              self.new_classes, training):
                self.indices.append(img_id)
        self.num_img = len(self.indices)
        print(f"number of images used for {'training' if is_train else 'testing'}: {self.num_img}")

    def _load_image(self, id):
        path = self.coco.loadImgs(id)[0]["file_name"]
        with Image.open(os.path.join(self.root, path)) as im:
            return np.array(im.convert("RGB"), dtype=np.uint8), path

    def __getitem__(self, index):
        id = self.indices[index]
        img, path = self._load_image(id)
        anno = self.coco.loadAnns(self.coco.getAnnIds(id))
        if self.is_train and not self.val_on_trainset:
            if self.masking:
                anno = image_annotation(anno, self.new_classes)
        else:
            masks = np.stack([self.coco.annToMask(a) for a in anno])
            cats = np.array([a["category_id"] for a in anno])
            seg = np.max(masks * cats[:, None, None], axis=0).astype(np.uint8)
            return img, seg, masks, cats, path
        masks = np.stack([self.coco.annToMask(a) for a in anno]).astype(np.int64)
        cats = np.array([a["category_id"] for a in anno])
        seg = np.max(masks * cats[:, None, None], axis=0).astype(np.uint8)
        inst = np.max(masks * (np.arange(len(anno)) + 1)[:, None, None], axis=0).astype(np.uint8)
        lbls = np.zeros((len(self.old_classes) + len(self.new_classes),))
        lbls[np.unique(cats) - 1] = 1
        if self.pseudo is not None:                      # instance pseudo labels of an earlier run (dataset/voc.py:309-322)
            name = path.split(".")[0]
            f = np.load(f"data/voc/{self.pseudo}/ins_seg_{self.pseudo}/{name}.npy", allow_pickle=True).item()
            m = f["mask"].astype(np.uint8)
            inst = np.max(m * (np.arange(m.shape[0], dtype=np.uint8) + 1)[:, None, None], axis=0)
            seg = np.max(m * (np.asarray(f["class"]).astype(np.uint8) + 1)[:, None, None], axis=0)
        return img, seg, inst, lbls

    def __len__(self):
        return len(self.indices)


class VOCInstanceSegmentationIncremental(IncrementalInstanceSegmentationDataset):
    """dataset/voc.py:334-337."""

    def make_dataset(self, root, ann_file, old_classes, new_classes, is_train, val_on_trainset, pseudo=None, overlap=True,
                     indices=None, masking=True):
        return VOCInstanceSegmentation(root, ann_file, old_classes, new_classes, is_train, val_on_trainset, pseudo,
                                       overlap=overlap, masking=masking)


class VOCasCOCOSegmentationIncremental(IncrementalInstanceSegmentationDataset):
    """dataset/voc.py:339-343 (coco-voc steps > 0: VOC images under the COCO class ids)."""

    def make_dataset(self, root, ann_file, old_classes, new_classes, is_train, val_on_trainset, pseudo=None, overlap=True,
                     indices=None, masking=True):
        return VOCInstanceSegmentation(root, ann_file, old_classes, new_classes, is_train, val_on_trainset, pseudo,
                                       overlap=overlap, as_coco=True, masking=masking)
