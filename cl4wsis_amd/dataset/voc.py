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
        # `as_coco` is stored and never read, exactly as in the reference: the coco-voc protocol uses VOCSegmentation below
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


class VOCSegmentation:
    """dataset/voc.py:46-172 — the file-list VOC source: `splits/{train_aug,val}[_ascoco].txt` rows "/JPEGImages/x.jpg
    /SegmentationClassAug[AsCoco]/x.png", instance ids from SegmentationObjectAug, image-level labels from
    `{voc,cocovoc}_1h_labels_{train,val}.npy`, optional image subset `indices`. With `as_coco` the semantic maps and the
    label vectors are the pre-remapped COCO-id versions (91-way vectors), which is what coco-voc steps > 0 train on.
    Train items: (image u8, seg u8, instance-id u8, labels); otherwise (image, seg, masks [R,H,W], mask labels [R], name)."""

    def __init__(self, root, train=True, transform=None, indices=None, as_coco=False, saliency=False, pseudo=None):
        self.root = os.path.expanduser(root)
        self.transform, self.is_train, self.pseudo, self.as_coco = transform, train, pseudo, as_coco
        self.image_set = "train" if train else "val"
        voc_root = os.path.join(self.root, "voc")
        if not os.path.isdir(voc_root):
            raise RuntimeError("Dataset not found or corrupted. Download it")
        assert os.path.exists(os.path.join(voc_root, "SegmentationClassAug")), "SegmentationClassAug not found"
        name = ("train_aug" if train else "val") + ("_ascoco" if as_coco else "") + ".txt"
        split_f = os.path.join(voc_root, "splits", name)
        if not os.path.exists(split_f):
            raise ValueError('Wrong image_set entered! Please use image_set="train" or image_set="trainval" or image_set="val"')
        with open(split_f) as f:
            rows = [x[:-1].split(" ") for x in f.readlines()]
        self.file_names = [x[0][1:].split("/")[1] for x in rows]
        ori = "SegmentationClassAugAsCoco" if as_coco else "SegmentationClassAug"
        self.images = [(os.path.join(voc_root, x[0][1:]), os.path.join(voc_root, x[1][1:]),
                        os.path.join(voc_root, x[1][1:].replace(ori, "SegmentationObjectAug"))) for x in rows]
        if pseudo is not None and train:
            rep = (ori, f"pseudo_data/{pseudo}/seg_{pseudo}_AsCoco") if as_coco else (ori, f"PseudoLabels/{pseudo}/rw/")
            self.images = [(a, b.replace(*rep), c) for a, b, c in self.images]
        lab = ("cocovoc" if as_coco else "voc") + f"_1h_labels_{self.image_set}.npy"
        self.img_lvl_labels = np.load(os.path.join(voc_root, lab))
        self.indices = indices if indices is not None else np.arange(len(self.images))

    def __getitem__(self, index):
        k = self.indices[index]
        with Image.open(self.images[k][0]) as im:
            img = np.array(im.convert("RGB"), dtype=np.uint8)
        with Image.open(self.images[k][1]) as im:
            seg = np.array(im)
        with Image.open(self.images[k][2]) as im:
            inst = np.array(im)
        name = self.file_names[k].split(".")[0]
        if not self.is_train:
            masks, labels = [], []
            for idx in np.unique(inst):
                if idx != 0 and idx != 255:
                    masks.append((inst == idx).astype(int))
                    cls = np.unique(seg[inst == idx])
                    assert len(cls) == 1, f"{cls}, {self.images[k][1]}, {self.images[k][2]}"
                    labels.append(cls[0])
            return img, seg.astype(np.uint8), np.stack(masks), np.array(labels), name
        if self.pseudo is not None:
            f = np.load(f"data/voc/{self.pseudo}/ins_seg_{self.pseudo}/{name}.npy", allow_pickle=True).item()
            m = f["mask"].astype(np.uint8)
            inst = np.max(m * (np.arange(m.shape[0], dtype=np.uint8) + 1)[:, None, None], axis=0)
        return img, seg.astype(np.uint8), inst.astype(np.uint8), self.img_lvl_labels[k]

    def __len__(self):
        return len(self.indices)


class VOCasCOCOSegmentationIncremental(IncrementalInstanceSegmentationDataset):
    """dataset/voc.py:339-343 — coco-voc steps > 0: VOC images whose semantic maps / label vectors carry COCO class ids
    (pre-remapped files, VOCSegmentation(as_coco=True)); the training subset is `indices` (coco-voc/voc/train-<step>.npy)."""

    def make_dataset(self, root, ann_file, old_classes, new_classes, is_train, val_on_trainset, pseudo=None, overlap=True,
                     indices=None, masking=True):
        return VOCSegmentation(root, is_train, transform=None, indices=indices, as_coco=True, saliency=None, pseudo=pseudo)
