"""Host side of the input pipeline (cl4wsis_amd/dataset): COCO-format index + mask decoding, the incremental image filter,
label mapping / LabelSelection and the augmentation plans — against the oracle restatement (oracle/dataset.py, pinned by the
reference fixtures in test_oracle_dataset.py). No GPU: RawSamples are inspected before the device stage."""
import random

import numpy as np

from cl4wsis_amd import argparser
from cl4wsis_amd.dataset import coco_index as ci
from oracle import dataset as ods
from util import make_mini_voc


def _opts(root, step=1, extra=()):
    p = argparser.get_argparser()
    o = p.parse_args(["--dataset", "voc", "--task", "15-5", "--step", str(step), "--data_root", str(root), "--crop_size", "65",
                      "--crop_size_val", "70", *extra])
    return argparser.modify_command_options(o)


def test_rle_codecs_roundtrip_and_polygon():
    rs = np.random.RandomState(0)
    m = (rs.rand(17, 23) > 0.6).astype(np.uint8)
    flat = m.T.reshape(-1)
    edges = np.flatnonzero(np.diff(np.concatenate([[0], flat.astype(np.int8), [0]])))
    counts = np.diff(np.concatenate([[0], edges, [flat.size]]))
    assert np.array_equal(ci.ann_to_mask({"size": [17, 23], "counts": counts.tolist()}, 17, 23), m)
    # compressed string: encode with the published scheme (5 bits per char, continuation 0x20, delta to cnts[i-2] from i > 2)
    s = ""
    for i, c in enumerate(counts.tolist()):
        x = c - int(counts[i - 2]) if i > 2 else c
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = not ((x == 0 and not (ch & 0x10)) or (x == -1 and (ch & 0x10)))
            s += chr((ch | (0x20 if more else 0)) + 48)
    assert np.array_equal(ci.ann_to_mask({"size": [17, 23], "counts": s}, 17, 23), m)
    # polygons: an axis-aligned rectangle covers [x0, x1) x [y0, y1); two polygons are united
    r = ci.ann_to_mask([[2, 3, 10, 3, 10, 8, 2, 8]], 12, 14)
    assert r.sum() == 40 and r[3:8, 2:10].all()
    u = ci.ann_to_mask([[2, 3, 10, 3, 10, 8, 2, 8], [0, 0, 3, 0, 3, 4, 0, 4]], 12, 14)
    assert u.sum() == 40 + 12 - 1 and u[0:4, 0:3].all()


def test_incremental_filter_and_items(tmp_path):
    from cl4wsis_amd.dataset import get_dataset
    truth = make_mini_voc(tmp_path)
    train, val, test, labels_cum, n = get_dataset(_opts(tmp_path))
    assert n == 21 and labels_cum == list(range(21))
    # overlap protocol, step 1: only photos that contain a new class (16..20) are kept for training; val keeps everything
    assert len(train) == 4 and len(val) == 3 and len(test) == 3
    kept = [s for k, s in enumerate(truth["train"]) if k % 3]
    mapping, order, labels = ods.label_mapping({0: list(range(16)), 1: list(range(16, 21))}, 1, True, True)
    assert np.array_equal(train.mapping, mapping) and train.total_classes == 20
    for k in range(len(train)):
        img, seg, inst, l1h = kept[k]
        random.seed(100 + k)
        raw = train[k]
        random.seed(100 + k)
        i, j, h, w = ods.random_resized_crop_params(img.shape[1], img.shape[0])
        flip = random.random() < 0.5
        assert raw.plan.crop == (i, j, h, w) and raw.plan.flip == flip and raw.plan.out_hw == (65, 65)
        assert np.array_equal(raw.image, img)
        # masking=True drops the old-class annotations before the maps are built (dataset/voc.py:291-293)
        new = seg >= 16
        assert np.array_equal(raw.label[:, :, 0], np.where(new, seg, 0))
        assert np.array_equal(raw.label[:, :, 1] > 0, new)
        assert np.array_equal(raw.l1h, ods.label_selection(l1h, order, labels, True))
    v = val[1]
    img, seg, inst, _ = truth["val"][1]
    assert v.plan.out_hw == ods.resize_output_size(img.shape[1], img.shape[0], 70) and v.plan.crop == (0, 0) + img.shape[:2]
    assert np.array_equal(v.label[:, :, 0], seg) and v.masks.shape[0] == len(v.mask_label) == len(np.unique(inst)) - 1
    assert v.path == "val_001.png"


def test_disjoint_protocol_rejects_future_classes(tmp_path):
    from cl4wsis_amd.dataset.voc import check_if_insert
    a = [{"category_id": 16}, {"category_id": 3}]
    assert check_if_insert(a, True, list(range(1, 21)), [16, 17])
    assert check_if_insert(a, False, [1, 2, 3, 16, 17], [16, 17])
    assert not check_if_insert(a + [{"category_id": 19}], False, [1, 2, 3, 16, 17], [16, 17])
    assert not check_if_insert([{"category_id": 3}], True, list(range(1, 21)), [16, 17])
    assert check_if_insert([{"category_id": 3}], True, list(range(1, 21)), [16, 17], is_train=False)


def test_coco_source_and_coco_voc_step0(tmp_path):
    """coco-voc step 0 (COCOIncremental over dataset/coco.py's COCO): split files, polygon + RLE masks, the 91-way image-level
    labels reduced to the protocol's class order, instances of classes outside the step gated away by the label map."""
    import os
    from cl4wsis_amd.dataset import get_dataset
    from cl4wsis_amd import tasks
    from util import make_mini_coco
    truth = make_mini_coco(tmp_path)
    p = argparser.get_argparser()
    o = argparser.modify_command_options(p.parse_args(["--dataset", "coco-voc", "--task", "voc", "--step", "0", "--data_root", str(tmp_path),
                                                       "--crop_size", "65", "--crop_size_val", "70"]))
    base = os.path.join(str(tmp_path), "coco-voc", "voc")
    os.makedirs(base, exist_ok=True)
    np.save(os.path.join(base, "train-0.npy"), np.arange(5))            # the protocol's image subset (here: everything)
    train, val, test, labels_cum, n = get_dataset(o)
    td = tasks.get_task_dict("coco-voc", "voc", 0)
    assert n == len(td[0]) == 61 and len(train) == 5 and len(val) == 2 and len(test) == 2
    import pickle
    for ds in (train, val, test):                       # loader workers are spawned (worker_options): the dataset has to pickle
        assert len(pickle.loads(pickle.dumps(ds))) == len(ds)
    random.seed(0)
    raw = train[0]
    H, W, boxes = truth["train"][0]
    assert raw.image.shape == (H, W, 3) and raw.label.shape == (H, W, 2) and raw.l1h.shape == (60,)
    for j, (cat, x0, y0, w, h) in enumerate(boxes):
        assert (raw.label[y0:y0 + h, x0:x0 + w, 0] == cat).all() and (raw.label[y0:y0 + h, x0:x0 + w, 1] == j + 1).all()
    assert (raw.label[:, :, 0] > 0).sum() == sum(w * h for (_, _, _, w, h) in boxes)
    # image-level labels: position of each present class in the protocol order (background dropped), classes of later steps masked
    order = [c for s in sorted(td) for c in td[s]]
    for cat in (8, 10, 11):
        assert raw.l1h[order.index(cat) - 1] == 1
    assert raw.l1h.sum() == 3
    # second photo holds a VOC class (id 1, a later step): its pixels map to 0 through the LabelTransform LUT
    raw1 = train[1]
    assert train.lut[1] == 0 and train.lut[8] == order.index(8)
    v = val[0]
    assert v.masks.shape[0] == len(v.mask_label) == len(truth["val"][0][2]) and v.path.endswith(".jpg")


def _make_mini_voc_ascoco(root, n_train=4, n_val=2):
    """The file-list VOC tree the coco-voc protocol reads for steps > 0 (reference dataset/voc.py:46-131 with as_coco=True):
    JPEGImages, SegmentationClassAugAsCoco (semantic maps in COCO ids), SegmentationObjectAug (instance ids),
    splits/{train_aug,val}_ascoco.txt, cocovoc_1h_labels_{train,val}.npy (91-way)."""
    import os
    from PIL import Image
    from cl4wsis_amd import tasks
    from oracle import synth
    coco_of_voc = tasks.get_task_dict("coco-voc", "voc", 1)[1]           # COCO ids of VOC classes 1..20, in VOC order
    voc = os.path.join(str(root), "voc")
    for d in ("JPEGImages", "SegmentationClassAug", "SegmentationClassAugAsCoco", "SegmentationObjectAug", "splits"):
        os.makedirs(os.path.join(voc, d), exist_ok=True)
    truth = {}
    for split, n, fname in (("train", n_train, "train_aug_ascoco.txt"), ("val", n_val, "val_ascoco.txt")):
        rows, labels, samples = [], [], []
        for k in range(n):
            img, seg, inst, l1h = synth.dataset_sample(900 + 10 * k + (split == "val"), 70 + 6 * k, 90 + 4 * k, n_inst=3,
                                                       classes=(1, 8, 15, 20), void=False)
            seg_coco = np.zeros_like(seg)
            for v in np.unique(seg):
                if v:
                    seg_coco[seg == v] = coco_of_voc[v - 1]
            name = f"{split}_{k:02d}"
            Image.fromarray(img).save(os.path.join(voc, "JPEGImages", name + ".png"))
            Image.fromarray(seg_coco).save(os.path.join(voc, "SegmentationClassAugAsCoco", name + ".png"))
            Image.fromarray(inst).save(os.path.join(voc, "SegmentationObjectAug", name + ".png"))
            rows.append(f"/JPEGImages/{name}.png /SegmentationClassAugAsCoco/{name}.png\n")
            lab = np.zeros(91)
            lab[[coco_of_voc[v - 1] - 1 for v in np.unique(seg) if v]] = 1
            labels.append(lab)
            samples.append((img, seg_coco, inst, lab))
        with open(os.path.join(voc, "splits", fname), "w") as f:
            f.writelines(rows)
        np.save(os.path.join(voc, f"cocovoc_1h_labels_{split}.npy"), np.stack(labels))
        truth[split] = samples
    return truth


def test_coco_voc_step1_reads_the_as_coco_voc_tree(tmp_path):
    """coco-voc step 1 (BASELINE config 5): the training source is the file-list VOC tree with COCO class ids
    (VOCasCOCOSegmentationIncremental -> VOCSegmentation(as_coco=True), dataset/voc.py:339-343), restricted to the protocol's
    image subset train-1.npy; image-level labels are the 91-way cocovoc vectors reduced by LabelSelection."""
    import os
    from cl4wsis_amd.dataset import get_dataset
    from cl4wsis_amd import tasks
    from util import make_mini_coco
    make_mini_coco(tmp_path)                                            # the test split of coco-voc is COCO val2017
    truth = _make_mini_voc_ascoco(tmp_path)
    p = argparser.get_argparser()
    o = argparser.modify_command_options(p.parse_args(["--dataset", "coco-voc", "--task", "voc", "--step", "1", "--data_root", str(tmp_path),
                                                       "--crop_size", "65", "--crop_size_val", "70"]))
    base = os.path.join(str(tmp_path), "coco-voc", "voc")
    os.makedirs(base, exist_ok=True)
    np.save(os.path.join(base, "train-1.npy"), np.array([2, 0, 3]))      # subset AND order of the protocol
    train, val, test, labels_cum, n = get_dataset(o)
    td = tasks.get_task_dict("coco-voc", "voc", 1)
    order = [c for s in sorted(td) for c in td[s]]
    assert n == 81 and len(train) == 3 and len(val) == 2 and len(test) == 2
    for pos, k in enumerate((2, 0, 3)):
        random.seed(7 + pos)
        raw = train[pos]
        img, seg_coco, inst, lab = truth["train"][k]
        assert np.array_equal(raw.image, img)
        assert np.array_equal(raw.label[:, :, 0], seg_coco) and np.array_equal(raw.label[:, :, 1], inst)
        want = np.zeros(80)
        for cid in np.flatnonzero(lab) + 1:
            want[order.index(cid) - 1] = 1                              # all present ids are VOC classes = the new step
        assert raw.l1h.shape == (80,) and np.array_equal(raw.l1h, want) and raw.l1h[:60].sum() == 0
    # label map: COCO id -> position in the protocol order (61 old classes incl. background first, then the 20 VOC classes)
    assert train.lut[1] == 61 and train.lut[order[-1]] == 80
    v = val[1]
    img, seg_coco, inst, lab = truth["val"][1]
    assert np.array_equal(v.label[:, :, 0], seg_coco) and v.masks.shape[0] == len(v.mask_label) == len(np.unique(inst)) - 1
    assert v.path == "val_01"


def test_loader_workers_are_spawned_and_the_dataset_travels_to_them(tmp_path):
    """cl4wsis_amd.dataset.loader.worker_options: worker processes are spawned (never forked from the GPU-initialised trainer process) and
    persistent, so the dataset object and the collate function must pickle; one epoch through one spawned worker yields the same packed
    batches (photo bytes, labels, image-level labels) as the in-process loader."""
    import torch.utils.data as data
    from cl4wsis_amd.dataset import get_dataset
    from cl4wsis_amd.dataset.loader import collate_raw, worker_options
    make_mini_voc(tmp_path)
    train = get_dataset(_opts(tmp_path))[0]
    opt = worker_options(1)
    assert opt == {"multiprocessing_context": "spawn", "persistent_workers": True} and worker_options(0) == {}
    got = {}
    for tag, kw in (("main", {}), ("worker", opt)):
        loader = data.DataLoader(train, batch_size=2, shuffle=False, num_workers=1 if kw else 0, collate_fn=collate_raw, **kw)
        got[tag] = [(pb.paths, pb.l1h.clone()) for pb in loader]
        del loader
    assert len(got["main"]) == 2 and [p for p, _ in got["main"]] == [p for p, _ in got["worker"]]
    assert all((a == b).all() for (_, a), (_, b) in zip(got["main"], got["worker"]))
