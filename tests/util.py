import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def p2_opts(crop=513, extra=()):
    """Options of BASELINE config 3 (VOC 15-5, step 1, phase 2), scripts/run.sh:39-40 + --crop_size."""
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "2", "--alpha", "0.5",
            "--loss_de", "1", "--lr_policy", "warmup", "--affinity", "--optim", "adam", "--weight_decay", "0",
            "--lr", "5e-5", "--crop_size", str(crop), "--batch_size", "16", "--no_pretrained"] + list(extra)
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))


class QuietLogger:
    def info(self, *a, **k):
        pass

    debug = add_scalar = info


def rel_err(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def p1_opts(crop=513, extra=()):
    """Options of BASELINE config 2 (VOC 15-5, step 1, phase 1), scripts/run.sh:31-33 + --crop_size."""
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "1", "--alpha", "0.5",
            "--loss_de", "1", "--lr_policy", "warmup", "--affinity", "--optim", "sgd", "--lr", "0.001", "--epochs", "40",
            "--crop_size", str(crop), "--batch_size", "16", "--no_pretrained"] + list(extra)
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))


def s0_opts(crop=513, extra=()):
    """Options of BASELINE config 1 (VOC 15-5, step 0, supervised), scripts/run.sh:24 + --crop_size."""
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "0", "--bce", "--lr", "5e-5", "--epochs", "100", "--optim", "adam",
            "--weight_decay", "0", "--crop_size", str(crop), "--batch_size", "16", "--no_pretrained"] + list(extra)
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))


def make_mini_voc(root, n_train=6, n_val=3, seed=0):
    """A tiny Pascal-VOC-shaped tree under `root` (voc/JPEGImages/*.png + COCO-format pascal_sbd_{train,val}.json with
    uncompressed-RLE instance masks) built from oracle.synth.dataset_sample. Returns {split: [(img, seg, inst, l1h), ...]}."""
    import json
    import os
    import numpy as np
    from PIL import Image
    from oracle import synth
    os.makedirs(os.path.join(root, "voc", "JPEGImages"), exist_ok=True)
    out = {}
    img_id = ann_id = 0
    for split, n in (("train", n_train), ("val", n_val)):
        images, anns, samples = [], [], []
        for k in range(n):
            H, W = 90 + 10 * (k % 3), 120 + 7 * k
            classes = (3, 16, 17, 20) if k % 3 else (2, 5)          # every third photo holds old classes only
            img, seg, inst, l1h = synth.dataset_sample(seed * 100 + img_id, H, W, n_inst=3, classes=classes, void=False)
            img_id += 1
            name = f"{split}_{k:03d}.png"
            Image.fromarray(img).save(os.path.join(root, "voc", "JPEGImages", name))
            images.append({"id": img_id, "file_name": name, "height": H, "width": W})
            keep = []
            for iid in np.unique(inst):
                if iid == 0:
                    continue
                m = inst == iid
                flat = m.T.reshape(-1).astype(np.int8)               # column-major
                edges = np.flatnonzero(np.diff(np.concatenate([[0], flat, [0]])))
                counts = np.diff(np.concatenate([[0], edges, [flat.size]])).tolist()
                ys, xs = np.where(m)
                ann_id += 1
                anns.append({"id": ann_id, "image_id": img_id, "category_id": int(seg[m][0]), "iscrowd": 0,
                             "bbox": [int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)],
                             "area": int(m.sum()), "segmentation": {"size": [H, W], "counts": counts}})
                keep.append(int(iid))
            # instance ids as the dataset numbers them: position of the annotation in the image's list, 1-based
            renum = np.zeros(256, dtype=np.uint8)
            for pos, iid in enumerate(keep):
                renum[iid] = pos + 1
            samples.append((img, seg, renum[inst], l1h))
        with open(os.path.join(root, "voc", f"pascal_sbd_{split}.json"), "w") as f:
            json.dump({"images": images, "annotations": anns, "categories": [{"id": i} for i in range(1, 21)]}, f)
        out[split] = samples
    return out


def make_mini_coco(root, n_train=5, n_val=2, seed=3):
    """A tiny COCO-2017-shaped tree under `root`/coco (split/*.txt, annotations/instances_*2017.json with polygon AND RLE
    masks, images/*2017/*.jpg). Category ids: 8, 10, 11 (non-VOC, step 0 of coco-voc) and 1, 2 (VOC classes)."""
    import json
    import os
    import numpy as np
    from PIL import Image
    out = {}
    rs = np.random.RandomState(seed)
    img_id = ann_id = 0
    for split, n in (("train", n_train), ("val", n_val)):
        os.makedirs(os.path.join(root, "coco", "split"), exist_ok=True)
        os.makedirs(os.path.join(root, "coco", "annotations"), exist_ok=True)
        os.makedirs(os.path.join(root, "coco", "images", f"{split}2017"), exist_ok=True)
        images, anns, names, truth = [], [], [], []
        for k in range(n):
            H, W = 80 + 8 * k, 100 + 5 * k
            img_id += 1
            name = f"{img_id:012d}"
            Image.fromarray(rs.randint(0, 256, (H, W, 3)).astype(np.uint8)).save(os.path.join(root, "coco", "images", f"{split}2017", name + ".jpg"), quality=95)
            images.append({"id": img_id, "file_name": name + ".jpg", "height": H, "width": W})
            names.append(name)
            boxes = []
            cats = [8, 10, 11] if k % 2 == 0 else [1, 8]
            for j, cat in enumerate(cats):                      # disjoint axis-aligned boxes: polygon for even j, RLE for odd j
                x0, y0, w, h = 5 + 30 * j, 10 + 5 * j, 20, 25 + 3 * k
                ann_id += 1
                if j % 2 == 0:
                    segm = [[x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h]]
                else:
                    m = np.zeros((H, W), dtype=np.int8)
                    m[y0:y0 + h, x0:x0 + w] = 1
                    flat = m.T.reshape(-1)
                    edges = np.flatnonzero(np.diff(np.concatenate([[0], flat, [0]])))
                    segm = {"size": [H, W], "counts": np.diff(np.concatenate([[0], edges, [flat.size]])).tolist()}
                anns.append({"id": ann_id, "image_id": img_id, "category_id": cat, "iscrowd": 0, "bbox": [x0, y0, w, h],
                             "area": w * h, "segmentation": segm})
                boxes.append((cat, x0, y0, w, h))
            truth.append((H, W, boxes))
        with open(os.path.join(root, "coco", "split", f"{split}.txt"), "w") as f:
            f.write("".join(nm + "\n" for nm in names))
        with open(os.path.join(root, "coco", "annotations", f"instances_{split}2017.json"), "w") as f:
            json.dump({"images": images, "annotations": anns, "categories": [{"id": i} for i in range(1, 92)]}, f)
        out[split] = truth
    return out


def label_inputs_for(gold):
    """Inputs of a tests/golden/label_*.npz fixture, regenerated from its seed (label_sXX) or scene name (label_edge_X)."""
    from oracle import synth
    if "edge_name" in gold:
        return synth.label_stage_edge_inputs(str(gold["edge_name"]))
    seed, b, H, W = (int(v) for v in gold["case"])
    return synth.label_stage_inputs(seed, b=b, H=H, W=W)


def planes_split_ref(x):
    """The two-plane split of the loaded library's build, restated in torch (csrc/planes_fmt.h): (hi, lo) as fp32 tensors with
    x ~ hi + lo. fp16x3 build (the default): hi = fp16(16 x) / 16, lo = fp16(16 x - fp16(16 x)) / 16 — saturating at |16 x| = 65504;
    bf16x3 build (CL4_F16_PLANES=0): hi = bf16(x), lo = bf16(x - hi)."""
    import torch
    from cl4wsis_amd import ops
    if ops.planes_f16():
        a = (x.float() * 16.0).clamp(-65504.0, 65504.0)
        hi = a.to(torch.float16).float()
        lo = (a - hi).to(torch.float16).float()
        return hi / 16.0, lo / 16.0
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def planes_raw_ref(x):
    """(hi, lo) of planes_split_ref as the 16-bit patterns the library stores (int16 views of fp16 / bf16)."""
    import torch
    from cl4wsis_amd import ops
    hi, lo = planes_split_ref(x)
    if ops.planes_f16():
        return (hi * 16.0).to(torch.float16).view(torch.int16), (lo * 16.0).to(torch.float16).view(torch.int16)
    return hi.to(torch.bfloat16).view(torch.int16), lo.to(torch.bfloat16).view(torch.int16)
