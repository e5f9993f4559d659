"""Device input pipeline timing: one training batch of 16 VOC-sized photos (375x500) -> 513x513 crops,
flip, ToTensor + Normalize, label map + LUT, and (supervised) label_generation — against the oracle (Pillow + numpy on the
host cores, what the reference's loader workers do per sample).
usage: bench_input.py [batch] [iters]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))

import os, sys, time, json, random
import numpy as np
import torch
from cl4wsis_amd import ops
from oracle import dataset as ods, synth

b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
S, H, W = 513, 375, 500
dev = torch.device("cuda:0")
samples = [synth.dataset_sample(k, H, W, n_inst=4, classes=tuple(range(1, 16)), void=True) for k in range(b)]
random.seed(0)
crops = [ods.random_resized_crop_params(W, H) + (int(random.random() < 0.5),) for _ in range(b)]
imgs = torch.from_numpy(np.concatenate([s[0].reshape(-1) for s in samples])).to(dev)
lbls = torch.from_numpy(np.concatenate([np.stack([s[1], s[2]], -1).reshape(-1) for s in samples])).to(dev)
descs = [(k * H * W * 3, k * H * W * 2, H, W) + crops[k] for k in range(b)]
mapping, order, labels = ods.label_mapping({0: list(range(16))}, 0, True, True)
lut = torch.from_numpy(mapping.astype(np.uint8)).to(dev)


def gpu_batch(supervised):
    img, seg, mask = ops.input_batch(imgs, descs, (S, S), labels=lbls, lut=lut, want_mask=supervised)
    if supervised:
        return ops.target_generation(seg, mask, 15, 8)
    return img


out = {}
for name, sup in (("weakly (image + label map)", False), ("supervised (+ label_generation)", True)):
    for _ in range(3):
        gpu_batch(sup)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gpu_batch(sup)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # algorithmic bytes: read the crop's uint8 photo + label once, write the float crop (+ int64 seg, + targets)
    crop_px = sum(c[2] * c[3] for c in crops)
    byts = crop_px * 5 + b * S * S * (12 + 8) + (b * S * S * (1 + 4 * (15 + 2 + 1) + 1) if sup else 0)
    out[name] = {"ms_per_batch": ms, "images_per_s": b / ms * 1e3, "algorithmic_MB": byts / 1e6, "GB_per_s": byts / ms / 1e6,
                 "frac_of_8TBps": byts / ms / 1e6 / 8000}
t0 = time.perf_counter()
n = min(b, 8)
for k in range(n):
    random.seed(k)
    ods.train_item(samples[k][0], samples[k][1], samples[k][2], samples[k][3], {0: list(range(16))}, 0, S, True)
cpu = (time.perf_counter() - t0) / n
out["oracle_cpu"] = {"ms_per_image_1core": cpu * 1e3, "images_per_s_1core": 1 / cpu, "sample": f"{n} supervised items (Pillow + numpy)"}
out["h2d_MB_per_batch"] = {"device_pipeline_uint8": b * H * W * 5 / 1e6, "reference_float_tensors": b * S * S * 4 * (3 + 15 + 2 + 1) / 1e6 + b * S * S * 8 / 1e6}
print(json.dumps(out, indent=1))
