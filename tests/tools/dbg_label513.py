"""Debug: peak_extract at 513x513 vs the reference fixture label_s07."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import label_inputs_for
from cl4wsis_amd import ops
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "label_s07.npz"))
gpu = torch.device("cuda:0")
seed, b, H, W = (int(v) for v in gold["case"])
inp = label_inputs_for(gold)
d = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in inp.items()}
oc = inp["old_classes"]
label_new = d["l1h"][:, oc - 1:].contiguous()
cam = ops.cam_smooth(d["int_masks"], d["pg_w"], d["pg_b"], label_new)
print("cam err", (cam.cpu() - torch.from_numpy(gold["cam_lo"][:, oc - 1:])).abs().max().item())
peaks = ops.peak_extract(cam, label_new, (H, W))
conf, ys, xs = (t.cpu().numpy() for t in peaks)
gc, gy, gx = gold["peak_conf"][:, oc - 1:], gold["peak_y"][:, oc - 1:], gold["peak_x"][:, oc - 1:]
for i in range(b):
    for c in range(conf.shape[1]):
        if label_new[i, c].item() == 0:
            continue
        print("img", i, "cls", c)
        print("  hip ", [(round(float(a), 5), int(y), int(x)) for a, y, x in zip(conf[i, c][:10], ys[i, c][:10], xs[i, c][:10])])
        print("  gold", [(round(float(a), 5), int(y), int(x)) for a, y, x in zip(gc[i, c][:10], gy[i, c][:10], gx[i, c][:10])])
