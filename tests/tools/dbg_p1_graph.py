"""Debug: phase-1 losses per iteration, eager vs eager vs graph replay (same seeds)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import QuietLogger, p1_opts  # noqa: E402
from cl4wsis_amd.train import Trainer  # noqa: E402
from oracle import weights  # noqa: E402

gpu = torch.device("cuda", 0)
crop = 65
g = torch.Generator().manual_seed(23)
batches = [(torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[15 + i % 5, 19 - i % 3]].clone()) for i in range(6)]


def run(use_graph):
    opts = p1_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), gpu, opts)
    tr._use_graph = use_graph
    tr._cur_epoch = opts.pseudo_ep
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=41 + i))
    tr._realias()
    np.random.seed(11)
    out = []
    for b in batches:
        st = np.random.get_state()[2]
        l = {k: float(v) for k, v in tr.step(b)["losses"].items()}
        out.append((st, l))
    return out


a, b, c = run(False), run(False), run(True)
for it in range(len(batches)):
    print("it", it, "rng pos", a[it][0], b[it][0], c[it][0])
    for k in a[it][1]:
        print(f"   {k:10s} eager {a[it][1][k]:.6e}  eager2 {b[it][1][k]:.6e}  graph {c[it][1][k]:.6e}")
