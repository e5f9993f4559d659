"""Is the phase-2 step deterministic run to run? Two plain single-GPU trainers from the same weights on the same three batches (the set-up of
tests/test_golden_gpu.py::test_collective_path_over_rccl_single_rank, no process group): weight and gradient differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from util import p2_opts
from cl4wsis_amd.train import Trainer
from oracle import weights

class _Log:
    def info(self, *a, **k): pass
    debug = info

gpu = torch.device("cuda:0")
crop = 97
g = torch.Generator().manual_seed(18)
batches = [(torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[16 + i % 4, 15]].clone()) for i in range(3)]
LR = 5e-6

def run():
    opts = p2_opts(crop)
    opts.lr = LR / 10
    opts.max_iters, opts.start_decay = 50, 1
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=31 + i))
    tr._realias()
    grads = []
    for b in batches:
        tr.step(b)
        torch.cuda.synchronize()
        grads.append(torch.cat([tr._grad_map[id(p)].flatten() for p in tr._branch.trainable_parameters()]).cpu().clone())
    w = torch.cat([p.detach().flatten() for p in tr._branch.trainable_parameters()]).cpu()
    return w, grads

runs = [run() for _ in range(4)]
for i in range(1, 4):
    d = (runs[0][0] - runs[i][0]).abs()
    gd = [float((a - b).norm() / a.norm()) for a, b in zip(runs[0][1], runs[i][1])]
    nz = [int(((a - b) != 0).sum()) for a, b in zip(runs[0][1], runs[i][1])]
    print(f"run 0 vs run {i}: weights max {float(d.max()) / LR:.2f} lr, beyond 0.4 lr {int((d > 0.4 * LR).sum())}; gradient rel-L2 difference per step {gd}; gradient elements that differ {nz}")
