"""Two-rank rehearsal of the phase-1 step on ONE GPU (gloo backend, both ranks on cuda:0): rank r trains on image r of a
batch of two with cross-rank ABN statistics and the bucketed gradient average; rank 0 then runs the same batch alone
(world-size-1 semantics emulated by a fresh Trainer on the full batch in a second launch) — see --single.
usage: torchrun --nproc-per-node 2 tools/rehearse_p1_ddp.py      |      python tools/rehearse_p1_ddp.py --single"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))

import os, sys
import torch
from util import p1_opts
from cl4wsis_amd.parallel import init_distributed
from cl4wsis_amd.train import Trainer
from oracle import weights

class _Log:
    def info(self, *a, **k): pass
    debug = info

single = "--single" in sys.argv
os.environ.setdefault("CL4_DIST_BACKEND", "gloo")
os.environ["CL4_AUTOTUNE"] = "0"
rank, world, local = (0, 1, 0) if single else init_distributed()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
crop = 65
opts = p1_opts(crop)
opts.max_iters, opts.start_decay = 1000, 1000
torch.manual_seed(0)
tr = Trainer(_Log(), dev, opts)
for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
    m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=21 + i))
tr._realias()
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 3, crop, crop, generator=g)
l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
if not single:
    x, l1h = x[rank:rank + 1], l1h[rank:rank + 1]
out = tr._step_phase1((x, l1h), cur_epoch=0, flac_angle=180)
torch.cuda.synchronize()
vec = torch.cat([p.detach().flatten()[:50] for p in list(tr.model.parameters())[::40]]).cpu()
if rank == 0:
    torch.save({"vec": vec, "rm": tr.model.body.mod5.block3.convs.bn3.running_mean.cpu(), "world": world},
               f"gpurun_out/p1_ddp_{'single' if single else 'two'}.pt")
    print("world", world, "lde", float(out["losses"]["lde"]), "vec norm", vec.norm().item())
