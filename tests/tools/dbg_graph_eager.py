import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import os, sys
import torch
from util import p2_opts
from cl4wsis_amd.train import Trainer
from oracle import weights
class _Log:
    def info(self, *a, **k): pass
    debug = info
gpu = torch.device("cuda:0")
crop = 97
g = torch.Generator().manual_seed(8)
batches = [(torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[16 + i % 4, 15]].clone()) for i in range(3)]
def run(graph, nsteps=3):
    os.environ["CL4_GRAPH"] = "1" if graph else "0"
    opts = p2_opts(crop)
    opts.max_iters, opts.start_decay = 50, 1
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=31 + i))
    tr._realias()
    ws, gs = [], []
    for x, l in batches[:nsteps]:
        tr.step((x, l))
        torch.cuda.synchronize()
        ws.append(torch.cat([p.detach().flatten() for p in tr._branch.trainable_parameters()]).cpu())
        gs.append(tr.optimizer.grads_fb.flat.detach().cpu().clone() if hasattr(tr.optimizer.grads_fb, "flat") else None)
    return ws, gs, tr
a, ga, tra = run(False)
b, gb, _ = run(False)
c, gc, _ = run(True)
for i in range(3):
    print("step", i, "eager-eager max", (a[i] - b[i]).abs().max().item(), "mean", (a[i] - b[i]).abs().mean().item(),
          "| eager-graph max", (a[i] - c[i]).abs().max().item(), "mean", (a[i] - c[i]).abs().mean().item())
    if ga[i] is not None:
        print("   grads eager-eager", (ga[i] - gb[i]).abs().max().item(), "eager-graph", (ga[i] - gc[i]).abs().max().item(), "gmax", ga[i].abs().max().item())
# which tensors differ
names = [n for n, _ in tra._branch.named_trainable()] if hasattr(tra._branch, "named_trainable") else None
