import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import os, sys
os.environ["CL4_AUTOTUNE"] = "0"
import torch
from util import p1_opts
from cl4wsis_amd.train import Trainer
from cl4wsis_amd.phase1 import Phase1Step
from oracle import step as ostep, weights, wss as W, model as M
class _Log:
    def info(self, *a, **k): pass
    debug = info
gpu = torch.device("cuda:0")
crop = 65
opts = p1_opts(crop); opts.max_iters, opts.start_decay = 1000, 1000
torch.manual_seed(0)
tr = Trainer(_Log(), gpu, opts)
sds = []
for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
    sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=21 + i)
    m.load_state_dict(sd); sds.append({k: v.clone() for k, v in sd.items()})
tr._realias()
new, old, pl, pg = sds
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 3, crop, crop, generator=g)
l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
tr._p1 = Phase1Step(tr)
tr.model.train()
dbg = {}
tr._p1.run(x.to(gpu), l1h.to(gpu), 0, flac_angle=180, debug=dbg)
torch.cuda.synchronize()
rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
for dt in (torch.float64, torch.float32):
    c = lambda d: {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in d.items()}
    n_, o_, p_, g_ = c(new), c(old), c(pl), c(pg)
    xx = x.to(dt)
    with torch.no_grad():
        outs_old = M.resnet_body(xx, o_)
        old_seg = M.incremental_classifier(M.deeplab_head(outs_old[-1], o_, pooling=opts.pooling), o_)
    res5 = M.resnet_body(xx, n_, training=True)[-1].detach()
    feat3 = torch.cat([res5, res5.flip(-1), W.rot90k(res5, 180)], 0)
    for k in p_: 
        if p_[k].is_floating_point(): p_[k].requires_grad_(False)
    A = M.pseudolabeler(feat3, p_, training=True).detach().requires_grad_(True)
    raw = A[:2]
    lab = l1h[:, 15:].to(dt)
    terms = {"flac": W.flac_loss(A, 2, 16, 180), "ngwp": W.bce_loss(raw, lab), "peak": W.peakgen_train_loss(raw, g_["extra_conv4.weight"], g_["extra_conv4.bias"], lab)[0],
             "lloc": W.l_loc(raw, old_seg, 16)}
    tot = sum(terms.values())
    tot.backward()
    print(str(dt), "res5", rel(dbg["res5"], res5), "res5_old", rel(dbg["res5_old"], outs_old[-1]), "old_seg", rel(dbg["old_seg"], old_seg),
          "A", rel(dbg["A"], A.detach()), "dA", rel(dbg["dA"], A.grad))
    # per-loss gradient using OUR A as input (isolates the loss kernels from upstream differences)
    A2 = dbg["A"].cpu().to(dt).requires_grad_(True)
    t2 = W.flac_loss(A2, 2, 16, 180) + W.bce_loss(A2[:2], lab) + W.peakgen_train_loss(A2[:2], g_["extra_conv4.weight"], g_["extra_conv4.bias"], lab)[0] + \
        W.l_loc(A2[:2], dbg["old_seg"].cpu().to(dt), 16)
    t2.backward()
    print("    dA given our A / old_seg:", rel(dbg["dA"], A2.grad))
