"""GPU: the metric workload at its REAL size (BASELINE config 3: 513x513, batch 16) — the size at which the autotuned
main + split-tail conv plans, the LDS-band choices of the label kernels and the device-side tables actually engage.
Smaller crops (65..193) are what the rest of the suite runs; bench.py runs this size but checks nothing."""
import numpy as np
import pytest
import torch

from util import QuietLogger, p2_opts, rel_err

pytestmark = pytest.mark.gpu

CROP = 513


@pytest.fixture(scope="module")
def gpu():
    from cl4wsis_amd import _hip
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _hip.load()
    return torch.device("cuda:0")


def _bench_batch(b, seed):
    """bench.py's synthetic batch: images ~ N(0,1); 1 new class w.p. 0.6 / 2 w.p. 0.4."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(b, 3, CROP, CROP, generator=g)
    l1h = torch.zeros(b, 20)
    for i in range(b):
        k = 1 if torch.rand(1, generator=g).item() < 0.6 else 2
        for c in torch.randperm(5, generator=g)[:k].tolist():
            l1h[i, 15 + c] = 1
    return images, l1h


def _trainer(gpu, lr=None, seed0=31):
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    opts = p2_opts(CROP)
    if lr is not None:
        opts.lr = lr / 10
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), gpu, opts)
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=seed0 + i)
        m.load_state_dict(sd)
        sds.append(sd)
    tr._realias()
    return tr, opts, sds


def test_full_size_step_graph_autotuned_matches_eager_heuristic(gpu, monkeypatch):
    """Trainer.step at 513x513, batch 16, the way bench.py and run.py execute it (iteration 0 eager + conv autotuning,
    iteration 1 captured into a HIP graph, iteration 2 replayed) against an eager run on the heuristic launch plans: same
    losses (1e-3), finite, no label-table overflow, weights moved."""
    LR = 5e-6     # see test_graph_replay_matches_eager: keeps the two runs' discrete label decisions aligned
    batches = [_bench_batch(16, 42 + i) for i in range(3)]

    def run(graph):
        monkeypatch.setenv("CL4_GRAPH", "1" if graph else "0")
        monkeypatch.setenv("CL4_AUTOTUNE", "1" if graph else "0")
        tr, opts, _ = _trainer(gpu, lr=LR)
        w0 = torch.cat([p.detach().flatten() for p in tr._branch.trainable_parameters()]).clone()
        losses = [tr.step(b)["losses"].clone() for b in batches]
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        counters = tr.check_label_overflow()          # raises on overflow
        w = torch.cat([p.detach().flatten() for p in tr._branch.trainable_parameters()])
        assert torch.isfinite(w).all() and not torch.equal(w, w0)
        return torch.stack(losses).cpu(), counters

    lg, cg = run(True)
    torch.cuda.empty_cache()
    le, ce = run(False)
    print("FULLSIZE losses graph", lg.tolist(), "eager", le.tolist(), "counters", cg, ce)
    assert torch.isfinite(lg).all() and torch.isfinite(le).all()
    assert cg["overflow"] == 0 and ce["overflow"] == 0
    assert cg["n_contour"] > 0
    assert torch.allclose(lg, le, rtol=1e-3, atol=1e-6), (lg, le)


def test_full_size_step_matches_oracle(gpu):
    """The whole phase-2 iteration at 513x513, batch 2 (one image with one new class, one with two) against the CPU oracle's
    p2_step on identical weights / inputs / Dropout mask: predictions, pseudo ground truth, four losses, Adam update."""
    from oracle import step as ostep
    tr, opts, (new, old, pl, pg) = _trainer(gpu, seed0=11)
    new, old, pl, pg = ({k: v.clone() for k, v in d.items()} for d in (new, old, pl, pg))
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, CROP, CROP, generator=g)
    l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
    keep = torch.rand(2, 256, 33, 33, generator=g) >= 0.5
    before = {k: v.clone() for k, v in new.items()}
    ref = ostep.p2_step(new, old, pl, pg, x, l1h, 16, pooling=opts.pooling, dropout_mask=keep.float(), lr=opts.lr * opts.lr_head)
    out = tr.step((x, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
    torch.cuda.synchronize()
    assert tr.check_label_overflow()["overflow"] == 0
    mism = out["seg_gt"].cpu().long() != ref["seg_gt"]
    if mism.any():
        # An arg-max may differ only where the oracle's own top-2 probabilities are closer than the network tolerance allows
        # to resolve (seg logits agree to ~1e-4 relative after 100+ layers; over 2 x 513^2 pixels the closest calls sit at a
        # gap of ~1e-5), and only on a handful of pixels.
        top2 = ref["soft"].topk(2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1])[mism]
        print("FULLSIZE seg_gt mismatches", int(mism.sum()), "max oracle top-2 gap", float(gap.max()))
        assert gap.max() < 5e-5 and int(mism.sum()) <= 16
    assert rel_err(out["center"].cpu(), ref["pred_center"]) < 1e-3
    assert rel_err(out["offset"].cpu(), ref["pred_offset"]) < 1e-3
    got = out["losses"].cpu().tolist()
    print("FULLSIZE oracle losses", got, [float(ref[k].detach()) for k in ("center_old", "offset_old", "center_new", "offset_new")])
    for v, k in zip(got, ("center_old", "offset_old", "center_new", "offset_new")):
        r = float(ref[k].detach())
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)
    params = dict(tr.model.named_parameters())
    agree, total = 0, 0
    for k in ostep.trainable_names(new):
        du = (params[k].detach().cpu() - before[k])
        dr = (new[k] - before[k])
        big = ref["grads"][k].abs() > 1e-3 * ref["grads"][k].abs().max()
        agree += int(((du - dr).abs()[big] < 0.05 * opts.lr * opts.lr_head).sum())
        total += int(big.sum())
    assert agree / total > 0.995, (agree, total)
