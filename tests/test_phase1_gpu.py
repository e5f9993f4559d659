"""Whole phase-1 iteration (train.py:241-566 with --phase 1; BASELINE config 2) on the GPU against the CPU oracle
(oracle/step.py::p1_step, PyTorch autograd + torch.optim.SGD arithmetic) on identical weights, inputs, FLAC angle and
RandDrop draws: every loss term within 1e-3 relative, and the SGD-Nesterov update of all 300+ trainable tensors
(body, DeepLabV3 head, classifiers, PseudoLabeler, PeakGenerator). Both regimes: before and after `pseudo_ep`."""
import pytest
import torch

from util import p1_opts

pytestmark = pytest.mark.gpu


class _Log:
    def info(self, *a, **k):
        pass
    debug = info


@pytest.mark.parametrize("cur_epoch,precision", [(0, 0), (5, 0), (5, 2)])
def test_phase1_step_matches_oracle(gpu, monkeypatch, cur_epoch, precision):
    """precision 0 = exact fp32 products (the reference's arithmetic; its rounding — and with it the discrete decisions of this
    ill-conditioned tiny-crop step — follows the CPU oracle's closely): the strict statistical bar. precision 2 = bf16x6, the
    default: per-conv error equal to fp32 (tests/test_conv_gpu.py, 2e-6) and the same distance to float64 on the body alone
    (tests/tools/dbg_x6_r101.py: 2.3e-4 forward vs 3.0e-4 for CPU fp32), but its rounding is independent of the CPU's, so at 2x5x5
    samples the thresholded pseudo labels / ReLU masks flip elsewhere: an rms bar relative to the CPU oracle's own distance."""
    from cl4wsis_amd import body_train
    from cl4wsis_amd.train import Trainer
    monkeypatch.setattr(body_train, "PRECISION", precision)
    monkeypatch.setattr(body_train, "WGRAD_PRECISION", 0 if precision == 0 else 1)      # the defaults that go with each precision
    from oracle import step as ostep, weights
    crop = 65
    opts = p1_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    assert tr.model.decoder is None and opts.flac and opts.randrop          # argparser.py:21-24
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=21 + i)
        m.load_state_dict(sd)
        sds.append({k: v.clone() for k, v in sd.items()})
    tr._realias()
    new, old, pl, pg = sds
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, crop, crop, generator=g)
    l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
    neg = torch.randint(0, 16, (2, 5, 5), generator=g)
    before = [{k: v.clone() for k, v in d.items()} for d in (new, pl, pg)]
    kw = dict(cur_epoch=cur_epoch, pseudo_epoch=opts.pseudo_ep, flac_angle=180, randrop_neg=neg, lr=opts.lr, lr_head=opts.lr_head,
              lr_pseudo=opts.lr_pseudo, weight_decay=opts.weight_decay, loss_de=opts.loss_de, alpha=opts.alpha,
              l_seg_weight=opts.l_seg, icarl_bkg=opts.icarl_bkg, pooling=opts.pooling)
    # the reference arithmetic (fp32 on the CPU) and the same step in float64 as the ground truth for the gradients
    to64 = lambda d: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in d.items()}
    new64, old64, pl64, pg64 = to64(new), to64(old), to64(pl), to64(pg)
    ostep.p1_step(new64, old64, pl64, pg64, x.double(), l1h.double(), 16, **kw)
    ref = ostep.p1_step(new, old, pl, pg, x, l1h, 16, **kw)
    out = tr._step_phase1((x, l1h), cur_epoch=cur_epoch, flac_angle=180, randrop_neg=neg.to(torch.int32).to(gpu))
    torch.cuda.synchronize()
    names = ["l_cam_new", "l_loc", "flac", "l_peak", "lde"] + (["l_seg", "l_cls", "l_rand"] if cur_epoch >= opts.pseudo_ep else [])
    for k in names:
        v, r = float(out["losses"][k]), float(ref[k])
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)
    # The update (after - before) of every trainable tensor. Gradients through ~100 batch-statistics ABN layers on 50
    # samples per channel are ill-conditioned at fp32 resolution: the train-mode res5 of BOTH fp32 evaluations
    # (CPU oracle, this GPU path) already sits ~3e-4 from float64, and individual tensors amplify that differently from run
    # to run (conv launch plans change the summation order). The bar is the reference's own arithmetic, statistically:
    # distance to the float64 update relative to the CPU-fp32 oracle's distance — median ratio < 2.5, at most 3 % of the
    # tensors beyond 3x, none beyond 0.2 absolute. (The ratio itself moves between 1.6 and 2.2 with any change of rounding
    # upstream — r02 measured 2.12 after the |gamma| + eps scale of InPlaceABN went in on both sides; the well-conditioned
    # version of this comparison, crop 193 / batch 4 in tests/test_acceptance_gpu.py, has the HIP path at 4.4e-2 and the CPU fp32
    # oracle at 4.8e-2 from float64: at 50 samples per channel this test can only bound, not rank.)
    ratios, worst, num_o, num_c, den = [], 0.0, 0.0, 0.0, 0.0
    for mod, after32, after64, sd_before in ((tr.model, new, new64, before[0]), (tr.pseudolabeler, pl, pl64, before[1]),
                                             (tr.peakgenerator, pg, pg64, before[2])):
        for k, p in mod.named_parameters():
            d64 = after64[k] - sd_before[k].double()
            if d64.norm() == 0:
                continue
            ours = ((p.detach().cpu().double() - sd_before[k].double() - d64).norm() / d64.norm()).item()
            cpu32 = ((after32[k].double() - sd_before[k].double() - d64).norm() / d64.norm()).item()
            ratios.append(ours / (cpu32 + 1e-5))
            worst = max(worst, ours)
            num_o += ours ** 2; num_c += cpu32 ** 2; den += 1
    ratios = torch.tensor(ratios)
    assert len(ratios) > 300
    print("P1STAT epoch", cur_epoch, "median ratio", ratios.median().item(), "q10/q90", ratios.quantile(0.1).item(), ratios.quantile(0.9).item(),
          "frac>3", (ratios > 3).float().mean().item(), "worst", worst, "rms ours", (num_o / den) ** 0.5, "rms cpu32", (num_c / den) ** 0.5)
    rms_ours, rms_cpu = (num_o / den) ** 0.5, (num_c / den) ** 0.5
    if precision == 0:
        assert ratios.median() < 2.5 and (ratios > 3).float().mean() < 0.035 and worst < 0.2, (ratios.median(), (ratios > 3).float().mean(), worst)
    else:
        assert rms_ours < 5 * rms_cpu + 1e-4 and worst < 0.3, (rms_ours, rms_cpu, worst)


def test_phase1_graph_replay_matches_eager(gpu):
    """r04: on one GPU the phase-1 step is HIP-graph replayed from its third call on — one graph per FLAC angle (drawn on the host,
    train.py:316) and side of `pseudo_ep`, RandDrop labels from a device-resident counter. Two eager runs of this tiny configuration
    (crop 65, batch 2, SGD at lr 1e-2 on the heads) already differ by several per cent in every loss from the third iteration on
    (tests/tools/dbg_p1_graph.py: the fp32 atomics of one step are amplified by the next), so trajectories cannot be compared; what is
    compared is ONE step from ONE state: a replay of the captured graph, then — weights, momentum buffers, running statistics and the
    RandDrop counter restored — the eager step with the same angle. Same losses to 1e-4, same updated weights."""
    import numpy as np
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    crop = 65
    g = torch.Generator().manual_seed(23)
    batches = [(torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[15 + i % 5, 19 - i % 3]].clone()) for i in range(5)]
    opts = p1_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    tr._cur_epoch = opts.pseudo_ep                  # full loss stack: PAMR, l_seg, l_cls, RandDrop
    mods = (tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)
    for i, m in enumerate(mods):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=41 + i))
    tr._realias()
    np.random.seed(11)
    for b in batches[:4]:
        out = tr.step(b)
        assert all(np.isfinite(float(v)) for v in out["losses"].values())
    assert tr._p1_static is not None and 1 <= len(tr._p1_static["graphs"]) <= 2      # iterations 2 and 3 replayed
    torch.cuda.synchronize()
    snap = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in mods]
    opt_snap = {k: v.clone() for k, v in enumerate(tr.optimizer.buf_fb.views)}
    seed = tr._p1_seed_dev.clone()
    x, l = batches[4][0].to(gpu), batches[4][1].to(gpu).float()
    flat = lambda: torch.cat([p.detach().flatten() for m in (tr.model, tr.pseudolabeler, tr.peakgenerator) for p in m.parameters()]).clone()

    def restore():
        for m, sd in zip(mods, snap):
            m.load_state_dict(sd)
        tr._realias()
        for k, v in opt_snap.items():
            tr.optimizer.buf_fb.views[k].copy_(v)
        tr._p1_seed_dev.copy_(seed)

    for angle in (90, 270):
        restore()
        lg = {k: float(v) for k, v in tr._replay_phase1(x, l, angle, True).items()}        # forward + backward + SGD, one replay
        torch.cuda.synchronize()
        wg = flat()
        restore()
        le = {k: float(v) for k, v in tr._p1.run(x, l, opts.pseudo_ep, flac_angle=angle).items()}
        for i in range(len(tr.buckets.fb.sizes)):
            tr.buckets.ready(i)
        tr.buckets.finish()
        tr.optimizer.step()
        torch.cuda.synchronize()
        we = flat()
        assert {"l_seg", "l_cls", "l_rand", "flac"} <= set(lg) and set(lg) == set(le)
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-4 * max(abs(le[k]), 1e-3), (angle, k, le[k], lg[k])
        rel = ((we - wg).norm() / (we - torch.cat([v.flatten() for m, sd in zip(mods, snap) if m is not tr.model_old
                                                   for k, v in sd.items() if k in dict(m.named_parameters())])).norm()).item()
        print("P1GRAPH angle", angle, "graphs", len(tr._p1_static["graphs"]), "update difference relative to the update", rel)
        assert rel < 2e-2, rel


def test_phase1_adam_step_count_survives_graph_replays(gpu):
    """`--phase 1 --optim adam` on one GPU is graph-replayed from the third call on; the host-side step count that
    `state_dict()` writes (and `load_state_dict` feeds back into the device counter on resume) must advance with every replay exactly
    like the device's `step_dev`, which the captured update kernels increment (train.py:537-541 with torch.optim.Adam's bias correction)."""
    import numpy as np
    from cl4wsis_amd.train import Trainer
    from cl4wsis_amd.optim import FusedAdam
    from oracle import weights
    crop = 65
    opts = p1_opts(crop)
    opts.optim = "adam"
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    assert isinstance(tr.optimizer, FusedAdam)
    mods = (tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)
    for i, m in enumerate(mods):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=41 + i))
    tr._realias()
    g = torch.Generator().manual_seed(5)
    np.random.seed(3)
    n = 9                                             # 2 eager + 7 replays over up to three FLAC-angle graphs (each captured once)
    for i in range(n):
        out = tr.step((torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[15 + i % 5, 19 - i % 3]].clone()))
    torch.cuda.synchronize()
    assert tr._p1_static is not None and len(tr._p1_static["graphs"]) >= 1
    assert all(np.isfinite(float(v)) for v in out["losses"].values())
    assert int(tr.optimizer.step_dev.item()) == n
    steps = {float(st["step"]) for st in tr.optimizer.state_dict()["state"].values()}
    assert steps == {float(n)}, steps
