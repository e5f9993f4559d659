"""GPU: the HIP path against the committed golden vectors of the upstream reference (tests/golden/*.npz) and an
end-to-end `Trainer.step` against the oracle's `p2_step` on the same weights, batch and Dropout mask."""
import glob
import os

import numpy as np
import pytest
import torch

from util import label_inputs_for, p2_opts, rel_err

pytestmark = pytest.mark.gpu

# Network outputs: the north-star tolerance (1e-3 relative, fp32). Measured: ~2e-5 with the exact fp32 MFMA kernels
# (CL4_CONV_PRECISION=0) and ~1.2e-4 with the default bf16x3 split kernels, through ~110 conv layers.
NET_TOL = 1e-3
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(gpu, crop=65, classes=(16, 5), seed=1):
    from cl4wsis_amd import segmentation_module as sm
    from cl4wsis_amd.wss import PeakGenerator, PseudoLabeler
    from oracle import weights
    classes = list(classes)
    tot = sum(classes)
    opts = p2_opts(crop)
    torch.manual_seed(0)
    mods = [sm.make_model(opts, classes=classes), sm.make_model(opts, classes=classes[:-1]),
            PseudoLabeler(2048, tot, sm.get_norm(opts)), PeakGenerator(tot, tot - 1, old_classes=tot - classes[-1] - 1)]
    for i, m in enumerate(mods):
        m.load_state_dict(weights.fill_(m.state_dict(), seed=seed + i))
        m.to(gpu)
    return opts, mods


# fixture -> (classes per step, model seed, input seed), as in tests/test_oracle_golden.py: the reference's own two-, three- and
# four-head networks (VOC 15-5 step 1; 10-5 step 2, scripts/run_10-5.sh:46-60; 15-1 step 3, tasks.py:108-117)
MODEL_FIXTURES = {"model_p2_c65.npz": ([16, 5], 1, 5), "model_p2_3h_c65.npz": ([11, 5, 5], 31, 35),
                  "model_p2_4h_c65.npz": ([16, 1, 1, 1], 41, 45)}


@pytest.mark.parametrize("fixture", sorted(MODEL_FIXTURES))
def test_networks_match_reference_golden(gpu, fixture):
    gold = np.load(os.path.join(GOLD, fixture))
    classes, seed, x_seed = MODEL_FIXTURES[fixture]
    oc1 = sum(classes) - classes[-1] - 1
    opts, (new, old, pl, pg) = _build(gpu, classes=classes, seed=seed)
    g = torch.Generator().manual_seed(x_seed)
    x = torch.randn(2, 3, 65, 65, generator=g).to(gpu)
    l1h = torch.from_numpy(gold["l1h"]).to(gpu)
    old.eval()
    o_old, _ = old(x, interpolate=False)
    for k in ("seg", "center", "offset"):
        assert rel_err(o_old[k].cpu(), torch.from_numpy(gold[f"old_{k}"])) < NET_TOL, k
    new.train(); new.body.eval(); new.head.eval(); new.cls.eval()
    x2 = torch.cat([x, x.flip(-1)], 0).contiguous()
    o2, f2 = new.forward_seg(x2, interpolate=False)
    assert rel_err(o2["seg"].cpu(), torch.from_numpy(gold["seg2b"])) < NET_TOL
    for k in ("res2", "res5"):
        if f"feat_{k}" in gold:
            assert rel_err(f2["features"][k][:2].cpu(), torch.from_numpy(gold[f"feat_{k}"])) < NET_TOL
    shape = tuple(gold["dropout_shape"])
    keep = torch.from_numpy(np.unpackbits(gold["dropout_keep"])[:int(np.prod(shape))].reshape(shape)).to(gpu)
    br = new.instance_branch()
    feats = {k: v[:2] for k, v in f2["features"].items()}
    pred, tape = br.forward_train(feats, dropout_keep=keep)
    assert rel_err(pred["center"].cpu(), torch.from_numpy(gold["train_center"])) < NET_TOL
    assert rel_err(pred["offset"].cpu(), torch.from_numpy(gold["train_offset"])) < NET_TOL
    gc = torch.randn(pred["center"].shape, generator=g).to(gpu)
    go = torch.randn(pred["offset"].shape, generator=g).to(gpu)
    grads = br.backward(tape, {"center": gc, "offset": go})
    params = dict(new.named_parameters())
    norms = np.array([grads[id(params[str(n)])].double().norm().item() for n in gold["grad_names"]])
    # batch-stat BN over 2x5x5 samples is ill-conditioned: the reference's own fp32 gradients sit 1e-3..1e-2 away from a
    # float64 evaluation (tests/test_model_gpu.py), and a 1e-4 change of the input features moves single tensors by <1 %.
    dev = np.abs(norms / gold["grad_norms"] - 1)
    for i in np.argsort(-dev)[:3]:
        print(f"GRADNORM worst {str(gold['grad_names'][i]):60s} ours {norms[i]:.6e} reference {gold['grad_norms'][i]:.6e} dev {dev[i]:.2e}")
    # One tensor is cancellation-dominated and gets its own bar: the pooling branch's weight gradient is a sum over images of (pixel sum of a
    # gradient that a train-mode BatchNorm has just made zero-mean over the batch) x (globally pooled features that are nearly the same for
    # every image) — FINDINGS.md finding 29: the kernels are exact for their inputs to 3e-6, the 1e-6-level element noise upstream is amplified
    # by the cancellation (measured here: 7e-2 on the four-head fixture, 8e-3 / 3e-3 on the others).
    # r05: at 2 x 5 x 5 samples per batch-normalised channel single tensors move by per cent with the last bits of the
    # features (1.7e-2 -> 2.1e-2 on one BN weight when the res2 kernels changed), so this configuration carries the MEDIAN bar and a sanity
    # ceiling only; the per-tensor 2e-2 bar against the reference's own gradients lives on the well-conditioned fixture (crop 193, batch 4):
    # test_gradient_norms_match_reference_well_conditioned below.
    assert (dev <= 0.15).all(), [(str(gold["grad_names"][i]), float(dev[i])) for i in np.flatnonzero(dev > 0.15)]
    med = float(np.median(np.abs(norms / gold["grad_norms"] - 1)))
    print("GRADNORM median", med, "max", float(np.abs(norms / gold["grad_norms"] - 1).max()))
    assert med < 2e-3, med
    pl.eval(); pg.eval()
    im = pl(f2["body"][:2])
    assert rel_err(im.cpu(), torch.from_numpy(gold["int_masks"])) < NET_TOL
    cam = pg.cam(im, l1h[:, oc1:].contiguous())
    assert rel_err(cam.cpu(), torch.from_numpy(gold["cam_lo"][:, oc1:])) < NET_TOL


@pytest.mark.parametrize("frozen_precision", [1, 2])
def test_gradient_norms_match_reference_well_conditioned(gpu, monkeypatch, frozen_precision):
    """The per-tensor gradient bar against the REFERENCE'S OWN gradients (tests/golden/model_p2_c193_grads.npz, made by
    tests/golden/make_golden.py --only wellcond from /root/reference): train-mode instance branch at crop 193, batch 4 (>= 676 samples per
    batch-normalised channel), recorded Dropout mask, seeded upstream gradients: every one of the 61 gradient norms within 2e-2 (the r01-r03
    bar), median within 2e-3, predictions within the network tolerance — in the default arithmetic (bf16x3 frozen
    networks feeding a bf16x6 trainable branch)."""
    from cl4wsis_amd import ops
    monkeypatch.setattr(ops, "CONV_PRECISION", frozen_precision)
    gold = np.load(os.path.join(GOLD, "model_p2_c193_grads.npz"))
    crop = int(gold["crop"])
    opts, (new, _, _, _) = _build(gpu, crop=crop, classes=tuple(int(c) for c in gold["classes"]), seed=int(gold["model_seed"]))
    g = torch.Generator().manual_seed(int(gold["x_seed"]))
    x = torch.randn(4, 3, crop, crop, generator=g).to(gpu)
    new.train(); new.body.eval(); new.head.eval(); new.cls.eval()
    _, f2 = new.forward_seg(x, interpolate=False)
    shape = tuple(gold["dropout_shape"])
    keep = torch.from_numpy(np.unpackbits(gold["dropout_keep"])[:int(np.prod(shape))].reshape(shape)).to(gpu)
    br = new.instance_branch()
    pred, tape = br.forward_train(dict(f2["features"]), dropout_keep=keep)
    assert rel_err(pred["center"][:, :, ::4, ::4].cpu(), torch.from_numpy(gold["train_center_s4"])) < NET_TOL
    assert rel_err(pred["offset"][:, :, ::4, ::4].cpu(), torch.from_numpy(gold["train_offset_s4"])) < NET_TOL
    gc = torch.randn(pred["center"].shape, generator=g).to(gpu)
    go = torch.randn(pred["offset"].shape, generator=g).to(gpu)
    grads = br.backward(tape, {"center": gc, "offset": go})
    params = dict(new.named_parameters())
    norms = np.array([grads[id(params[str(n)])].double().norm().item() for n in gold["grad_names"]])
    dev = np.abs(norms / gold["grad_norms"] - 1)
    for i in np.argsort(-dev)[:4]:
        print(f"GRADNORM193 worst {str(gold['grad_names'][i]):60s} ours {norms[i]:.6e} reference {gold['grad_norms'][i]:.6e} dev {dev[i]:.2e}")
    print(f"GRADNORM193 frozen precision {frozen_precision}: median", float(np.median(dev)), "max", float(dev.max()))
    # 2e-2 on EVERY tensor (measured r05, fp16x3 planes: max 6.6e-3 at the default, 4.0e-3 with bf16x6 frozen networks). A bf16x3 build of the
    # library (CL4_F16_PLANES=0: the r02-r04 arithmetic) puts two tensors at 3.2e-2 — the frozen features' 1e-4 — and is held to 4e-2.
    bar = 2e-2 if (ops.planes_f16() or frozen_precision != 1) else 4e-2
    assert len(dev) == 61 and (dev <= bar).all(), [(str(gold["grad_names"][i]), float(dev[i])) for i in np.flatnonzero(dev > bar)]
    assert float(np.median(dev)) < 2e-3


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "label_*.npz"))))
def test_label_stage_matches_reference_golden(gpu, path):
    """Bit-exact indices / labels against the reference's own outputs (not via the oracle)."""
    from cl4wsis_amd import ops
    from oracle import synth
    from oracle.labels import gaussian
    gold = np.load(path)
    seed, b, H, W = (int(v) for v in gold["case"])
    inp = label_inputs_for(gold)
    d = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in inp.items()}
    oc = inp["old_classes"]
    label_new = d["l1h"][:, oc - 1:].contiguous()
    cam = ops.cam_smooth(d["int_masks"], d["pg_w"], d["pg_b"], label_new)
    assert rel_err(cam.cpu(), torch.from_numpy(gold["cam_lo"][:, oc - 1:])) < 1e-5
    peaks = ops.peak_extract(cam, label_new, (H, W))
    conf, ys, xs = (t.cpu().numpy() for t in peaks)
    gc, gy, gx = gold["peak_conf"][:, oc - 1:], gold["peak_y"][:, oc - 1:], gold["peak_x"][:, oc - 1:]
    sel = gc >= 0.7                                   # the prefix the step consumes (train.py:455-458)
    assert ((conf >= 0.7) == sel).all()
    np.testing.assert_array_equal(ys[sel], gy[sel])
    np.testing.assert_array_equal(xs[sel], gx[sel])
    np.testing.assert_allclose(conf[sel], gc[sel], rtol=1e-5)
    seg_gt, old_w, prob_win = ops.seg_pseudo_gt(d["seg2b"], d["l1h"], oc, (H, W))
    assert float(gold["top2_gap_min"]) > 1e-6          # fixtures were chosen without fp32-level argmax ties
    np.testing.assert_array_equal(seg_gt.cpu().numpy(), gold["seg_gt"])
    ow = np.unpackbits(gold["old_w"])[:b * H * W].reshape(b, 1, H, W)
    np.testing.assert_array_equal(old_w.cpu().numpy().astype(np.uint8), ow)
    g = torch.from_numpy(gaussian(6).astype(np.float32)).to(gpu)
    pseudo, refined = ops.label_generation(seg_gt, prob_win, label_new, peaks, d["center_lo"], d["offset_lo"], g, oc)
    losses, _, _ = ops.p2_loss(d["center_lo"], d["offset_lo"], d["old_center_lo"], d["old_offset_lo"], old_w, pseudo,
                               refined, (H, W), with_grad=False)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pseudo["offset"].cpu().numpy(), gold["p_offset"])
    np.testing.assert_array_equal(pseudo["weight"].cpu().numpy().astype(np.uint8), gold["p_weight"])
    np.testing.assert_allclose(pseudo["center"].cpu().numpy(), gold["p_center"], atol=1e-7)
    np.testing.assert_array_equal(refined["offset"].cpu().numpy(), gold["r_offset"])
    np.testing.assert_allclose(refined["weight"].cpu().numpy(), gold["r_weight"], atol=1e-5)
    np.testing.assert_allclose(refined["center"].cpu().numpy(), gold["r_center"], atol=1e-7)
    np.testing.assert_allclose(losses.cpu().numpy().astype(np.float64), gold["losses"], rtol=1e-4)


class _Log:
    def info(self, *a, **k):
        pass
    debug = info


def test_trainer_step_matches_oracle(gpu):
    """Whole phase-2 iteration (train.py:241-541) at crop 97, batch 2: losses, the semantic pseudo GT, and the
    parameter update of Adam against the CPU oracle on identical weights / inputs / Dropout mask."""
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep, weights
    crop = 97
    opts = p2_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=11 + i)
        m.load_state_dict(sd)
        sds.append({k: v.clone() for k, v in sd.items()})
    tr._realias()
    new, old, pl, pg = sds
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, crop, crop, generator=g)
    l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
    keep = torch.rand(2, 256, 7, 7, generator=g) >= 0.5
    before = {k: v.clone() for k, v in new.items()}
    ref = ostep.p2_step(new, old, pl, pg, x, l1h, 16, pooling=opts.pooling, dropout_mask=keep.float(),
                        lr=opts.lr * opts.lr_head)
    out = tr.step((x, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
    torch.cuda.synchronize()
    mism = out["seg_gt"].cpu().long() != ref["seg_gt"]
    if mism.any():   # only admissible where the oracle's own top-2 softmax gap is at fp32 resolution
        top2 = ref["soft"].topk(2, dim=1).values
        assert (top2[:, 0] - top2[:, 1])[mism].max() < 1e-5
    assert rel_err(out["center"].cpu(), ref["pred_center"]) < NET_TOL
    assert rel_err(out["offset"].cpu(), ref["pred_offset"]) < NET_TOL
    got = out["losses"].cpu().tolist()
    for v, k in zip(got, ("center_old", "offset_old", "center_new", "offset_new")):
        assert abs(v - float(ref[k].detach())) <= 1e-3 * max(abs(float(ref[k].detach())), 1e-3), (k, v, float(ref[k].detach()))
    # Adam's first step moves every weight by ~lr * sign(g): compare the updates where the gradient is not ~0
    params = dict(tr.model.named_parameters())
    agree, total = 0, 0
    for k in ostep.trainable_names(new):
        du = (params[k].detach().cpu() - before[k])
        dr = (new[k] - before[k])
        big = ref["grads"][k].abs() > 1e-3 * ref["grads"][k].abs().max()
        agree += int(((du - dr).abs()[big] < 0.05 * opts.lr * opts.lr_head).sum())
        total += int(big.sum())
    assert agree / total > 0.995, (agree, total)


def test_graph_replay_matches_eager(gpu, monkeypatch):
    """Three iterations through the captured HIP graph (device-resident Dropout seed / Adam step / learning rate) give the
    same losses and weights as three eager iterations from the same state (differences: fp32 atomics order only)."""
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    crop = 97
    g = torch.Generator().manual_seed(8)
    batches = [(torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[16 + i % 4, 15]].clone()) for i in range(3)]

    # A small learning rate keeps the two runs' weights within ~1e-6 of each other, so that the discrete decisions of the label
    # stage (which sees the trainable branch's own predictions from iteration 2 on) do not flip between the runs — at the
    # reference's 5e-4 one run in four kept an instance the other dropped. The Adam arithmetic under test is scale-free.
    LR = 5e-6
    # both runs on the heuristic launch plans: the warm-up autotuner picks by timing and could give the two Trainers different
    # summation orders, which is not what this test is about
    monkeypatch.setenv("CL4_AUTOTUNE", "0")

    def run(graph):
        monkeypatch.setenv("CL4_GRAPH", "1" if graph else "0")
        opts = p2_opts(crop)
        opts.lr = LR / 10                            # the instance branch trains at 10 x lr (train.py:144-185)
        opts.max_iters, opts.start_decay = 50, 1     # the schedule changes the learning rate every iteration
        torch.manual_seed(0)
        tr = Trainer(_Log(), gpu, opts)
        for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
            m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=31 + i))
        tr._realias()
        w0 = torch.cat([p.detach().flatten() for p in tr._branch.trainable_parameters()]).cpu()
        losses = []
        for x, l in batches:
            losses.append(tr.step((x, l))["losses"].clone())
        torch.cuda.synchronize()
        assert (tr._graph is not None or tr._pipe is not None) == graph
        w = torch.cat([p.detach().flatten() for p in tr._branch.trainable_parameters()]).cpu()
        assert torch.isfinite(w).all() and not torch.equal(w, w0)
        return torch.stack(losses).cpu(), w, tr.optimizer.param_groups[0]["lr"]

    le, we, lre = run(False)
    lg, wg, lrg = run(True)
    assert lre == lrg and lre < LR
    assert torch.allclose(le, lg, rtol=2e-3, atol=1e-6), (le, lg)
    # Adam turns a gradient of ~1e-8 into a step of +-lr, so a weight whose gradient is numerically zero can move by up to
    # 3 * lr differently in two runs that differ only in the order of the fp32 atomics of the split-K weight gradient
    # (two EAGER runs show the same: tools/dbg_graph_eager.py). Such weights are a handful out of 15.7 M.
    d = (we - wg).abs()
    assert d.max() < 3.2 * LR
    assert (d > 0.4 * LR).float().mean() < 1e-5
    assert d.mean() < 0.02 * LR


def test_trainer_step_voc_10_10(gpu):
    """BASELINE config 4 (VOC 10-10 step 1 phase 2: 11 old + 10 new classes) through the same Trainer against the oracle."""
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep, weights
    crop = 97
    opts = p2_opts(crop)
    opts.task = "10-10"
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    assert tr.old_classes == 11 and tr.tot_classes == 21
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=61 + i)
        m.load_state_dict(sd)
        sds.append({k: v.clone() for k, v in sd.items()})
    tr._realias()
    new, old, pl, pg = sds
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, crop, crop, generator=g)
    l1h = torch.zeros(2, 20); l1h[0, 12] = 1; l1h[1, 10] = 1; l1h[1, 19] = 1
    keep = torch.rand(2, 256, 7, 7, generator=g) >= 0.5
    ref = ostep.p2_step(new, old, pl, pg, x, l1h, 11, pooling=opts.pooling, dropout_mask=keep.float(), lr=opts.lr * opts.lr_head)
    out = tr.step((x, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
    torch.cuda.synchronize()
    mism = out["seg_gt"].cpu().long() != ref["seg_gt"]
    if mism.any():
        top2 = ref["soft"].topk(2, dim=1).values
        assert (top2[:, 0] - top2[:, 1])[mism].max() < 1e-5
    for v, k in zip(out["losses"].cpu().tolist(), ("center_old", "offset_old", "center_new", "offset_new")):
        r = float(ref[k].detach())
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)


def test_collective_path_over_rccl_single_rank(gpu, monkeypatch):
    """The N > 1 training path (graph = forward .. backward; bucketed RCCL all-reduce (AVG) on the communication stream and Adam
    after the replay) rehearsed on ONE GPU: a world-size-1 `nccl` (= RCCL) process group with CL4_FORCE_COLLECTIVES=1. Averaging
    over one rank is the identity, so losses and weights must match the plain single-GPU run (Adam +-lr ambiguity aside)."""
    import torch.distributed as dist
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    crop = 97
    g = torch.Generator().manual_seed(18)
    batches = [(torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[16 + i % 4, 15]].clone()) for i in range(3)]
    LR = 5e-6

    def run(lookahead=False):
        opts = p2_opts(crop)
        opts.lr = LR / 10
        opts.max_iters, opts.start_decay = 50, 1
        torch.manual_seed(0)
        tr = Trainer(_Log(), gpu, opts)
        for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
            m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=31 + i))
        tr._realias()
        losses, g1 = [], None
        for i, b in enumerate(batches):
            losses.append(tr.step(b, next_batch=batches[i + 1] if lookahead and i + 1 < len(batches) else None)["losses"].clone())
            if i == 0:      # the first step's (averaged) gradients: every run computes them from the SAME weights
                torch.cuda.synchronize()
                g1 = torch.cat([tr._grad_map[id(p)].flatten() for p in tr._branch.trainable_parameters()]).cpu().clone()
        torch.cuda.synchronize()
        w = torch.cat([p.detach().flatten() for p in tr._branch.trainable_parameters()]).cpu()
        return torch.stack(losses).cpu(), w, tr, g1

    ls, ws, _, gs = run()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29577")
    monkeypatch.setenv("CL4_FORCE_COLLECTIVES", "1")
    monkeypatch.setenv("CL4_PIPELINE_MIN_BATCH", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        ld, wd, tr, gd = run()
        assert tr.buckets.distributed and tr.buckets.native_avg and tr._graph is not None
        lp, wp_, trp, gp = run(lookahead=True)           # the same with the next batch announced: (deep-)pipelined graphs + eager ASPP tail
        assert trp.buckets.distributed and trp._pipe is not None and trp._pipe["deep"]      # 2 images per GPU: deep pipelining
        monkeypatch.setenv("CL4_SPLIT_SIDE_GRAPH", "1")  # opt-in: the next batch's frozen stage as a graph of its own on the side stream
        l2, w2, tr2, g2 = run(lookahead=True)
        assert tr2._pipe.get("gB") and len(tr2._pipe["gB"]) == 2
    finally:
        dist.destroy_process_group()
    # (1) From the same weights — the first step — the collective path must reproduce the plain path's gradients: what is left is the order of
    # the weight-gradient kernels' atomic adds (measured 4e-8; two PLAIN runs differ by the same, tests/tools/dbg_plain_vs_plain.py).
    for name, g_ in (("collective path", gd), ("collective path, deep-pipelined", gp), ("collective path, deep-pipelined, two graphs per step", g2)):
        e = float((g_ - gs).norm() / gs.norm())
        print(f"COLLECTIVE1 {name}: first-step gradients vs the plain run, relative L2 {e:.2e}")
        assert e < 1e-6, (name, e)
    # (2) Three chained Adam steps. Adam turns every gradient into a step of ~lr whatever its size, so a weight whose gradient IS rounding
    # noise (taps of the dilated ASPP convolutions that see one or two pixels of a 7 x 7 map) moves +-lr by the order of those atomic adds,
    # and from the second step on the runs see different weights. r05: two plain runs in one process show the same spread (second-step
    # gradients 4e-6 apart, third-step up to 8e-4 when a label decision flips), in discrete states that depend on the launch plans the tuner
    # timed into place: max 0.00 / 0.01 / 0.12 / 0.22 / 0.39 / 1.66 lr with 0 ... 523 of 15.7 M weights beyond 0.4 lr. The bars bound that
    # spread (max, mean unchanged since r03; the count at 1e-4 of the weights, r03-r04: 1e-5 = the five smaller states only).
    for l_, w_ in ((ld, wd), (lp, wp_), (l2, w2)):
        assert torch.allclose(ls, l_, rtol=2e-3, atol=1e-6), (ls, l_)
        d = (ws - w_).abs()
        print(f"COLLECTIVE1 weights after 3 Adam steps vs the plain run: max {float(d.max()) / LR:.2f} lr, mean {float(d.mean()) / LR:.4f} lr, "
              f"fraction beyond 0.4 lr {float((d > 0.4 * LR).float().mean()):.2e} ({int((d > 0.4 * LR).sum())} of {d.numel()})")
        assert d.max() < 3.2 * LR and (d > 0.4 * LR).float().mean() < 1e-4 and d.mean() < 0.02 * LR


def test_pipelined_replay_matches_plain_graph_replay(gpu, monkeypatch):
    """Trainer.step(batch, next_batch=...) replays frozen work of the NEXT batch beside the current step's trainable branch and backward
    (two parities of static buffers, one graph each, the fork inside the graph) — shallow: the old model's forward; deep (r04, the default up
    to 8 images per GPU): the old model on one side stream and the new model's whole frozen stage (semantic forward, CAM, peaks, semantic
    pseudo ground truth) on a second one. Two alternating batches, eight steps (one of them without an announcement in mid-sequence): the losses of every step and the final weights equal those
    of the unpipelined graph path (same kernels; only the fp32 atomics of the split-K weight gradients may reorder)."""
    import copy
    from cl4wsis_amd.nn import CACHE
    from cl4wsis_amd.train import Trainer
    from util import p2_opts
    opts = p2_opts(65)
    opts.max_iters, opts.start_decay = 1000, 1000
    g = torch.Generator().manual_seed(12)
    batches = []
    for i in range(2):
        x = torch.randn(2, 3, 65, 65, generator=g)
        l1h = torch.zeros(2, 20); l1h[0, 15 + i] = 1; l1h[1, 18 + i] = 1
        batches.append((x.to(gpu), l1h.to(gpu)))

    class _L:
        def info(self, *a, **k):
            pass
        debug = info

    results = {}
    for mode in ("plain", "shallow", "deep"):
        monkeypatch.setenv("CL4_PIPELINE", "0" if mode == "plain" else "1")
        monkeypatch.setenv("CL4_DEEP_PIPE", "1" if mode == "deep" else "0")
        monkeypatch.setenv("CL4_PIPELINE_MIN_BATCH", "1")       # the default keeps a batch of 1 per GPU on the one-graph path
        torch.manual_seed(0)
        tr = Trainer(_L(), gpu, copy.deepcopy(opts))
        assert tr._pipeline == (mode != "plain") and tr._deep_pipe == (mode == "deep")
        # r05: every step starts from the SAME trainable state — weights, Adam moments and step count, decoder
        # BatchNorm statistics are put back after each step — so a step's losses and gradients depend on nothing but its batch and the frozen
        # results handed to it, which is what this test guards. (r04 chained eight steps and had to bound the chaotic growth of the
        # reordered fp32 atomics through the discrete label stage: 5e-2 from step 4 on.)
        opt = tr.optimizer
        snap = ([t.clone() for t in (opt.params_fb.flat, opt.m_fb.flat, opt.v_fb.flat, opt.step_dev)],
                {k: v.detach().clone() for k, v in tr.model.state_dict().items() if "running_" in k or "num_batches" in k})

        def restore():
            for dst, src in zip((opt.params_fb.flat, opt.m_fb.flat, opt.v_fb.flat, opt.step_dev), snap[0]):
                dst.copy_(src)
            sd = tr.model.state_dict()
            for k, v in snap[1].items():
                sd[k].copy_(v)
            for grp in opt.param_groups:
                for prm in grp["params"]:
                    CACHE.touch(prm)

        losses, gnorm = [], []
        for t in range(8):
            # step 2 announces nothing (the end of an epoch) and step 3 brings a DIFFERENT batch than the one last seen: the frozen
            # results prefetched by step 2 must not be taken for it
            b, nb = batches[t % 2], batches[(t + 1) % 2]
            out = tr.step(b, next_batch=None if t in (2, 7) else nb)
            losses.append(out["losses"].clone())
            gnorm.append(opt.grads_fb.flat.clone())
            restore()
        torch.cuda.synchronize()
        assert (tr._pipe is not None) == (mode != "plain") and (tr._graph is not None) == (mode == "plain")
        assert mode == "plain" or tr._pipe["deep"] == (mode == "deep")
        results[mode] = (torch.stack(losses).cpu(), torch.stack(gnorm).cpu())
    l0, g0 = results["plain"]
    for mode in ("shallow", "deep"):
        l1, g1 = results[mode]
        assert torch.isfinite(l1).all()
        dev = ((l0 - l1).abs() / l0.abs().clamp_min(1e-6)).max(1).values      # per step
        gdev = (g0 - g1).norm(dim=1) / g0.norm(dim=1)                          # per step: the whole flat gradient buffer
        print(f"PIPEREPLAY {mode}: per-step loss deviation {[f'{float(d):.1e}' for d in dev]}, gradient deviation {[f'{float(d):.1e}' for d in gdev]}")
        # same kernels on the same operands: only the fp32 atomics of the split-K weight gradients may reorder within ONE step; a wrong frozen
        # result at step 3 (the step after the unannounced one) would show there as a deviation of order one
        assert (dev < 1e-4).all() and (gdev < 1e-3).all(), (mode, dev, gdev)


