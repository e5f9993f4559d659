"""GPU: Trainer.validate_semseg (train.py:695-745) and the device-side confusion matrix (cl4_semseg_hist) against the
reference's formulae on the CPU: oracle forward in eval mode -> align_corners=True upsampling -> flip average -> softmax ->
arg-max -> np.bincount confusion matrix (metrics/stream_metrics.py:72-79) -> mIoU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import p2_opts

pytestmark = pytest.mark.gpu


class _Log:
    def info(self, *a, **k):
        pass
    debug = info


def _np_hist(lt, lp, n):
    mask = (lt >= 0) & (lt < n)
    return np.bincount(n * lt[mask].astype(int) + lp[mask], minlength=n * n).reshape(n, n)


@pytest.mark.parametrize("flip", [False, True])
def test_semseg_hist_kernel(gpu, flip):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(2)
    n, c, H, W = (2, 21, 37, 41)
    logits = torch.randn(2 if flip else n, c, H, W, generator=g) * 3
    nimg = 1 if flip else n
    labels = torch.randint(0, 21, (nimg, H, W), generator=g)
    labels[torch.rand(nimg, H, W, generator=g) < 0.1] = 255
    m = (logits[:1] + logits[1:].flip(-1)) / 2 if flip else logits
    pred = m.softmax(1).max(1)[1]
    ref = sum(_np_hist(labels[i].numpy().flatten(), pred[i].numpy().flatten(), 21) for i in range(nimg))
    hist = torch.zeros(21, 21, dtype=torch.int64, device=gpu)
    got_pred = ops.semseg_hist_(hist, logits.to(gpu), labels.to(gpu), 21, flip=flip, want_pred=True)
    assert torch.equal(got_pred.cpu().long(), pred)
    assert np.array_equal(hist.cpu().numpy(), ref)
    ops.semseg_hist_(hist, logits.to(gpu), labels.to(gpu), 21, flip=flip)      # accumulates
    assert np.array_equal(hist.cpu().numpy(), 2 * ref)


def test_validate_semseg_matches_reference_pipeline(gpu):
    from cl4wsis_amd.metrics import StreamSegMetrics
    from cl4wsis_amd.train import Trainer
    from oracle import model as M, weights
    crop = 65
    opts = p2_opts(crop, extra=["--val_flip"]) if False else p2_opts(crop)
    opts.val_flip = True
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    sd = weights.fill_({k: v.cpu().clone() for k, v in tr.model.state_dict().items()}, seed=7)
    tr.model.load_state_dict(sd)
    tr._realias()
    g = torch.Generator().manual_seed(4)
    loader = [(torch.randn(1, 3, crop, crop, generator=g), torch.randint(0, 21, (1, crop, crop), generator=g)) for _ in range(3)]
    metrics = StreamSegMetrics(21)
    res = tr.validate_semseg(loader, metrics, opts)
    ref = np.zeros((21, 21))
    mism = 0
    for img, seg in loader:
        x = torch.cat([img, img.flip(-1)], 0)
        with torch.no_grad():
            lo, _ = M.forward_seg(x, sd, pooling=opts.pooling)
            up = F.interpolate(lo["seg"], size=(crop, crop), mode="bilinear", align_corners=True)
            m = ((up[:1] + up[1:].flip(-1)) / 2).softmax(1)
        pred = m.max(1)[1]
        ref += _np_hist(seg[0].numpy().flatten(), pred[0].numpy().flatten(), 21)
    # arg-max may differ from the fp32 CPU evaluation only where the top-2 gap is at rounding level
    diff = np.abs(res["Confusion Matrix"] - ref).sum() / 2
    assert diff <= 3 * 1e-3 * crop * crop, diff
    assert res["Total samples"] == 3 and abs(res["Overall Acc"] - np.diag(ref).sum() / ref.sum()) < 2e-3
    iu = np.diag(ref) / (ref.sum(1) + ref.sum(0) - np.diag(ref) + 1e-6)
    assert abs(res["Mean IoU"] - iu[ref.sum(1) != 0].mean()) < 2e-3


def test_validate_instance_ap(gpu, monkeypatch):
    """Trainer.validate end to end on structured network outputs (the model's forward is replaced by the synthetic maps of
    oracle.synth.insmap_inputs so that the instances are meaningful): predictions equal the oracle's get_ins_map on the same
    maps, and the AP over IoU 0.5:0.95 equals the same accumulation run on the oracle's predictions."""
    from collections import OrderedDict
    from cl4wsis_amd import metrics as MT
    from cl4wsis_amd.train import Trainer
    from oracle import labels as L, synth
    crop = 65
    opts = p2_opts(crop)
    opts.val_flip = True
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    cases = [(3, 129, 129), (6, 161, 145), (5, 145, 129)]
    store = {}

    def fake_forward(images, **kw):
        seg, center, offset = store["cur"]
        return OrderedDict(seg=seg.to(gpu), center=center.to(gpu), offset=offset.to(gpu)), {}

    monkeypatch.setattr(tr.model, "forward", fake_forward)
    loader, ref_preds = [], []
    for seed, H, W in cases:
        seg, center, offset = synth.insmap_inputs(seed, H, W, True)
        _, lab, msk, sc = L.get_ins_map(seg, center, offset, val_flip=True)
        ref_preds.append((lab, msk, sc))
        # ground truth: the oracle's instances, one of them shifted (IoU < 1) and one dropped
        gt_masks, gt_labels = msk.copy(), lab.copy() + 1
        gt_masks[0] = np.roll(gt_masks[0], 3, axis=1)
        gt_masks, gt_labels = gt_masks[:-1], gt_labels[:-1]
        loader.append(((seg, center, offset), torch.zeros(1, H, W), torch.from_numpy(gt_masks)[None], torch.from_numpy(gt_labels)[None], "x"))

    class _Loader:
        def __iter__(self):
            for maps, seg_t, m, ml, fn in loader:
                store["cur"] = maps
                yield torch.zeros(1, 3, maps[0].shape[-2], maps[0].shape[-1]), seg_t, m, ml, fn

    res = tr.validate(_Loader(), opts)
    thr = np.arange(0.5, 0.95, 0.05).tolist()
    acc = [MT.DetectionAccumulator() for _ in thr]
    for (lab, msk, sc), (_, _, m, ml, _) in zip(ref_preds, loader):
        for a, t in zip(acc, thr):
            a.update(ml.numpy()[0] - 1, m.numpy()[0].astype(bool), lab, msk, sc, t)
    ap = np.full((len(thr), 20), np.nan)
    for i, a in enumerate(acc):
        v = a.ap()
        ap[i, :len(v)] = v[:20]
    with np.errstate(all="ignore"):
        ref_ap = np.nanmean(ap, axis=0)
    np.testing.assert_allclose(res["ap"], ref_ap, rtol=1e-6, equal_nan=True)
    assert 0.0 < res["map"] <= 1.0 and abs(res["map"] - np.nanmean(ref_ap)) < 1e-9


def test_eval_after_graph_training_sees_current_weights(gpu):
    """train (HIP-graph replays) -> validate -> train -> validate: each eval pass must score the weights and BN running
    statistics as they are NOW. During a replay the host-side cache invalidation of the optimizer does not run, so the eval-only
    derived tensors (packed weights, folded BN of the instance branch) built by the first validation would otherwise be reused."""
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    crop = 97
    opts = p2_opts(crop)
    opts.val_flip = False
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=31 + i))
    tr._realias()
    g = torch.Generator().manual_seed(8)
    batches = [(torch.randn(2, 3, crop, crop, generator=g), torch.eye(20)[[16 + i % 4, 15]].clone()) for i in range(4)]
    x_eval = torch.randn(1, 3, crop, crop, generator=g)

    def eval_center():
        tr.model.eval()
        tr.predict_instances(x_eval, (crop, crop), opts)       # the public eval entry (what validate() calls per image)
        c = tr.model(x_eval.to(gpu))[0]["center"].clone()
        tr.model.train()
        return c

    for b in batches[:2]:
        tr.step(b)
    assert tr._graph is not None
    c1 = eval_center()
    for b in batches[2:]:
        tr.step(b)
    c2 = eval_center()
    torch.cuda.synchronize()
    assert not torch.allclose(c1, c2, rtol=1e-4, atol=1e-6)      # two more updates at lr 5e-4 moved the head
    fresh = Trainer(_Log(), gpu, opts)
    fresh.model.load_state_dict({k: v.clone() for k, v in tr.model.state_dict().items()})
    fresh._realias()
    fresh.model.eval()
    ref = fresh.model(x_eval.to(gpu))[0]["center"]
    torch.cuda.synchronize()
    assert torch.allclose(c2, ref, rtol=1e-5, atol=1e-6), (c2 - ref).abs().max()


# (the eval-pass-between-replays check lives in tests/test_graph_safety_gpu.py::test_graph_steps_survive_regrow_short_batch_and_eval_pass)
