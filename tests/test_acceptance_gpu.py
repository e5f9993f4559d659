"""GPU: the two acceptance checks of BASELINE.json's north star that need more than one kernel:

  * a WELL-CONDITIONED gradient comparison — phase-2 and phase-1 steps at crop 193, batch 4 (>= 600 samples per
    batch-normalised channel at the coarsest level) against a float64 evaluation of the oracle, per tensor, relative L2 < 1e-3,
    no statistical escape hatch (the 65..97-pixel tests elsewhere have 50-100 samples per channel and can only be read
    statistically);
  * an mAP proxy for "<= 0.2 mAP after one epoch on fixed seeds" — the dataset is not available offline, so: K = 20 phase-2
    iterations on a mini VOC-shaped tree, identical weights / batches / Dropout masks through the HIP path and through the CPU
    oracle, then instance-segmentation validation (train.py:588-651) of both weight sets against the same ground truth.
"""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import QuietLogger, make_mini_voc, p1_opts, p2_opts

pytestmark = pytest.mark.gpu
CROP = 193


def _fill(tr, seed0):
    from oracle import weights
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=seed0 + i)
        m.load_state_dict(sd)
        sds.append(sd)
    tr._realias()
    return sds


def _to64(d):
    return {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in d.items()}


def _report(tag, errs, cpu):
    e, c = torch.tensor(errs), torch.tensor(cpu)
    print(f"WELLCOND {tag}: {len(errs)} tensors, HIP vs f64 rel-L2 median {e.median():.2e} max {e.max():.2e}; "
          f"CPU-fp32 oracle vs f64 median {c.median():.2e} max {c.max():.2e}")
    return e, c


def test_phase2_gradients_well_conditioned(gpu):
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep
    opts = p2_opts(CROP)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), gpu, opts)
    new, old, pl, pg = _fill(tr, 71)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 3, CROP, CROP, generator=g)
    l1h = torch.zeros(4, 20)
    for i, cs in enumerate(((15,), (16, 19), (17,), (18, 15))):
        l1h[i, list(cs)] = 1
    keep = torch.rand(4, 256, 13, 13, generator=g) >= 0.5
    cp = lambda d: {k: v.clone() for k, v in d.items()}
    r64 = ostep.p2_step(_to64(new), _to64(old), _to64(pl), _to64(pg), x.double(), l1h.double(), 16, pooling=opts.pooling,
                        dropout_mask=keep.double(), lr=opts.lr * opts.lr_head)
    r32 = ostep.p2_step(cp(new), cp(old), cp(pl), cp(pg), x, l1h, 16, pooling=opts.pooling, dropout_mask=keep.float(),
                        lr=opts.lr * opts.lr_head)
    out = tr.step((x, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
    torch.cuda.synchronize()
    # the three evaluations must have made the same discrete label decisions, or the gradients are not comparable
    assert (out["seg_gt"].cpu().long() != r64["seg_gt"]).float().mean() < 1e-4
    for v, k in zip(out["losses"].cpu().tolist(), ("center_old", "offset_old", "center_new", "offset_new")):
        r = float(r64[k].detach())
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)
    params = dict(tr.model.named_parameters())
    errs, cpu = [], []
    for k, g64 in r64["grads"].items():
        if float(g64.norm()) == 0:
            continue
        ours = tr._grad_map[id(params[k])].cpu().double()
        errs.append(((ours - g64).norm() / g64.norm()).item())
        cpu.append(((r32["grads"][k].double() - g64).norm() / g64.norm()).item())
    e, c = _report("phase 2", errs, cpu)
    assert len(errs) >= 55
    assert e.max() < 1e-3, (e.max(), c.max())


def test_phase1_gradients_well_conditioned(gpu):
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep
    opts = p1_opts(CROP)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), gpu, opts)
    new, old, pl, pg = _fill(tr, 81)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 3, CROP, CROP, generator=g)
    l1h = torch.zeros(4, 20)
    for i, cs in enumerate(((15,), (16, 19), (17,), (18, 15))):
        l1h[i, list(cs)] = 1
    kw = dict(cur_epoch=0, pseudo_epoch=opts.pseudo_ep, flac_angle=90, randrop_neg=None, lr=opts.lr, lr_head=opts.lr_head,
              lr_pseudo=opts.lr_pseudo, weight_decay=opts.weight_decay, loss_de=opts.loss_de, alpha=opts.alpha,
              l_seg_weight=opts.l_seg, icarl_bkg=opts.icarl_bkg, pooling=opts.pooling)
    cp = lambda d: {k: v.clone() for k, v in d.items()}
    n64, p64, g64 = _to64(new), _to64(pl), _to64(pg)
    r64 = ostep.p1_step(n64, _to64(old), p64, g64, x.double(), l1h.double(), 16, **kw)
    n32, p32, g32 = cp(new), cp(pl), cp(pg)
    r32 = ostep.p1_step(n32, cp(old), p32, g32, x, l1h, 16, **kw)
    out = tr._step_phase1((x, l1h), cur_epoch=0, flac_angle=90)
    torch.cuda.synchronize()
    for k in ("l_cam_new", "l_loc", "flac", "l_peak", "lde"):
        v, r = float(out["losses"][k]), float(r64[k])
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)
    errs, cpu = [], []
    for mod, d64, d32 in ((tr.model, n64, n32), (tr.pseudolabeler, p64, p32), (tr.peakgenerator, g64, g32)):
        for k, p in mod.named_parameters():
            gr64 = r64["grads"].get((id(d64), k))
            if gr64 is None or float(gr64.norm()) == 0:
                continue
            ours = tr._grad_map[id(p)].cpu().double()
            errs.append(((ours - gr64).norm() / gr64.norm()).item())
            cpu.append(((r32["grads"][(id(d32), k)].double() - gr64).norm() / gr64.norm()).item())
    e, c = _report("phase 1", errs, cpu)
    assert len(errs) > 300
    assert e.max() < 1e-3, (e.max(), c.max())


def _oracle_predict(sd, img, target, opts):
    """train.py:616-626 + dataset/utils.py:795-892 through the oracle (eval forward, two-stage resize, get_ins_map)."""
    from oracle import labels as L, model as M
    x = torch.cat([img, img.flip(-1)], 0) if opts.val_flip else img
    with torch.no_grad():
        o, _ = M.forward_full(x, sd, pooling=opts.pooling)
        o = {k: F.interpolate(v, size=img.shape[-2:], mode="bilinear", align_corners=True) for k, v in o.items()}
        o = {k: F.interpolate(v, size=target, mode="bilinear", align_corners=False) for k, v in o.items()}
    _, lab, msk, sc = L.get_ins_map(o["seg"], o["center"], o["offset"], val_flip=opts.val_flip, val_thresh=opts.val_thresh,
                                    val_kernel=opts.val_kernel, beta=opts.beta, val_ignore=opts.val_ignore)
    return lab, msk, sc


def test_map_proxy_after_training_both_paths(gpu, tmp_path):
    from cl4wsis_amd import metrics as MT
    from cl4wsis_amd.dataset import DeviceLoader, get_dataset
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep
    K, crop = 20, 97
    make_mini_voc(tmp_path, n_train=9, n_val=4)
    opts = p2_opts(crop, extra=["--data_root", str(tmp_path), "--crop_size_val", "97", "--batch_size", "2"])
    opts.max_iters, opts.start_decay = 10000, 10000          # constant learning rate over the K iterations
    opts.val_flip = False
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), gpu, opts)
    new, old, pl, pg = _fill(tr, 91)
    new, old, pl, pg = ({k: v.clone() for k, v in d.items()} for d in (new, old, pl, pg))
    random.seed(5)
    train_dst, val_dst, _, _, _ = get_dataset(opts)
    loader = DeviceLoader(train_dst, 2, gpu, shuffle=False, drop_last=True)
    g = torch.Generator().manual_seed(13)
    state, it = None, 0
    while it < K:
        for batch in loader:
            if it >= K:
                break
            images, l1h = batch[0], batch[-1].float()
            keep = torch.rand(2, 256, 7, 7, generator=g) >= 0.5
            tr.step((images, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
            r = ostep.p2_step(new, old, pl, pg, images.cpu(), l1h.cpu(), 16, pooling=opts.pooling, dropout_mask=keep.float(),
                              lr=opts.lr * opts.lr_head, adam_state=state, step=it + 1)
            state = r["adam_state"]
            it += 1
    torch.cuda.synchronize()
    # how far apart the two weight sets drifted over K Adam steps (each step moves a weight by ~lr)
    params = dict(tr.model.named_parameters())
    num = sum(float((params[k].detach().cpu() - new[k]).double().pow(2).sum()) for k in ostep.trainable_names(new))
    den = sum(float(new[k].double().pow(2).sum()) for k in ostep.trainable_names(new))
    print("MAPPROXY relative weight distance after", K, "iterations:", (num / den) ** 0.5)
    # validation of both weight sets against the same ground truth (the dataset's own instance masks)
    val_loader = DeviceLoader(val_dst, 1, gpu)
    thr = np.arange(0.5, 0.95, 0.05).tolist()
    acc = [MT.DetectionAccumulator() for _ in thr]
    samples = []
    for img, seg, mask, mask_label, fname in val_loader:
        samples.append((img.cpu(), np.asarray(mask)[0].astype(bool), np.asarray(mask_label)[0] - 1))
    res = tr.validate(DeviceLoader(val_dst, 1, gpu), opts)
    n_pred = 0
    for img, gt_mask, gt_label in samples:
        lab, msk, sc = _oracle_predict(new, img, gt_mask.shape[-2:], opts)
        n_pred += int((sc > 0).sum())
        for a, t in zip(acc, thr):
            a.update(gt_label, gt_mask, lab, msk, sc, t)
    ap = np.full((len(thr), 20), np.nan)
    for i, a in enumerate(acc):
        v = a.ap()
        ap[i, :len(v)] = v[:20]
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            ref_map = float(np.nanmean(np.nanmean(ap, axis=0)))
    print("MAPPROXY mAP HIP path", res["map"], "oracle path", ref_map, "oracle predictions", n_pred)
    assert n_pred > 0
    assert abs(100 * res["map"] - 100 * ref_map) <= 0.2          # north star: <= 0.2 mAP (points)
