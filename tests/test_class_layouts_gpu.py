"""GPU: class layouts with MORE THAN TWO classifier heads through `Trainer.step` against the CPU oracle.

The reference ships them: scripts/run_10-5.sh:46-60 trains VOC 10-5 step 2 (classes [11, 5, 5], old model [11, 5]) and tasks.py:108-117
defines 15-1 with six steps (step 3: [16, 1, 1, 1], old model [16, 1, 1]). With >= 3 heads the per-step 1x1 classifiers of
IncrementalClassifier (segmentation_module.py:107-113) and of IncrementalSinglePanopticDeepLabHead (modules/panoptic_deeplab.py:432-468)
write three / four channel slices, the old model itself is incremental, `old_classes` is no longer the first head's width, and in 15-1 the
one-hot label vector has tot - 1 = 18 columns (LabelSelection, dataset/dataset.py:269-284) with a single new class. The networks of both
layouts are pinned by the reference's own outputs (tests/golden/model_p2_{3h,4h}_c65.npz, tests/test_golden_gpu.py); here the whole
iteration — label stage slices (`l1h[:, oc - 1:]`), the old / new split of `cl4_p2_loss`, backward, Adam — runs against the oracle.
"""
import pytest
import torch

from util import QuietLogger, p1_opts, p2_opts

pytestmark = pytest.mark.gpu

LAYOUTS = {      # id -> (task, step, classes of the new model, labels per image (0-based, background excluded))
    "voc10-5_step2": ("10-5", 2, [11, 5, 5], ((17,), (15, 19))),
    "voc15-1_step3": ("15-1", 3, [16, 1, 1, 1], ((17,), (17,))),
    "voc15-1_step5": ("15-1", 5, [16, 1, 1, 1, 1, 1], ((19,), (19,))),
}


def _trainer(gpu, opts, seed0):
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), gpu, opts)
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=seed0 + i)
        m.load_state_dict(sd)
        sds.append({k: v.clone() for k, v in sd.items()})
    tr._realias()
    return tr, sds


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_phase2_step_with_three_or_more_heads_matches_oracle(gpu, layout):
    from oracle import step as ostep
    task, step, classes, labels = LAYOUTS[layout]
    crop = 97
    opts = p2_opts(crop)
    opts.task, opts.step = task, step
    opts.max_iters, opts.start_decay = 1000, 1000
    tr, (new, old, pl, pg) = _trainer(gpu, opts, 131)
    tot, oc = sum(classes), sum(classes[:-1])
    assert tr.classes == classes and tr.tot_classes == tot and tr.old_classes == oc
    assert len(tr.model.cls) == len(classes) and len(tr.model_old.cls) == len(classes) - 1
    assert [m.out_channels for m in tr.model.instance_head.classifier["center"]["cls"]] == [classes[0] - 1] + classes[1:]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, crop, crop, generator=g)
    l1h = torch.zeros(2, tot - 1)
    for i, cs in enumerate(labels):
        l1h[i, list(cs)] = 1
    keep = torch.rand(2, 256, 7, 7, generator=g) >= 0.5
    before = {k: v.clone() for k, v in new.items()}
    ref = ostep.p2_step(new, old, pl, pg, x, l1h, oc, pooling=opts.pooling, dropout_mask=keep.float(), lr=opts.lr * opts.lr_head)
    out = tr.step((x, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
    torch.cuda.synchronize()
    mism = out["seg_gt"].cpu().long() != ref["seg_gt"]
    if mism.any():
        top2 = ref["soft"].topk(2, dim=1).values
        assert (top2[:, 0] - top2[:, 1])[mism].max() < 1e-5
    for v, k in zip(out["losses"].cpu().tolist(), ("center_old", "offset_old", "center_new", "offset_new")):
        r = float(ref[k].detach())
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)
    assert out["center"].shape[1] == tot - 1
    err = ((out["center"].cpu() - ref["pred_center"]).abs().max() / ref["pred_center"].abs().max()).item()
    assert err < 1e-3, err
    # the label stage wrote the new-class channels only, and into the right ones
    for k in ("offset", "weight"):
        assert torch.equal(out["pseudo"][k].cpu().float(), ref["pseudo"][k].float()), k
    assert torch.allclose(out["pseudo"]["center"].cpu(), ref["pseudo"]["center"][:, oc - 1:], atol=1e-6)
    # gradients of every head: the per-step classifiers of the instance head receive the old-model term (old heads) or the pseudo-label
    # term (the newest head) — a wrong channel split shows up as a gradient of the wrong head
    params = dict(tr.model.named_parameters())
    worst = 0.0
    for k, gr in ref["grads"].items():
        if "cls." not in k or float(gr.norm()) == 0:
            continue
        ours = tr._grad_map[id(params[k])].cpu()
        e = ((ours - gr).norm() / gr.norm()).item()
        worst = max(worst, e)
        assert e < 2e-2, (k, e)                                  # batch-statistics BN at 2 x 25 x 25 samples: DESIGN.md section 2, finding (4)
    n_heads = sum(1 for k in ref["grads"] if ".center.cls." in k and k.endswith(".weight"))
    assert n_heads == len(classes)
    # Adam moved exactly the trainable set and left every old-step head's SEMANTIC classifier alone (freeze_seg)
    for k in before:
        if k.startswith("cls."):
            assert torch.equal(params[k].detach().cpu(), before[k]), k
    print(f"LAYOUT {layout}: losses {out['losses'].cpu().tolist()} worst head-gradient rel-L2 {worst:.2e}")


def test_phase1_step_with_three_heads_matches_oracle(gpu):
    """VOC 10-5 step 2, phase 1 (scripts/run_10-5.sh:46-52): the network trains all three classifier heads, the distillation terms read a
    two-head old model, `l_loc` / `l_seg` split at old_classes = 16 = 11 + 5."""
    from oracle import step as ostep
    crop = 65
    opts = p1_opts(crop)
    opts.task, opts.step = "10-5", 2
    opts.max_iters, opts.start_decay = 1000, 1000
    tr, (new, old, pl, pg) = _trainer(gpu, opts, 151)
    assert tr.classes == [11, 5, 5] and tr.old_classes == 16 and len(tr.model_old.cls) == 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, crop, crop, generator=g)
    l1h = torch.zeros(2, 20); l1h[0, 17] = 1; l1h[1, 15] = 1; l1h[1, 19] = 1
    neg = torch.randint(0, 16, (2, 5, 5), generator=g)
    for cur_epoch in (0, 5):
        kw = dict(cur_epoch=cur_epoch, pseudo_epoch=opts.pseudo_ep, flac_angle=90, randrop_neg=neg, lr=opts.lr, lr_head=opts.lr_head,
                  lr_pseudo=opts.lr_pseudo, weight_decay=opts.weight_decay, loss_de=opts.loss_de, alpha=opts.alpha,
                  l_seg_weight=opts.l_seg, icarl_bkg=opts.icarl_bkg, pooling=opts.pooling)
        cp = lambda d: {k: v.clone() for k, v in d.items()}
        for m, sd in zip((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator), (new, old, pl, pg)):
            m.load_state_dict(sd)
        tr._realias()
        ref = ostep.p1_step(cp(new), cp(old), cp(pl), cp(pg), x, l1h, 16, **kw)
        out = tr._step_phase1((x, l1h), cur_epoch=cur_epoch, flac_angle=90, randrop_neg=neg.to(torch.int32).to(gpu))
        torch.cuda.synchronize()
        names = ["l_cam_new", "l_loc", "flac", "l_peak", "lde"] + (["l_seg", "l_cls", "l_rand"] if cur_epoch >= opts.pseudo_ep else [])
        for k in names:
            v, r = float(out["losses"][k]), float(ref[k])
            assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (cur_epoch, k, v, r)
