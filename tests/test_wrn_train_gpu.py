"""GPU parity of the train-mode WiderResNet-38 A2 body (cl4wsis_amd/body_train.py: wrn_body_forward / wrn_body_backward — coco-voc
step 0 / phase 1, models/wide_resnet.py:82-207, modules/residual.py:101-181) against the oracle restatement differentiated by
PyTorch autograd in float64 on the CPU. Dropout2d masks (mod6 / mod7) are injected on both sides: the reference draws them from
torch's generator, which no other implementation can reproduce. Outputs and running statistics: 1e-4 relative; gradients: relative
L2 per tensor, max < 3e-2 and median < 5e-3 (the bar of tests/test_body_train_gpu.py, for the reason given there)."""
import pytest
import torch

from oracle import model as omodel
from oracle import weights

pytestmark = pytest.mark.gpu


def _iabn():
    from functools import partial
    from cl4wsis_amd.nn import InPlaceABNSync
    return partial(InPlaceABNSync, activation="leaky_relu", activation_param=0.01)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("structure,crop,partial_out", [((1, 1, 1, 1, 1, 1), 41, False), ((2, 1, 2, 1, 1, 1), 33, False),
                                                         ((1, 1, 1, 1, 1, 1), 33, True)])
def test_wrn_body_forward_backward(gpu, structure, crop, partial_out):
    from cl4wsis_amd import body_train
    from cl4wsis_amd.backbone import WiderResNetA2
    torch.manual_seed(0)
    body = WiderResNetA2(list(structure), norm_act=_iabn(), keep_outputs=True)
    sd = weights.fill_({k: v.clone() for k, v in body.state_dict().items()}, seed=13)
    body.load_state_dict(sd)
    body.to(gpu).train()
    g = torch.Generator().manual_seed(3)
    n = 3
    x = torch.randn(n, 3, crop, crop, generator=g)
    keeps = {"mod6.block1": (torch.rand(n, 1024, generator=g) > 0.3).to(torch.uint8),
             "mod7.block1": (torch.rand(n, 2048, generator=g) > 0.5).to(torch.uint8)}

    osd = {"body." + k: v.double().clone() for k, v in sd.items()}
    names = [k for k in osd if k.endswith("weight") or k.endswith("bias")]
    for k in names:
        osd[k].requires_grad_(True)
    outs_ref = omodel.wider_resnet_a2_body(x.double(), osd, structure=structure, training=True, keeps=keeps)
    douts = [torch.randn(o.shape, generator=g, dtype=torch.float64) / o.numel() ** 0.5 for o in outs_ref]
    used = [4] if partial_out else list(range(5))           # phase 1 reaches the body through the last output alone
    sum((outs_ref[i] * douts[i]).sum() for i in used).backward()

    outs, tape = body_train.wrn_body_forward(body, x.to(gpu), keeps={k: v.to(gpu) for k, v in keeps.items()})
    assert len(outs) == 5
    for o, r in zip(outs, outs_ref):
        assert o.shape == r.shape and _rel(o, r.detach()) < 1e-4
    grads = {}

    def grad_of(p):
        if id(p) not in grads:
            grads[id(p)] = torch.zeros_like(p, dtype=torch.float32)
        return grads[id(p)]

    body_train.wrn_body_backward(body, tape, {i: douts[i].float().to(gpu) for i in used}, grad_of)
    torch.cuda.synchronize()
    errs = []
    for name, p in body.named_parameters():
        ref = osd["body." + name].grad
        got = grad_of(p).double().cpu()
        errs.append(((got - ref).norm() / (ref.norm() + 1e-30)).item())
    errs = torch.tensor(errs)
    assert len(errs) == len(names)
    assert errs.max() < 3e-2 and errs.median() < 5e-3, (errs.max(), errs.median())
    for name, b in body.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            assert _rel(b, osd["body." + name]) < 1e-4, name


def test_dropout2d_matches_torch(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 10, 7, 5, generator=g)
    keep = (torch.rand(4, 10, generator=g) > 0.3).to(torch.uint8)
    y = ops.dropout2d(x.to(gpu), keep.to(gpu), 0.3).cpu()
    assert torch.allclose(y, torch.where(keep.bool()[:, :, None, None], x * (1.0 / (1.0 - 0.3)), torch.zeros(())), rtol=1e-6, atol=0)
    drawn = ops.dropout_mask((64, 512), 0.5, 1234, gpu)
    assert 0.45 < drawn.float().mean().item() < 0.55


class _Log:
    def info(self, *a, **k):
        pass
    debug = info


def _wrn_p1_opts(crop):
    from cl4wsis_amd import argparser
    args = ["--dataset", "coco-voc", "--task", "voc", "--step", "1", "--weakly", "--phase", "1", "--alpha", "0.9", "--loss_de", "1",
            "--lr_policy", "warmup", "--affinity", "--optim", "sgd", "--lr", "0.001", "--epochs", "40", "--batch_size", "32",
            "--no_pretrained"]
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(args))
    assert opts.backbone == "wider_resnet38_a2" and opts.output_stride == 8
    opts.crop_size, opts.pooling = crop, crop // 8
    return opts


@pytest.mark.parametrize("cur_epoch", [5])      # after pseudo_ep: every loss term of the stack; the other side runs with the ResNet body (test_phase1_gpu.py)
def test_trainer_phase1_step_with_wrn_body_matches_oracle(gpu, monkeypatch, cur_epoch):
    """The coco-voc task's phase-1 iteration (WiderResNet-38 A2 body in training mode, DeepLabV3 at output stride 8, 61 + 20 classes,
    PseudoLabeler / PeakGenerator, SGD-Nesterov) through Trainer.step's phase-1 path against oracle.step.p1_step on identical weights,
    inputs, FLAC angle, RandDrop draws and Dropout2d masks: every loss term within 1e-3, and the update of the trainable tensors no
    further from the float64 update than a few times the CPU-fp32 oracle is (the statistical bar of tests/test_phase1_gpu.py)."""
    from cl4wsis_amd import body_train
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep, weights
    monkeypatch.setattr(body_train, "PRECISION", 0)
    monkeypatch.setattr(body_train, "WGRAD_PRECISION", 0)
    crop = 49
    opts = _wrn_p1_opts(crop)
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    sds = []
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        sd = weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=31 + i)
        m.load_state_dict(sd)
        sds.append({k: v.clone() for k, v in sd.items()})
    tr._realias()
    new, old, pl, pg = sds
    g = torch.Generator().manual_seed(5)
    n = 2
    x = torch.randn(n, 3, crop, crop, generator=g)
    l1h = torch.zeros(n, 80); l1h[0, 63] = 1; l1h[1, 71] = 1; l1h[1, 79] = 1      # 60 COCO-only + 20 VOC foreground classes
    hs = (crop + 7) // 8
    neg = torch.randint(0, 61, (n, hs, hs), generator=g)
    keeps = {"mod6.block1": (torch.rand(n, 1024, generator=g) > 0.3).to(torch.uint8),
             "mod7.block1": (torch.rand(n, 2048, generator=g) > 0.5).to(torch.uint8)}
    monkeypatch.setattr(body_train, "WRN_KEEPS", {k: v.to(gpu) for k, v in keeps.items()})
    before = [{k: v.clone() for k, v in d.items()} for d in (new, pl, pg)]
    kw = dict(cur_epoch=cur_epoch, pseudo_epoch=opts.pseudo_ep, flac_angle=180, randrop_neg=neg, lr=opts.lr, lr_head=opts.lr_head,
              lr_pseudo=opts.lr_pseudo, weight_decay=opts.weight_decay, loss_de=opts.loss_de, alpha=opts.alpha,
              l_seg_weight=opts.l_seg, icarl_bkg=opts.icarl_bkg, pooling=opts.pooling, wrn_keeps=keeps)
    to64 = lambda d: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in d.items()}
    new64, old64, pl64, pg64 = to64(new), to64(old), to64(pl), to64(pg)
    ostep.p1_step(new64, old64, pl64, pg64, x.double(), l1h.double(), 61, **kw)
    ref = ostep.p1_step(new, old, pl, pg, x, l1h, 61, **kw)
    out = tr._step_phase1((x, l1h), cur_epoch=cur_epoch, flac_angle=180, randrop_neg=neg.to(torch.int32).to(gpu))
    torch.cuda.synchronize()
    names = ["l_cam_new", "l_loc", "flac", "l_peak", "lde"] + (["l_seg", "l_cls", "l_rand"] if cur_epoch >= opts.pseudo_ep else [])
    for k in names:
        v, r = float(out["losses"][k]), float(ref[k])
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)
    ratios, worst = [], 0.0
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
    ratios = torch.tensor(ratios)
    print("WRN P1STAT epoch", cur_epoch, "tensors", len(ratios), "median ratio", ratios.median().item(), "frac>3", (ratios > 3).float().mean().item(), "worst", worst)
    assert len(ratios) > 100
    assert ratios.median() < 2.5 and (ratios > 3).float().mean() < 0.05 and worst < 0.3, (ratios.median(), (ratios > 3).float().mean(), worst)


def test_trainer_step0_with_wrn_body_matches_oracle(gpu, monkeypatch):
    """coco-voc step 0 (scripts/coco.sh: supervised, Adam): the whole network with the WiderResNet-38 body in training mode — semantic
    head, classifier, instance decoder and heads back-propagating into the body's features — against oracle.step.s0_step: the three
    loss terms within 1e-3 and every gradient no further from float64 than a few times the CPU-fp32 oracle is."""
    from cl4wsis_amd import argparser, body_train, heads
    from cl4wsis_amd.train import Trainer
    from oracle import step as ostep, weights
    monkeypatch.setattr(body_train, "PRECISION", 0)
    monkeypatch.setattr(body_train, "WGRAD_PRECISION", 0)
    monkeypatch.setattr(heads, "TRAIN_PRECISION", 0)
    crop = 49
    args = ["--dataset", "coco-voc", "--task", "voc", "--step", "0", "--bce", "--lr", "5e-5", "--epochs", "100", "--optim", "adam",
            "--weight_decay", "0", "--batch_size", "32", "--no_pretrained"]
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(args))
    assert opts.backbone == "wider_resnet38_a2" and opts.output_stride == 8
    opts.crop_size, opts.pooling = crop, crop // 8
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(_Log(), gpu, opts)
    assert tr.model_old is None and tr.model.decoder is not None
    sd = weights.fill_({k: v.cpu().clone() for k, v in tr.model.state_dict().items()}, seed=43)
    tr.model.load_state_dict(sd)
    tr._realias()
    g = torch.Generator().manual_seed(8)
    b = 3
    ncls = tr.model.tot_classes                                # background + 60 COCO-only classes
    x = torch.randn(b, 3, crop, crop, generator=g)
    seg_maps = torch.randint(0, ncls, (b, crop, crop), generator=g)
    seg_maps[torch.rand(b, crop, crop, generator=g) < 0.1] = 255
    center_maps = torch.rand(b, ncls - 1, crop, crop, generator=g)
    offset_maps = torch.randn(b, 2, crop, crop, generator=g) * 5
    wts = (torch.rand(b, 1, crop, crop, generator=g) < 0.4).float() * torch.rand(b, 1, crop, crop, generator=g)
    hs = (crop + 7) // 8
    keep = torch.rand(b, 256, hs, hs, generator=g) >= 0.5
    keeps = {"mod6.block1": (torch.rand(b, 1024, generator=g) > 0.3).to(torch.uint8),
             "mod7.block1": (torch.rand(b, 2048, generator=g) > 0.5).to(torch.uint8)}
    monkeypatch.setattr(body_train, "WRN_KEEPS", {k: v.to(gpu) for k, v in keeps.items()})
    kw = dict(dropout_mask=keep.float(), lr=opts.lr, pooling=opts.pooling, wrn_keeps=keeps)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref64 = ostep.s0_step(sd64, x.double(), seg_maps, center_maps.double(), offset_maps.double(), wts.double(),
                          **{**kw, "dropout_mask": keep.double()})
    ref = ostep.s0_step({k: v.clone() for k, v in sd.items()}, x, seg_maps, center_maps, offset_maps, wts, **kw)
    out = tr.step((x, seg_maps, center_maps, offset_maps, wts, torch.zeros(b, ncls - 1)), dropout_keep=keep.to(torch.uint8).to(gpu))
    torch.cuda.synchronize()
    for k in ("seg_loss", "center_loss", "offset_loss"):
        v, r = float(out["losses"][k]), float(ref[k])
        assert abs(v - r) <= 1e-3 * max(abs(r), 1e-3), (k, v, r)
    ratios = []
    for k, p in tr.model.named_parameters():
        g64 = ref64["grads"][k]
        if g64.norm() == 0:
            continue
        ours = ((tr._grad_map[id(p)].double().cpu() - g64).norm() / g64.norm()).item()
        cpu32 = ((ref["grads"][k].double() - g64).norm() / g64.norm()).item()
        ratios.append(ours / (cpu32 + 1e-5))
    ratios = torch.tensor(ratios)
    print("WRN S0STAT tensors", len(ratios), "median ratio", ratios.median().item(), "frac>3", (ratios > 3).float().mean().item())
    assert len(ratios) > 150
    assert ratios.median() < 2 and (ratios > 3).float().mean() < 0.05, (ratios.median(), (ratios > 3).float().mean())
