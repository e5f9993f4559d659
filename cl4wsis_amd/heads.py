"""DeepLabV3 semantic head and the Panoptic-DeepLab instance decoder / heads on HIP kernels.

State-dict names follow the reference:
  head.{map_convs.N, map_bn, global_pooling_conv, global_pooling_bn, red_conv, pool_red_conv, red_bn}
  decoder.instance_decoder.{aspp.convs.N.{0,1}, aspp.convs.4.aspp_pooling.1, aspp.project.{0,1},
                            project.N.{0,1}, fuse.N.0.{0.0,0.1,1,2}}
  instance_head.classifier.{center,offset}.{fuse.0.{0.0,0.1,1,2}, cls.K}

The instance branch has two execution modes:
  * eval (frozen old model): BatchNorm folded into the conv epilogues, one kernel per conv;
  * train (phase-2 trainable sub-net): batch statistics, Dropout(0.5), and a hand-written backward
    (`InstanceBranch.forward_train` / `.backward`) that accumulates parameter gradients. `Trainer.step` drives the
    tape directly; called the reference's way (`model.decoder(feats)`, `model.instance_head(x)`, train.py:269-272) the
    two halves run under `cl4wsis_amd.autograd.TapeFunction`, so that ``loss.backward()`` works (train.py:537).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from .nn import ABN, CACHE, conv_abn, conv_abn_planes, conv_bias, packed_weight, planes_of, planes_ok

EPS = 1e-5
# Arithmetic of the trainable sub-net's convolutions. Gradients through batch-statistics BN + ReLU amplify operand noise
# (measured on a 75-sample batch: median gradient error 5e-4 with fp32-exact products vs 1e-2 with the bf16x3 split), so
# the default is "bf16x6" (2): three bf16 planes per operand, six products, fp32-level accuracy on the fast MFMA pipe
# (about twice the speed of the fp32 MFMA kernels). 0 = exact fp32 MFMA, 1 = bf16x3 (fastest, ~4e-6 per layer). The frozen,
# no-grad networks (old model, flip-TTA semantic forward, PseudoLabeler) run at ops.CONV_PRECISION (bf16x3 by default).
# r05: 3 (the default) = fp16x3 — three fp16 products per MAC on split-fp16 operands with power-of-two scales, the accuracy of bf16x6 at
# half its MFMA work (FINDINGS.md finding 47) — wherever the LDS-DMA kernel takes the shape, bf16x6 elsewhere.
TRAIN_PRECISION = int(__import__("os").environ.get("CL4_TRAIN_PRECISION", "3"))
# The trainable ASPP reads a three-plane (hi, mid, lo) copy of the frozen res5 features through the LDS-DMA planes kernel (conv_b.hip,
# plan impl 6): 0.82 -> 0.61 ms per dilated branch (r03); 0 = the role-split bf16x6 kernel on the fp32 features.
X6_PLANES = __import__("os").environ.get("CL4_X6_PLANES", "1") == "1"
POOL_GRAD_COMMUTED = __import__("os").environ.get("CL4_POOL_GRAD", "0") != "0"


def _dx_1x1(dz, conv):
    """Input gradient of a 1x1 conv. Exact fp32: the PyTorch weight [cout, cin] already is the packed [K=cout][M=cin] operand of
    the transposed GEMM; bf16x3 / bf16x6: the transposed weight in split planes (cached until the optimiser touches it)."""
    if TRAIN_PRECISION and ops.bf16x3_ok(conv.in_channels):
        wd = CACHE.get(conv, ("dgrad", TRAIN_PRECISION), (conv.weight,), lambda: ops.dgrad_weight(conv.weight.detach(), TRAIN_PRECISION))
        return ops.conv2d(dz, wd, 1, xdyn=True)
    return ops.conv2d(dz, conv.weight.detach().reshape(conv.out_channels, conv.in_channels), 1)


class DeeplabV3(nn.Module):
    """modules/deeplab.py:8-89."""

    def __init__(self, in_channels, out_channels, hidden_channels=256, out_stride=16, norm_act=ABN, pooling_size=None):
        super().__init__()
        self.pooling_size = pooling_size
        dil = [6, 12, 18] if out_stride == 16 else [12, 24, 32]
        self.hidden = hidden_channels
        self.map_convs = nn.ModuleList(
            [nn.Conv2d(in_channels, hidden_channels, 1, bias=False)] +
            [nn.Conv2d(in_channels, hidden_channels, 3, bias=False, dilation=d, padding=d) for d in dil])
        self.map_bn = norm_act(hidden_channels * 4)
        self.global_pooling_conv = nn.Conv2d(in_channels, hidden_channels, 1, bias=False)
        self.global_pooling_bn = norm_act(hidden_channels)
        self.red_conv = nn.Conv2d(hidden_channels * 4, out_channels, 1, bias=False)
        self.pool_red_conv = nn.Conv2d(hidden_channels, out_channels, 1, bias=False)
        self.red_bn = norm_act(out_channels)
        self.reset_parameters(self.map_bn.activation, self.map_bn.activation_param)

    def reset_parameters(self, activation, slope):
        gain = nn.init.calculate_gain(activation, slope)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight.data, gain)

    def forward(self, x):
        if self.training:        # batch statistics + global average pooling (modules/deeplab.py:54-76), differentiable
            from . import autograd as AG
            return AG.apply(AG.DeeplabRunner(self), [x], list(self.parameters()))
        n, _, h, w = x.shape
        hid = self.hidden
        xp = planes_of(x)
        use_planes = xp is not None and all(planes_ok(m) for m in self.map_convs) and planes_ok(self.red_conv) and hid % 8 == 0
        if use_planes:                            # the body handed its output out pre-split: the branches and red_conv stay in planes
            cat = ops.Planes(n, 4 * hid, h, w, x.device)
            for i, conv in enumerate(self.map_convs):
                conv_abn_planes(xp, conv, self.map_bn, bn_lo=i * hid, bn_hi=(i + 1) * hid, out_planes=cat, out_coff=i * hid)
        else:
            cat = torch.empty((n, 4 * hid, h, w), dtype=torch.float32, device=x.device)
            for i, conv in enumerate(self.map_convs):  # each branch writes its slice, map_bn folded per slice
                conv_abn(x, conv, self.map_bn, out=cat, out_coff=i * hid, bn_lo=i * hid, bn_hi=(i + 1) * hid)
        # eval-mode image pooling (deeplab.py:78-88): box average, stride 1, then replicate pad. The two 1x1
        # convs + ABN are pointwise, so they commute exactly with the replicate pad: run them on the
        # (h-k+1)x(w-k+1) map and pad afterwards.
        ps = self.pooling_size if isinstance(self.pooling_size, (tuple, list)) else (self.pooling_size, self.pooling_size)
        kh, kw = min(ps[0], h), min(ps[1], w)
        pool = ops.avgpool_s1(x, kh, kw)
        pool = conv_abn(pool, self.global_pooling_conv, self.global_pooling_bn)
        pool = ops.conv2d(pool, packed_weight(self.pool_red_conv), 1)
        pl, pt = (kw - 1) // 2, (kh - 1) // 2
        pr = pl if kw % 2 == 1 else pl + 1
        pb = pt if kh % 2 == 1 else pt + 1
        pool = ops.replicate_pad(pool, pl, pr, pt, pb)
        if use_planes:
            return conv_abn_planes(cat, self.red_conv, self.red_bn, pre=pool, want_fp32=True, want_planes=False)[0]
        return conv_abn(cat, self.red_conv, self.red_bn, pre=pool)


class IncrementalClassifier(nn.ModuleList):
    """segmentation_module.py:107-113 — per-step 1x1 classifiers, concatenated on C."""

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import autograd as AG
            return AG.apply(AG.ClassifierRunner(self), [x], list(self.parameters()))
        return self._forward(x)

    def _forward(self, x):
        n, _, h, w = x.shape
        tot = sum(m.out_channels for m in self)
        out = torch.empty((n, tot, h, w), dtype=torch.float32, device=x.device)
        off = 0
        for m in self:
            conv_bias(x, m, out=out, out_coff=off)
            off += m.out_channels
        return out


# ------------------------------------------------------------------------------------------------
# Panoptic-DeepLab building blocks (parameter holders with the reference's Sequential indices)
# ------------------------------------------------------------------------------------------------
def _conv_bn_relu(cin, cout, k, padding=0, dilation=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, padding=padding, dilation=dilation, groups=groups, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())


def _dw_separable(cin, cout):
    """stacked_conv(num_stack=1, depthwise_separable_conv) — modules/panoptic_deeplab.py:32-67:
    Sequential( Sequential( Sequential(dw5x5, BN, ReLU), conv1x1, BN, ReLU ) )."""
    inner = nn.Sequential(_conv_bn_relu(cin, cin, 5, padding=2, groups=cin),
                          nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())
    return nn.Sequential(inner)


class ASPPPooling(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.aspp_pooling = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False), nn.ReLU())


class ASPP(nn.Module):
    """modules/panoptic_deeplab.py:101-133."""

    def __init__(self, cin, cout, rates):
        super().__init__()
        mods = [_conv_bn_relu(cin, cout, 1)]
        mods += [_conv_bn_relu(cin, cout, 3, padding=r, dilation=r) for r in rates]
        mods.append(ASPPPooling(cin, cout))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(5 * cout, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(),
                                     nn.Dropout(0.5))


class SinglePanopticDeepLabDecoder(nn.Module):
    """modules/panoptic_deeplab.py:137-193 (parameters only; executed by InstanceBranch)."""

    def __init__(self, in_channels, feature_key, low_level_channels, low_level_key, low_level_channels_project,
                 decoder_channels, atrous_rates, aspp_channels=None):
        super().__init__()
        aspp_channels = aspp_channels or decoder_channels
        self.aspp = ASPP(in_channels, aspp_channels, atrous_rates)
        self.feature_key = feature_key
        self.low_level_key = list(low_level_key)
        self.decoder_stage = len(low_level_channels)
        self.project = nn.ModuleList([_conv_bn_relu(low_level_channels[i], low_level_channels_project[i], 1)
                                      for i in range(self.decoder_stage)])
        fuse = []
        for i in range(self.decoder_stage):
            cin = (aspp_channels if i == 0 else decoder_channels) + low_level_channels_project[i]
            fuse.append(_dw_separable(cin, decoder_channels))
        self.fuse = nn.ModuleList(fuse)


class IncrementalPanopticDeepLabDecoder(nn.Module):
    """modules/panoptic_deeplab.py:381-430 — only the instance branch exists on the CL4WSIS path
    (argparser.py:19-20 forces DeepLabV3 as the semantic branch)."""

    def __init__(self, in_channels, feature_key, low_level_channels, low_level_key, low_level_channels_project,
                 decoder_channels, atrous_rates, branch=("ins",), detach_instance=False):
        super().__init__()
        self.branch = list(branch)
        if "seg" in self.branch:
            raise NotImplementedError("Panoptic-DeepLab semantic decoder: never instantiated by the reference's CLI")
        self.detach_instance = detach_instance
        if "ins" in self.branch:
            self.instance_decoder = SinglePanopticDeepLabDecoder(
                in_channels, feature_key, low_level_channels, low_level_key, (64, 32, 16), 128, atrous_rates,
                aspp_channels=256)
        self._dropout_keep = None     # test hook: a recorded nn.Dropout keep mask for the next training forward

    def forward(self, features):
        """modules/panoptic_deeplab.py:413-430 — returns (semantic, instance) decoder features; the semantic decoder does not
        exist on the CL4WSIS path, so the first item is None. Training mode runs the hand-written forward under
        torch.autograd (cl4wsis_amd/autograd.py) so that `loss.backward()` of a foreign training loop works
        (train.py:269-272,537); `detach_instance` cuts the gradient to the backbone features exactly like the reference."""
        if "ins" not in self.branch:
            return None, None
        from . import autograd as AG
        d = self.instance_decoder
        branch = InstanceBranch(self, None)
        if not d.training:
            return None, branch.decoder_forward_eval(features)
        keys = [d.feature_key] + list(d.low_level_key)
        feats = [features[k].detach() if self.detach_instance else features[k] for k in keys]
        keep, self._dropout_keep = self._dropout_keep, None
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())      # drawn from torch's generator: follows torch.manual_seed
        runner = AG.DecoderRunner(branch, keys, keep, seed, detach=self.detach_instance)
        return None, AG.apply(runner, feats, list(d.parameters()))


class IncrementalSinglePanopticDeepLabHead(nn.Module):
    """modules/panoptic_deeplab.py:432-468 (parameters only; executed by InstanceBranch)."""

    def __init__(self, decoder_channels, head_channels, num_classes, class_key):
        super().__init__()
        classifier = {}
        for i, key in enumerate(class_key):
            ncls = num_classes[i] if isinstance(num_classes[i], (list, tuple)) else [num_classes[i]]
            classifier[key] = nn.ModuleDict({
                "fuse": _dw_separable(decoder_channels, head_channels[i]),
                "cls": nn.ModuleList([nn.Conv2d(head_channels[i], c, 1) for c in ncls]),
            })
        self.classifier = nn.ModuleDict(classifier)
        self.class_key = list(class_key)

    def forward(self, x):
        """modules/panoptic_deeplab.py:457-468 — OrderedDict(center [n, sum(classes), h, w], offset [n, 2, h, w])."""
        from . import autograd as AG
        branch = InstanceBranch(None, self)
        if not self.training:
            return branch._head_eval(x)
        runner = AG.HeadRunner(branch)
        outs = AG.apply(runner, [x], list(self.parameters()))
        if isinstance(outs, dict):
            return outs
        return OrderedDict(zip(self.class_key, outs))


# ------------------------------------------------------------------------------------------------
# Execution of the instance branch
# ------------------------------------------------------------------------------------------------
class _BnConvRec:
    """Saved tensors of one conv -> BN(train) -> ReLU group for the backward pass."""
    __slots__ = ("x", "x_coff", "cin", "z", "mean", "var", "keep", "y")


BN_FUSED_FINAL = __import__("os").environ.get("CL4_BN_FUSED_FINAL", "1") != "0"   # r03: statistics finalised inside the elementwise pass


def _bn_train(z, bn, out=None, out_coff=0, keep=None, keep_scale=1.0, partials=None):
    if BN_FUSED_FINAL:
        y, mean, var = ops.bn_train_fwd(z, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                        ops.ACT_RELU, keep=keep, keep_scale=keep_scale, out=out, out_coff=out_coff, partials=partials)
        CACHE.touch(bn.running_mean)
        CACHE.touch(bn.running_var)
        return y, mean, var
    mean, var = ops.bn_train_stats(z, running_mean=bn.running_mean, running_var=bn.running_var, momentum=bn.momentum)
    CACHE.touch(bn.running_mean)
    CACHE.touch(bn.running_var)
    y = ops.bn_apply(z, mean, var, bn.weight.detach(), bn.bias.detach(), bn.eps, ops.ACT_RELU, keep=keep,
                     keep_scale=keep_scale, out=out, out_coff=out_coff)
    return y, mean, var


class InstanceBranch:
    """Runs decoder.instance_decoder + instance_head of one model.

    eval: `forward_eval(feats)`; train: `forward_train(feats)` then `backward(dcenter, doffset)`.
    """

    def __init__(self, decoder, head):
        self.dec = decoder.instance_decoder if decoder is not None else None     # either half may be driven on its own
        self.head = head

    # ------------------------------------------------------------------ eval (frozen old model)
    def _dwsep_eval(self, x, seq):
        blk = seq[0]
        dw, bn0 = blk[0][0], blk[0][1]
        y = ops.dwconv5(x, dw.weight.detach())
        # a ReLU sits between the depthwise and the pointwise conv, so the first BN+ReLU is one affine pass
        # over the running statistics; the second BN+ReLU is the pointwise conv's epilogue.
        y = ops.bn_apply(y, bn0.running_mean, bn0.running_var, bn0.weight.detach(), bn0.bias.detach(), bn0.eps, ops.ACT_RELU)
        return conv_abn(y, blk[1], blk[2])

    @torch.no_grad()
    def forward_eval(self, feats):
        return self._head_eval(self.decoder_forward_eval(feats))

    @torch.no_grad()
    def decoder_forward_eval(self, feats):
        d = self.dec
        x = feats[d.feature_key]
        n, _, h, w = x.shape
        aspp = d.aspp
        cout = aspp.convs[0][0].out_channels
        cat = torch.empty((n, 5 * cout, h, w), dtype=torch.float32, device=x.device)
        xp = planes_of(x)
        for i in range(4):
            if xp is not None and planes_ok(aspp.convs[i][0]):     # res5 arrives with its planes twin: the branches read that (conv_s.hip)
                conv_abn_planes(xp, aspp.convs[i][0], aspp.convs[i][1], out=cat, out_coff=i * cout, want_planes=False)
            else:
                conv_abn(x, aspp.convs[i][0], aspp.convs[i][1], out=cat, out_coff=i * cout)
        g = ops.global_mean(x)
        pconv = aspp.convs[4].aspp_pooling[1]
        g = ops.conv2d(g, packed_weight(pconv), 1, act1=ops.ACT_RELU)
        ops.resize_bilinear(g, (h, w), True, out=cat, out_coff=4 * cout)
        y = conv_abn(cat, aspp.project[0], aspp.project[1])  # Dropout is identity in eval
        for i in range(d.decoder_stage):
            l = feats[d.low_level_key[i]]
            cp = d.project[i][0].out_channels
            cy = y.shape[1]
            hh, ww = l.shape[2:]
            cat = torch.empty((n, cy + cp, hh, ww), dtype=torch.float32, device=x.device)
            ops.resize_bilinear(y, (hh, ww), True, out=cat, out_coff=0)
            conv_abn(l, d.project[i][0], d.project[i][1], out=cat, out_coff=cy)
            y = self._dwsep_eval(cat, d.fuse[i])
        return y

    def _head_eval(self, y):
        pred = OrderedDict()
        n, _, h, w = y.shape
        for key in self.head.class_key:
            mod = self.head.classifier[key]
            z = self._dwsep_eval(y, mod["fuse"])
            tot = sum(m.out_channels for m in mod["cls"])
            out = torch.empty((n, tot, h, w), dtype=torch.float32, device=y.device)
            off = 0
            for m in mod["cls"]:
                conv_bias(z, m, out=out, out_coff=off)
                off += m.out_channels
            pred[key] = out
        return pred

    # ------------------------------------------------------------------ train (phase-2 trainable sub-net)
    def _dwsep_train(self, x, seq, tape):
        blk = seq[0]
        dw, bn0, pw, bn1 = blk[0][0], blk[0][1], blk[1], blk[2]
        if BN_FUSED_FINAL:        # the depthwise kernel leaves the partial sums of its output: BatchNorm reads z0 once
            z0, part, nparts = ops.dwconv5_stats(x, dw.weight.detach())
            a, m0, v0 = _bn_train(z0, bn0, partials=(part, nparts))
        else:
            z0 = ops.dwconv5(x, dw.weight.detach())
            a, m0, v0 = _bn_train(z0, bn0)
        z1 = ops.conv2d(a, packed_weight(pw, TRAIN_PRECISION), 1)
        y, m1, v1 = _bn_train(z1, bn1)
        tape.append(("dwsep", blk, x, z0, m0, v0, a, z1, m1, v1))
        return y

    def forward_train(self, feats, dropout_keep=None, seed=0):
        """Forward with batch statistics; returns (pred dict, tape). Inputs are treated as constants
        (IncrementalPanopticDeepLabDecoder.forward detaches them, modules/panoptic_deeplab.py:422-426).
        = decoder_forward_train + head_forward_train on one tape (the "fork" record separates the two halves)."""
        y, tape = self.decoder_forward_train(feats, dropout_keep, seed)
        pred, htape = self.head_forward_train(y)
        return pred, tape + htape

    def decoder_forward_train(self, feats, dropout_keep=None, seed=0):
        """SinglePanopticDeepLabDecoder.forward in training mode (modules/panoptic_deeplab.py:181-193): ASPP + three fusion
        stages. Returns (decoder features [n,128,h/4,w/4], tape)."""
        d = self.dec
        tape = []
        x = feats[d.feature_key]
        n, _, h, w = x.shape
        aspp = d.aspp
        cout = aspp.convs[0][0].out_channels
        cat5 = torch.empty((n, 5 * cout, h, w), dtype=torch.float32, device=x.device)
        recs = []
        # bf16x6 on the planes kernel (r03): the four ASPP branches read ONE three-plane copy of the (frozen, detached) res5 features —
        # 144-pixel tiles, no conversion in the K loop — instead of re-splitting the fp32 map per tile and tap in the r01 role-split kernel
        x3 = None
        if (X6_PLANES and TRAIN_PRECISION == 3 and ops.planes_f16() and x.is_contiguous() and ops.f16x3_ok(x.shape[1], aspp.convs[0][0].out_channels, 3)
                and all(ops.f16x3_ok(x.shape[1], aspp.convs[i][0].out_channels, 3) for i in range(4))):
            # r05, fp16x3: the four branches multiply the planes twin the frozen body left beside res5 (the build's two-plane format IS the
            # fp16x3 operand: fp16 hi / lo of 16 x) — three products per MAC at bf16x6's accuracy, and no conversion pass at all
            x3 = planes_of(x)
            if x3 is None or x3.planes != 2:
                x3 = ops.to_planes(x)
        elif (X6_PLANES and TRAIN_PRECISION >= 2 and x.is_contiguous() and x.shape[1] % 32 == 0
                and all(ops.bf16x3_ok(aspp.convs[i][0].out_channels) for i in range(4))):
            x3 = ops.to_planes(x, planes=3)
        for i in range(4):
            conv, bn = aspp.convs[i][0], aspp.convs[i][1]
            if x3 is not None and x3.planes == 2:           # (the 1x1 branch rides on the shared copy although it shrinks the channel count)
                wq = CACHE.get(conv, "wp3f", (conv.weight,), lambda: ops.pack_conv_weight(conv.weight.detach(), 3, force_f16=True))
                z = ops.conv2d_planes(x3, wq, conv.kernel_size[0], pad=conv.padding[0], dil=conv.dilation[0], want_fp32=True)[0]
            elif x3 is not None:
                z = ops.conv2d_planes(x3, packed_weight(conv, 2), conv.kernel_size[0], pad=conv.padding[0], dil=conv.dilation[0],
                                      want_fp32=True)[0]
            else:
                z = ops.conv2d(x, packed_weight(conv, TRAIN_PRECISION), conv.kernel_size[0], pad=conv.padding[0],
                               dil=conv.dilation[0])
            _, m, v = _bn_train(z, bn, out=cat5, out_coff=i * cout)
            recs.append((z, m, v))
        del x3
        g = ops.global_mean(x)
        pconv = aspp.convs[4].aspp_pooling[1]
        gp = ops.conv2d(g, packed_weight(pconv, TRAIN_PRECISION), 1, act1=ops.ACT_RELU)
        ops.resize_bilinear(gp, (h, w), True, out=cat5, out_coff=4 * cout)
        pz = ops.conv2d(cat5, packed_weight(aspp.project[0], TRAIN_PRECISION), 1)
        p = aspp.project[3].p
        if dropout_keep is None:
            dropout_keep = ops.dropout_mask((n, cout, h, w), p, seed, x.device)
        y, pm, pv = _bn_train(pz, aspp.project[1], keep=dropout_keep, keep_scale=1.0 / (1.0 - p))
        tape.append(("aspp", x, recs, g, gp, cat5, pz, pm, pv, dropout_keep, 1.0 / (1.0 - p)))
        for i in range(d.decoder_stage):
            l = feats[d.low_level_key[i]]
            conv, bn = d.project[i][0], d.project[i][1]
            cp, cy = conv.out_channels, y.shape[1]
            hh, ww = l.shape[2:]
            cat = torch.empty((n, cy + cp, hh, ww), dtype=torch.float32, device=x.device)
            ops.resize_bilinear(y, (hh, ww), True, out=cat, out_coff=0)
            lz = ops.conv2d(l, packed_weight(conv, TRAIN_PRECISION), 1)
            _, lm, lv = _bn_train(lz, bn, out=cat, out_coff=cy)
            tape.append(("stage", i, l, lz, lm, lv, cy, cp, y.shape[2:]))
            y = self._dwsep_train(cat, d.fuse[i], tape)
        return y, tape

    def head_forward_train(self, y):
        """IncrementalSinglePanopticDeepLabHead.forward in training mode (modules/panoptic_deeplab.py:457-468).
        Returns (OrderedDict center / offset, tape starting with the "fork" record)."""
        pred = OrderedDict()
        hn, _, hh, hw = y.shape
        tape = [("fork", y)]
        for key in self.head.class_key:
            mod = self.head.classifier[key]
            z = self._dwsep_train(y, mod["fuse"], tape)
            tot = sum(m.out_channels for m in mod["cls"])
            out = torch.empty((hn, tot, hh, hw), dtype=torch.float32, device=y.device)
            off = 0
            for m in mod["cls"]:
                conv_bias(z, m, out=out, out_coff=off)
                off += m.out_channels
            tape.append(("cls", key, z))
            pred[key] = out
        return pred, tape

    # ---- backward helpers ------------------------------------------------------------------------
    _on_ready = None
    _queue = ()

    def _grad(self, param, grads):
        """Gradient buffer of `param`. Every parameter of this sub-net receives its gradient from exactly one kernel,
        enqueued right after the (at most two) _grad calls that build its arguments. A parameter is therefore final on
        the compute stream once two later _grad calls have happened; that is when `on_ready` fires (the all-reduce
        stream orders itself behind the compute stream at launch, GradBuckets._launch)."""
        g = grads.get(id(param))
        if g is None:
            g = torch.zeros_like(param, dtype=torch.float32)
            grads[id(param)] = g
        if self._on_ready is not None:
            self._queue.append(param)
            while len(self._queue) > 2:
                self._on_ready(self._queue.pop(0))
        return g

    def _flush_ready(self):
        if self._on_ready is not None:
            for p in self._queue:
                self._on_ready(p)
        self._queue = []

    def _dwsep_bwd(self, rec, dy, grads, need_dx=True):
        _, blk, x, z0, m0, v0, a, z1, m1, v1 = rec
        dw, bn0, pw, bn1 = blk[0][0], blk[0][1], blk[1], blk[2]
        G = lambda p: self._grad(p, grads)
        dz1 = ops.bn_train_bwd(z1, dy, m1, v1, bn1.weight.detach(), bn1.bias.detach(), G(bn1.weight), G(bn1.bias), bn1.eps)
        ops.conv2d_wgrad(a, dz1, G(pw.weight), 1)
        # input gradient of a 1x1 conv: the PyTorch weight [cout, cin] already is the packed [K=cout][M=cin] operand
        da = _dx_1x1(dz1, pw)
        dz0 = ops.bn_train_bwd(z0, da, m0, v0, bn0.weight.detach(), bn0.bias.detach(), G(bn0.weight), G(bn0.bias), bn0.eps)
        ops.dwconv5_wgrad_(G(dw.weight), x, dz0)
        return ops.dwconv5(dz0, dw.weight.detach(), flip=True) if need_dx else None

    @ops.gradient_scope
    def backward(self, tape, dpred, grads=None, on_ready=None, feature_grads=None, stop_before_aspp=False):
        """Back-propagates d(loss)/d(pred[key]) through the recorded forward. Returns {id(param): grad}.
        `feature_grads` (a dict) additionally receives d(loss)/d(feature) for the backbone features the decoder read —
        step 0 trains the body through the instance branch; phases 1 / 2 detach them (modules/panoptic_deeplab.py:422-426).

        `grads` may map id(param) -> a pre-zeroed gradient view (the flat DP buffer); `on_ready(param)` is called right
        after the kernel that finalises a parameter's gradient has been enqueued (bucketed all-reduce hook).
        `stop_before_aspp`: stop after the decoder stages and return the gradient entering the ASPP block instead — the
        data-parallel graph mode replays everything up to here from a HIP graph and runs `backward_aspp` (90 % of the
        gradient bytes, the last kernels of the step) eagerly so that its buckets are all-reduced while it is still computing."""
        grads = {} if grads is None else grads
        self._on_ready = on_ready
        self._queue = []
        fork = max(i for i, r in enumerate(tape) if r[0] == "fork")
        dy = self._head_backward(tape[fork:], dpred, grads)
        dy = self._decoder_backward_stages(tape[:fork], dy, grads, feature_grads)
        if stop_before_aspp:
            self._flush_ready()
            return dy
        self._decoder_backward_aspp(tape[0], dy, grads, feature_grads)
        self._flush_ready()
        return grads

    @ops.gradient_scope
    def backward_aspp(self, tape, dy, grads, on_ready=None):
        """The ASPP part of `backward` on its own (see `stop_before_aspp`)."""
        self._on_ready = on_ready
        self._queue = []
        self._decoder_backward_aspp(tape[0], dy, grads, None)
        self._flush_ready()
        return grads

    def aspp_parameters(self):
        return list(self.dec.aspp.parameters())

    @ops.gradient_scope
    def head_backward(self, tape, dpred, grads=None):
        """Backward of head_forward_train alone: returns d(loss)/d(decoder features); parameter gradients go to `grads`."""
        grads = {} if grads is None else grads
        self._on_ready, self._queue = None, []
        return self._head_backward(tape, dpred, grads), grads

    @ops.gradient_scope
    def decoder_backward(self, tape, dy, grads=None, feature_grads=None):
        """Backward of decoder_forward_train alone."""
        grads = {} if grads is None else grads
        self._on_ready, self._queue = None, []
        dy = self._decoder_backward_stages(tape, dy, grads, feature_grads)
        self._decoder_backward_aspp(tape[0], dy, grads, feature_grads)
        return grads

    def _head_backward(self, tape, dpred, grads):
        G = lambda p: self._grad(p, grads)
        i = len(tape) - 1
        dfork = None
        # heads (recorded as dwsep, cls per key) ---------------------------------------------------
        while tape[i][0] != "fork":
            tag, key, z = tape[i]
            assert tag == "cls"
            rec = tape[i - 1]
            mod = self.head.classifier[key]
            dy = dpred[key]
            dz = None
            off = 0
            for m in mod["cls"]:
                co = m.out_channels
                ops.conv2d_wgrad(z, dy, G(m.weight), 1, dy_coff=off, cout=co)
                ops.bias_grad_(G(m.bias), dy, c=co, coff=off)
                # dz += W^T dy_slice : conv over the dy channel slice with the natural [cout][cin] weight
                part = ops.conv2d(dy, m.weight.detach().reshape(co, m.in_channels), 1, in_coff=off, cin=co)
                dz = part if dz is None else ops.axpy_(dz, part)
                off += co
            dx = self._dwsep_bwd(rec, dz, grads)
            dfork = dx if dfork is None else ops.axpy_(dfork, dx)
            i -= 2
        return dfork

    def _decoder_backward_stages(self, tape, dy, grads, feature_grads=None):
        G = lambda p: self._grad(p, grads)
        i = len(tape) - 1
        d = self.dec
        # decoder stages, last to first -----------------------------------------------------------------
        while tape[i][0] != "aspp":
            rec = tape[i]
            stage = tape[i - 1]
            _, si, l, lz, lm, lv, cy, cp, prev_hw = stage
            dcat = self._dwsep_bwd(rec, dy, grads)
            conv, bn = d.project[si][0], d.project[si][1]
            dlz = ops.bn_train_bwd(lz, dcat, lm, lv, bn.weight.detach(), bn.bias.detach(), G(bn.weight), G(bn.bias), bn.eps,
                                   dy_coff=cy)
            ops.conv2d_wgrad(l, dlz, G(conv.weight), 1)
            if feature_grads is not None:
                feature_grads[d.low_level_key[si]] = _dx_1x1(dlz, conv)
            dy = ops.resize_bilinear_bwd(dcat, tuple(prev_hw), True, dy_coff=0, c=cy)
            i -= 2
        assert i == 0
        return dy

    @staticmethod
    def _pool_branch_grad(dpz, dcat5, pconv0, cout):
        """Gradient of the image-pooling branch's [n, cout, 1, 1] output: the sum over the plane of the projection's input gradient in
        that branch's channels (the backward of `x.repeat(1, 1, h, w)`, modules/panoptic_deeplab.py:84-89). CL4_POOL_GRAD=1 (r04, VERDICT
        r03 item 2c) commutes the sum: sum_px (W^T dpz)[px] = W^T (sum_px dpz[px]) with the plane sum in fp64 — 256 rounded terms per
        output instead of 256 x 1089. Measured (FINDINGS.md finding 41): the pooling weight's distance from the float64 oracle does not
        move in the fourth digit (3.7603e-3 / 3.1669e-2 / 9.987e-3 at frozen precision 0 / 2 / 1, both orders), i.e. the noise is in dpz
        itself — sign decisions of the activation backward on a nearly cancelling sum — not in the order of the reduction, so the
        literal order stays the default."""
        if not POOL_GRAD_COMMUTED:
            return ops.resize_bilinear_bwd(dcat5, (1, 1), True, dy_coff=4 * cout, c=cout)
        ds = _dx_1x1(ops.plane_sum(dpz), pconv0)               # [n, 5 * cout, 1, 1]
        return ops.resize_bilinear_bwd(ds, (1, 1), True, dy_coff=4 * cout, c=cout)

    def _decoder_backward_aspp(self, rec, dy, grads, feature_grads=None):
        G = lambda p: self._grad(p, grads)
        d = self.dec
        _, x, recs, g, gp, cat5, pz, pm, pv, keep, kscale = rec
        aspp = d.aspp
        pbn = aspp.project[1]
        dpz = ops.bn_train_bwd(pz, dy, pm, pv, pbn.weight.detach(), pbn.bias.detach(), G(pbn.weight), G(pbn.bias), pbn.eps,
                               keep=keep, keep_scale=kscale)
        pconv0 = aspp.project[0]
        ops.conv2d_wgrad(cat5, dpz, G(pconv0.weight), 1)
        dcat5 = _dx_1x1(dpz, pconv0)
        cout = aspp.convs[0][0].out_channels
        dx = None
        h, w = x.shape[2:]
        pconv = aspp.convs[4].aspp_pooling[1]
        if feature_grads is None:
            # parameter order in the flat gradient buffer is convs.0 .. convs.4, project: finishing the small pooling branch
            # first and the four conv branches from the last to the first lets every bucket but the smallest (the 1x1 branch)
            # be all-reduced under the remaining weight-gradient kernels. The tensors are independent: same values.
            dgp = ops.relu_bwd(gp, self._pool_branch_grad(dpz, dcat5, pconv0, cout))
            ops.conv2d_wgrad(g, dgp, G(pconv.weight), 1)
        kcache = {}       # the three dilated branches share one pixel-major pack of x on the GEMM path of the weight gradient (ops._wgrad_gemm)
        for j in (range(4) if feature_grads is not None else (3, 2, 1, 0)):
            conv, bn = aspp.convs[j][0], aspp.convs[j][1]
            z, m, v = recs[j]
            dz = ops.bn_train_bwd(z, dcat5, m, v, bn.weight.detach(), bn.bias.detach(), G(bn.weight), G(bn.bias), bn.eps,
                                  dy_coff=j * cout)
            k, pad, dil = conv.kernel_size[0], conv.padding[0], conv.dilation[0]
            ops.conv2d_wgrad(x, dz, G(conv.weight), k, pad=pad, dil=dil, kcache=kcache)
            if feature_grads is not None:
                from .body_train import _dgrad_weight
                part = ops.conv2d_dgrad(dz, _dgrad_weight(conv), tuple(x.shape[2:]), k, pad=pad, dil=dil)
                dx = part if dx is None else ops.axpy_(dx, part)
        if feature_grads is not None:
            dgp = ops.relu_bwd(gp, self._pool_branch_grad(dpz, dcat5, pconv0, cout))
            ops.conv2d_wgrad(g, dgp, G(pconv.weight), 1)
            dg = _dx_1x1(dgp, pconv)
            feature_grads[d.feature_key] = ops.add_per_plane_(dx, dg, 1.0 / (h * w))

    def trainable_parameters(self):
        return list(self.dec.parameters()) + list(self.head.parameters())
