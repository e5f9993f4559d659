"""ORACLE (test infrastructure, not product code) — CPU restatement of the CL4WSIS networks.

Functional PyTorch-CPU fp32 restatement of the forward passes on the phase-2 hot path,
driven by a plain ``state_dict`` whose key names are the reference's checkpoint contract
(SURVEY.md Appendix B). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; the product (cl4wsis_amd) never does.

Pinned against outputs of the reference itself: tests/golden/make_golden.py imports the
upstream modules in the build container and the fixtures it writes are compared with this
restatement in tests/test_oracle_golden.py.

Third-party arithmetic that is NOT in the reference tree (parity unpinned at that boundary,
see DESIGN.md): ``inplace_abn`` is restated as BatchNorm(eps=1e-5) followed by
leaky-ReLU(0.01) / identity.

Each function cites the upstream file:line it follows.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

EPS = 1e-5
SLOPE = 0.01
RESNET101 = (3, 4, 23, 3)


IABN_ABS_GAMMA = True    # InPlaceABN(Sync) scales with |weight| + eps (believed behaviour of the absent inplace_abn package)


def _abn(x, sd, p, act=True, training=False):
    """InPlaceABN(Sync) — segmentation_module.py:17-29 (third-party, not vendored, unpinned): batch norm + leaky-ReLU 0.01 whose
    fused kernels read the scale as `abs(weight) + eps`; autograd of `abs` gives the package's d(weight) = sign(weight) * d(scale)."""
    w = sd[p + ".weight"]
    if IABN_ABS_GAMMA:
        w = w.abs() + EPS
    y = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], w, sd[p + ".bias"],
                     training, 0.1, EPS)
    return F.leaky_relu(y, SLOPE) if act else y


def _bn_relu(x, sd, p, training, relu=True):
    """nn.BatchNorm2d (+ReLU) of the Panoptic-DeepLab decoder — modules/panoptic_deeplab.py:16-29."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, 0.1, EPS)
    if training and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return F.relu(y) if relu else y


def bottleneck(x, sd, p, stride, dil, training=False):
    """ResidualBlock.forward — modules/residual.py:84-97 (bottleneck variant, :66-76)."""
    if (p + ".proj_conv.weight") in sd:
        r = F.conv2d(x, sd[p + ".proj_conv.weight"], stride=stride)
        r = _abn(r, sd, p + ".proj_bn", act=False, training=training)
    else:
        r = x
    y = _abn(F.conv2d(x, sd[p + ".convs.conv1.weight"]), sd, p + ".convs.bn1", training=training)
    y = _abn(F.conv2d(y, sd[p + ".convs.conv2.weight"], stride=stride, padding=dil, dilation=dil), sd, p + ".convs.bn2",
             training=training)
    y = _abn(F.conv2d(y, sd[p + ".convs.conv3.weight"]), sd, p + ".convs.bn3", act=False, training=training)
    return F.leaky_relu(y + r, SLOPE)


def resnet_body(x, sd, p="body", structure=RESNET101, output_stride=16, training=False):
    """ResNet.forward with keep_outputs — models/resnet.py:106-123; strides/dilations :49-55,101-104. training=True uses
    batch statistics (InPlaceABNSync in train mode: step 0 / phase 1) and updates the running statistics in `sd`."""
    dilation = [1, 1, 1, 2] if output_stride == 16 else [1, 1, 2, 4]
    y = F.conv2d(x, sd[p + ".mod1.conv1.weight"], stride=2, padding=3)
    y = _abn(y, sd, p + ".mod1.bn1", training=training)
    y = F.max_pool2d(y, 3, stride=2, padding=1)
    outs = [y]
    for mod_id, num in enumerate(structure):
        d = dilation[mod_id]
        for b in range(num):
            s = 2 if (d == 1 and b == 0 and mod_id > 0) else 1
            y = bottleneck(y, sd, f"{p}.mod{mod_id + 2}.block{b + 1}", s, d, training=training)
        outs.append(y)
    return outs


WRN_CHANNELS = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]
WRN38 = (3, 3, 6, 3, 1, 1)


def identity_block(x, sd, p, channels, stride, dil, training=False, drop_p=None, keep=None):
    """IdentityResidualBlock.forward — modules/residual.py:168-181. Returns (out, bn1). training=True: batch statistics and, in the
    modules that have one (models/wide_resnet.py:150-155), Dropout2d(drop_p) in front of the last convolution with the [n, c] keep
    mask `keep` (the reference draws it from torch's generator; tests inject it on both sides)."""
    a = _abn(x, sd, p + ".bn1", training=training)
    shortcut = F.conv2d(a, sd[p + ".proj_conv.weight"], stride=stride) if (p + ".proj_conv.weight") in sd else x

    def drop(y):
        if not training or drop_p is None:
            return y
        return y * keep.to(y.dtype)[:, :, None, None] / (1.0 - drop_p)

    if len(channels) == 2:
        y = _abn(F.conv2d(a, sd[p + ".convs.conv1.weight"], stride=stride, padding=dil, dilation=dil), sd, p + ".convs.bn2", training=training)
        y = F.conv2d(drop(y), sd[p + ".convs.conv2.weight"], padding=dil, dilation=dil)
    else:
        y = _abn(F.conv2d(a, sd[p + ".convs.conv1.weight"], stride=stride), sd, p + ".convs.bn2", training=training)
        y = _abn(F.conv2d(y, sd[p + ".convs.conv2.weight"], padding=dil, dilation=dil), sd, p + ".convs.bn3", training=training)
        y = F.conv2d(drop(y), sd[p + ".convs.conv3.weight"])
    return y + shortcut, a


WRN_DROPOUT = {4: 0.3, 5: 0.5}       # mod6, mod7 (models/wide_resnet.py:150-155)


def wider_resnet_a2_body(x, sd, p="body", structure=WRN38, training=False, keeps=None):
    """WiderResNetA2.forward with keep_outputs, dilation=True (output stride 8) — models/wide_resnet.py:122-207. training=True: batch
    statistics, Dropout2d masks from `keeps` ({"mod6.block1": uint8 [n, c], ...})."""
    out = F.conv2d(x, sd[p + ".mod1.conv1.weight"], padding=1)
    prevs = []
    for mod_id, num in enumerate(structure):
        if mod_id < 2:
            out = F.max_pool2d(out, 3, stride=2, padding=1)
        for b in range(num):
            dil = 2 if mod_id == 3 else (4 if mod_id > 3 else 1)
            stride = 2 if b == 0 and mod_id == 2 else 1
            key = f"mod{mod_id + 2}.block{b + 1}"
            out, a = identity_block(out, sd, f"{p}.{key}", WRN_CHANNELS[mod_id], stride, dil, training=training,
                                    drop_p=WRN_DROPOUT.get(mod_id), keep=None if keeps is None else keeps.get(key))
            if b == 0 and mod_id >= 2:
                prevs.append(a)
    return prevs + [_abn(out, sd, p + ".bn_out", training=training)]


def deeplab_head(x, sd, p="head", pooling=32, training=False, output_stride=16):
    """DeeplabV3.forward / _global_pooling — modules/deeplab.py:54-89."""
    dils = [6, 12, 18] if output_stride == 16 else [12, 24, 32]
    outs = [F.conv2d(x, sd[p + ".map_convs.0.weight"])]
    for i, d in enumerate(dils):
        outs.append(F.conv2d(x, sd[f"{p}.map_convs.{i + 1}.weight"], padding=d, dilation=d))
    out = _abn(torch.cat(outs, 1), sd, p + ".map_bn", training=training)
    out = F.conv2d(out, sd[p + ".red_conv.weight"])
    if training or pooling is None:
        pool = x.reshape(x.size(0), x.size(1), -1).mean(-1).reshape(x.size(0), x.size(1), 1, 1)
    else:
        kh, kw = min(pooling, x.shape[2]), min(pooling, x.shape[3])
        pad = ((kw - 1) // 2, (kw - 1) // 2 if kw % 2 == 1 else (kw - 1) // 2 + 1,
               (kh - 1) // 2, (kh - 1) // 2 if kh % 2 == 1 else (kh - 1) // 2 + 1)
        pool = F.pad(F.avg_pool2d(x, (kh, kw), stride=1), pad=pad, mode="replicate")
    pool = F.conv2d(pool, sd[p + ".global_pooling_conv.weight"])
    pool = _abn(pool, sd, p + ".global_pooling_bn", training=training)
    pool = F.conv2d(pool, sd[p + ".pool_red_conv.weight"])
    if training or pooling is None:
        pool = pool.repeat(1, 1, x.size(2), x.size(3))
    return _abn(out + pool, sd, p + ".red_bn", training=training)


def incremental_classifier(x, sd, p="cls"):
    """IncrementalClassifier.forward — segmentation_module.py:107-113."""
    outs, k = [], 0
    while f"{p}.{k}.weight" in sd:
        outs.append(F.conv2d(x, sd[f"{p}.{k}.weight"], sd[f"{p}.{k}.bias"]))
        k += 1
    return torch.cat(outs, 1)


def _dw_separable(x, sd, p, training):
    """depthwise_separable_conv (5x5 dw + BN + ReLU, 1x1 + BN + ReLU) — modules/panoptic_deeplab.py:32-47."""
    c = x.shape[1]
    y = F.conv2d(x, sd[p + ".0.0.weight"], padding=2, groups=c)
    y = _bn_relu(y, sd, p + ".0.1", training)
    y = F.conv2d(y, sd[p + ".1.weight"])
    return _bn_relu(y, sd, p + ".2", training)


def instance_decoder(feats, sd, p="decoder.instance_decoder", training=False, dropout_mask=None, rates=(3, 6, 9)):
    """SinglePanopticDeepLabDecoder.forward — modules/panoptic_deeplab.py:181-193 with
    ASPP.forward :128-133, ASPPConv :70-77, ASPPPooling :80-98. Inputs are detached by the caller
    (IncrementalPanopticDeepLabDecoder.forward :413-430, detach_instance)."""
    x = feats["res5"]
    a = p + ".aspp"
    res = [_bn_relu(F.conv2d(x, sd[a + ".convs.0.0.weight"]), sd, a + ".convs.0.1", training)]
    for i, r in enumerate(rates):
        y = F.conv2d(x, sd[f"{a}.convs.{i + 1}.0.weight"], padding=r, dilation=r)
        res.append(_bn_relu(y, sd, f"{a}.convs.{i + 1}.1", training))
    g = F.relu(F.conv2d(F.adaptive_avg_pool2d(x, 1), sd[a + ".convs.4.aspp_pooling.1.weight"]))
    res.append(F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=True))
    y = F.conv2d(torch.cat(res, 1), sd[a + ".project.0.weight"])
    y = _bn_relu(y, sd, a + ".project.1", training)
    if training:  # nn.Dropout(0.5): mask injected so that parity tests are RNG independent
        if dropout_mask is None:
            dropout_mask = (torch.rand_like(y) >= 0.5).to(y.dtype)
        y = y * dropout_mask * 2.0
    for i, key in enumerate(("res4", "res3", "res2")):
        l = F.conv2d(feats[key], sd[f"{p}.project.{i}.0.weight"])
        l = _bn_relu(l, sd, f"{p}.project.{i}.1", training)
        y = F.interpolate(y, size=l.shape[2:], mode="bilinear", align_corners=True)
        y = _dw_separable(torch.cat((y, l), 1), sd, f"{p}.fuse.{i}.0", training)
    return y


def instance_head(x, sd, p="instance_head.classifier", training=False):
    """IncrementalSinglePanopticDeepLabHead.forward — modules/panoptic_deeplab.py:458-468."""
    pred = OrderedDict()
    for key in ("center", "offset"):
        y = _dw_separable(x, sd, f"{p}.{key}.fuse.0", training)
        outs, k = [], 0
        while f"{p}.{key}.cls.{k}.weight" in sd:
            outs.append(F.conv2d(y, sd[f"{p}.{key}.cls.{k}.weight"], sd[f"{p}.{key}.cls.{k}.bias"]))
            k += 1
        pred[key] = torch.cat(outs, 1)
    return pred


def _body(x, sd):
    """ResNet (output stride 16) or, when the state dict holds `body.mod7`, WiderResNet-38 A2 (output stride 8)."""
    if any(k.startswith("body.mod7.") for k in sd):
        return wider_resnet_a2_body(x, sd), 8
    return resnet_body(x, sd), 16


def body_any(x, sd, structure=None, training=False, keeps=None):
    """(outputs, output stride) of whichever body the state dict holds, in eval or training mode (oracle/step.py)."""
    if any(k.startswith("body.mod7.") for k in sd):
        return wider_resnet_a2_body(x, sd, structure=structure or WRN38, training=training, keeps=keeps), 8
    return resnet_body(x, sd, structure=structure or RESNET101, training=training), 16


def forward_seg(x, sd, pooling=32, training=False):
    """IncrementalInstanceSegmentationModule.forward_seg(interpolate=False) — segmentation_module.py:400-426."""
    outs, os_ = _body(x, sd)
    feats = {f"res{i + 1}": o for i, o in enumerate(outs)}
    seg = incremental_classifier(deeplab_head(outs[-1], sd, pooling=pooling, training=training, output_stride=os_), sd)
    return OrderedDict(seg=seg), {"body": outs[-1], "features": feats}


def forward_full(x, sd, pooling=32, training=False, dropout_mask=None):
    """IncrementalInstanceSegmentationModule.forward(interpolate=False) — segmentation_module.py:360-398."""
    pred, aux = forward_seg(x, sd, pooling=pooling, training=training)
    det = {k: v.detach() for k, v in aux["features"].items()}
    ins = instance_decoder(det, sd, training=training, dropout_mask=dropout_mask)
    pred.update(instance_head(ins, sd, training=training))
    return pred, aux


def pseudolabeler(x, sd, p="", training=False):
    """PseudoLabeler.forward — wss/modules.py:335-343 (training=True: batch statistics, phase 1)."""
    y = _abn(F.conv2d(x, sd[p + "conv1.weight"], padding=1), sd, p + "norm1", training=training)
    y = _abn(F.conv2d(y, sd[p + "conv2.weight"], padding=1), sd, p + "norm2", training=training)
    return F.conv2d(y, sd[p + "cls.weight"], sd[p + "cls.bias"])


def peak_generator_eval(x, sd, label, old_classes, alpha=0.7, p=""):
    """PeakGenerator.forward in eval mode — wss/modules.py:390-434 with PAM.forward :354-368.
    `old_classes` excludes background (train.py:89). Returns (logit [B,C], cam [B,C,H,W])."""
    w = sd[p + "extra_conv4.weight"]
    new = w.shape[0]
    B, _, H, W = x.shape
    y = F.relu(x[:, -new:])
    peak = y.amax(dim=(2, 3), keepdim=True)
    y = torch.where(y < peak * alpha, torch.zeros_like(y), y)
    y = F.conv2d(y, w, sd[p + "extra_conv4.bias"])
    m = F.softmax(y, dim=1).reshape(B, new, -1)
    logit = (y.reshape(B, new, -1) * m).sum(-1) / (1.0 + m.sum(-1))
    if old_classes > 0:
        logit = torch.cat([logit.new_zeros(B, old_classes), logit], 1)
        y = torch.cat([y.new_zeros(B, old_classes, H, W), y], 1)
    cam = F.relu(y) * label[:, :, None, None]
    cam = F.interpolate(cam, size=(H, W), mode="bilinear", align_corners=False)
    cam = cam / (F.adaptive_max_pool2d(cam, 1) + 1e-5)
    return logit, cam
