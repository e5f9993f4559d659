"""ResNet-{50,101} body (output stride 8/16, five stage outputs) on the HIP conv kernels.

Mirrors the state-dict layout of the reference (models/resnet.py:11-123, modules/residual.py:7-97):
``mod1.conv1 / mod1.bn1``, ``mod{2..5}.block{i}.convs.{conv1,bn1,conv2,bn2,conv3,bn3}``,
``.proj_conv / .proj_bn``. Every conv + InPlaceABN + (residual add + leaky-ReLU) group is ONE
kernel launch (fused epilogue of cl4_conv2d_fwd).
"""
from collections import OrderedDict

import torch.nn as nn

from . import ops
from .nn import ABN, conv_abn, conv_abn_planes, gamma_of, packed_weight, planes_of, planes_ok


import os

INPLACE_TRUNK = os.environ.get("CL4_INPLACE_TRUNK", "1") != "0"   # identity blocks overwrite their residual operand (forward_planes)
STEM_FUSED = os.environ.get("CL4_STEM_FUSED", "1") != "0"      # 0: conv + ABN launch followed by a max-pool launch (the r02 path)


class _Holder(nn.Module):
    """A named container of sub-modules (keeps state_dict prefixes; has no forward of its own)."""

    def __init__(self, mods=None):
        super().__init__()
        for k, v in (mods or {}).items():
            self.add_module(k, v)


class ResidualBlock(nn.Module):
    """Bottleneck residual block — modules/residual.py:31-97."""

    def __init__(self, in_channels, channels, stride=1, dilation=1, norm_act=ABN):
        super().__init__()
        if len(channels) != 3:
            raise ValueError("only bottleneck blocks (three channel counts) are on the CL4WSIS path")
        bn3 = norm_act(channels[2])
        bn3.activation = "identity"
        self.convs = _Holder(OrderedDict([
            ("conv1", nn.Conv2d(in_channels, channels[0], 1, bias=False)),
            ("bn1", norm_act(channels[0])),
            ("conv2", nn.Conv2d(channels[0], channels[1], 3, stride=stride, padding=dilation, dilation=dilation, bias=False)),
            ("bn2", norm_act(channels[1])),
            ("conv3", nn.Conv2d(channels[1], channels[2], 1, bias=False)),
            ("bn3", bn3),
        ]))
        if stride != 1 or in_channels != channels[-1]:
            self.proj_conv = nn.Conv2d(in_channels, channels[-1], 1, stride=stride, bias=False)
            self.proj_bn = norm_act(channels[-1])
            self.proj_bn.activation = "identity"

    def forward(self, x):
        if self.training and any(p.requires_grad for p in self.parameters()):
            # a single block has no stand-alone training entry: the body trains as a whole (batch statistics, tape, backward)
            raise RuntimeError("train-mode ResidualBlock: call the owning model / body (cl4wsis_amd.body_train.body_forward, or "
                               "model(images) in training mode, which runs it under torch.autograd)")
        c = self.convs
        r = conv_abn(x, self.proj_conv, self.proj_bn) if hasattr(self, "proj_conv") else x
        y = conv_abn(x, c.conv1, c.bn1)
        y = conv_abn(y, c.conv2, c.bn2)
        act, slope = c.bn1.act_code()
        return conv_abn(y, c.conv3, c.bn3, post=r, act2=act, slope2=slope)

    def planes_capable(self):
        c = self.convs
        return all(planes_ok(m) for m in (c.conv1, c.conv2, c.conv3)) and (not hasattr(self, "proj_conv") or planes_ok(self.proj_conv))

    def forward_planes(self, x, xp, want_fp32, own_input=False):
        """Eval forward with planes activations (frozen networks only). `x`: fp32 input or None, `xp`: its planes twin or None (at
        least one). Returns (fp32 output or None, planes output). With a planes input the four convolutions stay in planes; a block
        that only has an fp32 input (the first eligible block after the 64-channel stage) runs its first convolutions on the fp32
        kernels and splits the 3x3 output once for conv3."""
        c = self.convs
        act, slope = c.bn1.act_code()
        if xp is None and x.shape[1] <= 64:                    # a thin fp32 input (the pooled stem output): split it once, stay in planes
            xp = ops.to_planes(x)
        if xp is not None:
            rp = conv_abn_planes(xp, self.proj_conv, self.proj_bn)[1] if hasattr(self, "proj_conv") else xp
            y = conv_abn_planes(xp, c.conv1, c.bn1)[1]
            y = conv_abn_planes(y, c.conv2, c.bn2)[1]
            # Identity blocks write the new trunk IN PLACE over the old one (their residual operand): every element is read and then
            # written by the same thread of conv3's epilogue, conv1 / conv2 of this block have consumed the old trunk (stream order), and
            # nobody else holds it (a stage's first block has a projection shortcut and is excluded; `own_input` is False for an input the
            # caller still needs). The live set of a res4 block at 32 images drops from 358 to 215 MB — inside the 256 MB Infinity Cache,
            # so the residual read and much of the write no longer go to HBM. CL4_INPLACE_TRUNK=0: a fresh output tensor per block.
            inplace = INPLACE_TRUNK and own_input and rp is xp
            return conv_abn_planes(y, c.conv3, c.bn3, post=rp, act2=act, slope2=slope, want_fp32=want_fp32,
                                   out_planes=rp if inplace else None)
        r = conv_abn(x, self.proj_conv, self.proj_bn) if hasattr(self, "proj_conv") else x
        y = conv_abn(x, c.conv1, c.bn1)
        y = conv_abn(y, c.conv2, c.bn2)
        return conv_abn_planes(ops.to_planes(y), c.conv3, c.bn3, post=r, act2=act, slope2=slope, want_fp32=want_fp32)


class ResNet(nn.Module):
    def __init__(self, structure, norm_act=ABN, output_stride=16, keep_outputs=False):
        super().__init__()
        if len(structure) != 4:
            raise ValueError("Expected a structure with four values")
        if output_stride not in (8, 16):
            raise ValueError("Output stride must be 8 or 16")
        self.structure = structure
        self.keep_outputs = keep_outputs
        self.dilation = [1, 1, 1, 2] if output_stride == 16 else [1, 1, 2, 4]
        self.mod1 = _Holder(OrderedDict([
            ("conv1", nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)),
            ("bn1", norm_act(64)),
        ]))
        in_ch, channels = 64, (64, 64, 256)
        for mod_id, num in enumerate(structure):
            blocks = OrderedDict()
            d = self.dilation[mod_id]
            for b in range(num):
                s = 2 if (d == 1 and b == 0 and mod_id > 0) else 1
                blocks[f"block{b + 1}"] = ResidualBlock(in_ch, channels, stride=s, dilation=d, norm_act=norm_act)
                in_ch = channels[-1]
            self.add_module(f"mod{mod_id + 2}", nn.Sequential(blocks))
            channels = tuple(c * 2 for c in channels)
        self.out_channels = in_ch

    def forward(self, x):
        frozen = not (self.training and any(p.requires_grad for p in self.parameters()))
        yp = None                                              # planes twin of y (nn.CONV_PLANES), carried across blocks and stages
        c1, b1 = self.mod1.conv1, self.mod1.bn1
        if (frozen and STEM_FUSED and ops.CONV_PRECISION == 1 and isinstance(b1, ABN) and b1.activation == "leaky_relu"
                and x.is_contiguous()):
            # conv 7x7/2 + ABN + max-pool in one kernel (csrc/stem.hip): the 257 x 257 x 64 map is never written
            from .nn import CACHE, CONV_PLANES, folded_bn
            wq = CACHE.get(c1, "stemq", (c1.weight,), lambda: ops.pack_stem_weight(c1.weight.detach()))
            scale, shift = folded_bn(b1)
            y0, yp = ops.stem_fused(x, wq, scale, shift, slope=float(b1.activation_param),
                                    want_planes=CONV_PLANES and self.mod2[0].planes_capable())
            outs = [y0]
        else:
            y = conv_abn(x, c1, b1)
            outs = [ops.maxpool3x3s2(y)]
        for m in (self.mod2, self.mod3, self.mod4, self.mod5):
            y = outs[-1]
            for i, blk in enumerate(m):
                if frozen and blk.planes_capable():
                    y, yp = blk.forward_planes(y, yp, want_fp32=(i == len(m) - 1), own_input=(i > 0))     # fp32 only where a stage output is handed out; from the second block on the input belongs to this loop
                else:
                    y, yp = blk(y if y is not None else ops.from_planes(yp)), None
            if yp is not None:
                y._cl4_planes = yp                             # consumers that can read planes (DeeplabV3) pick it up
            outs.append(y)
        return outs if self.keep_outputs else outs[-1]


_NETS = {"50": [3, 4, 6, 3], "101": [3, 4, 23, 3], "152": [3, 8, 36, 3]}


def net_resnet50(**kw):
    return ResNet(_NETS["50"], **kw)


def net_resnet101(**kw):
    return ResNet(_NETS["101"], **kw)


def net_resnet152(**kw):
    return ResNet(_NETS["152"], **kw)


# ---------------------------------------------------------------------------------------------------------------------
# WiderResNet-38 A2 (coco-voc configuration, argparser.py:13-17): pre-activation identity-mapping blocks
# ---------------------------------------------------------------------------------------------------------------------
def _abn_eval(x, bn):
    """Stand-alone InPlaceABN in eval mode (running statistics) — the pre-activation `bn1` of a block reads the SUM of the
    previous block, so it cannot ride in a producing convolution's epilogue."""
    act, slope = bn.act_code()
    return ops.bn_apply(x, bn.running_mean, bn.running_var, gamma_of(bn), bn.bias.detach(), eps=bn.eps, act=act, slope=slope)


class IdentityResidualBlock(nn.Module):
    """modules/residual.py:101-181 (eval path; Dropout2d of mod6 / mod7 is the identity in eval mode)."""

    def __init__(self, in_channels, channels, stride=1, dilation=1, norm_act=ABN, dropout=None):
        super().__init__()
        if len(channels) not in (2, 3):
            raise ValueError("channels must contain either two or three values")
        self.bn1 = norm_act(in_channels)
        if len(channels) == 2:
            layers = [("conv1", nn.Conv2d(in_channels, channels[0], 3, stride=stride, padding=dilation, dilation=dilation, bias=False)),
                      ("bn2", norm_act(channels[0])),
                      ("conv2", nn.Conv2d(channels[0], channels[1], 3, padding=dilation, dilation=dilation, bias=False))]
        else:
            layers = [("conv1", nn.Conv2d(in_channels, channels[0], 1, stride=stride, bias=False)),
                      ("bn2", norm_act(channels[0])),
                      ("conv2", nn.Conv2d(channels[0], channels[1], 3, padding=dilation, dilation=dilation, bias=False)),
                      ("bn3", norm_act(channels[1])),
                      ("conv3", nn.Conv2d(channels[1], channels[2], 1, bias=False))]
        self.convs = _Holder(OrderedDict(layers))
        if stride != 1 or in_channels != channels[-1]:
            self.proj_conv = nn.Conv2d(in_channels, channels[-1], 1, stride=stride, bias=False)

    def forward(self, x, ret_prev=False):
        if self.training and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("train-mode IdentityResidualBlock: call the owning model / body (cl4wsis_amd.body_train.body_forward, or "
                               "model(images) in training mode, which runs it under torch.autograd)")
        c = self.convs
        a = _abn_eval(x, self.bn1)
        if hasattr(self, "proj_conv"):
            pc = self.proj_conv
            shortcut = ops.conv2d(a, packed_weight(pc), 1, stride=pc.stride[0])
        else:
            shortcut = x
        y = conv_abn(a, c.conv1, c.bn2)
        if hasattr(c, "conv3"):
            y = conv_abn(y, c.conv2, c.bn3)
            last = c.conv3
        else:
            last = c.conv2
        out = ops.conv2d(y, packed_weight(last), last.kernel_size[0], pad=last.padding[0], dil=last.dilation[0], post=shortcut)
        return (out, a) if ret_prev else out


class WiderResNetA2(nn.Module):
    """models/wide_resnet.py:82-207 — returns [bn1 of the first block of mod4..mod7, bn_out] with keep_outputs."""

    def __init__(self, structure, norm_act=ABN, classes=0, dilation=True, output_stride=None, keep_outputs=False):
        super().__init__()
        if len(structure) != 6:
            raise ValueError("Expected a structure with six values")
        self.structure, self.dilation, self.keep_outputs = structure, dilation, keep_outputs
        self.mod1 = _Holder(OrderedDict([("conv1", nn.Conv2d(3, 64, 3, stride=1, padding=1, bias=False))]))
        in_ch = 64
        channels = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]
        for mod_id, num in enumerate(structure):
            blocks = OrderedDict()
            for b in range(num):
                if not dilation:
                    dil, stride = 1, (2 if b == 0 and 2 <= mod_id <= 4 else 1)
                else:
                    dil = 2 if mod_id == 3 else (4 if mod_id > 3 else 1)
                    stride = 2 if b == 0 and mod_id == 2 else 1
                blocks[f"block{b + 1}"] = IdentityResidualBlock(in_ch, channels[mod_id], stride=stride, dilation=dil, norm_act=norm_act)
                in_ch = channels[mod_id][-1]
            self.add_module(f"mod{mod_id + 2}", nn.Sequential(blocks))
        self.out_channels = in_ch
        self.bn_out = norm_act(in_ch)

    def forward(self, img):
        mc = self.mod1.conv1
        out = ops.conv2d(img, packed_weight(mc), 3, pad=1)
        for name in ("mod2", "mod3"):               # pool2 / pool3: MaxPool2d(3, 2, 1) in front of mod2 / mod3
            out = ops.maxpool3x3s2(out)
            for blk in getattr(self, name):
                out = blk(out)
        prevs = []
        for name in ("mod4", "mod5", "mod6", "mod7"):
            for i, blk in enumerate(getattr(self, name)):
                if i == 0:
                    out, a = blk(out, ret_prev=True)
                    prevs.append(a)
                else:
                    out = blk(out)
        out7 = _abn_eval(out, self.bn_out)
        return prevs + [out7] if self.keep_outputs else out7


_WNETS = {"16": [1, 1, 1, 1, 1, 1], "20": [1, 1, 1, 3, 1, 1], "38": [3, 3, 6, 3, 1, 1]}


def net_wider_resnet38_a2(**kw):
    return WiderResNetA2(_WNETS["38"], **kw)


def net_wider_resnet20_a2(**kw):
    return WiderResNetA2(_WNETS["20"], **kw)


def net_wider_resnet16_a2(**kw):
    return WiderResNetA2(_WNETS["16"], **kw)
