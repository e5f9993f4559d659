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

    def planes_capable(self, min_cout=None):
        c = self.convs
        return (all(planes_ok(m, min_cout) for m in (c.conv1, c.conv2, c.conv3))
                and (not hasattr(self, "proj_conv") or planes_ok(self.proj_conv, min_cout)))

    def fusable(self):
        """Block that csrc/conv_f.hip runs as ONE launch (frozen networks, planes trunk): c_in -> cm -> cm (3x3, stride 1, dilation 1)
        -> 4 cm with leaky-ReLU ABNs after conv1 / conv2, an identity-activation ABN after conv3 and leaky-ReLU on the sum; either an identity
        block (c_in = 4 cm) or — width 64 only — a stage's first block with a stride-1 projection shortcut from c_in = cm channels (res2's
        block 1, models/resnet.py:68-77)."""
        c = self.convs
        cm = c.conv1.out_channels
        ok = (ops.FUSED_BLOCK and cm in ops.FUSED_CM and planes_ok(c.conv3)
              and c.conv1.kernel_size == (1, 1) and c.conv3.kernel_size == (1, 1) and c.conv2.kernel_size == (3, 3)
              and c.conv2.stride == (1, 1) and c.conv2.dilation == (1, 1) and c.conv2.padding == (1, 1) and c.conv2.in_channels == cm
              and c.conv2.out_channels == cm and c.conv3.out_channels == 4 * cm
              and all(isinstance(b, ABN) for b in (c.bn1, c.bn2, c.bn3))
              and c.bn1.activation == "leaky_relu" and c.bn2.activation == "leaky_relu" and c.bn3.activation == "identity")
        if not ok:
            return False
        if not hasattr(self, "proj_conv"):
            return c.conv1.in_channels == 4 * cm
        pc, pb = self.proj_conv, self.proj_bn
        return (ops.FUSED_PROJ and cm == 64 and c.conv1.in_channels == cm and pc.in_channels == cm and pc.out_channels == 4 * cm and pc.kernel_size == (1, 1)
                and pc.stride == (1, 1) and isinstance(pb, ABN) and pb.activation == "identity")

    def forward_fused(self, xp, want_fp32, f32_images=None):
        """Eval forward of a fusable block on a planes input: (fp32 output or None, planes output), one launch."""
        import torch
        from .nn import CACHE, folded_bn
        c = self.convs
        bns = (c.bn1, c.bn2, c.bn3) + ((self.proj_bn,) if hasattr(self, "proj_conv") else ())
        srcs = tuple(t for b in bns for t in (b.weight, b.bias, b.running_mean, b.running_var))
        aff = CACHE.get(self, "faff", srcs, lambda: torch.cat([t for b in bns for t in folded_bn(b)]).contiguous())
        return ops.bottleneck_fused(xp, packed_weight(c.conv1, 1), packed_weight(c.conv2, 1), packed_weight(c.conv3, 1), aff,
                                    float(c.bn1.activation_param), float(c.bn2.activation_param), float(c.bn1.activation_param),
                                    want_fp32=want_fp32, f32_images=f32_images,
                                    wp=packed_weight(self.proj_conv, 1) if hasattr(self, "proj_conv") else None)

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
    # Hint of the caller (Trainer, phase 2): only the first `f32_images` images of the batch need the fp32 copies of the low-level stage
    # outputs res2 / res3 (train.py:265-267 hands the un-flipped half of the flip-TTA batch to the decoder; nothing reads the other
    # half's low-level features). None: all. Honoured by the fused blocks only; the planes twins always cover every image.
    f32_images = None

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

    @staticmethod
    def _stage_min_cout(stage):
        """Thinnest layer kept on the planes path in this stage: the default threshold, lowered to the stage's middle width when its
        identity blocks run fused (they need a planes trunk, so the stage's first block has to produce one)."""
        fused = [b.convs.conv1.out_channels for b in stage if b.fusable()]
        return min(fused) if fused else None

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
                                    want_planes=CONV_PLANES and self.mod2[0].planes_capable(self._stage_min_cout(self.mod2)))
            outs = [y0]
        else:
            y = conv_abn(x, c1, b1)
            outs = [ops.maxpool3x3s2(y)]
        for m in (self.mod2, self.mod3, self.mod4, self.mod5):
            y = outs[-1]
            lim = self._stage_min_cout(m) if frozen else None
            for i, blk in enumerate(m):
                if frozen and yp is not None and blk.fusable():
                    y, yp = blk.forward_fused(yp, want_fp32=(i == len(m) - 1), f32_images=self.f32_images)                          # conv1 -> conv2 -> conv3 + residual, one launch (csrc/conv_f.hip)
                elif frozen and blk.planes_capable(lim):
                    y, yp = blk.forward_planes(y, yp, want_fp32=(i == len(m) - 1), own_input=(i > 0))     # fp32 only where a stage output is handed out; from the second block on the input belongs to this loop
                else:
                    # with the `f32_images` hint a fused stage's fp32 copy holds the first images only: the planes twin is the whole map
                    partial = yp is not None and self.f32_images is not None
                    y, yp = blk(ops.from_planes(yp) if (y is None or partial) else y), None
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
