"""Train-mode forward and hand-written backward of the semantic head (DeepLabV3), the incremental classifier and the
PseudoLabeler (SURVEY.md §8 A3, A4, A8 with gradients; phase 1 / step 0). Same tape conventions as body_train.py."""
import torch

from . import ops
from . import body_train as BT
from .body_train import _Rec, _dgrad_weight, conv_abn_backward, conv_abn_train
from .nn import CACHE, gamma_grad, gamma_of, packed_weight


def _abn_fwd(z, bn, group, emit_planes=False):
    act, slope = bn.act_code()
    y, mean, var, count = ops.abn_train_forward(z, gamma_of(bn), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                bn.momentum, bn.eps, act, slope, group, want_planes=emit_planes and BT._twin_pays(z.shape[1]))
    CACHE.touch(bn.running_mean)
    CACHE.touch(bn.running_var)
    return y, (z, mean, var, count)


def _abn_bwd(rec, bn, dy, grad_of, group):
    z, mean, var, count = rec
    act, slope = bn.act_code()
    gw, gsign = gamma_grad(bn, grad_of)
    dz = ops.abn_train_backward(z, dy, mean, var, gamma_of(bn), bn.bias.detach(), gw, grad_of(bn.bias), count,
                                bn.eps, act, slope, group, gamma_sign=gsign)
    return dz


def _conv1x1_bwd(x, conv, dz, grad_of, need_dx=True):
    ops.conv2d_wgrad(x, dz, grad_of(conv.weight), 1, precision=BT.WGRAD_PRECISION)
    if not need_dx:
        return None
    if BT.PRECISION:
        return ops.conv2d(dz, _dgrad_weight(conv), 1, x6=True)
    return ops.conv2d(dz, conv.weight.detach().reshape(conv.out_channels, conv.in_channels), 1)


# ---------------------------------------------------------------------------------------------------------------------
# DeepLabV3 (modules/deeplab.py:54-76, training branch: global average pooling, `repeat`)
# ---------------------------------------------------------------------------------------------------------------------
def deeplab_forward(head, x, group=None):
    n, _, h, w = x.shape
    hid = head.hidden
    cat = torch.empty((n, 4 * hid, h, w), dtype=torch.float32, device=x.device)
    shared = {}                                                       # fp16x3: the four branches read ONE converted copy of x
    for i, conv in enumerate(head.map_convs):
        ops.conv2d(x, packed_weight(conv, BT.PRECISION), conv.kernel_size[0], pad=conv.padding[0], dil=conv.dilation[0], out=cat, out_coff=i * hid, x6=True,
                   xs_cache=shared)
    a, r_map = _abn_fwd(cat, head.map_bn, group, emit_planes=True)
    out = ops.conv2d(a, packed_weight(head.red_conv, BT.PRECISION, force_f16=BT.PRECISION == 3 and getattr(a, "_cl4_planes", None) is not None), 1, x6=True)
    g = ops.global_mean(x)                                            # [n, cin, 1, 1]
    pz = ops.conv2d(g, packed_weight(head.global_pooling_conv, BT.PRECISION), 1)
    pa, r_pool = _abn_fwd(pz, head.global_pooling_bn, group)
    pr = ops.conv2d(pa, packed_weight(head.pool_red_conv, BT.PRECISION), 1)      # [n, cout, 1, 1]
    ops.add_per_plane_(out, pr)                                       # out += pool.repeat(...)
    y, r_red = _abn_fwd(out, head.red_bn, group)
    return y, (x, a, r_map, g, pa, r_pool, r_red)


@ops.gradient_scope
def deeplab_backward(head, tape, dy, grad_of, group=None):
    x, a, r_map, g, pa, r_pool, r_red = tape
    n, cin, h, w = x.shape
    hid = head.hidden
    dout = _abn_bwd(r_red, head.red_bn, dy, grad_of, group)
    # image-pooling branch: the broadcast add sums the gradient over the plane
    dpr = ops.global_mean(dout)
    ops.scale_(dpr, float(h * w))
    dpa = _conv1x1_bwd(pa, head.pool_red_conv, dpr, grad_of)
    dpz = _abn_bwd(r_pool, head.global_pooling_bn, dpa, grad_of, group)
    dg = _conv1x1_bwd(g, head.global_pooling_conv, dpz, grad_of)
    # map branch
    da = _conv1x1_bwd(a, head.red_conv, dout, grad_of)
    dcat = _abn_bwd(r_map, head.map_bn, da, grad_of, group)
    dx = None
    kcache = {}           # one pixel-major pack of x for the dilated branches (GEMM path of the weight gradient)
    for i, conv in enumerate(head.map_convs):
        k, p, d = conv.kernel_size[0], conv.padding[0], conv.dilation[0]
        ops.conv2d_wgrad(x, dcat, grad_of(conv.weight), k, pad=p, dil=d, dy_coff=i * hid, cout=hid, precision=BT.WGRAD_PRECISION, kcache=kcache)
        if k == 1:
            part = ops.conv2d(dcat, conv.weight.detach().reshape(hid, cin), 1, in_coff=i * hid, cin=hid)
        else:
            part = ops.conv2d(dcat, _dgrad_weight(conv), k, pad=d * (k - 1) - p, dil=d, in_coff=i * hid, cin=hid)
        dx = part if dx is None else ops.axpy_(dx, part)
    return ops.add_per_plane_(dx, dg, 1.0 / (h * w))                  # d(mean) spreads evenly over the plane


# ---------------------------------------------------------------------------------------------------------------------
# IncrementalClassifier (segmentation_module.py:107-113)
# ---------------------------------------------------------------------------------------------------------------------
def cls_forward(cls, x):
    return cls._forward(x), x


@ops.gradient_scope
def cls_backward(cls, x, dy, grad_of, need_dx=True):
    dx, off = None, 0
    for m in cls:
        co = m.out_channels
        if m.weight.requires_grad:
            ops.conv2d_wgrad(x, dy, grad_of(m.weight), 1, dy_coff=off, cout=co, precision=BT.WGRAD_PRECISION)
            ops.bias_grad_(grad_of(m.bias), dy, c=co, coff=off)
        if need_dx:
            part = ops.conv2d(dy, m.weight.detach().reshape(co, m.in_channels), 1, in_coff=off, cin=co)
            dx = part if dx is None else ops.axpy_(dx, part)
        off += co
    return dx


# ---------------------------------------------------------------------------------------------------------------------
# PseudoLabeler (wss/modules.py:322-343)
# ---------------------------------------------------------------------------------------------------------------------
def pseudolabeler_forward(pl, x, group=None):
    a1, r1 = conv_abn_train(x, pl.conv1, pl.norm1, group, emit_planes=True)
    a2, r2 = conv_abn_train(a1, pl.conv2, pl.norm2, group)
    from .nn import conv_bias
    return conv_bias(a2, pl.cls), (r1, r2, a2)


@ops.gradient_scope
def pseudolabeler_backward(pl, tape, dy, grad_of, need_dx=True, group=None):
    r1, r2, a2 = tape
    ops.conv2d_wgrad(a2, dy, grad_of(pl.cls.weight), 1, precision=BT.WGRAD_PRECISION)
    ops.bias_grad_(grad_of(pl.cls.bias), dy)
    da2 = ops.conv2d(dy, pl.cls.weight.detach().reshape(pl.cls.out_channels, pl.cls.in_channels), 1)
    da1 = conv_abn_backward(r2, pl.conv2, pl.norm2, da2, grad_of, group=group)
    return conv_abn_backward(r1, pl.conv1, pl.norm1, da1, grad_of, need_dx=need_dx, group=group)
