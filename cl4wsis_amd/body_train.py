"""Train-mode forward and hand-written backward of the ResNet body (SURVEY.md §8 A1 + A2 with gradients: step 0 and
phase 1 train the body; phase 2 freezes it and never comes here).

Forward: conv (exact fp32 MFMA) -> InPlaceABNSync with batch statistics over all ranks -> leaky-ReLU / identity, residual
add + in-place leaky-ReLU (modules/residual.py:84-97), stem + max-pool (models/resnet.py:59-63,106-123). Every conv->ABN
group records (input, conv output, batch mean / var, global count, output) on a tape; `backward` walks it in reverse:
ABN backward (cross-rank sums), weight gradient, input gradient as a forward convolution with the transposed / flipped
weight (strided convs: zero-inserted output gradient). Gradients go to the buffers returned by `grad_of(param)`.
"""
import torch

import os

from . import ops
from .nn import CACHE, PackBatch, gamma_grad, gamma_of, packed_weight

# Arithmetic of the convolutions of the step-0 / phase-1 training paths (forward and input gradients): 3 (default, r05) = fp16x3 — split-fp16
# operands with power-of-two scales, THREE MFMA products per MAC at fp32-level accuracy (FINDINGS.md finding 47) wherever the LDS-DMA kernel
# takes the shape, bf16x6 elsewhere; 2 = bf16x6 everywhere (three bf16 planes per operand, six products: the r01-r04 default); 0 = exact fp32
# MFMA; 1 = bf16x3 (~4e-6 relative per layer, fastest — measured in r05 to put the phase-1 gradients 4.5x further from float64 than the CPU
# fp32 oracle, bar 1.5x: tests/test_acceptance_gpu.py).
PRECISION = int(os.environ.get("CL4_TRAIN_PRECISION", "3"))
# Weight gradients: a dW rounding error stays in that tensor (it does not propagate through BN / ReLU masks like a forward or
# input-gradient error does), so the 4e-6 of bf16x3 is harmless there and twice as fast as either fp32-level kernel — the same
# choice phase 2 makes (ops.WGRAD_PRECISION). 0 keeps exact fp32 products with PRECISION == 0.
WGRAD_PRECISION = 0 if PRECISION == 0 else 1


_wrn_calls = 0
WRN_KEEPS = None      # tests: {block name: uint8 [n, c]} Dropout2d masks used instead of drawn ones (the reference's draws cannot be reproduced)


class _Rec:
    __slots__ = ("x", "z", "mean", "var", "count", "y")


def _dgrad_weight(conv, force_f16=False):
    key = ("dgrad", PRECISION, bool(force_f16))
    wd = CACHE.get(conv, key, (conv.weight,), lambda: ops.dgrad_weight(conv.weight.detach(), PRECISION, force_f16=force_f16))
    if PackBatch.ACTIVE is not None and PRECISION == 3:
        PackBatch.ACTIVE.register(conv, key, True, wd)
    return wd


def _grad_twin_pays(conv):
    """Does the ABN backward behind `conv` hand its gradient on as fp16 planes (r05, ops.abn_train_backward(want_planes=True))? When the
    input-gradient convolution of `conv` can read them: fp16x3 arithmetic, stride 1 (a strided layer's gradient is zero-inserted first), a
    shape the LDS-DMA kernel takes — channel-shrinking 1x1 gradients included, since no conversion pass is left to outweigh the gain."""
    return (PRECISION == 3 and ops.GRAD_PLANES and ops.planes_f16() and conv.stride[0] == 1
            and ops.f16x3_ok(conv.out_channels, conv.in_channels, 3))


FUSE_BLOCK_TAIL = os.environ.get("CL4_FUSE_BLOCK_TAIL", "1") != "0"      # r05: bn3 + add + activation in one pass; shortcut gradient in the dgrad epilogue


def _twin_pays(c):
    """Does an activation of c channels get a planes twin from its producer? fp16x3 arithmetic, a planes-kernel shape downstream."""
    return PRECISION == 3 and ops.PRODUCER_PLANES and ops.planes_f16() and c % 32 == 0 and c >= 128


def conv_abn_train(x, conv, bn, group=None, xs_cache=None, emit_planes=False, post=None, act2=ops.ACT_NONE, slope2=0.01):
    """conv -> ABN (batch statistics) -> activation. Returns (y, record). `emit_planes`: the ABN apply also writes the planes twin of y
    (the next convolution is an fp16x3 launch: no conversion pass)."""
    k = conv.kernel_size[0]
    r = _Rec()
    r.x = x
    has_twin = getattr(x, "_cl4_planes", None) is not None
    st = []           # r05: (partial sums, nparts) of z from the convolution's own epilogue, when its launch plan can leave them
    r.z = ops.conv2d(x, packed_weight(conv, PRECISION, force_f16=PRECISION == 3 and has_twin), k, stride=conv.stride[0], pad=conv.padding[0],
                     dil=conv.dilation[0], x6=True, xs_cache=xs_cache, stats_out=st)
    act, slope = bn.act_code()
    r.y, r.mean, r.var, r.count = ops.abn_train_forward(r.z, gamma_of(bn), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                        bn.momentum, bn.eps, act, slope, group,
                                                        want_planes=emit_planes and _twin_pays(conv.out_channels), post=post, act2=act2, slope2=slope2,
                                                        partials=st[0] if st else None)
    CACHE.touch(bn.running_mean)
    CACHE.touch(bn.running_var)
    return r.y, r


def conv_abn_backward(r, conv, bn, dy, grad_of, need_dx=True, group=None, post=None):
    act, slope = bn.act_code()
    gw, gsign = gamma_grad(bn, grad_of)
    twin = need_dx and _grad_twin_pays(conv)
    dz = ops.abn_train_backward(r.z, dy, r.mean, r.var, gamma_of(bn), bn.bias.detach(), gw, grad_of(bn.bias),
                                r.count, bn.eps, act, slope, group, want_planes=twin, gamma_sign=gsign)
    k = conv.kernel_size[0]
    ops.conv2d_wgrad(r.x, dz, grad_of(conv.weight), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], precision=WGRAD_PRECISION)
    if not need_dx:
        return None
    return ops.conv2d_dgrad(dz, _dgrad_weight(conv, force_f16=twin and getattr(dz, "_cl4_planes", None) is not None), tuple(r.x.shape[2:]), k,
                            stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], x6=True, post=post)


def block_forward(blk, x, group=None):
    """ResidualBlock.forward in training mode — modules/residual.py:84-97."""
    c = blk.convs
    shared = {} if hasattr(blk, "proj_conv") else None      # fp16x3: conv1 and the projection read one converted copy of x (same stride only)
    a1, r1 = conv_abn_train(x, c.conv1, c.bn1, group, xs_cache=shared, emit_planes=True)      # -> conv2 (3x3)
    a2, r2 = conv_abn_train(a1, c.conv2, c.bn2, group, emit_planes=True)                      # -> conv3 (1x1, expanding)
    rp = None
    sc = x
    if hasattr(blk, "proj_conv"):
        sc, rp = conv_abn_train(x, blk.proj_conv, blk.proj_bn, group, xs_cache=shared)
    act, slope = c.bn1.act_code()
    cout = c.conv3.out_channels
    if FUSE_BLOCK_TAIL and cout % 8 == 0 and sc.is_contiguous():
        # r05: bn3 (identity activation) + shortcut add + the block's activation in ONE pass — the bn3 output is never written — which also
        # leaves the planes twin for the next block's conv1 / projection and the heads
        out, r3 = conv_abn_train(a2, c.conv3, c.bn3, group, emit_planes=True, post=sc, act2=act, slope2=slope)
    else:
        b3, r3 = conv_abn_train(a2, c.conv3, c.bn3, group)
        out = ops.add_act(b3, sc, act, slope, want_planes=_twin_pays(cout))
    return out, (r1, r2, r3, rp, out)


def block_backward(blk, rec, dout, grad_of, group=None):
    r1, r2, r3, rp, out = rec
    c = blk.convs
    act, slope = c.bn1.act_code()
    dsum = ops.act_bwd(out, dout, act, slope)
    d = conv_abn_backward(r3, c.conv3, c.bn3, dsum, grad_of, group=group)
    d = conv_abn_backward(r2, c.conv2, c.bn2, d, grad_of, group=group)
    if FUSE_BLOCK_TAIL:      # r05: the shortcut's gradient rides in the epilogue of the last input-gradient convolution (no axpy pass)
        if rp is None:
            return conv_abn_backward(r1, c.conv1, c.bn1, d, grad_of, group=group, post=dsum)
        dx = conv_abn_backward(r1, c.conv1, c.bn1, d, grad_of, group=group)
        return conv_abn_backward(rp, blk.proj_conv, blk.proj_bn, dsum, grad_of, group=group, post=dx)
    dx = conv_abn_backward(r1, c.conv1, c.bn1, d, grad_of, group=group)
    if rp is not None:
        return ops.axpy_(dx, conv_abn_backward(rp, blk.proj_conv, blk.proj_bn, dsum, grad_of, group=group))
    return ops.axpy_(dx, dsum)


def body_forward(body, x, group=None, seed=None):
    """Train-mode forward of the body with batch statistics: ResNet (models/resnet.py) or WiderResNet-38 A2 (models/wide_resnet.py;
    `seed` feeds its Dropout2d masks, default: a per-call counter). Returns (list of the 5 outputs, tape)."""
    if hasattr(body, "bn_out"):
        global _wrn_calls
        _wrn_calls += 1
        return wrn_body_forward(body, x, group, keeps=WRN_KEEPS, seed=_wrn_calls * 0x9E3779B1 if seed is None else seed)
    y, r0 = conv_abn_train(x, body.mod1.conv1, body.mod1.bn1, group)
    outs = [ops.maxpool3x3s2(y)]
    tape = [("stem", r0)]
    for name in ("mod2", "mod3", "mod4", "mod5"):
        y = outs[-1]
        for blk in getattr(body, name):
            y, rec = block_forward(blk, y, group)
            tape.append(("block", blk, rec))
        tape.append(("out", len(outs)))
        outs.append(y)
    return outs, tape


@ops.gradient_scope
def body_backward(body, tape, douts, grad_of, group=None):
    """douts: {index of a module output (0..4): gradient or None}. Accumulates the parameter gradients; the input image
    needs no gradient (the stem's input gradient is skipped)."""
    if hasattr(body, "bn_out"):
        return wrn_body_backward(body, tape, douts, grad_of, group)
    dy = None
    for item in reversed(tape):
        if item[0] == "out":
            g = douts.get(item[1])
            if g is not None:
                dy = g if dy is None else ops.axpy_(dy, g)
        elif item[0] == "block":
            if dy is not None:
                dy = block_backward(item[1], item[2], dy, grad_of, group)
        else:  # stem: max-pool, conv 7x7 stride 2 + ABN
            g0 = douts.get(0)
            if g0 is not None:
                dy = g0 if dy is None else ops.axpy_(dy, g0)
            if dy is None:
                return
            r0 = item[1]
            dpool = ops.maxpool3x3s2_bwd(r0.y, dy)
            conv_abn_backward(r0, body.mod1.conv1, body.mod1.bn1, dpool, grad_of, need_dx=False, group=group)


# ---------------------------------------------------------------------------------------------------------------------
# WiderResNet-38 A2 in training mode (coco-voc step 0 / phase 1: models/wide_resnet.py:82-207, modules/residual.py:101-181).
# Pre-activation identity blocks: bn1 (ABN with batch statistics) on the block input, convolutions WITHOUT a following ABN at the
# end of the residual branch and on the projection shortcut, Dropout2d(0.3 / 0.5) in mod6 / mod7, max-pooling in front of mod2 and
# mod3, a final ABN (bn_out). The same tape discipline as the ResNet body above.
# ---------------------------------------------------------------------------------------------------------------------
def _abn_train(x, bn, group=None):
    """Stand-alone train-mode ABN on x (the pre-activation bn1 / bn_out). Returns (y, record)."""
    r = _Rec()
    r.x = r.z = x
    act, slope = bn.act_code()
    r.y, r.mean, r.var, r.count = ops.abn_train_forward(x, gamma_of(bn), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                        bn.momentum, bn.eps, act, slope, group)
    CACHE.touch(bn.running_mean)
    CACHE.touch(bn.running_var)
    return r.y, r


def _abn_backward(r, bn, dy, grad_of, group=None):
    act, slope = bn.act_code()
    gw, gsign = gamma_grad(bn, grad_of)
    dx = ops.abn_train_backward(r.z, dy, r.mean, r.var, gamma_of(bn), bn.bias.detach(), gw, grad_of(bn.bias), r.count, bn.eps,
                                act, slope, group, gamma_sign=gsign)
    return dx


def _conv_train(x, conv):
    k = conv.kernel_size[0]
    return ops.conv2d(x, packed_weight(conv, PRECISION), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], x6=True)


def _conv_backward(x, conv, dz, grad_of, need_dx=True):
    k = conv.kernel_size[0]
    ops.conv2d_wgrad(x, dz, grad_of(conv.weight), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], precision=WGRAD_PRECISION)
    if not need_dx:
        return None
    return ops.conv2d_dgrad(dz, _dgrad_weight(conv), tuple(x.shape[2:]), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], x6=True)


def identity_block_forward(blk, x, group=None, drop_p=None, keep=None, seed=0):
    """IdentityResidualBlock.forward in training mode. `drop_p`: the Dropout2d rate of the block's module (None: no dropout);
    `keep`: an injected uint8 [n, c] mask (tests), otherwise drawn on the device from `seed`. Returns (out, bn1 output, record)."""
    c = blk.convs
    a, ra = _abn_train(x, blk.bn1, group)
    three = hasattr(c, "conv3")
    z1 = _conv_train(a, c.conv1)
    y1, r2 = _abn_train(z1, c.bn2, group)
    r3 = z2 = y2 = None
    if three:
        z2 = _conv_train(y1, c.conv2)
        y2, r3 = _abn_train(z2, c.bn3, group)
        last_in, last = y2, c.conv3
    else:
        last_in, last = y1, c.conv2
    dropped = last_in
    if drop_p is not None:
        if keep is None:
            keep = ops.dropout_mask((last_in.shape[0], last_in.shape[1]), drop_p, seed, last_in.device)
        dropped = ops.dropout2d(last_in, keep, drop_p)
    out = _conv_train(dropped, last)
    if hasattr(blk, "proj_conv"):
        ops.axpy_(out, _conv_train(a, blk.proj_conv))
    else:
        ops.axpy_(out, x)
    return out, a, (x, ra, a, z1, r2, y1, z2, r3, y2, dropped, keep, drop_p)


def identity_block_backward(blk, rec, dout, grad_of, da_extra=None, group=None):
    """dout: gradient of the block output (None: no gradient reaches it); da_extra: gradient that reaches the block's bn1 output from
    outside (the first blocks of mod4..mod7 hand it out as a feature, models/wide_resnet.py:168-197). Returns the gradient of the
    block input (None when nothing reaches the block)."""
    x, ra, a, z1, r2, y1, z2, r3, y2, dropped, keep, drop_p = rec
    c = blk.convs
    da = None
    if dout is not None:
        three = r3 is not None
        last = c.conv3 if three else c.conv2
        d = _conv_backward(dropped, last, dout, grad_of)
        if drop_p is not None:
            d = ops.dropout2d(d, keep, drop_p)
        if three:
            d = _abn_backward(r3, c.bn3, d, grad_of, group)
            d = _conv_backward(y1, c.conv2, d, grad_of)
        d = _abn_backward(r2, c.bn2, d, grad_of, group)
        da = _conv_backward(a, c.conv1, d, grad_of)
        if hasattr(blk, "proj_conv"):
            ops.axpy_(da, _conv_backward(a, blk.proj_conv, dout, grad_of))
    if da_extra is not None:
        da = da_extra.clone() if da is None else ops.axpy_(da, da_extra)
    if da is None:
        return None
    dx = _abn_backward(ra, blk.bn1, da, grad_of, group)
    if dout is not None and not hasattr(blk, "proj_conv"):
        ops.axpy_(dx, dout)
    return dx


_WRN_DROP = {"mod6": 0.3, "mod7": 0.5}


def wrn_body_forward(body, x, group=None, keeps=None, seed=0):
    """WiderResNetA2.forward (keep_outputs) with batch statistics. Returns ([out3, out4, out5, out6, out7], tape). `keeps`: optional
    {block name: uint8 [n, c]} Dropout2d masks (tests); otherwise they are drawn on the device from `seed`."""
    out = _conv_train(x, body.mod1.conv1)
    tape = [("stem", x)]
    prevs = []
    for name in ("mod2", "mod3", "mod4", "mod5", "mod6", "mod7"):
        if name in ("mod2", "mod3"):
            tape.append(("pool", out))
            out = ops.maxpool3x3s2(out)
        for i, blk in enumerate(getattr(body, name)):
            key = f"{name}.block{i + 1}"
            out, a, rec = identity_block_forward(blk, out, group, _WRN_DROP.get(name), None if keeps is None else keeps.get(key),
                                                 seed + 7919 * len(tape))
            hand_out = i == 0 and name in ("mod4", "mod5", "mod6", "mod7")
            if hand_out:
                prevs.append(a)
            tape.append(("block", blk, rec, len(prevs) - 1 if hand_out else None))
    out7, r7 = _abn_train(out, body.bn_out, group)
    tape.append(("bn_out", r7))
    return prevs + [out7], tape


@ops.gradient_scope
def wrn_body_backward(body, tape, douts, grad_of, group=None):
    """douts: {index into [out3, out4, out5, out6, out7]: gradient or None}. The input image needs no gradient."""
    dy = None
    for item in reversed(tape):
        if item[0] == "bn_out":
            g = douts.get(4)
            if g is None:
                continue
            dy = _abn_backward(item[1], body.bn_out, g, grad_of, group)
        elif item[0] == "block":
            _, blk, rec, prev_idx = item
            extra = douts.get(prev_idx) if prev_idx is not None else None
            dy = identity_block_backward(blk, rec, dy, grad_of, extra, group)
        elif item[0] == "pool":
            if dy is not None:
                dy = ops.maxpool3x3s2_bwd(item[1], dy)
        else:  # stem conv: weight gradient only
            if dy is not None:
                _conv_backward(item[1], body.mod1.conv1, dy, grad_of, need_dx=False)
