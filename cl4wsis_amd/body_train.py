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
from .nn import CACHE, gamma_grad, gamma_of, packed_weight

# Arithmetic of the convolutions of the step-0 / phase-1 training paths (forward, input and weight gradients): 2 (default) = bf16x6, three bf16 planes per operand and six products — fp32-level
# accuracy (the gradient parity bar of the tests) at about twice the speed of the fp32 MFMA kernels; 0 = exact fp32 MFMA;
# 1 = bf16x3 (~4e-6 relative per layer — still far tighter than the fp16 autocast the reference trains under on GPUs,
# train.py:252 — fastest, outside the strict gradient bar).
PRECISION = int(os.environ.get("CL4_TRAIN_PRECISION", "2"))
# Weight gradients: a dW rounding error stays in that tensor (it does not propagate through BN / ReLU masks like a forward or
# input-gradient error does), so the 4e-6 of bf16x3 is harmless there and twice as fast as either fp32-level kernel — the same
# choice phase 2 makes (ops.WGRAD_PRECISION). 0 keeps exact fp32 products with PRECISION == 0.
WGRAD_PRECISION = 0 if PRECISION == 0 else 1


class _Rec:
    __slots__ = ("x", "z", "mean", "var", "count", "y")


def _dgrad_weight(conv):
    return CACHE.get(conv, ("dgrad", PRECISION), (conv.weight,), lambda: ops.dgrad_weight(conv.weight.detach(), PRECISION))


def conv_abn_train(x, conv, bn, group=None):
    """conv -> ABN (batch statistics) -> activation. Returns (y, record)."""
    k = conv.kernel_size[0]
    r = _Rec()
    r.x = x
    r.z = ops.conv2d(x, packed_weight(conv, PRECISION), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0])
    act, slope = bn.act_code()
    r.y, r.mean, r.var, r.count = ops.abn_train_forward(r.z, gamma_of(bn), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                        bn.momentum, bn.eps, act, slope, group)
    CACHE.touch(bn.running_mean)
    CACHE.touch(bn.running_var)
    return r.y, r


def conv_abn_backward(r, conv, bn, dy, grad_of, need_dx=True, group=None):
    act, slope = bn.act_code()
    gw, fin = gamma_grad(bn, grad_of)
    dz = ops.abn_train_backward(r.z, dy, r.mean, r.var, gamma_of(bn), bn.bias.detach(), gw, grad_of(bn.bias),
                                r.count, bn.eps, act, slope, group)
    if fin is not None:
        fin()
    k = conv.kernel_size[0]
    ops.conv2d_wgrad(r.x, dz, grad_of(conv.weight), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], precision=WGRAD_PRECISION)
    if not need_dx:
        return None
    return ops.conv2d_dgrad(dz, _dgrad_weight(conv), tuple(r.x.shape[2:]), k, stride=conv.stride[0], pad=conv.padding[0],
                            dil=conv.dilation[0])


def block_forward(blk, x, group=None):
    """ResidualBlock.forward in training mode — modules/residual.py:84-97."""
    c = blk.convs
    a1, r1 = conv_abn_train(x, c.conv1, c.bn1, group)
    a2, r2 = conv_abn_train(a1, c.conv2, c.bn2, group)
    b3, r3 = conv_abn_train(a2, c.conv3, c.bn3, group)
    rp = None
    sc = x
    if hasattr(blk, "proj_conv"):
        sc, rp = conv_abn_train(x, blk.proj_conv, blk.proj_bn, group)
    act, slope = c.bn1.act_code()
    out = ops.add_act(b3, sc, act, slope)
    return out, (r1, r2, r3, rp, out)


def block_backward(blk, rec, dout, grad_of, group=None):
    r1, r2, r3, rp, out = rec
    c = blk.convs
    act, slope = c.bn1.act_code()
    dsum = ops.act_bwd(out, dout, act, slope)
    d = conv_abn_backward(r3, c.conv3, c.bn3, dsum, grad_of, group=group)
    d = conv_abn_backward(r2, c.conv2, c.bn2, d, grad_of, group=group)
    dx = conv_abn_backward(r1, c.conv1, c.bn1, d, grad_of, group=group)
    if rp is not None:
        return ops.axpy_(dx, conv_abn_backward(rp, blk.proj_conv, blk.proj_bn, dsum, grad_of, group=group))
    return ops.axpy_(dx, dsum)


def body_forward(body, x, group=None):
    """ResNet.forward (keep_outputs) with batch statistics. Returns (list of the 5 module outputs, tape)."""
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


def body_backward(body, tape, douts, grad_of, group=None):
    """douts: {index of a module output (0..4): gradient or None}. Accumulates the parameter gradients; the input image
    needs no gradient (the stem's input gradient is skipped)."""
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
