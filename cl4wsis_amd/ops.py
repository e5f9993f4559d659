"""Tensor-level wrappers over the C ABI (include/cl4wsis.h).

Each function checks shapes/dtypes on the host, allocates outputs from PyTorch's caching
allocator (PyTorch is only the memory/stream plumbing here) and enqueues the HIP kernel on
the current stream. Nothing in this module computes with torch ops.
"""
import ctypes as C

import torch

from . import _hip
from ._hip import ACT_LEAKY, ACT_NONE, ACT_RELU, ConvDesc, call, ptr, stream

__all__ = ["ACT_NONE", "ACT_LEAKY", "ACT_RELU"]


def _f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def pack_conv_weight(w):
    """[cout, cin, kh, kw] -> [cin*kh*kw, cout] (the layout cl4_conv2d_fwd streams)."""
    _f32(w, "pack_conv_weight")
    cout, cin, kh, kw = w.shape
    wp = torch.empty((cin * kh * kw, cout), dtype=torch.float32, device=w.device)
    call("cl4_pack_conv_weight", ptr(w.contiguous()), ptr(wp), cout, cin, kh, kw, stream())
    return wp


def bn_fold(gamma, beta, mean, var, eps=1e-5):
    c = mean.numel()
    scale = torch.empty(c, dtype=torch.float32, device=mean.device)
    shift = torch.empty_like(scale)
    call("cl4_bn_fold", ptr(gamma), ptr(beta), ptr(mean), ptr(var), C.c_float(eps), ptr(scale), ptr(shift), c, stream())
    return scale, shift


def _conv_desc(x, cin, in_coff, cout, k, stride, pad, dil, out_ctot, out_coff, act1, slope1, act2, slope2):
    n, in_ctot, hin, win = x.shape
    d = ConvDesc()
    d.n, d.cin, d.hin, d.win = n, cin, hin, win
    d.cout, d.kh, d.kw = cout, k, k
    d.stride, d.pad, d.dil = stride, pad, dil
    d.hout = conv_out_size(hin, k, stride, pad, dil)
    d.wout = conv_out_size(win, k, stride, pad, dil)
    d.in_ctot, d.in_coff = in_ctot, in_coff
    d.out_ctot, d.out_coff = out_ctot, out_coff
    d.act1, d.slope1, d.act2, d.slope2 = act1, slope1, act2, slope2
    d.precision = 0
    return d


def conv2d(x, wp, k, stride=1, pad=0, dil=1, scale=None, shift=None, pre=None, post=None,
           act1=ACT_NONE, slope1=0.01, act2=ACT_NONE, slope2=0.01,
           out=None, out_coff=0, in_coff=0, cin=None):
    """Fused conv (+affine +act +residual +act). `wp` comes from pack_conv_weight.

    `out`/`out_coff` write into a channel slice of a wider tensor (replaces torch.cat);
    `in_coff`/`cin` read a channel slice of `x`.
    """
    _f32(x, "conv2d.x")
    n, in_ctot, hin, win = x.shape
    if cin is None:
        cin = in_ctot - in_coff
    K, cout = wp.shape
    if K != cin * k * k:
        raise RuntimeError(f"conv2d: packed weight K={K} does not match cin*k*k={cin * k * k}")
    hout = conv_out_size(hin, k, stride, pad, dil)
    wout = conv_out_size(win, k, stride, pad, dil)
    if out is None:
        out = torch.empty((n, cout, hout, wout), dtype=torch.float32, device=x.device)
        out_coff = 0
    elif tuple(out.shape[2:]) != (hout, wout) or out.shape[0] != n:
        raise RuntimeError(f"conv2d: out shape {tuple(out.shape)} incompatible with {(n, cout, hout, wout)}")
    for t, nm in ((pre, "pre"), (post, "post")):
        if t is not None and tuple(t.shape) != (n, cout, hout, wout):
            raise RuntimeError(f"conv2d: {nm} shape {tuple(t.shape)} != {(n, cout, hout, wout)}")
    for t, nm in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != cout:
            raise RuntimeError(f"conv2d: {nm} has {t.numel()} elements, expected {cout}")
    d = _conv_desc(x, cin, in_coff, cout, k, stride, pad, dil, out.shape[1], out_coff, act1, slope1, act2, slope2)
    call("cl4_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(pre), ptr(post), ptr(out), stream())
    return out


def conv2d_wgrad(x, dy, dw, k, stride=1, pad=0, dil=1, in_coff=0, cin=None, dy_coff=0, cout=None):
    """dw[cout,cin,k,k] += conv weight gradient. dw is accumulated into (zero it first)."""
    n, in_ctot, hin, win = x.shape
    if cin is None:
        cin = in_ctot - in_coff
    if cout is None:
        cout = dy.shape[1] - dy_coff
    if tuple(dw.shape) != (cout, cin, k, k):
        raise RuntimeError(f"conv2d_wgrad: dw shape {tuple(dw.shape)} != {(cout, cin, k, k)}")
    d = _conv_desc(x, cin, in_coff, cout, k, stride, pad, dil, dy.shape[1], dy_coff, 0, 0.0, 0, 0.0)
    if tuple(dy.shape[2:]) != (d.hout, d.wout) or dy.shape[0] != n:
        raise RuntimeError("conv2d_wgrad: dy spatial shape does not match the convolution geometry")
    call("cl4_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), stream())
    return dw
