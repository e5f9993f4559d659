"""GPU parity of the unified-wave bf16x3 forward kernel (csrc/conv_u.hip: 1x1 / 3x3 stride-1 "same" convolutions with
cin % 32 == 0 on 128x128 plans) against a float64 CPU convolution, on the shapes that exercise its edges: pixel counts that are
not multiples of 4 or 128, output-channel counts that are not multiples of 128, dilations that push most taps out of the image,
odd K-tile counts, channel slices of input and output, every operand of the fused epilogue, K-split slabs and main + split-tail
plans. Tolerance: the bf16x3 arithmetic's 5e-5 relative to the output scale (north-star 1e-3). Every case also asserts that the
launch really went to the unified-wave kernel (cl4_conv_debug_u_launches)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 5e-5


def _rel_err(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _u_launches():
    from cl4wsis_amd import _hip
    return _hip.load().cl4_conv_debug_u_launches()


def _run(gpu, n, cin, h, w, cout, k, dil, plan, seed, epilogue=False, in_ctot=None, in_coff=0, out_ctot=None, out_coff=0):
    from cl4wsis_amd import _hip, ops
    g = torch.Generator().manual_seed(seed)
    in_ctot = in_ctot or cin
    out_ctot = out_ctot or cout
    x = torch.randn(n, in_ctot, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pad = dil * (k // 2)
    ref = F.conv2d(x[:, in_coff:in_coff + cin].double(), wt.double(), padding=pad, dilation=dil).float()
    kw = {}
    if epilogue:
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        pre, post = torch.randn(n, cout, h, w, generator=g), torch.randn(n, cout, h, w, generator=g)
        ref = F.relu(F.leaky_relu((ref + pre) * scale[None, :, None, None] + shift[None, :, None, None], 0.01) + post)
        kw = dict(scale=scale.to(gpu), shift=shift.to(gpu), pre=pre.to(gpu), post=post.to(gpu), act1=ops.ACT_LEAKY, slope1=0.01,
                  act2=ops.ACT_RELU)
    xd = x.to(gpu)
    wp = ops.pack_conv_weight(wt.to(gpu), 1)
    out = torch.full((n, out_ctot, h, w), 7.0, device=gpu)
    d = ops._conv_desc(xd, cin, in_coff, cout, k, 1, pad, dil, out_ctot, out_coff, kw.get("act1", 0), kw.get("slope1", 0.0),
                       kw.get("act2", 0), 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    before = _u_launches()
    try:
        ops.conv2d(xd, wp, k, pad=pad, dil=dil, in_coff=in_coff, cin=cin, out=out, out_coff=out_coff, **kw)
        torch.cuda.synchronize()
    finally:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    assert _u_launches() > before, "the launch did not go to the unified-wave kernel"
    y = out[:, out_coff:out_coff + cout].cpu()
    assert _rel_err(y, ref) < TOL, (_rel_err(y, ref), (n, cin, h, w, cout, k, dil, plan))
    if out_ctot != cout:
        assert (out[:, :out_coff] == 7).all() and (out[:, out_coff + cout:] == 7).all()      # neighbouring channels untouched


# (n, cin, h, w, cout, k, dil)
SHAPES = [
    (2, 64, 17, 17, 256, 1, 1),       # 289 pixels per image: hw % 4 = 1
    (3, 32, 10, 10, 128, 1, 1),       # one K-tile (odd count), hw % 4 = 0
    (1, 96, 33, 33, 160, 3, 6),       # ragged cout, 27 K-tiles (odd), dilation 6
    (2, 64, 33, 33, 136, 3, 18),      # dilation 18: most taps outside the image
    (2, 160, 21, 19, 192, 3, 2),      # rows of 19: 4-pixel groups straddle rows
    (1, 128, 9, 7, 144, 3, 1),        # 63 pixels: less than one tile, cout just above one M tile
    (5, 64, 65, 67, 256, 3, 1),       # 342 pixel tiles x 2: more tiles than resident workgroups
    (2, 256, 33, 33, 1024, 1, 1),     # res4 conv3 at small batch
]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv_u_plain(gpu, shape):
    _run(gpu, *shape, plan=(2, 128, 1), seed=sum(shape))


@pytest.mark.parametrize("shape", SHAPES)
def test_conv_u_fused_epilogue(gpu, shape):
    _run(gpu, *shape, plan=(2, 128, 1), seed=1 + sum(shape), epilogue=True)


@pytest.mark.parametrize("plan", [(2, 128, 2), (2, 128, 5), (2, 128, -8), (2, 128, -1)])
@pytest.mark.parametrize("shape", [(5, 64, 65, 67, 256, 3, 1), (2, 160, 21, 19, 192, 3, 2), (4, 256, 33, 33, 256, 1, 1)])
def test_conv_u_split_plans(gpu, shape, plan):
    """K-split slabs (plan splits > 1) and main + split-tail / small-tile-tail plans (splits < 0) through the same kernel."""
    _run(gpu, *shape, plan=plan, seed=2 + sum(shape), epilogue=True)


def test_conv_u_channel_slices(gpu):
    _run(gpu, 2, 64, 15, 15, 128, 3, 2, plan=(2, 128, 1), seed=5, in_ctot=100, in_coff=20, out_ctot=200, out_coff=24)
    _run(gpu, 2, 32, 15, 15, 160, 1, 1, plan=(2, 128, 1), seed=6, in_ctot=40, in_coff=8, out_ctot=161, out_coff=1, epilogue=True)


def test_conv_u_input_at_the_end_of_an_allocation(gpu):
    """The 3x3 taps of the last rows of the last channel of the last image point past the tensor: every address goes through the
    buffer descriptor's bounds check, so a tensor that ends exactly at the end of its allocation is read correctly (and nothing
    beyond it is touched)."""
    from cl4wsis_amd import _hip, ops
    g = torch.Generator().manual_seed(9)
    n, cin, h, w, cout = 1, 32, 16, 16, 128                 # 32 KB of input: carve it out of the END of a 2 MiB block
    block = torch.empty(2 * 1024 * 1024 // 4, device=gpu)
    x = block[-n * cin * h * w:].view(n, cin, h, w)
    x.copy_(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    ref = F.conv2d(x.cpu().double(), wt.double(), padding=4, dilation=4).float()
    wp = ops.pack_conv_weight(wt.to(gpu), 1)
    d = ops._conv_desc(x, cin, 0, cout, 3, 1, 4, 4, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    before = _u_launches()
    try:
        y = ops.conv2d(x, wp, 3, pad=4, dil=4)
        torch.cuda.synchronize()
    finally:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    assert _u_launches() > before
    assert _rel_err(y.cpu(), ref) < TOL
