"""GPU parity of the planes format and of the planes-input x3 convolution (csrc/conv_s.hip, cl4_conv2d_fwd_planes) against
float64 CPU convolutions. The planes format stores x as two 16-bit halves (csrc/planes_fmt.h): fp16 hi + lo of 16 x in the default
fp16x3 build (22 significant bits, r05), bf16 hi + lo of x in the bf16x3 build (16-17 bits) — the tolerance is the looser arithmetic's
5e-5 relative to the output scale (north-star 1e-3); the fp16x3 build is measured at < 2e-6 in tests/test_f16x3_gpu.py.
to_planes / from_planes themselves are checked BIT-EXACTLY against a torch restatement of the split (tests/util.py)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 5e-5


def _rel_err(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _split_ref(x):
    from util import planes_split_ref
    return planes_split_ref(x)


def test_planes_round_trip_is_the_bf16_split(gpu):
    """to_planes / from_planes against a torch restatement of the build's two-plane split (tests/util.py::planes_split_ref), bit for
    bit, over the format's range: fp16x3 (the default build) holds 16 x in two fp16 halves — 22 significant bits for 2^-7 <= |x| < 4094,
    graceful below (the lo half goes subnormal: absolute error <= 2^-29), saturating above; the bf16x3 build holds any fp32 exponent to 16-17 bits."""
    from cl4wsis_amd import ops
    from util import planes_raw_ref
    f16 = ops.planes_f16()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 40, 9, 7, generator=g) * (torch.logspace(-4, 2, 40) if f16 else torch.logspace(-6, 6, 40))[None, :, None, None]
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.5])
    if f16:
        x[1, 8, 0, :3] = torch.tensor([5000.0, -1e9, 1e-9])          # beyond the range: saturates at +-65504 / 16 + its lo half; below it: flushed
    ps = ops.to_planes(x.to(gpu), in_coff=8, c=24)
    hi, lo = _split_ref(x[:, 8:32])
    back = ops.from_planes(ps).cpu()
    assert torch.equal(back, hi + lo)
    raw = ps.data.cpu().view(torch.int16).view(3, 3, 2, 63, 8)          # [n][c/8][plane][hw][8]
    rh, rl = planes_raw_ref(x[:, 8:32])
    want_hi = rh.view(3, 3, 8, 63).permute(0, 1, 3, 2).contiguous()
    want_lo = rl.view(3, 3, 8, 63).permute(0, 1, 3, 2).contiguous()
    assert torch.equal(raw[:, :, 0], want_hi) and torch.equal(raw[:, :, 1], want_lo)
    out = torch.full((3, 40, 9, 7), 5.0, device=gpu)
    ops.from_planes(ps, out=out, out_coff=16)
    assert torch.equal(out[:, 16:40].cpu(), hi + lo) and (out[:, :16] == 5).all()
    xr = x[:, 8:32]
    if f16:
        ok = (xr.abs() < 4000) & (xr.abs() >= 2.0 ** -7)       # both halves normal fp16 numbers: 22 significant bits
        assert ((back - xr).abs()[ok] <= xr.abs()[ok] * 2.0 ** -21).all() and ((back - xr).abs()[xr.abs() < 2.0 ** -7] <= 2.0 ** -28).all()
        assert back[1, 0, 0, 0] == 4094.0 and back[1, 0, 0, 1] == -4094.0 and back[1, 0, 0, 2] == 0     # saturation at +-65504 / 16, flush below 2^-24 / 16
    else:
        assert ((back - xr).abs() <= xr.abs() * 2.0 ** -16).all()


def _run(gpu, n, cin, h, w, cout, k, stride, pad, dil, plan, seed, epilogue=False, planes_post=False, in_ctot=None, in_coff=0,
         out_ctot=None, out_coff=0, counter="cl4_conv_debug_s_launches", keep=None):
    from cl4wsis_amd import _hip, ops
    g = torch.Generator().manual_seed(seed)
    in_ctot = in_ctot or cin
    out_ctot = out_ctot or cout
    x = torch.randn(n, in_ctot, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x[:, in_coff:in_coff + cin].double(), wt.double(), stride=stride, padding=pad, dilation=dil).float()
    ho, wo = ref.shape[2:]
    kw = {}
    if epilogue:
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        pre, post = torch.randn(n, cout, ho, wo, generator=g), torch.randn(n, cout, ho, wo, generator=g)
        ref = F.relu(F.leaky_relu((ref + pre) * scale[None, :, None, None] + shift[None, :, None, None], 0.01) + post)
        postd = ops.to_planes(post.to(gpu)) if planes_post else post.to(gpu)
        kw = dict(scale=scale.to(gpu), shift=shift.to(gpu), pre=pre.to(gpu), post=postd, act1=ops.ACT_LEAKY, slope1=0.01, act2=ops.ACT_RELU)
    xs = ops.to_planes(x.to(gpu))
    wp = ops.pack_conv_weight(wt.to(gpu), 1)
    out = torch.full((n, out_ctot, ho, wo), 7.0, device=gpu)
    outp = ops.Planes(n, out_ctot, ho, wo, gpu) if out_ctot % 8 == 0 and out_coff % 8 == 0 and cout % 8 == 0 else None
    if outp is not None:
        outp.data.fill_(0)
    d = ops.ConvDescFor(n, cin, h, w, in_ctot, in_coff, cout, k, stride, pad, dil, out_ctot, out_coff, 0, 0.0, 0, 0.0, 3)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    before = getattr(_hip.load(), counter)()
    try:
        ops.conv2d_planes(xs, wp, k, stride=stride, pad=pad, dil=dil, in_coff=in_coff, cin=cin, out=out, out_coff=out_coff,
                          out_planes=outp, **kw)
        torch.cuda.synchronize()
    finally:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    assert getattr(_hip.load(), counter)() > before
    y = out[:, out_coff:out_coff + cout].cpu()
    if keep is not None:
        keep.append(out.cpu())
    assert _rel_err(y, ref) < TOL, (_rel_err(y, ref), (n, cin, h, w, cout, k, stride, pad, dil, plan))
    if out_ctot != cout:
        assert (out[:, :out_coff] == 7).all() and (out[:, out_coff + cout:] == 7).all()
    if outp is not None:                                   # the planes output is the bf16 split of the fp32 output, bit for bit
        hi, lo = _split_ref(y)
        yp = ops.from_planes(outp).cpu()
        assert torch.equal(yp[:, out_coff:out_coff + cout], hi + lo)
        if out_ctot != cout:
            assert (yp[:, :out_coff] == 0).all() and (yp[:, out_coff + cout:] == 0).all()


# (n, cin, h, w, cout, k, stride, pad, dil)
SHAPES = [
    (2, 64, 17, 17, 256, 1, 1, 0, 1),
    (3, 32, 10, 10, 128, 1, 1, 0, 1),       # one K-tile
    (1, 96, 33, 33, 160, 3, 1, 6, 6),       # ragged cout, odd K-tile count, dilation 6
    (2, 64, 33, 33, 136, 3, 1, 18, 18),     # most taps outside the image
    (2, 160, 21, 19, 192, 3, 1, 2, 2),
    (2, 128, 33, 33, 128, 3, 2, 1, 1),      # strided 3x3 (res3 / res4 entry blocks)
    (2, 256, 33, 31, 512, 1, 2, 0, 1),      # strided 1x1 (downsample branch)
    (1, 128, 9, 7, 144, 3, 1, 1, 1),        # 63 pixels
    (5, 64, 65, 67, 256, 3, 1, 1, 1),       # more tiles than resident workgroups
    (1, 64, 12, 12, 64, 3, 1, 0, 1),        # no padding ("valid"), cout = 64 on a 128-row tile
    (20, 64, 33, 33, 1024, 1, 1, 0, 1),     # 1368 tiles = 2.7 rounds of resident workgroups
]


TILES = [64, 128, 144]  # pixels per tile (144: r03, one round of the 33 x 33 maps; two column parts of 80 + 64 pixels per wave pair. 64: r03, three workgroups per CU)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_planes_plain(gpu, shape, tile):
    _run(gpu, *shape, plan=(2, tile, 1), seed=sum(shape))


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("planes_post", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_planes_fused_epilogue(gpu, shape, planes_post, tile):
    _run(gpu, *shape, plan=(2, tile, 1), seed=1 + sum(shape), epilogue=True, planes_post=planes_post)


@pytest.mark.parametrize("plan", [(2, 128, 2), (2, 128, 5), (2, 128, -8), (2, 144, 2), (2, 144, 3), (2, 144, -8), (2, 64, 2), (2, 64, -8)])
@pytest.mark.parametrize("shape", [(5, 64, 65, 67, 256, 3, 1, 1, 1), (2, 160, 21, 19, 192, 3, 1, 2, 2), (4, 256, 33, 33, 256, 1, 1, 0, 1),
                                   (3, 96, 33, 33, 136, 3, 1, 18, 18),      # taps dropped per tile (rows outside the image): K splits divide the surviving K-tiles
                                   (2, 64, 33, 33, 128, 3, 1, 40, 40)])     # dilation > map: only the centre tap survives
def test_conv_planes_split_plans(gpu, shape, plan):
    _run(gpu, *shape, plan=plan, seed=2 + sum(shape), epilogue=True, planes_post=True)


@pytest.mark.parametrize("tile", TILES)
def test_conv_planes_channel_slices(gpu, tile):
    _run(gpu, 2, 64, 15, 15, 128, 3, 1, 2, 2, plan=(2, tile, 1), seed=5, in_ctot=104, in_coff=24, out_ctot=200, out_coff=24)
    _run(gpu, 2, 32, 15, 15, 160, 1, 1, 0, 1, plan=(2, tile, 1), seed=6, in_ctot=40, in_coff=8, out_ctot=176, out_coff=8, epilogue=True)
    _run(gpu, 2, 32, 15, 15, 132, 1, 1, 0, 1, plan=(2, tile, 1), seed=7, out_ctot=133, out_coff=1)      # fp32 output only (cout % 8 != 0)


def test_conv_planes_both_mfma_shapes_agree_with_the_fp32_input_kernel(gpu):
    """The planes kernel and cl4_conv2d_fwd's bf16x3 kernels form the same three products from the same hi / lo operands: their
    results differ only through the summation order inside the MFMA pipe (a few ulps)."""
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 256, 33, 33, generator=g).to(gpu)
    wt = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(gpu)
    wp = ops.pack_conv_weight(wt, 1)
    y0 = ops.conv2d(x, wp, 3, pad=1)
    y1, _ = ops.conv2d_planes(ops.to_planes(x), wp, 3, pad=1)
    torch.cuda.synchronize()
    # bf16x3 build: the same three products from the same operands (a few ulps of summation order); fp16x3 build (r05): the planes kernel
    # is ~10x closer to fp32 than the fp32-input bf16x3 kernel it is compared with, whose own 4e-6 is then the difference
    assert _rel_err(y1.cpu(), y0.cpu()) < (2e-5 if ops.planes_f16() else 2e-6)


# The big-tile LDS-DMA kernel (csrc/conv_b.hip, plan impl 4): 256 x 144 tiles, operands straight into LDS, three stages.
# (n, cin, h, w, cout, k, stride, pad, dil)
B_SHAPES = [
    (2, 64, 17, 17, 256, 1, 1, 0, 1),       # two K-tiles (fewer than the three stages)
    (3, 32, 10, 10, 256, 1, 1, 0, 1),       # one K-tile
    (1, 96, 33, 33, 264, 3, 1, 6, 6),       # ragged cout (a second row tile with 8 live rows), 27 K-tiles, dilation 6
    (2, 64, 33, 33, 256, 3, 1, 18, 18),     # most taps outside the image: tiles with different K-tile counts
    (2, 160, 21, 19, 320, 3, 1, 2, 2),
    (2, 128, 33, 33, 256, 3, 2, 1, 1),      # strided 3x3
    (2, 256, 33, 31, 512, 1, 2, 0, 1),      # strided 1x1
    (1, 128, 9, 7, 256, 3, 1, 1, 1),        # 63 pixels: half a tile
    (3, 64, 65, 67, 256, 3, 1, 1, 1),       # 91 tiles
    (9, 64, 33, 33, 1024, 1, 1, 0, 1),      # 272 tiles: more than one per CU
    (2, 512, 33, 33, 512, 3, 1, 2, 2),      # res5 conv2: 144 K-tiles
]


@pytest.mark.parametrize("impl", [4, 5, 6])  # 256-row tiles; 128-row tiles with four / eight waves
@pytest.mark.parametrize("shape", B_SHAPES)
def test_conv_big_tile_lds_dma_kernel_plain(gpu, shape, impl):
    _run(gpu, *shape, plan=(impl, 144, 1), seed=sum(shape), counter="cl4_conv_debug_b_launches")


@pytest.mark.parametrize("impl", [4, 6])
@pytest.mark.parametrize("planes_post", [False, True])
@pytest.mark.parametrize("shape", B_SHAPES)
def test_conv_big_tile_lds_dma_kernel_fused_epilogue(gpu, shape, planes_post, impl):
    a, b = [], []
    _run(gpu, *shape, plan=(impl, 144, 1), seed=1 + sum(shape), epilogue=True, planes_post=planes_post, counter="cl4_conv_debug_b_launches", keep=a)
    _run(gpu, *shape, plan=(2, 144, 1), seed=1 + sum(shape), epilogue=True, planes_post=planes_post, keep=b)
    assert torch.equal(a[0], b[0])          # the same products in the same order as conv_s: bit-identical


def test_conv_big_tile_kernel_channel_slices(gpu):
    _run(gpu, 2, 64, 15, 15, 256, 3, 1, 2, 2, plan=(4, 144, 1), seed=5, in_ctot=104, in_coff=24, out_ctot=328, out_coff=24, counter="cl4_conv_debug_b_launches")
    _run(gpu, 2, 32, 15, 15, 288, 1, 1, 0, 1, plan=(4, 144, 1), seed=6, in_ctot=40, in_coff=8, out_ctot=304, out_coff=8, epilogue=True,
         counter="cl4_conv_debug_b_launches")
    _run(gpu, 2, 32, 15, 15, 128, 1, 1, 0, 1, plan=(4, 144, 1), seed=7)      # cout < 256: the plan falls back to conv_s (144-pixel tiles)
    _run(gpu, 2, 64, 15, 15, 136, 3, 1, 2, 2, plan=(6, 144, 1), seed=9, in_ctot=104, in_coff=24, out_ctot=208, out_coff=24, epilogue=True, planes_post=True,
         counter="cl4_conv_debug_b_launches")
    _run(gpu, 2, 64, 15, 15, 136, 3, 1, 2, 2, plan=(5, 144, 1), seed=8, in_ctot=104, in_coff=24, out_ctot=208, out_coff=24, epilogue=True, planes_post=True,
         counter="cl4_conv_debug_b_launches")


# (n, cin, h, w, cout): 1x1 layers on the persistent pipelined kernel (csrc/conv_p.hip, plan impl 3): planes in, planes out
P_SHAPES = [
    (9, 256, 33, 33, 1024),      # res4 conv3: 8 K-tiles; 77 pixel tiles x 8 = 616 tiles = 2.4 per workgroup, ragged last pixel tile
    (3, 128, 33, 31, 512),       # 4 K-tiles (the minimum): the four epilogue chunks fill the whole next K loop
    (2, 160, 17, 19, 256),       # 5 K-tiles, fewer tiles than CUs: most workgroups own one tile (drain path only)
    (5, 512, 20, 20, 128),       # 16 K-tiles, one channel tile
    (4, 192, 33, 33, 384),       # 6 K-tiles
    (4, 224, 33, 33, 384),       # 7 K-tiles (odd count after the peeled steps)
]


@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("shape", P_SHAPES)
def test_conv_persistent_pipelined_planes_kernel(gpu, shape, residual):
    """Same results as the plain planes kernel, bit for bit on the planes output: both form the same products in the same K order
    (one K-tile after the other, two row-block halves per tile) and apply the same epilogue arithmetic."""
    from cl4wsis_amd import _hip, ops
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    post = torch.randn(n, cout, h, w, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double()) * scale[None, :, None, None] + shift[None, :, None, None], 0.01)
    if residual:
        ref = F.leaky_relu(ref + post.double(), 0.01)
    xs, wp = ops.to_planes(x.to(gpu)), ops.pack_conv_weight(wt.to(gpu), 1)
    kw = dict(scale=scale.to(gpu), shift=shift.to(gpu), act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    if residual:
        kw.update(post=ops.to_planes(post.to(gpu)), act2=ops.ACT_LEAKY, slope2=0.01)
    d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, 1, 1, 0, 1, cout, 0, 0, 0.0, 0, 0.0, 3)
    outs = {}
    for impl in (2, 3):
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), impl, 128, 1)
        try:
            outs[impl] = ops.conv2d_planes(xs, wp, 1, **kw)[1]
            torch.cuda.synchronize()
        finally:
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    y = ops.from_planes(outs[3]).cpu()
    assert _rel_err(y, ref.float()) < TOL, _rel_err(y, ref.float())
    assert torch.equal(outs[3].data, outs[2].data)


def test_conv_persistent_kernel_in_place_and_channel_slice(gpu):
    """The two ways the networks call it: output written in place over the residual (identity blocks of a frozen stage), and output
    into a channel slice of a wider planes tensor."""
    from cl4wsis_amd import _hip, ops
    g = torch.Generator().manual_seed(5)
    n, cin, h, cout = 6, 128, 33, 256
    x, post = torch.randn(n, cin, h, h, generator=g), torch.randn(n, cout, h, h, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    xs, wp = ops.to_planes(x.to(gpu)), ops.pack_conv_weight(wt.to(gpu), 1)
    d = ops.ConvDescFor(n, cin, h, h, cin, 0, cout, 1, 1, 0, 1, cout, 0, 0, 0.0, 0, 0.0, 3)
    res = {}
    for impl in (2, 3):
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), impl, 128, 1)
        try:
            ps = ops.to_planes(post.to(gpu))
            out = ops.conv2d_planes(xs, wp, 1, post=ps, act2=ops.ACT_RELU, want_fp32=False, out_planes=ps)[1]
            torch.cuda.synchronize()
            res[impl] = out.data.clone()
        finally:
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    assert torch.equal(res[2], res[3])
    ref = F.relu(F.conv2d(x.double(), wt.double()) + post.double()).float()
    assert _rel_err(ops.from_planes(ops.Planes(n, cout, h, h, gpu, data=res[3])).cpu(), ref) < TOL
    # channel slice: 256 channels into [64, 320) of a 384-channel planes tensor
    d2 = ops.ConvDescFor(n, cin, h, h, cin, 0, cout, 1, 1, 0, 1, 384, 64, 0, 0.0, 0, 0.0, 3)
    wide = {}
    for impl in (2, 3):
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d2), impl, 128, 1)
        try:
            big = ops.Planes(n, 384, h, h, gpu)
            big.data.fill_(0)
            ops.conv2d_planes(xs, wp, 1, want_fp32=False, out_planes=big, out_coff=64)
            torch.cuda.synchronize()
            wide[impl] = big.data.clone()
        finally:
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d2), 0, 64, 1)
    assert torch.equal(wide[2], wide[3])


# (n, cin, h, w, cout, k, pad, dil)
X6_SHAPES = [
    (2, 64, 33, 33, 128, 3, 6, 6),
    (3, 96, 33, 31, 256, 3, 18, 18),      # most taps outside the image; ragged last pixel tile
    (2, 128, 17, 19, 192, 1, 0, 1),       # 1x1, cout not a multiple of 128
    (5, 32, 20, 20, 64, 3, 1, 1),         # one K-tile per tap, 64 output channels on a 128-row tile
]


@pytest.mark.parametrize("plan", [(2, 144, 1), (2, 144, 2), (2, 144, -4), (6, 144, 1)])      # 6 = the LDS-DMA kernel (conv_b.hip), 128 rows x 8 waves
@pytest.mark.parametrize("shape", X6_SHAPES)
def test_conv_planes_three_plane_bf16x6(gpu, shape, plan):
    """The bf16x6 variant of the planes kernel (three planes per operand, six MFMA products; descriptor precision 4): fp32-level
    agreement with a float64 convolution (the same bar as the bf16x6 instantiation of cl4_conv2d_fwd), output into a channel slice."""
    from cl4wsis_amd import _hip, ops
    n, cin, h, w, cout, k, pad, dil = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g) * torch.logspace(-2, 2, cin)[None, :, None, None]
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), padding=pad, dilation=dil)
    x3 = ops.to_planes(x.to(gpu), planes=3)
    assert x3.planes == 3 and x3.data.numel() * 4 == n * cin * h * w * 6
    wp = ops.pack_conv_weight(wt.to(gpu), 2)
    out = torch.full((n, cout + 16, h, w), 7.0, device=gpu)
    d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, dil, cout + 16, 8, 0, 0.0, 0, 0.0, 4)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    counter = "cl4_conv_debug_b_launches" if plan[0] == 6 and cout >= 128 else "cl4_conv_debug_s_launches"     # cout < 128: conv_b falls back to conv_s
    before = getattr(_hip.load(), counter)()
    try:
        ops.conv2d_planes(x3, wp, k, pad=pad, dil=dil, out=out, out_coff=8)
        torch.cuda.synchronize()
    finally:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    assert getattr(_hip.load(), counter)() > before
    y = out[:, 8:8 + cout].cpu().double()
    if plan[0] == 6:        # the same six products in the same order as the planes kernel: bit-identical
        out2 = torch.full_like(out, 7.0)
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 144, 1)
        try:
            ops.conv2d_planes(x3, wp, k, pad=pad, dil=dil, out=out2, out_coff=8)
            torch.cuda.synchronize()
        finally:
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
        assert torch.equal(out, out2)
    err = ((y - ref).norm() / ref.norm()).item()
    y6 = ops.conv2d(x.to(gpu), wp, k, pad=pad, dil=dil).cpu().double()       # the role-split bf16x6 kernel on the fp32 input
    err6 = ((y6 - ref).norm() / ref.norm()).item()
    assert err < 5e-7 and err <= 3 * err6 + 1e-8, (err, err6)
    assert (out[:, :8] == 7).all() and (out[:, 8 + cout:] == 7).all()


def test_three_plane_split_is_exact_to_24_bits(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 16, 5, 7, generator=g) * torch.logspace(-5, 5, 16)[None, :, None, None]
    ps = ops.to_planes(x.to(gpu), planes=3)
    raw = ps.data.cpu().view(torch.int16).view(2, 2, 3, 35, 8)            # [n][c/8][plane][hw][8]
    parts = [raw[:, :, p].contiguous().view(torch.bfloat16).float() for p in range(3)]
    back = (parts[0] + parts[1] + parts[2]).permute(0, 1, 3, 2).reshape(2, 16, 5, 7)
    assert ((back - x).abs() <= x.abs() * 2.0 ** -23).all()
    hi = x.to(torch.bfloat16).float()
    mid = (x - hi).to(torch.bfloat16).float()
    lo = (x - hi - mid).to(torch.bfloat16).float()
    want = [t.view(2, 2, 8, 35).permute(0, 1, 3, 2) for t in (hi, mid, lo)]
    for p in range(3):
        assert torch.equal(parts[p], want[p].contiguous())


@pytest.mark.parametrize("shape", [(2, 128, 17, 19, 128, 3, 1), (2, 256, 9, 9, 512, 1, 1), (1, 160, 12, 11, 136, 3, 2), (2, 512, 9, 9, 256, 1, 1)])
def test_x6_route_scope_sends_plain_bf16x6_convolutions_through_three_planes(gpu, shape):
    """ops.x6_route_scope (phase 1 / step 0): a plain precision-2 convolution with a long K loop runs as cl4_to_planes3 + the three-plane
    LDS-DMA kernel — same six bf16 products per MAC, so fp32-level against float64 (2e-6 of the output scale) and against the role-split
    kernel it replaces; the narrowing 1x1 (last shape) and everything outside the scope stay on the role-split kernel."""
    from cl4wsis_amd import _hip, ops
    n, cin, h, w, cout, k, dil = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pad = dil * (k - 1) // 2
    ref = F.conv2d(x.double(), wt.double(), padding=pad, dilation=dil).float()
    wp = ops.pack_conv_weight(wt.to(gpu), 2)
    lib = _hip.load()
    y0 = ops.conv2d(x.to(gpu), wp, k, pad=pad, dil=dil)
    b0 = lib.cl4_conv_debug_b_launches() + lib.cl4_conv_debug_s_launches()
    with ops.x6_route_scope():
        y1 = ops.conv2d(x.to(gpu), wp, k, pad=pad, dil=dil)
    routed = lib.cl4_conv_debug_b_launches() + lib.cl4_conv_debug_s_launches() - b0
    assert routed == (0 if (k == 1 and cout < cin) else 1)
    assert _rel_err(y1.cpu(), ref) < 2e-6 and _rel_err(y0.cpu(), ref) < 2e-6
    assert ops._x6_scope == 0


# (n, cin, h, w, cout, k, pad, dil, splits)
B_SPLIT_SHAPES = [
    (4, 256, 33, 33, 256, 3, 1, 1, 4),        # res4 conv2 at 2 images per GPU + flip: 62 tiles of 72 K-tiles
    (2, 1024, 33, 33, 256, 1, 0, 1, 3),       # 1x1, 32 K-tiles over 3 ranges (11 + 11 + 10)
    (3, 96, 33, 31, 136, 3, 18, 18, 8),       # most taps dropped, more ranges than some tiles have K-tiles, partial M tile
]


@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("shape", B_SPLIT_SHAPES)
def test_conv_b_split_k_equals_the_planes_kernel_split_the_same_way(gpu, shape, residual):
    """r05: K-split slabs on the LDS-DMA kernel (plan impl 6, splits > 1) for launches that cannot fill the chip. The K ranges, the
    products inside a range and the slab sum (conv_splitk_epilogue_s) are those of the planes kernel's split plan (2, 144, s): planes
    output bit-identical; against float64 within the x3 tolerance."""
    from cl4wsis_amd import _hip, ops
    n, cin, h, w, cout, k, pad, dil, splits = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    post = torch.randn(n, cout, h, w, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), padding=pad, dilation=dil) * scale[None, :, None, None] + shift[None, :, None, None], 0.01)
    if residual:
        ref = F.relu(ref + post.double())
    xs, wp = ops.to_planes(x.to(gpu)), ops.pack_conv_weight(wt.to(gpu), 1)
    d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, dil, cout, 0, ops.ACT_LEAKY, 0.01, ops.ACT_RELU if residual else 0, 0.0, 3)
    outs = {}
    for plan in ((6, 144, splits), (2, 144, splits), (6, 144, 1)):
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
        b0, s0 = _hip.load().cl4_conv_debug_b_launches(), _hip.load().cl4_conv_debug_s_launches()
        try:
            kw = dict(scale=scale.to(gpu), shift=shift.to(gpu), act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=True, want_planes=True)
            if residual:
                kw.update(post=ops.to_planes(post.to(gpu)), act2=ops.ACT_RELU)
            y, ys = ops.conv2d_planes(xs, wp, k, pad=pad, dil=dil, **kw)
            torch.cuda.synchronize()
        finally:
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
        launched_b = _hip.load().cl4_conv_debug_b_launches() > b0
        assert launched_b == (plan[0] == 6), plan
        outs[plan] = (y.clone(), ys.data.clone())
    yb, ysb = outs[(6, 144, splits)]
    assert torch.equal(ysb, outs[(2, 144, splits)][1]) and torch.equal(yb, outs[(2, 144, splits)][0])
    assert _rel_err(yb.cpu(), ref.float()) < TOL and _rel_err(outs[(6, 144, 1)][0].cpu(), ref.float()) < TOL
