"""GPU parity of the implicit-GEMM conv kernels (cl4_conv2d_fwd / cl4_conv2d_wgrad) against a
plain PyTorch fp32 CPU reference of the same op (floating-point kernel: tolerance 1e-4 relative
to the output scale; the north-star tolerance is 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (n, cin, h, w, cout, k, stride, pad, dil) — the layer shapes of ResNet101 / DeepLabV3 / decoder at small size
CASES = [
    (2, 64, 17, 17, 256, 1, 1, 0, 1),      # bottleneck 1x1 expand
    (2, 128, 33, 33, 128, 3, 2, 1, 1),     # strided 3x3
    (1, 96, 33, 33, 160, 3, 1, 6, 6),      # ASPP dilation 6
    (1, 64, 33, 33, 72, 3, 1, 18, 18),     # dilation 18: most taps out of bounds
    (2, 3, 65, 65, 64, 7, 2, 3, 1),        # stem
    (2, 256, 9, 9, 21, 1, 1, 0, 1),        # classifier, cout < 32
    (1, 37, 13, 11, 50, 3, 1, 1, 1),       # ragged everything
    (1, 5, 33, 33, 5, 1, 1, 0, 1),         # PeakGenerator 5->5
    (3, 512, 5, 5, 130, 1, 1, 0, 1),       # cout just above one M tile
]


def _rel_err(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


# precision 0: exact fp32 MFMA; 1: bf16x3 split; 2: bf16x6 (both fall back to fp32 for cout < 64 / cout % 4 != 0)
TOL = {0: 1e-5, 1: 5e-5, 2: 2e-6}


@pytest.mark.parametrize("precision", [0, 1, 2])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_plain(gpu, case, precision):
    from cl4wsis_amd import ops
    n, cin, h, w, cout, k, s, p, d = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), stride=s, padding=p, dilation=d).float()
    y = ops.conv2d(x.to(gpu), ops.pack_conv_weight(wt.to(gpu), precision), k, stride=s, pad=p, dil=d)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert _rel_err(y.cpu(), ref) < TOL[precision]


@pytest.mark.parametrize("precision", [0, 1, 2])
@pytest.mark.parametrize("plan", [(2, 64, 1), (2, 64, 3), (2, 128, 1), (2, 128, 4), (1, 128, 1)])
def test_conv_fwd_pinned_plans(gpu, plan, precision):
    """Every launch plan (kernel variant, tile, K split) computes the same convolution."""
    import ctypes as C
    from cl4wsis_amd import _hip, ops
    if plan[0] == 1 and precision != 0:
        pytest.skip("the register-staged kernel is fp32 only")
    g = torch.Generator().manual_seed(21)
    n, cin, h, w, cout, k, pad, dil = 3, 160, 21, 19, 192, 3, 2, 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * 9) ** 0.5
    post = torch.randn(n, cout, h, w, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), padding=pad, dilation=dil).float() + post, 0.01)
    xd = x.to(gpu)
    wp = ops.pack_conv_weight(wt.to(gpu), precision)
    d = ops._conv_desc(xd, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    try:
        y = ops.conv2d(xd, wp, k, pad=pad, dil=dil, post=post.to(gpu), act2=ops.ACT_LEAKY, slope2=0.01)
        torch.cuda.synchronize()
    finally:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    assert _rel_err(y.cpu(), ref) < TOL[precision]


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_autotuner_keeps_the_result(gpu, precision):
    """ops.AUTOTUNE times the candidate plans of a new shape and pins the fastest (what Trainer / bench do during warm-up);
    whatever it picks, the convolution is the same within the arithmetic's tolerance, and clear_plans forgets the choice."""
    from cl4wsis_amd import _hip, ops
    g = torch.Generator().manual_seed(4)
    n, cin, h, cout, k = 4, 256, 33, 256, 3
    x = torch.randn(n, cin, h, h, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * 9) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), padding=1).float()
    xd, wp = x.to(gpu), ops.pack_conv_weight(wt.to(gpu), precision)
    y0 = ops.conv2d(xd, wp, k, pad=1)
    ops.AUTOTUNE = True
    try:
        y1 = ops.conv2d(xd, wp, k, pad=1)
    finally:
        ops.AUTOTUNE = False
    _hip.call("cl4_conv2d_fwd_clear_plans")
    y2 = ops.conv2d(xd, wp, k, pad=1)
    assert _rel_err(y1.cpu(), ref) < TOL[precision] and _rel_err(y0.cpu(), ref) < TOL[precision]
    assert torch.equal(y0, y2)                      # heuristic plan again: bit-identical to the first call


@pytest.mark.parametrize("precision", [1, 2])
@pytest.mark.parametrize("plan", [(2, 64, -4), (2, 128, -8), (2, 64, -16), (2, 128, -1)])
def test_conv_fwd_main_plus_split_tail(gpu, plan, precision):
    """Plans with splits < 0: the whole rounds of tiles run unsplit, the last partial round is split — same convolution
    (fused epilogue on both parts), on a grid with more tiles than resident-workgroup slots and a ragged last pixel tile."""
    import ctypes as C
    from cl4wsis_amd import _hip, ops
    g = torch.Generator().manual_seed(8)
    n, cin, h, w, cout, k = 5, 64, 65, 67, 256, 3          # 21775 pixels: 4 x 341 tiles of 64, 2 x 171 tiles of 128
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * 9) ** 0.5
    post = torch.randn(n, cout, h, w, generator=g)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), padding=1).float() * scale[None, :, None, None] + shift[None, :, None, None]) + post
    xd = x.to(gpu)
    wp = ops.pack_conv_weight(wt.to(gpu), precision)
    d = ops._conv_desc(xd, cin, 0, cout, k, 1, 1, 1, cout, 0, ops.ACT_RELU, 0.0, 0, 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    ws = _hip.load().cl4_conv2d_fwd_workspace_bytes(C.byref(d))
    assert (ws > 0) == (plan[2] != -1)          # split tails need slabs; the small-tile tail (-1) does not
    y = ops.conv2d(xd, wp, k, pad=1, scale=scale.to(gpu), shift=shift.to(gpu), post=post.to(gpu), act1=ops.ACT_RELU)
    torch.cuda.synchronize()
    assert _rel_err(y.cpu(), ref) < TOL[precision]


@pytest.mark.parametrize("k,cin", [(1, 64), (3, 32), (1, 40)])
@pytest.mark.parametrize("tile", [64, 128])
def test_conv_fwd_persistent_tiles(gpu, tile, k, cin):
    """More output tiles than resident workgroups: persistent workgroups walk several tiles each (uneven counts: 546 tiles of
    128x128 on 512 workgroups), K-tile stream continuous across tile boundaries, fused residual epilogue per tile.
    cin = 40 takes the generic (not tap-major) loader."""
    import ctypes as C
    from cl4wsis_amd import _hip, ops
    g = torch.Generator().manual_seed(12)
    n, h, w, cout = 8, 65, 67, 256                          # 34840 pixels
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    post = torch.randn(n, cout, h, w, generator=g)
    shift = torch.randn(cout, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), padding=k // 2).float() + shift[None, :, None, None] + post, 0.01)
    xd = x.to(gpu)
    wp = ops.pack_conv_weight(wt.to(gpu), 1)
    d = ops._conv_desc(xd, cin, 0, cout, k, 1, k // 2, 1, cout, 0, 0, 0.0, ops.ACT_LEAKY, 0.01, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, tile, 1)
    y = ops.conv2d(xd, wp, k, pad=k // 2, shift=shift.to(gpu), post=post.to(gpu), act2=ops.ACT_LEAKY, slope2=0.01)
    torch.cuda.synchronize()
    assert _rel_err(y.cpu(), ref) < TOL[1]


@pytest.mark.parametrize("case", [(2, 128, 65, 65, 15), (1, 37, 40, 33, 2), (2, 512, 33, 33, 32), (3, 64, 129, 129, 5), (1, 16, 32, 32, 17)])
def test_conv_thin_1x1(gpu, case):
    """cout <= 32 1x1 convs (heads, low-level projections; the register-staged fp32 kernel): every epilogue operand, input and
    output channel slices, ragged cin and pixel counts. (A scalar-weight streaming kernel for these layers was measured and
    dropped: 313.7 vs 327.2 images/s — its per-FMA scalar weight loads serialise the wave.)"""
    from cl4wsis_amd import ops
    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(cin * 7 + cout)
    xfull = torch.randn(n, cin + 3, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    pre, post = torch.randn(n, cout, h, w, generator=g), torch.randn(n, cout, h, w, generator=g)
    x = xfull[:, 2:2 + cin]
    ref = F.conv2d(x.double(), wt.double()).float()
    ref = F.leaky_relu((ref + pre) * scale[None, :, None, None] + shift[None, :, None, None], 0.2)
    ref = F.relu(ref + post)
    out = torch.full((n, cout + 4, h, w), 7.0, device=gpu)
    wp = ops.pack_conv_weight(wt.to(gpu), 0)
    ops.conv2d(xfull.to(gpu), wp, 1, scale=scale.to(gpu), shift=shift.to(gpu), pre=pre.to(gpu), post=post.to(gpu),
               act1=ops.ACT_LEAKY, slope1=0.2, act2=ops.ACT_RELU, out=out, out_coff=1, in_coff=2, cin=cin)
    torch.cuda.synchronize()
    assert _rel_err(out[:, 1:1 + cout].cpu(), ref) < 1e-5
    assert (out[:, 0] == 7).all() and (out[:, 1 + cout:] == 7).all()          # neighbouring channels untouched


def test_conv_fwd_fused_epilogue(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(7)
    n, cin, h, w, cout = 2, 48, 19, 19, 80
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    mean, var = torch.randn(cout, generator=g), torch.rand(cout, generator=g) + 0.2
    pre = torch.randn(n, cout, h, w, generator=g)
    post = torch.randn(n, cout, h, w, generator=g)
    z = F.conv2d(x, wt, padding=1) + pre
    z = F.leaky_relu(F.batch_norm(z, mean, var, gamma, beta, False, 0.0, 1e-5), 0.01)
    ref = F.relu(z + post)
    scale, shift = ops.bn_fold(gamma.to(gpu), beta.to(gpu), mean.to(gpu), var.to(gpu), 1e-5)
    for precision in (0, 1, 2):
        y = ops.conv2d(x.to(gpu), ops.pack_conv_weight(wt.to(gpu), precision), 3, pad=1, scale=scale, shift=shift,
                       pre=pre.to(gpu), post=post.to(gpu), act1=ops.ACT_LEAKY, slope1=0.01, act2=ops.ACT_RELU)
        torch.cuda.synchronize()
        assert _rel_err(y.cpu(), ref) < TOL[precision]


def test_conv_channel_slices(gpu):
    """Reading a channel slice of x and writing a channel slice of y (the torch.cat replacement)."""
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 40, 15, 15, generator=g)
    w1 = torch.randn(24, 16, 1, 1, generator=g)
    w2 = torch.randn(40, 16, 3, 3, generator=g)
    ref = torch.cat([F.conv2d(x[:, 8:24], w1), F.conv2d(x[:, 8:24], w2, padding=2, dilation=2)], 1)
    xd = x.to(gpu)
    out = torch.full((2, 64, 15, 15), float("nan"), device=gpu)
    ops.conv2d(xd, ops.pack_conv_weight(w1.to(gpu)), 1, in_coff=8, cin=16, out=out, out_coff=0)
    ops.conv2d(xd, ops.pack_conv_weight(w2.to(gpu)), 3, pad=2, dil=2, in_coff=8, cin=16, out=out, out_coff=24)
    torch.cuda.synchronize()
    assert _rel_err(out.cpu(), ref) < 5e-5


@pytest.mark.parametrize("case", [
    (2, 64, 17, 17, 256, 1, 1, 0, 1),
    (2, 96, 33, 33, 160, 3, 1, 3, 3),
    (3, 40, 21, 19, 50, 3, 1, 1, 1),
    (2, 128, 33, 33, 2, 1, 1, 0, 1),
    (2, 3, 65, 65, 64, 7, 2, 3, 1),        # stem: 49 taps, stride 2
    (2, 32, 33, 33, 48, 3, 2, 1, 1),       # strided 3x3
    (1, 24, 33, 33, 40, 3, 1, 12, 12),     # ASPP dilation
])
@pytest.mark.parametrize("precision", [0, 1, 2])
def test_conv_wgrad(gpu, case, precision):
    from cl4wsis_amd import ops
    n, cin, h, w, cout, k, s, p, d = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
    wt = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, wt, stride=s, padding=p, dilation=d)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    ref = wt.grad.float()
    dw = torch.zeros(cout, cin, k, k, device=gpu)
    ops.conv2d_wgrad(x.float().to(gpu), dy.float().to(gpu), dw, k, stride=s, pad=p, dil=d, precision=precision)
    torch.cuda.synchronize()
    assert _rel_err(dw.cpu(), ref) < {0: 2e-6, 1: 1e-5, 2: 2e-6}[precision]


def test_conv_wgrad_many_k_tiles_and_accumulate(gpu):
    """Pixel counts that are not a multiple of the K tile pair (odd number of tiles, ragged last tile) through the two-set
    register pipeline of the bf16 kernels, and accumulation into a non-zero dw."""
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(3)
    for n, h, w in ((1, 5, 7), (1, 8, 8), (3, 9, 11), (5, 33, 33)):
        x = torch.randn(n, 72, h, w, generator=g, dtype=torch.float64)
        wt = torch.zeros(136, 72, 3, 3, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(n, 136, h, w, generator=g, dtype=torch.float64)
        F.conv2d(x, wt, padding=1).backward(dy)
        for precision in (1, 2):
            dw = torch.ones(136, 72, 3, 3, device=gpu)
            ops.conv2d_wgrad(x.float().to(gpu), dy.float().to(gpu), dw, 3, pad=1, precision=precision)
            assert _rel_err(dw.cpu() - 1, wt.grad.float()) < 2e-5, (n, h, w, precision)


def test_bad_arguments_raise(gpu):
    from cl4wsis_amd import ops
    x = torch.zeros(1, 4, 8, 8, device=gpu)
    wp = torch.zeros(4 * 25, 8, device=gpu)
    with pytest.raises(RuntimeError):
        ops.conv2d(x, wp, 5, pad=2)  # 5x5 dense conv is not part of the path


@pytest.mark.parametrize("k,d", [(1, 1), (3, 1), (3, 2), (3, 6)])
@pytest.mark.parametrize("precision", [1, 2])
def test_conv_wgrad_channel_slices_vector_loader(gpu, k, d, precision):
    """The stride-1 "same" weight gradients run on the 16-byte loader (K = the per-image padded pixel space): channel slices of both
    operands, map sizes whose pixel count is not a multiple of 4, widths of 4 and 5 (every 4-pixel group wraps a row), against autograd;
    CL4_WGRAD_VEC=0 (the dword loader) must give the same numbers."""
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(100 + k * 10 + d)
    for n, h, w in ((2, 9, 13), (3, 6, 4), (1, 7, 5), (2, 33, 33)):
        x = torch.randn(n, 40, h, w, generator=g, dtype=torch.float64)
        dy = torch.randn(n, 72, h, w, generator=g, dtype=torch.float64)
        wt = torch.zeros(48, 24, k, k, dtype=torch.float64, requires_grad=True)
        F.conv2d(x[:, 8:32], wt, padding=d * (k // 2), dilation=d).backward(dy[:, 16:64])
        dw = torch.zeros(48, 24, k, k, device=gpu)
        ops.conv2d_wgrad(x.float().to(gpu), dy.float().to(gpu), dw, k, pad=d * (k // 2), dil=d, in_coff=8, cin=24, dy_coff=16, cout=48,
                         precision=precision)
        tol = 2e-6 if precision == 2 and k > 1 else 1e-5
        assert _rel_err(dw.cpu(), wt.grad.float()) < tol, (n, h, w)
