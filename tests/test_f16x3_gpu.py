"""fp16x3 (r05): the differentiated convolutions as THREE fp16 MFMA products per MAC on split-fp16 operands (x = h + l, 22 significant
bits, power-of-two operand scales) instead of bf16x6's six bf16 products — cl4_pack_conv_weight_f16x3, cl4_to_planes_f16,
cl4_conv2d_fwd_planes_f16 (csrc/conv_b.hip, F16). Every case is compared with a float64 convolution and with the bf16x6 kernel on the
same operands. The bar is fp32-level agreement: 1e-6 relative L2 and no worse than 2.5x bf16x6 (measured 0.8e-7 ... 5.3e-7 against
bf16x6's 0.5e-7 ... 3.6e-7 and bf16x3's 4.4e-6: three products keep terms down to 2^-22, six down to 2^-24; an fp32 FMA chain of the
same K = 2304 dot products sits at 8e-7, tools/probes/f16x3_probe.hip)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # n, cin, h, w, cout, k, stride, pad, dil
    (2, 256, 33, 33, 256, 3, 1, 1, 1),
    (3, 1024, 17, 19, 256, 1, 1, 0, 1),
    (2, 256, 33, 33, 1024, 1, 1, 0, 1),
    (1, 2048, 33, 33, 256, 3, 1, 12, 12),
    (2, 128, 65, 65, 128, 3, 2, 1, 1),
    (2, 160, 21, 21, 136, 1, 1, 0, 1),           # partial M tile (cout 136), cin % 32 == 0 only
]


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("plan", [(6, 144, 1), (4, 144, 1)])
def test_f16x3_conv_matches_float64_like_bf16x6(gpu, shape, plan):
    from cl4wsis_amd import _hip, ops
    n, cin, h, w, cout, k, stride, pad, dil = shape
    if plan[0] == 4 and cout < 256:
        pytest.skip("256-row tiles need cout >= 256")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g) * torch.logspace(-2, 2, cin)[None, :, None, None]
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), stride=stride, padding=pad, dilation=dil)
    wp = ops.pack_conv_weight(wt.to(gpu), 3, force_f16=True)
    assert wp.precision == 3
    xs = ops.to_planes_f16(x.to(gpu))
    assert float(xs.slot[0]) == 16.0
    ho, wo = ref.shape[2:]
    out = torch.full((n, cout + 16, ho, wo), 7.0, device=gpu)
    d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, stride, pad, dil, cout + 16, 8, 0, 0.0, 0, 0.0, 5)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    before = _hip.load().cl4_conv_debug_b_launches()
    try:
        ops.conv2d_planes_f16(xs, wp, k, stride=stride, pad=pad, dil=dil, out=out, out_coff=8)
        torch.cuda.synchronize()
    finally:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    assert _hip.load().cl4_conv_debug_b_launches() > before
    assert (out[:, :8] == 7).all() and (out[:, 8 + cout:] == 7).all()
    err = _rel(out[:, 8:8 + cout].cpu().double(), ref)
    y6 = ops.conv2d(x.to(gpu), ops.pack_conv_weight(wt.to(gpu), 2), k, stride=stride, pad=pad, dil=dil).cpu().double()
    err6 = _rel(y6, ref)
    y3 = ops.conv2d(x.to(gpu), ops.pack_conv_weight(wt.to(gpu), 1), k, stride=stride, pad=pad, dil=dil).cpu().double()
    print(f"F16X3 {shape} plan {plan}: fp16x3 {err:.2e}  bf16x6 {err6:.2e}  bf16x3 {_rel(y3, ref):.2e}")
    assert err < 1e-6 and err <= 2.5 * err6 + 1e-8, (err, err6)


@pytest.mark.parametrize("magnitude", [1e-7, 1e-3, 1.0, 3e3])
def test_f16x3_dynamic_scale_makes_the_error_scale_free(gpu, magnitude):
    """Gradients have any magnitude: with the scale taken from max |x| the relative error does not depend on it (the conv route of
    ops.conv2d(..., xdyn=True), what conv2d_dgrad uses); the static 2^4 of the forward activations would lose the small ones."""
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 33, 33, generator=g) * magnitude
    wt = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    wp = ops.pack_conv_weight(wt.to(gpu), 3)
    y = ops.conv2d(x.to(gpu), wp, 3, pad=1, xdyn=True).cpu().double()
    xs = ops.to_planes_f16(x.to(gpu), dynamic=True)
    s = float(xs.slot[0])
    amax = float(x.abs().max())
    assert s == 2.0 ** round(torch.log2(torch.tensor(s)).item()) and 2.0 ** 13 <= amax * s < 2.0 ** 14, (s, amax)
    err = _rel(y, ref)
    print(f"F16X3 dynamic scale, |x| ~ {magnitude:g}: scale 2^{int(torch.log2(torch.tensor(s)).item())}, error {err:.2e}")
    assert err < 1e-6, err          # (measured 2e-7 ... 5.3e-7 over the four magnitudes; bf16x3 sits at 4.4e-6 on this shape)


def test_f16x3_zero_input_and_fused_epilogue(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 128, 19, 23, generator=g)
    wt = torch.randn(256, 128, 1, 1, generator=g) / 11.0
    sc, sh = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    post = torch.randn(2, 256, 19, 23, generator=g)
    wp = ops.pack_conv_weight(wt.to(gpu), 3)
    y = ops.conv2d(x.to(gpu), wp, 1, scale=sc.to(gpu), shift=sh.to(gpu), post=post.to(gpu), act1=ops.ACT_LEAKY, slope1=0.01, act2=ops.ACT_RELU).cpu().double()
    ref = F.relu(F.leaky_relu(F.conv2d(x.double(), wt.double()) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None], 0.01) + post.double())
    assert _rel(y, ref) < 5e-7
    z = ops.conv2d(torch.zeros(2, 128, 19, 23, device=gpu), wp, 1, xdyn=True)       # max |x| = 0: scale 1, output exactly 0
    assert (z == 0).all()


@pytest.mark.parametrize("case", [(2, 256, 33, 256, 3, 1, 1, 1), (2, 512, 17, 128, 1, 1, 0, 1), (2, 128, 33, 256, 3, 2, 1, 1), (2, 256, 33, 256, 3, 1, 2, 2)])
def test_f16x3_input_gradient(gpu, case):
    """conv2d_dgrad with fp16x3 dgrad weights (transposed-flipped, cl4_pack_conv_weight_f16x3) against float64 autograd, gradient
    magnitudes of 1e-5: the scale comes from the data."""
    from cl4wsis_amd import ops
    n, cin, h, cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, h, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    y = F.conv2d(x, wt.double(), stride=stride, padding=pad, dilation=dil)
    dy = torch.randn(y.shape, generator=g) * 1e-5
    (y * dy.double()).sum().backward()
    wd = ops.dgrad_weight(wt.to(gpu), 3)
    # 1x1 input gradients that shrink the channel count, or whose K loop (= cout of the layer) is shorter than 256, keep bf16x6
    assert wd.precision == (3 if (k > 1 or (cin >= cout and cout >= ops.F16X3_MIN_K1)) else 2)
    dx = ops.conv2d_dgrad(dy.to(gpu), wd, (h, h), k, stride=stride, pad=pad, dil=dil).cpu().double()
    wd6 = ops.dgrad_weight(wt.to(gpu), 2)
    dx6 = ops.conv2d_dgrad(dy.to(gpu), wd6, (h, h), k, stride=stride, pad=pad, dil=dil).cpu().double()
    err, err6 = _rel(dx, x.grad), _rel(dx6, x.grad)
    print(f"F16X3 dgrad {case}: fp16x3 {err:.2e} bf16x6 {err6:.2e}")
    assert err < 1e-6 and err <= 2.5 * err6 + 1e-8, (err, err6)


def test_f16x3_falls_back_to_bf16x6_for_shapes_the_route_does_not_take(gpu):
    from cl4wsis_amd import ops
    for cout, cin in ((64, 256), (256, 144), (16, 256), (256, 1024), (256, 160)):      # (the last: a K loop of 160 does not carry the conversion)
        wp = ops.pack_conv_weight(torch.randn(cout, cin, 1, 1, device=gpu), 3)
        assert wp.precision in (2, 0), (cout, cin, wp.precision)
    assert ops.pack_conv_weight(torch.randn(256, 160, 1, 1, device=gpu), 3, force_f16=True).precision == 3      # ... unless the producer left a twin
    assert ops.pack_conv_weight(torch.randn(512, 256, 1, 1, device=gpu), 3).precision == 3
    assert ops.pack_conv_weight(torch.randn(128, 1024, 3, 3, device=gpu), 3).precision == 3       # every 3x3 layer is routed


def test_batched_weight_packs_equal_the_single_layer_launches(gpu):
    """cl4_pack_conv_weights_f16x3_batched (r05: all packs of a training iteration in one launch) against cl4_pack_conv_weight_f16x3 and
    cl4_transpose_flip_weight + cl4_pack_conv_weight_f16x3, bit for bit, through nn.PackBatch (registration, in-place refresh, cache marks)."""
    import torch.nn as tnn
    from cl4wsis_amd import body_train, nn as cnn, ops
    if not ops.planes_f16() or body_train.PRECISION != 3:
        pytest.skip("fp16x3 training arithmetic only")
    g = torch.Generator().manual_seed(21)
    shapes = [(256, 128, 1), (64 + 64, 256, 3), (1024, 256, 1), (256, 1024, 1), (160, 128, 3)]     # cout, cin, k
    convs = []
    for cout, cin, k in shapes:
        c = tnn.Conv2d(cin, cout, k, bias=False)
        c.weight.data = torch.randn(cout, cin, k, k, generator=g) * 0.05
        convs.append(c.to(gpu))
    batch = cnn.PackBatch()
    with batch:
        first = [(cnn.packed_weight(c, 3, force_f16=True), body_train._dgrad_weight(c, force_f16=True)) for c in convs]
    assert len(batch.entries) == 2 * len(convs) and all(a.precision == 3 and b.precision == 3 for a, b in first)
    with torch.no_grad():
        for c in convs:
            c.weight.mul_(1.5).add_(0.01)                    # an "optimizer step": bumps the version, the cache entries go stale
    with batch:
        assert batch.refresh() and batch.launches == 1
        again = [(cnn.packed_weight(c, 3, force_f16=True), body_train._dgrad_weight(c, force_f16=True)) for c in convs]
    for (a0, b0), (a1, b1), c in zip(first, again, convs):
        assert a1 is a0 and b1 is b0                         # refreshed in place, the lazy path found the entries current
        ref_f = ops.pack_conv_weight(c.weight.detach(), 3, force_f16=True)
        ref_d = ops.dgrad_weight(c.weight.detach(), 3, force_f16=True)
        assert torch.equal(a1.data, ref_f.data), tuple(c.weight.shape)
        assert torch.equal(b1.data, ref_d.data), tuple(c.weight.shape)


@pytest.mark.parametrize("plan", [(6, 144, 1), (4, 144, 1), (2, 128, 1), (2, 144, 1), (2, 64, 1), (3, 128, 1), (2, 144, 3)])
@pytest.mark.parametrize("shape", [(2, 256, 33, 33, 256, 3, 1), (3, 128, 21, 19, 136, 1, 0), (2, 64, 40, 40, 256, 1, 0)])
def test_convolution_leaves_the_batchnorm_partial_sums(gpu, shape, plan):
    """r05 (cl4_conv2d_fwd_planes_stats): the workgroups of an unsplit planes launch add up the rows of their parked tile, so the train-mode
    ABN behind the convolution needs no statistics pass. Per channel the partials must sum to the sums of the fp32 output the same launch
    wrote (1e-6: fp32 row sums of <= 144 elements, folded in double); a K-split plan reports 0 parts and the caller falls back."""
    from cl4wsis_amd import _hip, ops
    if not ops.planes_f16():
        pytest.skip("fp16 planes build only")
    n, cin, h, w, cout, k, pad = shape
    if plan[0] == 4 and cout < 256:
        pytest.skip("256-row tiles need cout >= 256")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g) + 0.3
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    xs = ops.to_planes(x.to(gpu))
    wp = ops.pack_conv_weight(wt.to(gpu), 3, force_f16=True)
    d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, 1, cout, 0, 0, 0.0, 0, 0.0, 3)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    try:
        parts = _hip.load().cl4_conv2d_fwd_planes_stats_parts(C.byref(d))
        st = []
        y = ops.conv2d_planes(xs, wp, k, pad=pad, want_fp32=True, stats_out=st)[0]
        torch.cuda.synchronize()
    finally:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    if plan[2] > 1:
        assert parts == 0 and not st
        return
    tile = {6: 144, 4: 144, 3: 128}.get(plan[0], plan[1])
    assert parts == -(-(n * h * w) // tile) and len(st) == 1 and st[0][1] == parts
    part = st[0][0].view(cout, parts, 2).sum(1).cpu()
    yd = y.double().cpu()
    ref = torch.stack([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))], 1)
    assert ((part - ref).abs() <= 1e-6 * ref.abs() + 1e-6 * yd.abs().sum((0, 2, 3))[:, None] * 1e-1 + 1e-9).all(), (part - ref).abs().max()
    ref64 = F.conv2d(x.double(), wt.double(), padding=pad)
    assert _rel(yd, ref64) < 1e-6


@pytest.mark.parametrize("one_rank", [True, False])
def test_train_mode_abn_reads_the_convolutions_partial_sums(gpu, monkeypatch, one_rank):
    """conv -> train-mode ABN of the differentiated networks (body_train.conv_abn_train) with and without the fused statistics: the same
    mean / variance (1e-6 relative to the standard deviation), running statistics and activation — on the one-rank path
    (cl4_bn_stats_from_partials) and on the two-phase path of N > 1 (cl4_bn_sync_stats_from_partials -> exchange -> final)."""
    import torch.nn as tnn
    from cl4wsis_amd import body_train, nn as cnn, ops
    if not ops.planes_f16() or body_train.PRECISION != 3:
        pytest.skip("fp16x3 training arithmetic only")
    g = torch.Generator().manual_seed(31)
    conv = tnn.Conv2d(256, 256, 3, padding=1, bias=False)
    conv.weight.data = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    conv = conv.to(gpu)
    x = (torch.randn(4, 256, 33, 33, generator=g) * 1.3 + 0.2).to(gpu)
    outs = []
    monkeypatch.setattr(ops, "ABN_ONE_RANK", one_rank)
    for fused in (False, True):
        monkeypatch.setattr(ops, "CONV_STATS", fused)
        bn = cnn.InPlaceABNSync(256).to(gpu)
        bn.weight.data = (torch.rand(256, generator=torch.Generator().manual_seed(5)) + 0.5).to(gpu)
        bn.train()
        y, r = body_train.conv_abn_train(x, conv, bn)
        torch.cuda.synchronize()
        outs.append((y, r.mean, r.var, bn.running_mean.clone(), bn.running_var.clone()))
    (y0, m0, v0, rm0, rv0), (y1, m1, v1, rm1, rv1) = outs
    sd = v0.sqrt()
    assert ((m0 - m1).abs() <= 1e-6 * sd + 1e-7).all() and ((v0 - v1).abs() <= 2e-6 * v0).all()
    assert torch.allclose(rm0, rm1, rtol=1e-5, atol=1e-7) and torch.allclose(rv0, rv1, rtol=1e-5)
    assert torch.allclose(y0, y1, rtol=1e-4, atol=1e-5)
