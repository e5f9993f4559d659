"""GPU: the fused identity bottleneck (csrc/conv_f.hip, cl4_bottleneck_fused) — ResidualBlock.forward of modules/residual.py:84-97 for an
identity block of a frozen network as ONE launch — against (a) the same block as three planes launches (cl4_conv2d_fwd_planes: the fused
kernel makes the same products in the same order, so the comparison is bit for bit), (b) a float64 evaluation of the reference's formula,
and (c) through the ResNet body with the fusion switched on and off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _block(cm, seed, gpu):
    g = torch.Generator().manual_seed(seed)
    c4 = 4 * cm
    w1 = torch.randn(cm, c4, 1, 1, generator=g) * (1.0 / c4) ** 0.5
    w2 = torch.randn(cm, cm, 3, 3, generator=g) * (1.0 / (9 * cm)) ** 0.5
    w3 = torch.randn(c4, cm, 1, 1, generator=g) * (1.0 / cm) ** 0.5
    aff = [torch.rand(c, generator=g) * 1.5 + 0.25 if i % 2 == 0 else torch.randn(c, generator=g) * 0.3
           for i, c in enumerate((cm, cm, cm, cm, c4, c4))]            # scale1, shift1, scale2, shift2, scale3, shift3
    return [t.to(gpu) for t in (w1, w2, w3)], [t.to(gpu) for t in aff]


def _reference64(x, ws, aff, slope=0.01):
    w1, w2, w3 = (w.double() for w in ws)
    s1, b1, s2, b2, s3, b3 = (t.double()[None, :, None, None] for t in aff)
    y = F.leaky_relu(F.conv2d(x, w1) * s1 + b1, slope)
    y = F.leaky_relu(F.conv2d(y, w2, padding=1) * s2 + b2, slope)
    y = F.conv2d(y, w3) * s3 + b3
    return F.leaky_relu(y + x, slope)


# the last three shapes have more workgroups than the chip holds at once (> 512): the second round runs under memory load, which is
# what exposed the store-data hazard of FINDINGS.md finding 37 (wrong first dwords of output chunks in those workgroups only)
@pytest.mark.parametrize("cm,n,h,w,want_fp32", [(64, 2, 33, 33, True), (64, 1, 29, 31, False), (64, 3, 13, 13, True), (64, 1, 40, 17, True),
                                               (64, 2, 14, 27, False), (64, 1, 129, 129, True), (64, 6, 129, 129, False),
                                               (64, 24, 65, 65, True), (64, 70, 33, 33, False),
                                               (128, 2, 33, 33, True), (128, 1, 29, 31, False), (128, 3, 13, 13, True), (128, 2, 65, 65, True),
                                               (128, 1, 14, 27, False), (128, 24, 65, 65, False), (128, 40, 33, 33, True)])
def test_fused_bottleneck_matches_three_planes_launches_and_float64(gpu, cm, n, h, w, want_fp32):
    from cl4wsis_amd import ops
    ws, aff = _block(cm, 7 * n + h, gpu)
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(n, 4 * cm, h, w, generator=g).to(gpu)
    xp = ops.to_planes(x)
    x_seen = ops.from_planes(xp)                              # what every bf16x3 kernel sees of x (hi + lo)
    p1, p2, p3 = (ops.pack_conv_weight(t, 1) for t in ws)
    s1, b1, s2, b2, s3, b3 = aff
    # (a) three launches, pinned to unsplit plans: a K split sums partial slabs in another order, and a last-bit difference of an fp32
    # pre-activation moves the hi + lo split of an intermediate by up to 2^-17 (measured: 6e-5 absolute on the output at n = 2, 33 x 33)
    import ctypes as C
    for cin, cout, k, pad in ((4 * cm, cm, 1, 0), (cm, cm, 3, 1), (cm, 4 * cm, 1, 0)):
        d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, 1, cout, 0, ops.ACT_LEAKY, 0.01, ops.ACT_NONE, 0.01, 3)
        ops.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    _, y1 = ops.conv2d_planes(xp, p1, 1, scale=s1, shift=b1, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    _, y2 = ops.conv2d_planes(y1, p2, 3, pad=1, scale=s2, shift=b2, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    ref_f, ref_p = ops.conv2d_planes(y2, p3, 1, scale=s3, shift=b3, post=xp, act2=ops.ACT_LEAKY, slope2=0.01, want_fp32=True,
                                     want_planes=True)
    # (b) one launch
    before = ops._hip.load().cl4_conv_debug_f_launches()
    out_f, out_p = ops.bottleneck_fused(xp, p1, p2, p3, torch.cat(aff).contiguous(), 0.01, 0.01, 0.01, want_fp32=want_fp32)
    torch.cuda.synchronize()
    assert ops._hip.load().cl4_conv_debug_f_launches() == before + 1
    assert (out_f is not None) == want_fp32
    got = ops.from_planes(out_p)
    want = ops.from_planes(ref_p)
    assert torch.equal(out_p.data, ref_p.data), f"planes output differs from the three-launch path: max |d| {(got - want).abs().max().item():.3e}"
    if want_fp32:
        assert torch.equal(out_f, ref_f)
        assert torch.equal(ops.to_planes(out_f).data, out_p.data)     # the two outputs of one launch hold the same values
    r64 = _reference64(x_seen.double().cpu(), [t.cpu() for t in ws], [t.cpu() for t in aff])
    err = ((got.double().cpu() - r64).abs().max() / r64.abs().max()).item()
    assert err < 2e-5, err                                       # three bf16x3 layers: ~4e-6 each


@pytest.mark.parametrize("n,h,w,want_fp32", [(2, 33, 33, True), (1, 29, 31, False), (3, 13, 13, True), (1, 129, 129, True), (7, 129, 129, False),
                                            (40, 40, 37, True)])
def test_fused_first_block_with_projection_shortcut(gpu, n, h, w, want_fp32):
    """The stage's first block (modules/residual.py:79-82, 90-91: conv -> bn shortcut from the cm-channel input; res2 block 1) as one launch
    against the four planes launches (shortcut, conv1, conv2, conv3 + shortcut), bit for bit, and against float64."""
    import ctypes as C
    from cl4wsis_amd import ops
    cm = 64
    g = torch.Generator().manual_seed(1000 * n + h)
    w1 = (torch.randn(cm, cm, 1, 1, generator=g) / cm ** 0.5).to(gpu)
    w2 = (torch.randn(cm, cm, 3, 3, generator=g) / (9 * cm) ** 0.5).to(gpu)
    w3 = (torch.randn(4 * cm, cm, 1, 1, generator=g) / cm ** 0.5).to(gpu)
    wp = (torch.randn(4 * cm, cm, 1, 1, generator=g) / cm ** 0.5).to(gpu)
    aff = [(torch.rand(c, generator=g) * 1.5 + 0.25 if i % 2 == 0 else torch.randn(c, generator=g) * 0.3).to(gpu)
           for i, c in enumerate((cm, cm, cm, cm, 4 * cm, 4 * cm, 4 * cm, 4 * cm))]
    s1, b1, s2, b2, s3, b3, sp, bp = aff
    x = torch.randn(n, cm, h, w, generator=g).to(gpu)
    xp = ops.to_planes(x)
    p1, p2, p3, pp = (ops.pack_conv_weight(t, 1) for t in (w1, w2, w3, wp))
    for cin, cout, k, pad in ((cm, cm, 1, 0), (cm, cm, 3, 1), (cm, 4 * cm, 1, 0)):
        d = ops.ConvDescFor(n, cin, h, w, cin, 0, cout, k, 1, pad, 1, cout, 0, ops.ACT_LEAKY, 0.01, ops.ACT_NONE, 0.01, 3)
        ops.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    _, rp = ops.conv2d_planes(xp, pp, 1, scale=sp, shift=bp, want_fp32=False, want_planes=True)
    _, y1 = ops.conv2d_planes(xp, p1, 1, scale=s1, shift=b1, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    _, y2 = ops.conv2d_planes(y1, p2, 3, pad=1, scale=s2, shift=b2, act1=ops.ACT_LEAKY, slope1=0.01, want_fp32=False, want_planes=True)
    ref_f, ref_p = ops.conv2d_planes(y2, p3, 1, scale=s3, shift=b3, post=rp, act2=ops.ACT_LEAKY, slope2=0.01, want_fp32=True, want_planes=True)
    out_f, out_p = ops.bottleneck_fused(xp, p1, p2, p3, torch.cat(aff).contiguous(), 0.01, 0.01, 0.01, want_fp32=want_fp32, wp=pp)
    torch.cuda.synchronize()
    got, want = ops.from_planes(out_p), ops.from_planes(ref_p)
    assert torch.equal(out_p.data, ref_p.data), f"planes output differs from the four-launch path: max |d| {(got - want).abs().max().item():.3e}"
    if want_fp32:
        assert torch.equal(out_f, ref_f)
    xs64 = ops.from_planes(xp).double().cpu()
    c = lambda t: t.double().cpu()
    v = lambda t: t.double().cpu()[None, :, None, None]
    y = F.leaky_relu(F.conv2d(xs64, c(w1)) * v(s1) + v(b1), 0.01)
    y = F.leaky_relu(F.conv2d(y, c(w2), padding=1) * v(s2) + v(b2), 0.01)
    r64 = F.leaky_relu(F.conv2d(y, c(w3)) * v(s3) + v(b3) + F.conv2d(xs64, c(wp)) * v(sp) + v(bp), 0.01)
    err = ((got.double().cpu() - r64).abs().max() / r64.abs().max()).item()
    assert err < 3e-5, err


def test_fused_bottleneck_fp32_copy_can_be_limited_to_the_first_images(gpu):
    """train.py:265-267 reads the low-level features of the un-flipped half of the flip-TTA batch only: the fp32 copy of a stage output is
    written for images [0, f32_images); the planes output always covers the whole batch."""
    from cl4wsis_amd import ops
    ws, aff = _block(64, 3, gpu)
    x = torch.randn(4, 256, 20, 15, generator=torch.Generator().manual_seed(8)).to(gpu)
    xp = ops.to_planes(x)
    ps = [ops.pack_conv_weight(t, 1) for t in ws]
    full_f, full_p = ops.bottleneck_fused(xp, *ps, torch.cat(aff).contiguous(), 0.01, 0.01, 0.01, want_fp32=True)
    part_f, part_p = ops.bottleneck_fused(xp, *ps, torch.cat(aff).contiguous(), 0.01, 0.01, 0.01, want_fp32=True, f32_images=2)
    torch.cuda.synchronize()
    assert torch.equal(part_p.data, full_p.data)
    assert torch.equal(part_f[:2], full_f[:2])
    assert torch.equal(ops.to_planes(full_f).data, full_p.data)


def test_fused_bottleneck_rejects_what_it_cannot_do(gpu):
    from cl4wsis_amd import ops
    ws, aff = _block(64, 1, gpu)
    xp = ops.to_planes(torch.randn(1, 256, 13, 13, device=gpu))
    p1, p2, p3 = (ops.pack_conv_weight(t, 1) for t in ws)
    with pytest.raises(RuntimeError):
        ops.bottleneck_fused(xp, p1, p2, p3, torch.cat(aff), 0.01, 0.01, 0.01, out_planes=xp)        # in place: neighbours read the halo
    with pytest.raises(RuntimeError):
        ops.bottleneck_fused(xp, p1, p3, p2, torch.cat(aff), 0.01, 0.01, 0.01)                       # weights in the wrong roles
    ws32, aff32 = _block(32, 2, gpu)              # width 32 is not built (64 and 128 are)
    xp32 = ops.to_planes(torch.randn(1, 128, 13, 13, device=gpu))
    with pytest.raises(RuntimeError):
        ops.bottleneck_fused(xp32, *(ops.pack_conv_weight(t, 1) for t in ws32), torch.cat(aff32), 0.01, 0.01, 0.01)   # width not built


def test_resnet_body_is_unchanged_by_the_fusion(gpu, monkeypatch):
    """The frozen ResNet101 body (models/resnet.py:106-123) with the identity blocks of res2 fused against the same body with every
    block as separate launches: the five stage outputs agree to the last bit of what the planes trunk holds, and the fused path ran."""
    from cl4wsis_amd import backbone, nn as cnn, ops
    from oracle import weights
    torch.manual_seed(0)
    body = backbone.net_resnet101(norm_act=cnn.InPlaceABNSync, output_stride=16, keep_outputs=True)
    body.load_state_dict(weights.fill_(body.state_dict(), seed=5))
    body.to(gpu).eval()
    for p in body.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 3, 97, 97, generator=torch.Generator().manual_seed(3)).to(gpu)
    lib = ops._hip.load()
    outs = {}
    monkeypatch.setattr(ops, "FUSED_CM", (64, 128))            # res3 too (off by default: not faster, see ops.FUSED_CM)
    for fused in (False, True):
        monkeypatch.setattr(ops, "FUSED_BLOCK", fused)
        n0 = lib.cl4_conv_debug_f_launches()
        with torch.no_grad():
            outs[fused] = [t.clone() for t in body(x)]
        torch.cuda.synchronize()
        assert (lib.cl4_conv_debug_f_launches() - n0) == (6 if fused else 0)      # res2 blocks 1 (projection), 2, 3 and res3 blocks 2, 3, 4
    for a, b in zip(outs[False], outs[True]):
        assert a.shape == b.shape
        err = ((a - b).abs().max() / a.abs().max()).item()
        assert err < 1e-4, err          # other launch plans (K splits) for res2 re-associate fp32 sums; measured 1.5e-5 through the 33 blocks
