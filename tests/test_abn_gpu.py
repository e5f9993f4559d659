"""GPU parity of the train-mode InPlaceABNSync path (cl4_bn_sync_*: SURVEY §8 A2) against PyTorch-CPU autograd of
F.batch_norm(training=True) + leaky-ReLU: one rank, and two shards whose statistics are combined the way the all-reduce
does (the second shard's sums are computed with the same kernels and added) — the result must equal full-batch BN."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _reference(x, gamma, beta, dy, slope):
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(x.shape[1]), torch.ones(x.shape[1])
    y = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    y = F.leaky_relu(y, slope) if slope is not None else y
    y.backward(dy)
    return y.detach(), xr.grad, gr.grad, br.grad, rm, rv


@pytest.mark.parametrize("slope", [0.01, None])
def test_abn_train_single_rank(gpu, slope):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 37, 17, 13, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(37, generator=g) + 0.5, torch.randn(37, generator=g)
    dy = torch.randn(x.shape, generator=g)
    y_ref, dx_ref, dg_ref, db_ref, rm_ref, rv_ref = _reference(x, gamma, beta, dy, slope)
    act = ops.ACT_LEAKY if slope is not None else ops.ACT_NONE
    rm, rv = torch.zeros(37, device=gpu), torch.ones(37, device=gpu)
    xd, gd, bd = x.to(gpu), gamma.to(gpu), beta.to(gpu)
    y, mean, var, count = ops.abn_train_forward(xd, gd, bd, rm, rv, act=act, slope=slope or 0.0)
    dgamma, dbeta = torch.zeros(37, device=gpu), torch.zeros(37, device=gpu)
    dx = ops.abn_train_backward(xd, dy.to(gpu), mean, var, gd, bd, dgamma, dbeta, count, act=act, slope=slope or 0.0)
    assert _rel(y, y_ref) < 1e-5 and _rel(rm, rm_ref) < 1e-6 and _rel(rv, rv_ref) < 1e-6
    assert _rel(dx, dx_ref) < 2e-5 and _rel(dgamma, dg_ref) < 2e-5 and _rel(dbeta, db_ref) < 2e-5


def test_abn_train_two_shards_equal_full_batch(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(6, 24, 9, 11, generator=g) + torch.arange(6).float()[:, None, None, None] * 0.3   # shards differ in mean
    gamma, beta = torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g)
    dy = torch.randn(x.shape, generator=g)
    y_ref, dx_ref, dg_ref, db_ref, _, _ = _reference(x, gamma, beta, dy, 0.01)
    gd, bd = gamma.to(gpu), beta.to(gpu)
    shards = [x[:2].contiguous().to(gpu), x[2:].contiguous().to(gpu)]       # unequal shard sizes
    dys = [dy[:2].contiguous().to(gpu), dy[2:].contiguous().to(gpu)]
    c = 24
    # "rank" r's local statistics, as the all-reduce would deliver them to the other rank
    stat = []
    for xs in shards:
        s = torch.empty(2 * c + 1, dtype=torch.float64, device=gpu)
        ws = ops._workspace(gpu, ops.bn_workspace_bytes(c))
        ops.call("cl4_bn_sync_stats_local", ops.ptr(xs), xs.shape[0], c, 99, c, 0, ops.ptr(s), ops.ptr(ws), ws.numel(), ops.stream())
        s[2 * c] = float(xs.shape[0] * 99)
        stat.append(s)
    fwd = [ops.abn_train_forward(xs, gd, bd, extra_sums=stat[1 - r]) for r, xs in enumerate(shards)]
    assert _rel(torch.cat([f[0] for f in fwd]), y_ref) < 1e-5
    dgamma, dbeta = torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
    # first pass collects each shard's local backward sums, second pass applies with the combined sums
    local = []
    for r, xs in enumerate(shards):
        _, mean, var, count = fwd[r]
        _, ls = ops.abn_train_backward(xs, dys[r], mean, var, gd, bd, torch.zeros(c, device=gpu), torch.zeros(c, device=gpu), count,
                                       return_sums=True)
        local.append(ls)
    dxs = []
    for r, xs in enumerate(shards):
        _, mean, var, count = fwd[r]
        dxs.append(ops.abn_train_backward(xs, dys[r], mean, var, gd, bd, dgamma, dbeta, count, extra_sums=local[1 - r]))
    assert _rel(torch.cat(dxs), dx_ref) < 2e-5
    assert _rel(dgamma, dg_ref) < 2e-5 and _rel(dbeta, db_ref) < 2e-5       # sum over shards = full-batch gradient


def test_dropout_masks_of_successive_steps_are_independent(gpu):
    """The device-resident Dropout seed advances once per launch (graph-replay safe). Successive masks must be independent
    draws like nn.Dropout's — not the previous mask shifted by one element, which is what a seed advance equal to the
    per-element counter stride produced."""
    import torch
    from cl4wsis_amd import ops
    seed = torch.tensor([123456789], dtype=torch.int64, device=gpu)
    shape = (2, 256, 33, 33)
    m = [ops.dropout_mask_dev(shape, 0.5, seed, 1, gpu).flatten().float() for _ in range(3)]
    torch.cuda.synchronize()
    assert int(seed.item()) == 123456789 + 3
    for a in m:
        assert abs(a.mean().item() - 0.5) < 5e-3
    n = m[0].numel()
    for a, b in ((m[0], m[1]), (m[1], m[2]), (m[0], m[2])):
        for shift in (0, 1, 2, -1):
            x = a[max(shift, 0):n + min(shift, 0)]
            y = b[max(-shift, 0):n + min(-shift, 0)]
            agree = (x == y).float().mean().item()
            assert abs(agree - 0.5) < 5e-3, (shift, agree)       # an (anti)shifted copy would agree on 100 % / 0 %
    # same seed -> same mask (the captured graph and an eager run draw identical masks)
    s1 = torch.tensor([77], dtype=torch.int64, device=gpu)
    s2 = torch.tensor([77], dtype=torch.int64, device=gpu)
    assert torch.equal(ops.dropout_mask_dev(shape, 0.5, s1, 1, gpu), ops.dropout_mask_dev(shape, 0.5, s2, 1, gpu))


def test_inplace_abn_scales_with_abs_gamma_plus_eps(gpu):
    """InPlaceABN(Sync) — the norm every shipped configuration selects (`--norm_act iabn_sync`) — computes with |weight| + eps
    and returns d(weight) = sign(weight) * d(scale) (believed behaviour of the absent mapillary/inplace_abn package). A real
    `*_iabn_sync` checkpoint holds negative weights, so eval folding, train-mode forward and backward are checked with a
    mixed-sign weight vector against autograd of that formula; plain ABN keeps the signed weight."""
    import torch.nn as nn
    from cl4wsis_amd import body_train, nn as N, ops
    g = torch.Generator().manual_seed(5)
    c = 40
    conv = nn.Conv2d(24, c, 3, padding=1, bias=False).to(gpu)
    x = torch.randn(3, 24, 19, 17, generator=g)
    dy = torch.randn(3, c, 19, 17, generator=g)
    w = conv.weight.detach().cpu()

    def make(cls):
        bn = cls(c, activation="leaky_relu", activation_param=0.01).to(gpu)
        with torch.no_grad():
            bn.weight.copy_((torch.rand(c, generator=g) + 0.5) * torch.where(torch.arange(c) % 3 == 0, -1.0, 1.0))
            bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        return bn

    for cls, absg in ((N.InPlaceABNSync, True), (N.InPlaceABN, True), (N.ABN, False)):
        bn = make(cls)
        assert bn.abs_gamma == absg and (bn.weight < 0).any()
        gam = lambda t: (t.abs() + bn.eps) if absg else t
        # eval: folded into the conv epilogue
        with torch.no_grad():
            y = N.conv_abn(x.to(gpu), conv, bn)
        ref = F.leaky_relu(F.batch_norm(F.conv2d(x, w, padding=1), bn.running_mean.cpu(), bn.running_var.cpu(), gam(bn.weight.detach().cpu()),
                                        bn.bias.detach().cpu(), False, 0.1, bn.eps), 0.01)
        assert _rel(y, ref) < 2e-5
        # train: batch statistics, backward with the sign rule
        gr = bn.weight.detach().cpu().clone().requires_grad_(True)
        br = bn.bias.detach().cpu().clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        yr = F.leaky_relu(F.batch_norm(F.conv2d(x, wr, padding=1), torch.zeros(c), torch.ones(c), gam(gr), br, True, 0.1, bn.eps), 0.01)
        yr.backward(dy)
        saved = body_train.PRECISION, body_train.WGRAD_PRECISION
        body_train.PRECISION, body_train.WGRAD_PRECISION = 0, 0          # exact fp32 kernels: the comparison is about the formula
        try:
            grads = {}

            def grad_of(p):
                return grads.setdefault(id(p), torch.zeros_like(p))

            yt, rec = body_train.conv_abn_train(x.to(gpu), conv, bn)
            body_train.conv_abn_backward(rec, conv, bn, dy.to(gpu), grad_of, need_dx=False)
        finally:
            body_train.PRECISION, body_train.WGRAD_PRECISION = saved
        assert _rel(yt, yr.detach()) < 2e-5
        assert _rel(grads[id(bn.weight)], gr.grad) < 1e-4, cls.__name__
        assert _rel(grads[id(bn.bias)], br.grad) < 1e-4
        assert _rel(grads[id(conv.weight)], wr.grad) < 1e-4


def _decode_f16_planes(ps):
    """hi + lo of an fp16 two-plane tensor [n][c/8][hi, lo][h*w][8], still multiplied by its scale (float64 [n, c, h, w])."""
    n, c, h, w = ps.shape
    raw = ps.data.view(torch.float16).reshape(n, c // 8, 2, h * w, 8).double()
    return (raw[:, :, 0] + raw[:, :, 1]).permute(0, 1, 3, 2).reshape(n, c, h, w)


@pytest.mark.parametrize("magnitude", [1e-6, 1.0, 4e3])
@pytest.mark.parametrize("slope", [0.01, None])
def test_abn_backward_hands_the_gradient_on_as_fp16_planes(gpu, slope, magnitude):
    """r05 (cl4_bn_sync_bwd_local_max / _apply_planes): dx is bit-identical to the plain launch; its fp16 planes twin carries a
    power-of-two scale taken from a BOUND on |dx| (known before dx is), so nothing saturates, the bound is tight enough to keep the
    headroom of the data-derived scale (max |dx| s >= 2^12), and hi + lo reproduce dx to 2^-21 of the tensor's maximum."""
    from cl4wsis_amd import ops
    if not ops.planes_f16():
        pytest.skip("fp16 planes build only")
    g = torch.Generator().manual_seed(11)
    c = 72
    x = torch.randn(3, c, 19, 23, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    dy = torch.randn(x.shape, generator=g) * magnitude * torch.logspace(-1, 1, c)[None, :, None, None]
    act = ops.ACT_LEAKY if slope is not None else ops.ACT_NONE
    rm, rv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
    xd, gd, bd, dyd = x.to(gpu), gamma.to(gpu), beta.to(gpu), dy.to(gpu)
    _, mean, var, count = ops.abn_train_forward(xd, gd, bd, rm, rv, act=act, slope=slope or 0.0)
    out = []
    for twin in (False, True):
        dgamma, dbeta = torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
        dx = ops.abn_train_backward(xd, dyd, mean, var, gd, bd, dgamma, dbeta, count, act=act, slope=slope or 0.0, want_planes=twin)
        out.append((dx, dgamma, dbeta))
    (dx0, dg0, db0), (dx1, dg1, db1) = out
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert getattr(dx0, "_cl4_planes", None) is None
    ps = dx1._cl4_planes
    s = float(ps.slot[0])
    amax = float(dx1.abs().max())
    assert s > 0 and s == 2.0 ** round(torch.log2(torch.tensor(s)).item()), s
    assert 2.0 ** 12 <= amax * s < 65504.0, (amax, s, amax * s)
    dec = _decode_f16_planes(ps).cpu() / s
    err = (dec - dx1.double().cpu()).abs().max().item()
    print(f"ABN gradient twin, |dy| ~ {magnitude:g}: scale 2^{round(torch.log2(torch.tensor(s)).item())}, max |dx| s = {amax * s:.0f}, max error / max |dx| = {err / amax:.2e}")
    assert err <= 2.0 ** -21 * amax, (err, amax)


def test_input_gradient_convolution_reads_the_abn_twin(gpu):
    """The fp16x3 input-gradient convolution fed by the twin equals the one fed by a converted copy of the same fp32 gradient up to the
    two roundings' difference (both fp32-level), and channel-shrinking 1x1 gradients are routed to fp16x3 when the twin exists."""
    from cl4wsis_amd import ops
    if not ops.planes_f16():
        pytest.skip("fp16 planes build only")
    g = torch.Generator().manual_seed(12)
    cin, cout = 256, 1024                                    # conv3 of a res4 bottleneck: its input gradient maps 1024 -> 256 channels
    x = torch.randn(2, cout, 33, 33, generator=g)
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    dy = torch.randn(x.shape, generator=g) * 1e-4
    wt = torch.randn(cout, cin, 1, 1, generator=g) / 16.0
    rm, rv = torch.zeros(cout, device=gpu), torch.ones(cout, device=gpu)
    xd, gd, bd = x.to(gpu), gamma.to(gpu), beta.to(gpu)
    _, mean, var, count = ops.abn_train_forward(xd, gd, bd, rm, rv, act=ops.ACT_NONE)
    dz = ops.abn_train_backward(xd, dy.to(gpu), mean, var, gd, bd, torch.zeros(cout, device=gpu), torch.zeros(cout, device=gpu), count,
                                act=ops.ACT_NONE, want_planes=True)
    wd = ops.dgrad_weight(wt.to(gpu), 3, force_f16=True)
    assert wd.precision == 3 and ops.dgrad_weight(wt.to(gpu), 3).precision == 2
    before = ops._hip.load().cl4_conv_debug_b_launches()
    with_twin = ops.conv2d_dgrad(dz, wd, (33, 33), 1)
    assert ops._hip.load().cl4_conv_debug_b_launches() == before + 1
    plain = dz.clone()                                       # no twin: converted from the fp32 tensor with the data-derived scale
    converted = ops.conv2d_dgrad(plain, wd, (33, 33), 1)
    ref = F.conv_transpose2d(dz.double().cpu(), wt.double())
    e1 = ((with_twin.double().cpu() - ref).norm() / ref.norm()).item()
    e2 = ((converted.double().cpu() - ref).norm() / ref.norm()).item()
    print(f"input gradient 1024 -> 256 from the ABN twin {e1:.2e}, from a converted copy {e2:.2e}")
    assert e1 < 1e-6 and e2 < 1e-6


@pytest.mark.parametrize("twin", [False, True])
def test_one_rank_abn_equals_the_two_phase_pair_and_applies_the_sign_rule(gpu, monkeypatch, twin):
    """r05: with nobody to exchange sums with, forward statistics come straight from the reduction's partials (cl4_bn_train_stats) and the
    backward is cl4_bn_train_bwd_act (two launches, three with the planes twin); same numbers as the two-phase cl4_bn_sync_* pair. With
    `gamma_sign` (InPlaceABN: the kernels see |weight| + eps) dgamma receives sign(weight) * d(gamma), what cl4_sign_axpy did as a pass."""
    from cl4wsis_amd import ops
    if twin and not ops.planes_f16():
        pytest.skip("fp16 planes build only")
    g = torch.Generator().manual_seed(13)
    c = 64
    x = (torch.randn(3, c, 21, 17, generator=g) * 1.5 + 0.3).to(gpu)
    weight = torch.randn(c, generator=g).to(gpu)                       # negative entries: the sign rule matters
    gamma = ops.abs_eps(weight, 1e-5)
    beta = torch.randn(c, generator=g).to(gpu)
    dy = torch.randn(x.shape, generator=g).to(gpu)
    res = []
    for one_rank in (False, True):
        monkeypatch.setattr(ops, "ABN_ONE_RANK", one_rank)
        rm, rv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
        y, mean, var, count = ops.abn_train_forward(x, gamma, beta, rm, rv, act=ops.ACT_LEAKY, slope=0.01)
        dgamma, dbeta = torch.full((c,), 0.25, device=gpu), torch.full((c,), -0.5, device=gpu)      # accumulated into, not overwritten
        dx = ops.abn_train_backward(x, dy, mean, var, gamma, beta, dgamma, dbeta, count, act=ops.ACT_LEAKY, slope=0.01, want_planes=twin, gamma_sign=weight)
        res.append((y, mean, var, rm, rv, dx, dgamma, dbeta, getattr(dx, "_cl4_planes", None)))
    a, b = res
    for i, name in enumerate(("y", "mean", "var", "running_mean", "running_var", "dx")):
        assert torch.equal(a[i], b[i]), name
    # dgamma / dbeta: the one-rank kernels sum the 32 partials by a butterfly, the two-phase pair sequentially (float64 both): equal to 1e-6
    assert _rel(a[6], b[6]) < 1e-6 and _rel(a[7], b[7]) < 1e-6
    if twin:
        assert float(a[8].slot[0]) == float(b[8].slot[0]) and torch.equal(a[8].data, b[8].data)
    # the sign rule against the plain chain rule
    dg_plain, db_plain = torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
    ops.abn_train_backward(x, dy, a[1], a[2], gamma, beta, dg_plain, db_plain, float(x.numel() // c), act=ops.ACT_LEAKY, slope=0.01)
    expect = 0.25 + torch.where(weight > 0, dg_plain, -dg_plain)
    assert _rel(b[6], expect) < 1e-6 and _rel(b[7], db_plain - 0.5) < 1e-6
