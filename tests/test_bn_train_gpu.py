"""Train-mode nn.BatchNorm2d + ReLU (+ Dropout) of the instance decoder / heads (modules/panoptic_deeplab.py:26-28,44-46,74-76): the
two-launch forward (cl4_bn_train_fwd: statistics finalised inside the elementwise pass) and the backward (cl4_bn_train_bwd: channel
sums taken from the partials inside the elementwise pass) against float64 autograd; channel slices on both sides, a Dropout mask,
running statistics, accumulation into dgamma / dbeta; and against the three-launch forms they replace."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [(2, 24, 9, 11, 0, 24, False), (3, 40, 33, 33, 8, 16, True), (16, 8, 5, 5, 0, 8, False), (1, 64, 65, 67, 32, 32, True)])
def test_bn_train_forward_backward_fused_final(gpu, case):
    from cl4wsis_amd import ops
    n, ctot, h, w, coff, c, drop = case
    g = torch.Generator().manual_seed(n * 100 + ctot)
    x = torch.randn(n, ctot, h, w, generator=g, dtype=torch.float64) * 2 + 0.5
    gamma = torch.rand(c, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(c, generator=g, dtype=torch.float64) * 0.3
    keep = (torch.rand(n, c, h, w, generator=g) >= 0.3).to(torch.uint8) if drop else None
    ks = 1.0 / 0.7 if drop else 1.0
    dy = torch.randn(n, c + 8, h, w, generator=g, dtype=torch.float64)
    rm0, rv0 = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    # float64 reference through autograd
    xr = x[:, coff:coff + c].clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = F.relu(F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5))
    if drop:
        yr = yr * keep.double() * ks
    yr.backward(dy[:, 4:4 + c])
    xd = x.float().to(gpu)
    rmd, rvd = rm0.float().to(gpu), rv0.float().to(gpu)
    kd = keep.to(gpu) if drop else None
    out = torch.full((n, c + 5, h, w), 7.0, device=gpu)
    y, mean, var = ops.bn_train_fwd(xd, gamma.float().to(gpu), beta.float().to(gpu), rmd, rvd, 0.1, 1e-5, ops.ACT_RELU, keep=kd, keep_scale=ks,
                                    c=c, x_coff=coff, out=out, out_coff=3)
    torch.cuda.synchronize()
    assert (out[:, :3] == 7).all() and (out[:, 3 + c:] == 7).all()
    assert (out[:, 3:3 + c].cpu().double() - yr.detach()).abs().max().item() < 2e-5
    assert (rmd.cpu().double() - rm).abs().max().item() < 1e-6 and (rvd.cpu().double() - rv).abs().max().item() < 1e-5
    bm = xr.detach().mean(dim=(0, 2, 3))
    assert (mean.cpu().double() - bm).abs().max().item() < 1e-6
    # the three-launch forward gives the same statistics and output
    m2, v2 = ops.bn_train_stats(xd, c=c, coff=coff)
    y2 = ops.bn_apply(xd, m2, v2, gamma.float().to(gpu), beta.float().to(gpu), 1e-5, ops.ACT_RELU, keep=kd, keep_scale=ks, c=c, x_coff=coff)
    assert torch.allclose(m2, mean, rtol=0, atol=1e-7) and torch.allclose(v2, var, rtol=1e-6, atol=1e-7)
    assert (y2 - out[:, 3:3 + c]).abs().max().item() < 1e-5
    # backward (fused final), accumulated into non-zero dgamma / dbeta
    dg, db = torch.full((c,), 0.25, device=gpu), torch.full((c,), -0.5, device=gpu)
    dx = ops.bn_train_bwd(xd, dy.float().to(gpu), mean, var, gamma.float().to(gpu), beta.float().to(gpu), dg, db, 1e-5, ops.ACT_RELU, keep=kd,
                          keep_scale=ks, c=c, x_coff=coff, dy_coff=4)
    torch.cuda.synchronize()
    ref_dx = xr.grad
    err = (dx[:, coff:coff + c].cpu().double() - ref_dx).abs().max().item() / ref_dx.abs().max().item()
    assert err < 2e-5, err
    assert ((dg.cpu().double() - 0.25) - gr.grad).abs().max().item() < 2e-4 * max(1.0, gr.grad.abs().max().item())
    assert ((db.cpu().double() + 0.5) - br.grad).abs().max().item() < 2e-4 * max(1.0, br.grad.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 24, 33, 33), (3, 8, 129, 65), (16, 5, 9, 7), (1, 40, 130, 190)])
def test_dwconv5_leaves_the_partial_sums_batchnorm_needs(gpu, shape):
    """cl4_dwconv5_fwd_stats + cl4_bn_train_fwd_partials (the depthwise 5x5 -> train-mode BatchNorm -> ReLU pairs of the trainable
    decoder / heads, modules/panoptic_deeplab.py:26-28,44-46) against the separate kernels: same convolution bit for bit, same statistics."""
    from cl4wsis_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(n, c, h, w, generator=g) * 1.5 + 0.2).to(gpu)
    wt = (torch.randn(c, 1, 5, 5, generator=g) * 0.2).to(gpu)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(gpu), (torch.randn(c, generator=g) * 0.1).to(gpu)
    z_ref = ops.dwconv5(x, wt)
    z, part, nparts = ops.dwconv5_stats(x, wt)
    assert torch.equal(z, z_ref)
    rm, rv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
    y, mean, var = ops.bn_train_fwd(z, gamma, beta, rm, rv, 0.1, 1e-5, ops.ACT_RELU, partials=(part, nparts))
    rm2, rv2 = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
    y2, mean2, var2 = ops.bn_train_fwd(z_ref, gamma, beta, rm2, rv2, 0.1, 1e-5, ops.ACT_RELU)
    torch.cuda.synchronize()
    zd = z_ref.double()
    assert (mean.double() - zd.mean(dim=(0, 2, 3))).abs().max().item() < 1e-6
    assert (var.double() - zd.var(dim=(0, 2, 3), unbiased=False)).abs().max().item() < 1e-5 * max(1.0, var.max().item())
    assert torch.allclose(mean, mean2, rtol=0, atol=1e-7) and torch.allclose(var, var2, rtol=1e-6, atol=1e-7)
    assert torch.allclose(rm, rm2, rtol=0, atol=1e-7) and torch.allclose(rv, rv2, rtol=1e-6, atol=1e-7)
    assert (y - y2).abs().max().item() < 1e-5
