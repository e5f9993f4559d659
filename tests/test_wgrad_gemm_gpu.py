"""The 3x3 weight gradient as one K-major GEMM on the LDS-DMA planes kernel (csrc/wgrad_gemm.hip: cl4_kpack x 2, cl4_gemm_kplanes,
cl4_wgrad_scatter; autograd of nn.Conv2d, reference train.py:537) against a float64 autograd evaluation and against the role-split
weight-gradient kernel it replaces for the wide layers: channel slices of both operands, dilations that push most taps outside the map,
output maps smaller than the input, pixel counts that are not a multiple of the 32-deep K-tile, accumulation into a non-zero dw, and the
shared pack of x."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel_err(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


# (n, in_ctot, in_coff, cin, h, w, dy_ctot, dy_coff, cout, pad, dil)
CASES = [
    (3, 128, 0, 128, 19, 21, 32, 0, 32, 2, 2),          # kq = 1197: a ragged last K-tile
    (2, 160, 16, 128, 33, 33, 80, 16, 48, 18, 18),      # channel slices; dilation 18 on 33 x 33: most taps outside
    (2, 256, 0, 256, 17, 15, 64, 0, 64, 0, 1),          # pad 0: the output map is smaller than the input
    (1, 128, 0, 128, 9, 9, 16, 0, 16, 1, 1),            # a single tile, 81 pixels
    (4, 384, 0, 384, 33, 33, 128, 0, 128, 6, 6),        # more than one tile in both directions (3 x 8)
]


@pytest.mark.parametrize("case", CASES)
def test_wgrad_gemm_matches_autograd_and_the_role_split_kernel(gpu, case, monkeypatch):
    from cl4wsis_amd import ops
    n, in_ctot, in_coff, cin, h, w, dy_ctot, dy_coff, cout, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, in_ctot, h, w, generator=g, dtype=torch.float64)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x[:, in_coff:in_coff + cin], wt, padding=pad, dilation=dil)
    dy = torch.randn(n, dy_ctot, y.shape[2], y.shape[3], generator=g, dtype=torch.float64)
    y.backward(dy[:, dy_coff:dy_coff + cout])
    ref = wt.grad.float()
    xd, dyd = x.float().to(gpu), dy.float().to(gpu)
    kw = dict(pad=pad, dil=dil, in_coff=in_coff, cin=cin, dy_coff=dy_coff, cout=cout, precision=1)
    monkeypatch.setattr(ops, "_wgrad_gemm_ok", lambda *a: False)
    dw_old = torch.full((cout, cin, 3, 3), 0.5, device=gpu)
    ops.conv2d_wgrad(xd, dyd, dw_old, 3, **kw)
    monkeypatch.setattr(ops, "_wgrad_gemm_ok", lambda *a: True)
    cache = {}
    dw = torch.full((cout, cin, 3, 3), 0.5, device=gpu)          # accumulated into
    ops.conv2d_wgrad(xd, dyd, dw, 3, kcache=cache, **kw)
    dw2 = torch.zeros(cout, cin, 3, 3, device=gpu)
    ops.conv2d_wgrad(xd, dyd, dw2, 3, kcache=cache, **kw)         # x's pack comes from the cache the second time
    torch.cuda.synchronize()
    assert len(cache) == 1
    assert _rel_err(dw.cpu() - 0.5, ref) < 1e-5, _rel_err(dw.cpu() - 0.5, ref)
    assert _rel_err(dw2.cpu(), ref) < 1e-5
    assert _rel_err(dw.cpu() - 0.5, ref) < 3 * _rel_err(dw_old.cpu() - 0.5, ref) + 1e-6      # no further from float64 than the kernel it replaces
    dw3 = torch.zeros(cout, cin, 3, 3, device=gpu)
    ops.conv2d_wgrad(xd, dyd, dw3, 3, **kw)                       # no cache: packs x itself; no atomics anywhere, so bit-identical
    assert torch.equal(dw3, dw2)


def test_wgrad_gemm_is_picked_for_the_aspp_shape_only(gpu):
    from cl4wsis_amd import ops
    assert ops._wgrad_gemm_ok(16, 2048, 33, 33, 256, 3, 1, 1)
    assert not ops._wgrad_gemm_ok(16, 2048, 33, 33, 256, 1, 1, 1)       # 1x1
    assert not ops._wgrad_gemm_ok(16, 2048, 33, 33, 256, 3, 1, 0)       # exact-fp32 arithmetic (step 0 / phase 1)
    assert not ops._wgrad_gemm_ok(16, 256, 33, 33, 256, 3, 1, 1)        # too few rows to fill the chip
    assert not ops._wgrad_gemm_ok(2, 2048, 33, 33, 256, 3, 1, 1)        # short K
