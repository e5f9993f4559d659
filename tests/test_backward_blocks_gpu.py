"""GPU parity of the backward building blocks (csrc/backward.hip + the conv kernels run as input-gradient convolutions)
against PyTorch-CPU autograd: conv input gradients for the stride / padding / dilation combinations of ResNet101 +
DeepLabV3 + PseudoLabeler, max-pool backward including ties, residual add + leaky-ReLU and its derivative."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("case", [
    (2, 24, 17, 17, 40, 1, 1, 0, 1),     # 1x1
    (2, 24, 17, 17, 40, 1, 2, 0, 1),     # 1x1 stride 2 (proj_conv of the first block of res3 / res4)
    (2, 16, 18, 18, 32, 1, 2, 0, 1),     # even input size
    (2, 24, 17, 19, 24, 3, 1, 1, 1),     # 3x3
    (2, 24, 17, 17, 24, 3, 2, 1, 1),     # 3x3 stride 2
    (1, 16, 18, 20, 24, 3, 2, 1, 1),     # 3x3 stride 2, even sizes
    (1, 16, 21, 21, 24, 3, 1, 2, 2),     # res5 dilation 2
    (1, 32, 33, 33, 16, 3, 1, 12, 12),   # ASPP dilation 12
    (1, 32, 33, 33, 16, 3, 1, 18, 18),   # dilation 18: most taps fall outside
])
def test_conv_input_gradient(gpu, case):
    from cl4wsis_amd import ops
    n, cin, h, w, cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64) / (cin * k * k) ** 0.5
    y = F.conv2d(x, wt, stride=s, padding=p, dilation=d)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    wd = ops.dgrad_weight(wt.float().to(gpu))
    dx = ops.conv2d_dgrad(dy.float().to(gpu), wd, (h, w), k, stride=s, pad=p, dil=d)
    torch.cuda.synchronize()
    assert dx.shape == x.shape
    assert _rel(dx, x.grad.float()) < 1e-5


@pytest.mark.parametrize("shape", [(2, 3, 257, 257), (1, 5, 7, 9), (3, 2, 8, 6), (1, 1, 2, 2), (2, 4, 65, 33), (1, 2, 130, 300)])
def test_maxpool_forward(gpu, shape):
    """nn.MaxPool2d(3, stride=2, padding=1) (models/resnet.py:63) on odd / even sizes, widths beyond one workgroup pass, all-negative
    inputs (the implicit padding is -inf, not 0); bit-exact."""
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) - 3.0
    ref = F.max_pool2d(x, 3, stride=2, padding=1)
    out = ops.maxpool3x3s2(x.to(gpu))
    assert tuple(out.shape) == tuple(ref.shape)
    assert torch.equal(out.cpu(), ref)


def test_maxpool_backward_with_ties(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(3)
    for shape in [(2, 5, 17, 17), (1, 3, 18, 20), (1, 2, 5, 4), (1, 2, 65, 67), (2, 1, 48, 33)]:       # (several 16-row tiles of the r05 kernel)
        x = torch.randint(0, 4, shape, generator=g).float()     # many equal values inside every window
        x[0, 0, :2] = float("-inf")
        xr = x.clone().requires_grad_(True)
        y = F.max_pool2d(xr, 3, 2, 1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        dx = ops.maxpool3x3s2_bwd(x.to(gpu), dy.to(gpu))
        assert torch.equal(dx.cpu(), xr.grad)


def test_add_act_and_derivative(gpu):
    from cl4wsis_amd import ops
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(3, 7, 9, 5, generator=g), torch.randn(3, 7, 9, 5, generator=g)
    dy = torch.randn(a.shape, generator=g)
    for act, slope, fn in ((ops.ACT_LEAKY, 0.01, lambda t: F.leaky_relu(t, 0.01)), (ops.ACT_RELU, 0.0, F.relu),
                           (ops.ACT_NONE, 0.0, lambda t: t)):
        ar = a.clone().requires_grad_(True)
        y = fn(ar + b)
        y.backward(dy)
        yd = ops.add_act(a.to(gpu), b.to(gpu), act, slope)
        assert torch.equal(yd.cpu(), y.detach())
        assert torch.equal(ops.act_bwd(yd, dy.to(gpu), act, slope).cpu(), ar.grad)


@pytest.mark.parametrize("case", [
    (5, 5, 65, 65, False),       # step 0 / validation: logits to image size, factor 13
    (33, 33, 513, 513, False),   # the full-size case, factor 15.5
    (9, 9, 17, 17, True),        # decoder upsampling (exact 2x: the dedicated gather)
    (33, 33, 65, 65, True),
    (65, 65, 129, 129, True),
    (3, 5, 5, 9, True),
    (7, 9, 33, 21, True),
    (129, 129, 513, 513, True),  # the centre / offset maps of the phase-2 step (row-band kernel, three column passes, 9 bands)
    (129, 129, 513, 513, False),
    (17, 40, 35, 300, True),     # bands and column passes that do not divide
    (16, 16, 17, 16, False),     # one axis unchanged
    (1, 1, 33, 33, True),        # ASPP image pooling broadcast back over the map (plane-sum kernel)
    (1, 1, 7, 5, False),
    (17, 13, 9, 7, False),       # downsampling
    (4, 4, 4, 4, False),         # identity
])
def test_resize_bilinear_backward(gpu, case):
    from cl4wsis_amd import ops
    hin, win, hout, wout, align = case
    g = torch.Generator().manual_seed(hin * 7 + hout)
    x = torch.randn(2, 3, hin, win, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.interpolate(x, size=(hout, wout), mode="bilinear", align_corners=align)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dx = ops.resize_bilinear_bwd(dy.float().to(gpu), (hin, win), align)
    assert _rel(dx, x.grad.float()) < 1e-5
    yd = ops.resize_bilinear(x.detach().float().to(gpu), (hout, wout), align)
    assert _rel(yd, y.detach().float()) < 1e-5


@pytest.mark.parametrize("shape", [(2, 5, 129, 129), (1, 3, 7, 9), (3, 2, 33, 61), (1, 4, 64, 120), (2, 1, 1, 1)])
def test_dwconv5_forward_and_input_gradient(gpu, shape):
    """5x5 depthwise conv (streaming strip kernels: 60-column strips, row chunks, halo lanes) against torch: forward, the
    flipped-kernel input gradient and the weight gradient, on sizes that are / are not multiples of the strip and chunk sizes."""
    import torch.nn.functional as F
    from cl4wsis_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h * 131 + w)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, 1, 5, 5, generator=g)
    ref = F.conv2d(x.double(), wt.double(), padding=2, groups=c)
    y = ops.dwconv5(x.to(gpu), wt.to(gpu))
    assert ((y.double().cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-6
    dy = torch.randn(n, c, h, w, generator=g)
    xr = x.double().clone().requires_grad_(True)
    F.conv2d(xr, wt.double(), padding=2, groups=c).backward(dy.double())
    dx = ops.dwconv5(dy.to(gpu), wt.to(gpu), flip=True)
    assert ((dx.double().cpu() - xr.grad).abs().max() / xr.grad.abs().max()).item() < 1e-6
    wr = wt.double().clone().requires_grad_(True)
    F.conv2d(x.double(), wr, padding=2, groups=c).backward(dy.double())
    dw = ops.dwconv5_wgrad_(torch.ones(c, 1, 5, 5, device=gpu), x.to(gpu), dy.to(gpu))      # accumulates into dw
    assert ((dw.double().cpu() - 1 - wr.grad).abs().max() / wr.grad.abs().max()).item() < 2e-5
