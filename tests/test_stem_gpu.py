"""The fused stem (csrc/stem.hip: conv 7x7/2 + folded eval ABN + leaky-ReLU + max-pool 3x3/2, models/resnet.py:59-63) against a float64
torch evaluation and against the two-launch path it replaces; the planes twin must be the bf16 split of the fp32 output bit for bit."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, scale, shift, slope):
    y = F.conv2d(x.double(), w.double(), stride=2, padding=3)
    y = F.leaky_relu(y * scale.double()[None, :, None, None] + shift.double()[None, :, None, None], slope)
    return F.max_pool2d(y, 3, stride=2, padding=1).float()


@pytest.mark.parametrize("shape", [(2, 65, 65), (1, 97, 129), (3, 33, 21), (2, 513, 513), (1, 64, 70), (1, 7, 9)])
def test_stem_fused_matches_float64_and_the_two_launch_path(gpu, shape):
    from cl4wsis_amd import ops
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = torch.randn(n, 3, h, w, generator=g)
    x[0, :, : min(h, 5), : min(w, 5)] = 3.0                     # a flat corner: ties inside pooling windows
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    scale = torch.rand(64, generator=g) + 0.5
    scale[::7] *= -1.0                                          # negative scales: the maximum is taken AFTER the affine map
    shift = torch.randn(64, generator=g)
    ref = _ref(x, wt, scale, shift, 0.01)
    xd, wd, sd, hd = x.to(gpu), wt.to(gpu), scale.to(gpu), shift.to(gpu)
    y, ys = ops.stem_fused(xd, ops.pack_stem_weight(wd), sd, hd, slope=0.01)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape)
    err = ((y.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err
    from util import planes_split_ref
    hi, lo = planes_split_ref(y.cpu())                          # the build's two-plane split of the fp32 output, bit for bit
    assert torch.equal(ops.from_planes(ys).cpu(), hi + lo)
    # the path it replaces: bf16x3 conv + ABN epilogue, then the max-pool kernel
    y2 = ops.maxpool3x3s2(ops.conv2d(xd, ops.pack_conv_weight(wd, 1), 7, stride=2, pad=3, scale=sd, shift=hd, act1=ops.ACT_LEAKY, slope1=0.01))
    assert ((y - y2).abs().max() / ref.abs().max()).item() < 2e-5
    only_planes = ops.stem_fused(xd, ops.pack_stem_weight(wd), sd, hd, slope=0.01, want_fp32=False)
    assert only_planes[0] is None and torch.equal(only_planes[1].data, ys.data)


def test_resnet_body_uses_the_fused_stem(gpu, monkeypatch):
    """The frozen ResNet body with and without the fused stem: same five stage outputs (bf16x3 level)."""
    from cl4wsis_amd import backbone
    torch.manual_seed(3)
    body = backbone.net_resnet50(keep_outputs=True).to(gpu).eval()
    for m in body.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
    x = torch.randn(2, 3, 97, 97, device=gpu)
    with torch.no_grad():
        monkeypatch.setattr(backbone, "STEM_FUSED", True)
        a = [t.clone() for t in body(x)]
        monkeypatch.setattr(backbone, "STEM_FUSED", False)
        b = body(x)
    for u, v in zip(a, b):
        assert tuple(u.shape) == tuple(v.shape)
        assert ((u - v).abs().max() / v.abs().max()).item() < 5e-5


def test_inplace_trunk_gives_identical_stage_outputs(gpu, monkeypatch):
    """Identity bottleneck blocks overwrite their residual operand (backbone.INPLACE_TRUNK): same kernels, same operands, so the five
    stage outputs must be bit-identical to the out-of-place run — also in iteration 0, when the launch-plan tuner re-runs every
    convolution (it must not touch the residual it would then read again)."""
    from cl4wsis_amd import backbone, ops
    torch.manual_seed(4)
    body = backbone.net_resnet50(keep_outputs=True).to(gpu).eval()
    for m in body.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
    x = torch.randn(3, 3, 129, 129, device=gpu)
    with torch.no_grad():
        monkeypatch.setattr(backbone, "INPLACE_TRUNK", False)
        ref = [t.clone() for t in body(x)]
        monkeypatch.setattr(backbone, "INPLACE_TRUNK", True)
        monkeypatch.setattr(ops, "AUTOTUNE", True)
        tuned = body(x)                                 # tuner active: plans may change, the operands must not
        monkeypatch.setattr(ops, "AUTOTUNE", False)
        again = body(x)
    for r, t, a in zip(ref, tuned, again):
        assert ((r - t).abs().max() / r.abs().max()).item() < 1e-4      # another launch plan sums in another order (50 layers deep)
        assert torch.equal(t, a)
        assert planes_equal(t, a)


def planes_equal(t, a):
    from cl4wsis_amd.nn import planes_of
    pt, pa = planes_of(t), planes_of(a)
    return (pt is None and pa is None) or torch.equal(pt.data, pa.data)
