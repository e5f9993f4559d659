"""The two poolings of DeepLabV3's image branch (modules/deeplab.py:84-92: F.avg_pool2d(stride 1) at eval, the global mean in training)
against a float64 torch evaluation — both kernels of each (one wave per plane for maps of a few thousand pixels and at most four output
windows, one workgroup per plane otherwise)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [(3, 5, 33, 33, 32, 32), (2, 7, 33, 33, 33, 33), (1, 9, 17, 21, 16, 21), (2, 3, 33, 33, 30, 30),
                                  (1, 2, 65, 65, 64, 64), (1, 3, 100, 100, 99, 99), (5, 1, 4, 4, 3, 3)])
def test_avgpool_stride1(gpu, case):
    from cl4wsis_amd import ops
    n, c, h, w, kh, kw = case
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(h * 100 + kh)) + 0.3
    ref = F.avg_pool2d(x.double(), (kh, kw), stride=1).float()
    y = ops.avgpool_s1(x.to(gpu), kh, kw)
    assert tuple(y.shape) == tuple(ref.shape)
    assert (y.cpu() - ref).abs().max().item() < 2e-6


@pytest.mark.parametrize("case", [(3, 5, 33, 33), (1, 70, 1, 1), (2, 3, 90, 91), (1, 2, 129, 129), (4, 1, 7, 3)])
def test_global_mean(gpu, case):
    from cl4wsis_amd import ops
    n, c, h, w = case
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(h * 100 + w)) + 0.3
    ref = x.double().mean(dim=(2, 3), keepdim=True).float()
    y = ops.global_mean(x.to(gpu))
    assert tuple(y.shape) == tuple(ref.shape)
    assert (y.cpu() - ref).abs().max().item() < 2e-6
