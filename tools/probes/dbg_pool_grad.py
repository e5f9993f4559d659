"""Attribution of the one gradient tensor that sits 18x further from float64 than the CPU-fp32 oracle in the well-conditioned phase-2
case at exact-fp32 arithmetic (tests/test_acceptance_gpu.py, precision 0: decoder.instance_decoder.aspp.convs.4.aspp_pooling.1.weight,
1.45e-2 vs 7.8e-4): the ASPP image-pooling branch receives, per image, the SUM over all pixels of the gradient of its 256 broadcast
channels. That gradient comes out of the train-mode BatchNorm behind aspp.project, whose backward removes the batch mean — the sum
over images AND pixels is exactly zero, the per-image sums are what is left of a cancellation. This script captures the HIP path's own
operands of that reduction and redoes it in float64 on the host: if the float64 redo agrees with the HIP kernels, the kernels are
exact for their inputs and the deviation is element noise of the incoming gradient map amplified by the cancellation factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import QuietLogger, p2_opts
from cl4wsis_amd import heads, ops
from cl4wsis_amd.train import Trainer
from oracle import weights

ops.CONV_PRECISION = 0
heads.TRAIN_PRECISION = 0
ops.WGRAD_PRECISION = 0
CROP = 193
gpu = torch.device("cuda:0")
opts = p2_opts(CROP)
opts.max_iters, opts.start_decay = 1000, 1000
torch.manual_seed(0)
tr = Trainer(QuietLogger(), gpu, opts)
for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
    m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=71 + i))
tr._realias()
g = torch.Generator().manual_seed(7)
x = torch.randn(4, 3, CROP, CROP, generator=g)
l1h = torch.zeros(4, 20)
for i, cs in enumerate(((15,), (16, 19), (17,), (18, 15))):
    l1h[i, list(cs)] = 1
keep = torch.rand(4, 256, 13, 13, generator=g) >= 0.5

cap = {}
orig_rb, orig_relu, orig_wg = ops.resize_bilinear_bwd, ops.relu_bwd, ops.conv2d_wgrad


def rb(dy, in_size, align, dy_coff=0, c=None, **kw):
    out = orig_rb(dy, in_size, align, dy_coff=dy_coff, c=c, **kw)
    if tuple(in_size) == (1, 1):
        cap["dcat5_slice"] = dy[:, dy_coff:dy_coff + c].detach().clone()
        cap["pooled_sum_hip"] = out.detach().clone()
    return out


def relu(y, dy):
    out = orig_relu(y, dy)
    if dy.shape[-2:] == (1, 1):
        cap["gp"], cap["dgp"] = y.detach().clone(), out.detach().clone()
    return out


def wg(xx, dy, dw, k, **kw):
    if xx.shape[-2:] == (1, 1) and k == 1:
        cap["g"] = xx.detach().clone()
        before = dw.detach().clone()
        out = orig_wg(xx, dy, dw, k, **kw)
        cap["dW_hip"] = (dw.detach() - before).clone()
        return out
    return orig_wg(xx, dy, dw, k, **kw)


ops.resize_bilinear_bwd, ops.relu_bwd, ops.conv2d_wgrad = rb, relu, wg
tr.step((x, l1h), dropout_keep=keep.to(torch.uint8).to(gpu))
torch.cuda.synchronize()
d = cap["dcat5_slice"].double().cpu()                     # [n, 256, h, w]
s64 = d.sum((2, 3))
s_hip = cap["pooled_sum_hip"].double().cpu().reshape(s64.shape)
canc = (d.abs().sum((2, 3)) / s64.abs().clamp_min(1e-300))
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"per-image pixel sums of the broadcast channels' gradient: HIP kernel vs float64 redo of ITS OWN input: rel-L2 {rel(s_hip, s64):.2e}")
print(f"cancellation factor sum|d| / |sum d| per (image, channel): median {float(canc.median()):.1f}, 90th pct {float(canc.flatten().kthvalue(int(0.9 * canc.numel())).values):.1f}, max {float(canc.max()):.1f}")
tot = d.sum((0, 2, 3))
print(f"sum over images AND pixels (exactly 0 behind a train-mode BN): max |.| {float(tot.abs().max()):.3e} against per-image sums of typical size {float(s64.abs().median()):.3e}")
gp = cap["gp"].double().cpu().reshape(s64.shape)
dgp64 = s64 * (gp > 0)
dW64 = dgp64.t() @ cap["g"].double().cpu().reshape(4, -1)                # [256, 2048]
dW_hip = cap["dW_hip"].double().cpu().reshape(dW64.shape)
print(f"weight gradient of the pooling branch: HIP kernels vs float64 redo from the HIP path's own dcat5: rel-L2 {rel(dW_hip, dW64):.2e}")
print("=> if both rel-L2 figures are ~1e-6, the reduction and the weight-gradient kernel are exact for their inputs; the 1.45e-2 against the "
      "float64 ORACLE is the element-level noise of dcat5 (1e-6 .. 1e-5 relative per element after BN backward + the 1x1 input-gradient conv) "
      "times the cancellation factor above.")
