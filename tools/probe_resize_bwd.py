"""Times cl4_resize_bilinear_bwd on the shapes of the phase-2 bench step (b = 16), and checks the channel-slice + accumulate mode of the
row-band kernel against the plain gather (CL4_RESIZE_BWD_ROWS=0 in a second process gives the old kernel's times)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops

dev = torch.device("cuda", 0)
shapes = [(16, 20, 129, 513, True), (16, 2, 129, 513, True), (16, 128, 65, 129, True), (16, 256, 33, 65, True), (16, 128, 33, 65, True), (16, 64, 65, 129, True)]
for n, c, hin, hout, align in shapes:
    dy = torch.randn(n, c, hout, hout, device=dev)
    dx = ops.resize_bilinear_bwd(dy, (hin, hin), align)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.resize_bilinear_bwd(dy, (hin, hin), align, dx=dx, accumulate=False)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = dy.numel() * 4 / 1e9
    print(f"n {n} c {c} {hin}->{hout}: {ms * 1e3:8.1f} us  ({gb / (ms * 1e-3):7.1f} GB/s of dy)  checksum {dx.double().sum().item():.6e}")
# slice + accumulate: channels 3..7 of a 9-channel dy into channels 2..6 of an 8-channel dx that already holds values
dy = torch.randn(2, 9, 131, 77, device=dev)
base = torch.randn(2, 8, 33, 20, device=dev)
out = base.clone()
ops.resize_bilinear_bwd(dy, (33, 20), True, dy_coff=3, c=5, dx=out, dx_coff=2, accumulate=True)
ref = ops.resize_bilinear_bwd(dy[:, 3:8].contiguous(), (33, 20), True)
exp = base.clone()
exp[:, 2:7] += ref
print("slice + accumulate max abs diff", (out - exp).abs().max().item())
assert torch.allclose(out, exp, rtol=1e-5, atol=1e-5)
