"""Calibration launches for the HBM-traffic counters (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/prof_pmc.sh):
the conv kernel itself on a shape whose minimal traffic is known and far larger than the 256 MB Infinity Cache — 1x1, n = 80,
1024 -> 128 channels at 33x33: ONE M-tile, so every activation byte is needed exactly once (357 MB in), 44.6 MB out, 0.5 MB of
weights. MI355X_MICROARCH.md states a x2 correction for FETCH_SIZE on 16-byte-per-lane streaming reads and calls other widths
uncalibrated; the two launches here (4-byte-per-lane gather with CL4_CONV_VEC=0, 16-byte-per-lane with =1) measure the factor
for THIS kernel's loader. The grid (z-dimension of the key) identifies them in the counter file: 697 workgroups."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import _hip, ops
dev = torch.device("cuda:0")
n, cin, h, cout = 80, 1024, 33, 128
x = torch.randn(n, cin, h, h, device=dev)
w = torch.randn(cout, cin, 1, 1, device=dev) * 0.02
wp = ops.pack_conv_weight(w, 1)
y = torch.empty(n, cout, h, h, device=dev)
d = ops._conv_desc(x, cin, 0, cout, 1, 1, 0, 1, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
_hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for _ in range(3):
    flush.zero_()                      # evict the Infinity Cache between launches
    ops.conv2d(x, wp, 1, out=y)
torch.cuda.synchronize()
print("calibration: input bytes", x.numel() * 4, "output bytes", y.numel() * 4, "weight bytes", cin * cout * 4, "vec", os.environ.get("CL4_CONV_VEC", "1"))
