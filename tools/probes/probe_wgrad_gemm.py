"""Pieces of the GEMM weight gradient (csrc/wgrad_gemm.hip) on the ASPP shape: x pack, dy pack, GEMM, scatter; and the role-split kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cl4wsis_amd import _hip, ops
from cl4wsis_amd.ops import call, ptr, stream
dev = torch.device("cuda:0")
n, cin, h, cout, dil = 16, 2048, 33, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = torch.randn(n, cin, h, h, device=dev); dy = torch.randn(n, cout, h, h, device=dev); dw = torch.zeros(cout, cin, 3, 3, device=dev)
lib = _hip.load()
kq = n * h * h; kpad = (kq + 31) // 32 * 32
xk = torch.empty(lib.cl4_kpack_bytes(cin, kq) // 4, dtype=torch.int32, device=dev)
dyk = torch.empty(lib.cl4_kpack_bytes(9 * cout, kq) // 4, dtype=torch.int32, device=dev)
y = torch.empty(cin, 9 * cout, device=dev)
def t(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * kq * cout * cin * 9
print(f"x pack     {t(lambda: call('cl4_kpack', ptr(x), ptr(xk), n, cin, 0, cin, h, h, h, h, 1, 1, 0, stream())):8.1f} us")
print(f"dy pack    {t(lambda: call('cl4_kpack', ptr(dy), ptr(dyk), n, cout, 0, cout, h, h, h, h, 3, dil, dil, stream())):8.1f} us")
us = t(lambda: call('cl4_gemm_kplanes', ptr(xk), ptr(dyk), ptr(y), cin, 9 * cout, kpad, stream()))
print(f"gemm       {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s")
print(f"scatter    {t(lambda: call('cl4_wgrad_scatter', ptr(y), ptr(dw), cin, cout, 9, stream())):8.1f} us")
ops.WGRAD_GEMM = False
us = t(lambda: ops.conv2d_wgrad(x, dy, dw, 3, pad=dil, dil=dil, precision=1))
print(f"role-split {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s")
ops.WGRAD_GEMM = True
us = t(lambda: ops.conv2d_wgrad(x, dy, dw, 3, pad=dil, dil=dil, precision=1, kcache={}))
print(f"gemm path  {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s (x packed every time)")
