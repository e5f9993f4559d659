"""bf16x6 kernel with compile-time knock-outs (CL4_LIB=build/probe/lib_dbgN.so: 1 no loads, 2 no convert/LDS stores, 4 no MFMA)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cl4wsis_amd import _hip, ops
dev = torch.device("cuda:0")
tag = (os.environ.get("CL4_LIB") or "product").split("/")[-1]
out = []
for (n, cin, h, cout, k, dil, plan) in ((16, 2048, 33, 256, 3, 6, (2, 64, 4)), (16, 2048, 33, 256, 3, 6, (2, 128, 4)),
                                         (30, 4096, 33, 256, 1, 1, (2, 64, 1)), (30, 4096, 33, 256, 1, 1, (2, 128, 1))):
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    pad = dil * (k - 1) // 2
    wp = ops.pack_conv_weight(w, 2)
    d = ops._conv_desc(x, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, 2)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    for _ in range(2):
        ops.conv2d(x, wp, k, pad=pad, dil=dil)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d(x, wp, k, pad=pad, dil=dil)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out.append(f"{ms:.3f}ms/{2.0 * n * h * h * cout * cin * k * k / ms / 1e9:4.0f}TF")
    _hip.call("cl4_conv2d_fwd_clear_plans")
print(f"{tag:14s} aspp3x3 t64s4 {out[0]} | t128s4 {out[1]} | K4096 t64 {out[2]} | t128 {out[3]}", flush=True)
