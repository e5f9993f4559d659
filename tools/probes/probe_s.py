"""Planes-input kernel (conv_s.hip) against the fp32-input bf16x3 kernels on the bench step's dense shapes, under SUSTAINED load
(each shape is launched back to back for ~1 s; the last 100 launches are timed), with the in-kernel clock and cycles per K-tile from
one stamped launch. CL4_CONV_S_MFMA=16|32 selects the MFMA shape of the planes kernel. usage: probe_s.py [planes|fp32] [n ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cl4wsis_amd import _hip, ops

dev = torch.device("cuda:0")


def run(kind, name, n, cin, h, cout, k, stride, pad, dil, post, plan=(2, 128, 1), planes_out=False):
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, 1)
    ho = ops.conv_out_size(h, k, stride, pad, dil)
    res = torch.randn(n, cout, ho, ho, device=dev) if post else None
    sc = (torch.rand(cout, device=dev) + 0.5) if os.environ.get("CL4_PROBE_AFFINE", "1") != "0" else None     # folded ABN of the real layers
    sh = torch.randn(cout, device=dev) if sc is not None else None
    if kind == "planes":
        xs = ops.to_planes(x)
        resd = ops.to_planes(res) if (post and planes_out) else res
        y = None if planes_out else torch.empty(n, cout, ho, ho, device=dev)
        yp = ops.Planes(n, cout, ho, ho, dev) if planes_out else None
        d = ops.ConvDescFor(n, cin, h, h, cin, 0, cout, k, stride, pad, dil, cout, 0, 0, 0.0, 0, 0.0, 3)
        go = lambda: ops.conv2d_planes(xs, wp, k, stride=stride, pad=pad, dil=dil, scale=sc, shift=sh, act1=ops.ACT_LEAKY, post=resd, out=y, want_fp32=not planes_out, out_planes=yp)
    else:
        y = torch.empty(n, cout, ho, ho, device=dev)
        d = ops._conv_desc(x, cin, 0, cout, k, stride, pad, dil, cout, 0, 0, 0.0, 0, 0.0, 1)
        go = lambda: ops.conv2d(x, wp, k, stride=stride, pad=pad, dil=dil, scale=sc, shift=sh, act1=ops.ACT_LEAKY, post=res, out=y)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    ntile = ((cout + 127) // 128) * ((n * ((ho * ho + 3) & ~3) + 127) // 128)
    tl = torch.zeros(ntile * 8 * max(1, abs(plan[2])) + 64, dtype=torch.int64, device=dev)
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.time() - t0 < 1.0:
        e0.record()
        for _ in range(100):
            go()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10.0
    _hip.call("cl4_conv_debug_timeline", C.c_void_p(tl.data_ptr()))
    go()
    torch.cuda.synchronize()
    _hip.call("cl4_conv_debug_timeline", None)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    t = tl.cpu().numpy()[:-64].reshape(-1, 8)
    t = t[(t[:, 4] > 0) & (t[:, 7] > 0) & (t[:, 5] > t[:, 7])]
    tf = 2.0 * n * ho * ho * cout * cin * k * k / us / 1e6
    msg = f"{kind:6s} {name:26s} n={n:3d} {us:7.1f} us {tf:5.0f} TF/s"
    if len(t):
        cyc = (t[:, 4] - t[:, 0]).astype(np.float64)
        ghz = np.median(cyc / (t[:, 5] - t[:, 7]).astype(np.float64) * 0.1)
        nk = cin * k * k // 32
        kl = np.median((t[:, 2] - t[:, 1]).astype(np.float64))
        msg += (f" | clock {ghz:.2f} GHz, prologue {np.median(t[:, 1] - t[:, 0]):6.0f}  K loop {kl:8.0f} = {kl / nk:6.0f} / K-tile"
                f"  park {np.median(t[:, 3] - t[:, 2]):6.0f}  epilogue {np.median(t[:, 4] - t[:, 3]):6.0f}")
    print(msg, flush=True)


kind = sys.argv[1] if len(sys.argv) > 1 else "planes"
print("kind", kind, "CL4_CONV_S_MFMA =", os.environ.get("CL4_CONV_S_MFMA", "32"), "CL4_CONV_U =", os.environ.get("CL4_CONV_U", "1"))
for n in [int(v) for v in sys.argv[2:]] or [30, 15]:
    run(kind, "conv2 3x3 256->256", n, 256, 33, 256, 3, 1, 1, 1, False)
    run(kind, "conv1 1x1 1024->256", n, 1024, 33, 256, 1, 1, 0, 1, False)
    run(kind, "conv3 1x1 256->1024 +res", n, 256, 33, 1024, 1, 1, 0, 1, True)
    run(kind, "aspp 3x3 d12 2048->256", n, 2048, 33, 256, 3, 1, 12, 12, False)
    run(kind, "res5 3x3 d2 512->512", n, 512, 33, 512, 3, 1, 2, 2, False)
    if kind == "planes":
        run(kind, "conv2 -> planes out", n, 256, 33, 256, 3, 1, 1, 1, False, planes_out=True)
        run(kind, "conv3 +res -> planes out", n, 256, 33, 1024, 1, 1, 0, 1, True, planes_out=True)
