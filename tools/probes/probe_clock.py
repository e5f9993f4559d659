"""In-kernel shader clock and cycles per K-tile of the unified-wave kernel under SUSTAINED load: the shape is launched back to back
for ~1.5 s, then one more launch records s_memtime / s_memrealtime stamps (cl4_conv_debug_timeline) — MI355X_MICROARCH.md,
'DVFS give-back' item 6. usage: probe_clock.py [n]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cl4wsis_amd import _hip, ops

dev = torch.device("cuda:0")


def run(name, n, cin, h, cout, k, pad, dil, post):
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, 1)
    res = torch.randn(n, cout, h, h, device=dev) if post else None
    y = torch.empty(n, cout, h, h, device=dev)
    d = ops._conv_desc(x, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    hw_pad = (h * h + 3) & ~3
    ntile = ((cout + 127) // 128) * ((n * hw_pad + 127) // 128)
    tl = torch.zeros(ntile * 8 + 64, dtype=torch.int64, device=dev)
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 0
    while time.time() - t0 < 1.5:
        e0.record()
        for _ in range(100):
            ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
        e1.record()
        iters += 100
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10.0                          # last batch of 100
    _hip.call("cl4_conv_debug_timeline", C.c_void_p(tl.data_ptr()))
    ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
    torch.cuda.synchronize()
    _hip.call("cl4_conv_debug_timeline", None)
    t = tl.cpu().numpy()[:ntile * 8].reshape(ntile, 8)
    t = t[t[:, 4] > 0]
    cyc = (t[:, 4] - t[:, 0]).astype(np.float64)
    rt = (t[:, 5] - t[:, 7]).astype(np.float64)              # 100 MHz ticks
    ghz = np.median(cyc / np.maximum(rt, 1) * 0.1)
    nk = cin * k * k // 32
    kl = (t[:, 2] - t[:, 1]).astype(np.float64)
    tf = 2.0 * y.numel() * cin * k * k / us / 1e6
    print(f"{name:26s} n={n:3d} {us:7.1f} us {tf:5.0f} TF/s | {len(t)} WGs, lifetime {np.median(cyc):8.0f} cycles, clock {ghz:.2f} GHz, "
          f"prologue {np.median(t[:, 1] - t[:, 0]):6.0f}  K loop {np.median(kl):8.0f} = {np.median(kl) / nk:6.0f} / K-tile  park+epilogue {np.median(t[:, 4] - t[:, 2]):6.0f}", flush=True)


print("CL4_CONV_U =", os.environ.get("CL4_CONV_U", "1"), "CL4_LIB =", os.environ.get("CL4_LIB", "default"))
for n in [int(v) for v in sys.argv[1:]] or [30, 15]:
    run("conv2 3x3 256->256", n, 256, 33, 256, 3, 1, 1, False)
    run("conv1 1x1 1024->256", n, 1024, 33, 256, 1, 0, 1, False)
    run("conv3 1x1 256->1024 +res", n, 256, 33, 1024, 1, 0, 1, True)
    run("aspp 3x3 d12 2048->256", n, 2048, 33, 256, 3, 12, 12, False)
