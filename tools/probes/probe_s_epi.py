"""Multi-round / epilogue-heavy shapes of the planes kernel under sustained load (r03: up-front residual loads, anti-phase stagger).
Run once per CL4_S_STAGGER setting (the library reads it at its first launch). usage: probe_s_epi.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from cl4wsis_amd import _hip, ops

dev = torch.device("cuda:0")


PLAN = tuple(int(v) for v in os.environ.get("PLAN", "2,128,1").split(","))


def run(name, n, cin, h, cout, k, pad, dil, post, planes_out=True, plan=None):
    plan = plan or PLAN
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, 1)
    res = ops.to_planes(torch.randn(n, cout, h, h, device=dev)) if post else None
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    xs = ops.to_planes(x)
    yp = ops.Planes(n, cout, h, h, dev) if planes_out else None
    y = None if planes_out else torch.empty(n, cout, h, h, device=dev)
    d = ops.ConvDescFor(n, cin, h, h, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, 3)
    go = lambda: ops.conv2d_planes(xs, wp, k, pad=pad, dil=dil, scale=sc, shift=sh, act1=ops.ACT_LEAKY, post=res, act2=ops.ACT_LEAKY if post else ops.ACT_NONE,
                                   out=y, want_fp32=not planes_out, out_planes=yp)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    while time.time() - t0 < 0.8:
        e0.record()
        for _ in range(50):
            go()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 20.0)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    npx = n * h * h
    tiles = -(-cout // 128) * -(-npx // plan[1]) * max(1, abs(plan[2]))
    fl = 2.0 * npx * cout * cin * k * k
    by = 4.0 * npx * (cin + cout * (2 if post else 1))
    print(f"{name:30s} n={n:3d} tiles {tiles:5d} ({tiles / 512:5.2f} rounds) {best:8.1f} us {fl / best / 1e6:6.0f} TF/s {by / best / 1e6:6.2f} TB/s", flush=True)


print("plan", os.environ.get("PLAN", "2,128,1"))
for n in [int(v) for v in sys.argv[1:]] or [32, 16]:
    run("res4 conv3 256->1024 +res", n, 256, 33, 1024, 1, 0, 1, True)
    run("res5 conv3 512->2048 +res", n, 512, 33, 2048, 1, 0, 1, True)
    run("res3 conv3 128->512 +res", n, 128, 65, 512, 1, 0, 1, True)
    run("res5 conv1 2048->512", n, 2048, 33, 512, 1, 0, 1, False)
    run("res5 conv2 3x3 d2 512->512", n, 512, 33, 512, 3, 2, 2, False)
    run("res4 conv1 1024->256", n, 1024, 33, 256, 1, 0, 1, False)
    run("res4 conv2 3x3 256->256", n, 256, 33, 256, 3, 1, 1, False)
    run("res4 conv3 fp32 out", n, 256, 33, 1024, 1, 0, 1, True, planes_out=False)
    run("aspp 3x3 d12 2048->256", n, 2048, 33, 256, 3, 12, 12, False)
    run("aspp 1x1 2048->256", n, 2048, 33, 256, 1, 0, 1, False)
