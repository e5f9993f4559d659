"""K / M sweep of cl4_conv2d_fwd (bf16x3) at the res4 geometry: separates the per-launch fixed cost from the K-proportional
cost. usage: sweep_conv.py [b]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import _hip, ops

dev = torch.device("cuda:0")


def run(n, cin, h, cout, k, pad, dil, plan, post=False, iters=20):
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, 1)
    res = torch.randn(n, cout, h, h, device=dev) if post else None
    y = torch.empty(n, cout, h, h, device=dev)
    d = ops._conv_desc(x, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
    if plan:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    for _ in range(3):
        ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
    e1.record(); torch.cuda.synchronize()
    if plan:
        _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, 2.0 * y.numel() * cin * k * k / ms / 1e9


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    print("1x1, M=256, tile128 split1: K sweep")
    for cin in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        us, tf = run(b, cin, 33, 256, 1, 0, 1, (2, 128, 1))
        print(f"  K={cin:5d} {us:8.1f} us {tf:7.1f} TF/s", flush=True)
    print("1x1, M=256, tile64 split1: K sweep")
    for cin in (32, 256, 1024, 4096):
        us, tf = run(b, cin, 33, 256, 1, 0, 1, (2, 64, 1))
        print(f"  K={cin:5d} {us:8.1f} us {tf:7.1f} TF/s", flush=True)
    print("1x1, K=256, tile128 split1: M sweep (with / without residual)")
    for cout in (128, 256, 512, 1024, 2048):
        us, tf = run(b, 256, 33, cout, 1, 0, 1, (2, 128, 1))
        us2, tf2 = run(b, 256, 33, cout, 1, 0, 1, (2, 128, 1), post=True)
        print(f"  M={cout:5d} {us:8.1f} us {tf:7.1f} TF/s | +post {us2:8.1f} us {tf2:7.1f} TF/s", flush=True)
    print("3x3 256->256 plans")
    for plan in ((2, 128, 1), (2, 128, 2), (2, 128, 3), (2, 64, 1), (2, 64, 2)):
        us, tf = run(b, 256, 33, 256, 3, 1, 1, plan)
        print(f"  plan={plan} {us:8.1f} us {tf:7.1f} TF/s", flush=True)
    print("batch sweep 1x1 1024->256 tile128 split1")
    for n in (4, 8, 15, 16, 30, 32, 60, 64):
        us, tf = run(n, 1024, 33, 256, 1, 0, 1, (2, 128, 1))
        print(f"  n={n:3d} tiles={2 * ((n * 1089 + 127) // 128):5d} {us:8.1f} us {tf:7.1f} TF/s", flush=True)
