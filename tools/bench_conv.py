"""Micro-benchmark of cl4_conv2d_fwd on the layer shapes of the phase-2 step (b images)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops

def bench(n, cin, h, cout, k, s, p, d, iters=10, precision=0):
    dev = torch.device("cuda:0")
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, int(os.environ.get('PREC', ops.CONV_PRECISION)))
    ops.AUTOTUNE = os.environ.get("TUNE", "0") == "1"
    y = ops.conv2d(x, wp, k, stride=s, pad=p, dil=d)
    ops.AUTOTUNE = False
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, wp, k, stride=s, pad=p, dil=d, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * y.numel() * cin * k * k
    return ms, flops / ms / 1e9

if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    shapes = [
        ("stem7x7s2", 3, 513, 64, 7, 2, 3, 1),
        ("res2.1x1 64->256", 64, 129, 256, 1, 1, 0, 1),
        ("res2.3x3 64", 64, 129, 64, 3, 1, 1, 1),
        ("res3.3x3 128", 128, 65, 128, 3, 1, 1, 1),
        ("res4.1x1 1024->256", 1024, 33, 256, 1, 1, 0, 1),
        ("res4.3x3 256", 256, 33, 256, 3, 1, 1, 1),
        ("res4.1x1 256->1024", 256, 33, 1024, 1, 1, 0, 1),
        ("res5.3x3 512 d2", 512, 33, 512, 3, 1, 2, 2),
        ("aspp 3x3 2048->256 d12", 2048, 33, 256, 3, 1, 12, 12),
        ("aspp 1x1 2048->256", 2048, 33, 256, 1, 1, 0, 1),
    ]
    for name, cin, h, cout, k, s, p, d in shapes:
        ms, tf = bench(b, cin, h, cout, k, s, p, d)
        plan = [v for kk, v in ops._tuned.items() if kk[1] == cin and kk[4] == cout and kk[5] == k]
        print(f"{name:28s} b={b} {ms:8.3f} ms {tf:8.2f} TFLOP/s {plan[-1][0] if plan else ''}", flush=True)
