"""Launch plans of the frozen networks' planes convolutions at small per-GPU batches, timed alone (sustained, 50 launches): the planes kernel
(impl 2; 128 / 144 / 64-pixel tiles; unsplit and K-split) against the LDS-DMA kernel (impl 6; unsplit and K-split, r05).
usage: python tools/probe_small_plans.py [images=4]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import _hip, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
shapes = [("res4 conv2 3x3 256->256", 256, 33, 256, 3, 1, False), ("res4 conv1 1x1 1024->256", 1024, 33, 256, 1, 0, False),
          ("res4 conv3 1x1 256->1024 + residual", 256, 33, 1024, 1, 0, True), ("res5 conv2 3x3 512->512 dil 2", 512, 33, 512, 3, 2, False),
          ("res3 conv2 3x3 128->128 @65", 128, 65, 128, 3, 1, False)]
for name, cin, h, cout, k, pad, res in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, h, generator=g).to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    xs, wp = ops.to_planes(x), ops.pack_conv_weight(wt, 1)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    d = ops.ConvDescFor(n, cin, h, h, cin, 0, cout, k, 1, pad, pad if k == 3 else 1, cout, 0, ops.ACT_LEAKY, 0.01, ops.ACT_LEAKY if res else 0, 0.01, 3)
    post = ops.to_planes(torch.randn(n, cout, h, h, generator=g).to(dev)) if res else None
    rows = []
    for plan in [(2, t, s) for t in (128, 144, 64) for s in (1, 2, 3, 4, 6, 8, 12)] + [(6, 144, s) for s in (1, 2, 3, 4, 6, 8, 12)]:
        try:
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
            run = lambda: ops.conv2d_planes(xs, wp, k, pad=pad, dil=pad if k == 3 else 1, scale=sc, shift=sh, act1=ops.ACT_LEAKY, slope1=0.01,
                                            post=post, act2=ops.ACT_LEAKY if res else ops.ACT_NONE, want_fp32=False, want_planes=True)
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            e1.synchronize()
            rows.append((e0.elapsed_time(e1) / 50 * 1e3, plan))
        except Exception as ex:
            rows.append((float("inf"), plan))
        finally:
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    rows.sort()
    print(f"{name}, {n} images: " + "  ".join(f"{p}: {us:.1f} us" for us, p in rows[:8]))
    print("      LDS-DMA plans: " + "  ".join(f"{p}: {us:.1f}" for us, p in sorted(rows, key=lambda r: r[1]) if p[0] == 6))
