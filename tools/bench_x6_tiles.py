"""bf16x6 conv: 64x64 (48 KB LDS, 3 WG/CU) vs 128x128 (96 KB LDS, 1 WG/CU) tiles on the trainable-path shapes (b=16)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import ops, _hip
dev = torch.device("cuda:0")
shapes = [(16, 2048, 33, 256, 3, 6), (16, 2048, 33, 256, 1, 1), (16, 1280, 33, 256, 1, 1), (16, 256, 33, 1280, 1, 1),
          (16, 256, 33, 256, 3, 1), (16, 1024, 33, 256, 1, 1), (16, 256, 33, 1024, 1, 1), (16, 512, 33, 512, 3, 2),
          (16, 256, 33, 2048, 3, 6), (16, 128, 129, 128, 1, 1), (16, 64, 129, 64, 3, 1), (16, 128, 65, 128, 3, 1)]
for (n, cin, h, cout, k, dil) in shapes:
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    pad = dil * (k - 1) // 2
    wp = ops.pack_conv_weight(w, 2)
    d = ops._conv_desc(x, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, 2)
    fl = 2.0 * n * h * h * cout * cin * k * k
    res = []
    for tile in (64, 128):
        for splits in (1, 2, 4):
            _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, tile, splits)
            for _ in range(2):
                y = ops.conv2d(x, wp, k, pad=pad, dil=dil)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.conv2d(x, wp, k, pad=pad, dil=dil)
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 5
            res.append(f"t{tile}s{splits} {ms:.3f}ms {fl / ms / 1e9:5.0f}TF")
    _hip.call("cl4_conv2d_fwd_clear_plans")
    print((n, cin, h, cout, k, dil), " | ".join(res), flush=True)
