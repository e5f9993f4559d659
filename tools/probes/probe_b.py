"""Where the residual-closing / trunk-reading 1x1 layers of the frozen networks spend their time, on the LDS-DMA planes
kernel (conv_b) as the step launches them: planes in, planes out, planes residual IN PLACE, folded ABN, under sustained load (each shape
back to back for ~0.6 s, HIP events around the last 100 launches). usage: probe_b.py [stagger ticks ...]
Run it against library variants (CL4_LIB; tools/run_probe_b.sh): CL4_B_DBG=8 no K loop, 16 no epilogue traffic, 1 no LDS-DMA, 4 no MFMA.
`stagger`: first-round workgroups on odd CUs start that many REFCLK ticks (10 ns) late (cl4_conv_debug_stagger)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cl4wsis_amd import _hip, ops

dev = torch.device("cuda:0")
lib = _hip.load()
lib.cl4_conv_debug_stagger.argtypes = [C.c_int]
lib.cl4_conv_debug_stagger.restype = None

SHAPES = [  # name, cin, h, cout, k, pad, dil, residual
    ("res4 conv3 256->1024 +res", 256, 33, 1024, 1, 0, 1, True),
    ("res4 conv1 1024->256", 1024, 33, 256, 1, 0, 1, False),
    ("res4 conv2 3x3 256->256", 256, 33, 256, 3, 1, 1, False),
    ("res5 conv3 512->2048 +res", 512, 33, 2048, 1, 0, 1, True),
    ("res5 conv1 2048->512", 2048, 33, 512, 1, 0, 1, False),
    ("res3 conv3 128->512 +res", 128, 65, 512, 1, 0, 1, True),
    ("res3 conv1 512->128", 512, 65, 128, 1, 0, 1, False),
]


def run(name, n, cin, h, cout, k, pad, dil, post, plan):
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, 1)
    xs = ops.to_planes(x)
    del x
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    yp = ops.to_planes(torch.randn(n, cout, h, h, device=dev)) if post else ops.Planes(n, cout, h, h, dev)
    d = ops.ConvDescFor(n, cin, h, h, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, 3)
    go = lambda: ops.conv2d_planes(xs, wp, k, pad=pad, dil=dil, scale=sc, shift=sh, act1=ops.ACT_NONE if post else ops.ACT_LEAKY,
                                   act2=ops.ACT_LEAKY if post else ops.ACT_NONE, post=yp if post else None, want_fp32=False, out_planes=yp)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    while time.time() - t0 < 0.6:
        e0.record()
        for _ in range(100):
            go()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10.0
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 0, 64, 1)
    px = n * h * h
    mb = 4e-6 * (px * cin + px * cout * (2 if post else 1) + cin * cout * k * k)
    tf = 2.0 * px * cout * cin * k * k / us / 1e6
    rows = 256 if plan[0] == 4 else 128
    tiles = -(-cout // rows) * -(-px // plan[1])
    print(f"{name:28s} n={n:3d} plan {plan[0]}/{plan[1]}/{plan[2]} tiles {tiles:5d} ({tiles / 256:5.2f}/CU) {us:7.1f} us {tf:5.0f} TF/s {mb / us:5.2f} TB/s", flush=True)
    return us


staggers = [int(v) for v in sys.argv[1:]] or [0]
plans = [int(v) for v in os.environ.get("PROBE_B_PLANS", "4,6").split(",")]
closing_only = os.environ.get("PROBE_B_SHAPES") == "closing"
print("library", os.environ.get("CL4_LIB", "default"), "CL4_CONV_B_STAGES", os.environ.get("CL4_CONV_B_STAGES", "3"))
for st in staggers:
    lib.cl4_conv_debug_stagger(st)
    print(f"== stagger {st} shader cycles ({'odd CUs of the first round' if st > 0 else 'second workgroup of every CU' if st < 0 else 'off'})")
    for n in (32, 16):
        for (name, cin, h, cout, k, pad, dil, post) in SHAPES:
            if (st or closing_only) and not post:
                continue
            for impl in plans:
                if impl == 4 and cout < 256:
                    continue
                run(name, n, cin, h, cout, k, pad, dil, post, (impl, 144, 1))
lib.cl4_conv_debug_stagger(0)
