"""Who waits for whom inside a bf16x3 launch? Needs a CL4_TL2 build of the library (cl4wsis_amd/build.py --variant tl2 CL4_TL2=1,
run with CL4_LIB=cl4wsis_amd/libcl4wsis_tl2.so): loader wave 0 and MFMA wave 0 of every workgroup sum the s_memtime spent in each
phase of their K-tile loop. Printed per K-tile and workgroup, in shader cycles and (calibrated on the launch time) in ns.
usage: probe_phases.py [tag]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cl4wsis_amd import _hip, ops

dev = torch.device("cuda:0")


def run(name, n, cin, h, cout, k, pad, dil, plan, post):
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, 1)
    res = torch.randn(n, cout, h, h, device=dev) if post else None
    y = torch.empty(n, cout, h, h, device=dev)
    d = ops._conv_desc(x, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), *plan)
    for _ in range(3):
        ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
    torch.cuda.synchronize()
    T = plan[1]
    ntile = ((cout + T - 1) // T) * ((n * h * h + T - 1) // T)
    tl = torch.zeros(ntile * 24 + 64, dtype=torch.int64, device=dev)
    _hip.call("cl4_conv_debug_timeline", C.c_void_p(tl.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
    e1.record()
    torch.cuda.synchronize()
    _hip.call("cl4_conv_debug_timeline", None)
    us = e0.elapsed_time(e1) * 1e3
    a = tl.cpu().numpy()
    t = a[:ntile * 8].reshape(ntile, 8)
    p = a[ntile * 8:ntile * 24].reshape(ntile, 16)
    hw = t[:, 6]
    cu_key = (hw >> 32) * 65536 + ((hw & 0xffffffff) >> 8 & 0xff) + ((hw & 0xffffffff) >> 12 & 0xf) * 256
    spans = []
    for kk in np.unique(cu_key):
        sel = cu_key == kk
        spans.append(t[sel][:, [4, 5]].max() - t[sel, 0].min())
    ns_tick = us * 1e3 / float(np.median(spans))
    tot = np.maximum(p[:, 5], 1)
    tf = 2.0 * y.numel() * cin * k * k / us / 1e6
    print(f"== {name}: n={n} {ntile} workgroups on {len(np.unique(cu_key))} CUs, {us:.1f} us = {tf:.0f} TFLOP/s, {ns_tick:.3f} ns/tick, K-tiles per workgroup {int(tot[0])}")
    wg = (t[:, 4] - t[:, 0]).mean()
    print(f"   workgroup lifetime {wg * ns_tick / 1e3:6.2f} us: prologue {(t[:, 1] - t[:, 0]).mean() * ns_tick / 1e3:5.2f}  K loop {(t[:, 2] - t[:, 1]).mean() * ns_tick / 1e3:6.2f}"
          f"  park {(t[:, 3] - t[:, 2]).mean() * ns_tick / 1e3:5.2f}  epilogue {(t[:, 4] - t[:, 3]).mean() * ns_tick / 1e3:5.2f}")
    names = ["L wait loads", "L cvt+ds_write", "L issue loads", "L lds drain", "L barrier", None, None, None, "M read+mfma", "M barrier"]
    per = {}
    for i, nm in enumerate(names):
        if nm is None:
            continue
        v = p[:, i] / tot
        per[nm] = v.mean()
        print(f"   {nm:16s} {v.mean():8.0f} cycles/K-tile  ({v.mean() * ns_tick:7.0f} ns)   p10 {np.percentile(v, 10):7.0f}  p90 {np.percentile(v, 90):7.0f}")
    print(f"   loader sum {sum(per[k_] for k_ in names[:5]):.0f}   mfma-wave sum {per['M read+mfma'] + per['M barrier']:.0f}   (24 MFMAs = 768 cycles)")


tag = sys.argv[1] if len(sys.argv) > 1 else ""
print("#### build:", os.environ.get("CL4_LIB", "default"), tag)
P = (2, 128, 1)
for n in (15, 30):
    run("conv1 1x1 1024->256", n, 1024, 33, 256, 1, 0, 1, P, False)
    run("conv2 3x3 256->256", n, 256, 33, 256, 3, 1, 1, P, False)
    run("conv3 1x1 256->1024 +res", n, 256, 33, 1024, 1, 0, 1, P, True)
run("aspp 3x3 d12 2048->256", 30, 2048, 33, 256, 3, 12, 12, P, False)
run("res2 1x1 64->256 @129 +res", 8, 64, 129, 256, 1, 0, 1, P, True)
