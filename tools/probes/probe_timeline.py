"""Where does a bf16x3 forward launch spend its time? In-kernel s_memtime stamps per workgroup (cl4_conv_debug_timeline):
phase durations (prologue / K loop / park / epilogue), workgroups resident per CU over time, idle gaps per CU slot.
usage: probe_timeline.py  (res4 shapes at the bench batch)"""
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
    ntile = ((cout + plan[1] - 1) // plan[1]) * ((n * h * h + plan[1] - 1) // plan[1])
    tl = torch.zeros(ntile * 8 + 64, dtype=torch.int64, device=dev)
    _hip.call("cl4_conv_debug_timeline", C.c_void_p(tl.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
    e1.record()
    torch.cuda.synchronize()
    _hip.call("cl4_conv_debug_timeline", None)
    t = tl.cpu().numpy()[:ntile * 8].reshape(ntile, 8).astype(np.int64)
    print("   raw column min/max:", [(int(t[:, i].min()), int(t[:, i].max())) for i in range(8)], "rows with slot0==0:", int((t[:, 0] == 0).sum()))
    ok = (t[:, 4] > 0) & (t[:, 0] > 0)
    t = t[ok]
    hw = t[:, 6]
    cu_key = (hw >> 32) * 65536 + ((hw & 0xffffffff) >> 8 & 0xff) * 1 + ((hw & 0xffffffff) >> 12 & 0xf) * 256   # xcc, cu_id, sh/se
    keys = np.unique(cu_key)
    spans = []
    for kk in keys:                                     # s_memtime bases differ between CUs: compare within a CU only
        sel = cu_key == kk
        t[sel, :6] -= t[sel, 0].min()
        spans.append(t[sel][:, [4, 5]].max())
    t0 = 0
    span = float(np.median(spans))                      # a CU is busy for (almost) the whole launch
    tick_us = e0.elapsed_time(e1) * 1e3 / span
    ph = {"prologue": t[:, 1] - t[:, 0], "kloop": t[:, 2] - t[:, 1], "park": t[:, 3] - t[:, 2], "epilogue": t[:, 4] - t[:, 3],
          "total": t[:, 4] - t[:, 0]}
    print(f"== {name}: {ntile} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us, median CU span {span:.0f} ticks ({tick_us * 1e3:.3f} ns/tick; "
          f"CU spans p10 {np.percentile(spans, 10):.0f} p90 {np.percentile(spans, 90):.0f})")
    for k_, v in ph.items():
        print(f"   {k_:9s} mean {v.mean() * tick_us:7.2f} us  p10 {np.percentile(v, 10) * tick_us:7.2f}  p90 {np.percentile(v, 90) * tick_us:7.2f}")
    occ, kl, ep = [], [], []
    for kk in keys:
        rows = t[cu_key == kk]
        occ.append((rows[:, 4] - rows[:, 0]).sum() / span)              # average resident workgroups on this CU
        kl.append((rows[:, 2] - rows[:, 1]).sum() / span)
        ep.append((rows[:, 4] - rows[:, 3]).sum() / span)
    print(f"   {len(keys)} CUs seen; per CU: workgroups {np.mean([ (cu_key == kk).sum() for kk in keys]):.1f}, avg resident {np.mean(occ):.2f}, "
          f"avg in K loop {np.mean(kl):.2f}, avg in epilogue {np.mean(ep):.2f}")
    # one CU's timeline
    kk = keys[len(keys) // 2]
    rows = t[cu_key == kk]
    rows = rows[np.argsort(rows[:, 0])]
    print("   one CU (us from launch start): " + " | ".join(
        f"{(r[0] - t0) * tick_us:5.1f}>{(r[1] - t0) * tick_us:5.1f}>{(r[2] - t0) * tick_us:5.1f}>{(r[3] - t0) * tick_us:5.1f}>{(r[4] - t0) * tick_us:5.1f}" for r in rows[:12]))


for n in (32,):
    run("conv3 256->1024 +res", n, 256, 33, 1024, 1, 0, 1, (2, 128, 1), True)
    run("conv3 256->1024 no res", n, 256, 33, 1024, 1, 0, 1, (2, 128, 1), False)
    run("conv1 1024->256", n, 1024, 33, 256, 1, 0, 1, (2, 128, 1), False)
    run("conv2 3x3 256->256", n, 256, 33, 256, 3, 1, 1, (2, 128, 1), False)
    run("res2 64->256 @129 +res", n, 64, 129, 256, 1, 0, 1, (2, 128, 1), True)
