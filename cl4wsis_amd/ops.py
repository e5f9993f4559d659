"""Tensor-level wrappers over the C ABI (include/cl4wsis.h).

Each function checks shapes/dtypes on the host, allocates outputs from PyTorch's caching
allocator (PyTorch is only the memory/stream plumbing here) and enqueues the HIP kernel on
the current stream. Nothing in this module computes with torch ops.
"""
import ctypes as C
import os

import torch

from . import _hip
from ._hip import ACT_LEAKY, ACT_NONE, ACT_RELU, ConvDesc, call, ptr, stream

__all__ = ["ACT_NONE", "ACT_LEAKY", "ACT_RELU"]


class KernelTimer:
    """Optional live timing of the implicit-GEMM conv launches with HIP events recorded on the launch stream
    (bench.py's roofline leg). Disabled (None) by default: zero overhead on the product path."""

    def __init__(self):
        self.records = []   # (kind, flops, start_event, end_event)
        self.shapes = []    # per record: (n, cin, hin, cout, k, stride, dil) of the launch
        self.epilogue_bytes = {}   # kind -> bytes of the epilogue operands (residual / pre-activation addend) the launches read

    def span(self, kind, flops, shape=None, epilogue_bytes=0):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        self.records.append((kind, flops, e0, e1))
        self.shapes.append(shape)
        self.epilogue_bytes[kind] = self.epilogue_bytes.get(kind, 0) + epilogue_bytes
        return e0, e1

    def by_shape(self):
        """{(kind, shape): (launches, total_flops, total_ms)} — call after a device synchronisation."""
        out = {}
        for (kind, fl, e0, e1), shp in zip(self.records, self.shapes):
            n, f, t = out.get((kind, shp), (0, 0.0, 0.0))
            out[(kind, shp)] = (n + 1, f + fl, t + e0.elapsed_time(e1))
        return out

    def summary(self):
        """{kind: (launches, total_flops, total_ms)} — call after a device synchronisation."""
        out = {}
        for kind, fl, e0, e1 in self.records:
            n, f, t = out.get(kind, (0, 0.0, 0.0))
            out[kind] = (n + 1, f + fl, t + e0.elapsed_time(e1))
        return out


TIMER = None


def _f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


# Arithmetic of the dense forward convolutions: 0 = exact fp32 MFMA, 1 = "bf16x3" (fp32 operands split into bf16
# hi + lo, three bf16 MFMA products with fp32 accumulation; ~4e-6 relative error per layer, 3/16 of the MFMA cycles),
# 2 = "bf16x6" (hi + mid + lo, six products: fp32-level accuracy at 6/16 of the fp32 MFMA cycles).
CONV_PRECISION = int(os.environ.get("CL4_CONV_PRECISION", "1"))
# r04: plain bf16x6 convolutions with a long K loop via cl4_to_planes3 + the three-plane LDS-DMA kernel; thresholds
# from tools/probes/probe_x6_planes.py (the extra pass over the input has to be paid back by the kernel)
# Scope: the call sites of the differentiated body / DeepLabV3 head / PseudoLabeler (body_train.py, heads_train.py: phase 1, step 0 and the
# autograd module surface) ask for it with `x6=True`; `x6_route_scope` switches it on for everything inside a `with`. Phase 2's instance
# branch does not: its precision-2 convolutions share the chip with the frozen networks' stream and the extra pass costs more than it
# returns (measured 464.0 -> 459.2 images/s with the route on everywhere; phase 1: 88.8 -> 83.7-85.6 ms per step).
X6_ROUTE = os.environ.get("CL4_X6_ROUTE", "1") != "0"
_x6_scope = 0
X6_ROUTE_MIN_K3 = int(os.environ.get("CL4_X6_ROUTE_MIN_K3", "1152"))     # 3x3: from 128 input channels (measured gain 1.07-1.27)
X6_ROUTE_MIN_K1 = int(os.environ.get("CL4_X6_ROUTE_MIN_K1", "256"))      # 1x1: widening ones only, cout >= cin (1.11-1.25; narrowing 0.65-0.96)


class x6_route_scope:
    """with ops.x6_route_scope(): plain bf16x6 convolutions inside take the three-plane route (see X6_ROUTE)."""

    def __enter__(self):
        global _x6_scope
        _x6_scope += 1

    def __exit__(self, *exc):
        global _x6_scope
        _x6_scope -= 1


class PackedWeight:
    """Device-side operand of cl4_conv2d_fwd: `data` in the layout of `precision` for a [cout, cin, k, k] weight. `planes`: the operand of
    the kernels that read a two-plane activation (conv_s / conv_b / conv_p / the fused bottleneck) when that differs from `data` — an
    fp16x3 build of the library (cl4_planes_format() == 1, the default since r05) multiplies fp16 hi / lo planes of 4096 w there, while the
    fp32-input bf16x3 kernels (conv_u, conv_igemm) keep the bf16 split in `data`."""
    __slots__ = ("data", "K", "cout", "precision", "planes")

    def __init__(self, data, K, cout, precision, planes=None):
        self.data, self.K, self.cout, self.precision, self.planes = data, K, cout, precision, planes

    def for_planes(self):
        return self.planes if self.planes is not None else self.data

    @property
    def shape(self):
        return (self.K, self.cout)


def bf16x3_ok(cout):
    return cout % 4 == 0 and cout >= 64


_PLANES_F16 = None


def planes_f16():
    """True when the loaded library's two-plane format / arithmetic is fp16x3 (cl4_planes_format(), csrc/planes_fmt.h)."""
    global _PLANES_F16
    if _PLANES_F16 is None:
        _PLANES_F16 = bool(_hip.load().cl4_planes_format())
    return _PLANES_F16


F16X3_MIN_COUT = int(os.environ.get("CL4_F16X3_MIN_COUT", "128"))


def f16x3_ok(cin, cout, k=3):
    """Shapes the fp16x3 route takes: what conv_b needs (32-channel K-tiles, 8-channel output groups, at least one 128-row tile), and only
    where the conversion pass of the input pays — every 3x3 layer, 1x1 layers that do not shrink the channel count (a 1024 -> 256 1x1
    convolution reads 4x the elements it writes: converting them costs more than the bf16x6 launch it replaces, profiles/r05_f16x3_ab.txt)
    and have at least F16X3_MIN_K1 input channels."""
    return cin % 32 == 0 and cout % 8 == 0 and cout >= F16X3_MIN_COUT and (k > 1 or (cout >= cin and cin >= F16X3_MIN_K1) or F16X3_ALL_1X1)


F16X3_ALL_1X1 = os.environ.get("CL4_F16X3_ALL_1X1", "0") != "0"
# ... and 1x1 layers with a K loop long enough to carry the conversion of their input: the decoder's 128 -> 128 / 144 layers at 129 x 129 took
# 0.36 / 0.27 ms through the route against 0.23 / 0.15 as bf16x6 launches (profiles/r05_f16_planes_cost_per_layer.txt); with a producer-made
# twin (force_f16) there is no conversion and the limit does not apply
F16X3_MIN_K1 = int(os.environ.get("CL4_F16X3_MIN_K1", "256"))


def pack_conv_weight(w, precision=None, force_f16=False):
    """[cout, cin, kh, kw] -> the operand layout cl4_conv2d_fwd streams (fp32 [K][cout], or the bf16x3 split planes)."""
    _f32(w, "pack_conv_weight")
    cout, cin, kh, kw = w.shape
    K = cin * kh * kw
    if precision is None:
        precision = CONV_PRECISION
    if precision == 3 and not planes_f16():
        precision = 2
    if precision == 3:
        # fp16x3 (r05): fp16 hi / lo of w * 2^12 for the LDS-DMA planes kernel; shapes it does not take keep bf16x6 (the same accuracy)
        if f16x3_ok(cin, cout, kh) or (force_f16 and f16x3_ok(cin, cout, 3)):
            wq = torch.empty((K // 8) * 2 * cout * 16, dtype=torch.uint8, device=w.device)
            call("cl4_pack_conv_weight_f16x3", ptr(w.contiguous()), ptr(wq), cout, cin, kh, kw, stream())
            return PackedWeight(wq, K, cout, 3)
        precision = 2
    if precision in (1, 2) and bf16x3_ok(cout):
        planes = 2 if precision == 1 else 3
        wq = torch.empty(((K + 7) // 8) * planes * cout * 16, dtype=torch.uint8, device=w.device)
        call("cl4_pack_conv_weight_bf16x3" if precision == 1 else "cl4_pack_conv_weight_bf16x6", ptr(w.contiguous()), ptr(wq),
             cout, cin, kh, kw, stream())
        wq16 = None
        if precision == 1 and planes_f16() and cin % 32 == 0:       # the planes kernels of an fp16x3 build read their own pack of the same weight
            wq16 = torch.empty((K // 8) * 2 * cout * 16, dtype=torch.uint8, device=w.device)
            call("cl4_pack_conv_weight_f16x3", ptr(w.contiguous()), ptr(wq16), cout, cin, kh, kw, stream())
        return PackedWeight(wq, K, cout, precision, planes=wq16)
    wp = torch.empty((K, cout), dtype=torch.float32, device=w.device)
    call("cl4_pack_conv_weight", ptr(w.contiguous()), ptr(wp), cout, cin, kh, kw, stream())
    return PackedWeight(wp, K, cout, 0)


def abs_eps(w, eps):
    out = torch.empty_like(w)
    call("cl4_abs_eps", ptr(w), ptr(out), w.numel(), C.c_float(eps), stream())
    return out


def sign_axpy_(y, x, w):
    """y += sign(w) * x with sign(0) = -1 (inplace_abn's convention)."""
    call("cl4_sign_axpy", ptr(y), ptr(x), ptr(w), y.numel(), stream())
    return y


def bn_fold(gamma, beta, mean, var, eps=1e-5):
    c = mean.numel()
    scale = torch.empty(c, dtype=torch.float32, device=mean.device)
    shift = torch.empty_like(scale)
    call("cl4_bn_fold", ptr(gamma), ptr(beta), ptr(mean), ptr(var), C.c_float(eps), ptr(scale), ptr(shift), c, stream())
    return scale, shift


class _Scratch:
    """Scratch buffers that kernels address by raw pointer — and that captured HIP graphs therefore keep addressing on every
    replay. Rules (r03; the r02 caches replaced a buffer on growth and so freed memory a live graph still wrote to):

    * one buffer per (device, stream) key: kernels of different streams run concurrently and must not share scratch;
    * a buffer that has to grow is RETIRED, never freed: the old tensor moves to `retired` and stays alive, so an address baked
      into a captured kernel remains this module's memory for the life of the process (growth is geometric, so the retired
      buffers of a key sum to less than the live one; `release_retired()` frees them once the caller knows no graph is alive);
    * a key seen for the first time (the capture stream of torch.cuda.graph is one) starts at the largest size any stream of the
      device has needed so far, so a capture that follows an eager warm-up step never grows a buffer inside the capture."""

    keep_retired = True         # tests only: False restores the r02 behaviour (the replaced buffer is dropped) to show they have teeth

    def __init__(self, min_bytes):
        self.min_bytes = min_bytes
        self.cur = {}
        self.retired = []
        self.peak = {}          # device index -> largest request so far
        self.events = []        # (stream key, bytes, allocated while capturing, address of the buffer it retired or 0): diagnostics

    def get(self, device, nbytes, per_stream=True):
        idx = device.index if device.type == "cuda" else -1
        key = (device.type, idx, torch.cuda.current_stream().cuda_stream if (per_stream and device.type == "cuda") else 0)
        nbytes = int(nbytes)
        peak = self.peak[idx] = max(self.peak.get(idx, 0), nbytes)
        ws = self.cur.get(key)
        if ws is None or ws.numel() < nbytes:
            size = max(peak, self.min_bytes)
            if ws is not None:
                if self.keep_retired:
                    self.retired.append(ws)
                size = max(size, ws.numel() + ws.numel() // 2)
            old = ws.data_ptr() if ws is not None else 0
            ws = torch.empty(size, dtype=torch.uint8, device=device)
            self.cur[key] = ws
            self.events.append((key[2], size, device.type == "cuda" and torch.cuda.is_current_stream_capturing(), old))
        return ws

    def live(self):
        return list(self.cur.values()) + list(self.retired)

    def release_retired(self):
        self.retired.clear()


_conv_scratch = _Scratch(64 << 20)


def _conv_workspace(device, nbytes):
    """Split-K scratch of the forward conv — one buffer PER STREAM; it never aliases the BN / loss workspaces."""
    return _conv_scratch.get(device, nbytes)


def workspace_snapshot():
    """Strong references to every scratch buffer of this module (live and retired). The owner of a captured graph may hold the
    list next to the graph; with the retire-on-growth rule of `_Scratch` it is a second lock on the same door."""
    return _conv_scratch.live() + _misc_scratch.live() + _input_scratch.live() + list(_label_ws.values())


def release_retired_workspaces():
    """Free the buffers that growth has retired. Only safe when no captured graph of this process is alive."""
    for s in (_conv_scratch, _misc_scratch, _input_scratch):
        s.release_retired()


AUTOTUNE = False      # set True during warm-up (Trainer / bench): time the launch plans of every new conv shape once
_tuned = {}
_SPLITS = [int(v) for v in os.environ.get("CL4_TUNE_SPLITS", "1,2,3,4,6,8,12,16,24,-1,-4,-8,-16").split(",")]   # experiments: restrict the K splits tried
_CANDIDATES = [(1, 128, 1)] + [(2, t, s) for t in (128, 64) for s in _SPLITS]   # s < 0: main + split tail (-1: small-tile tail)


SPLIT_MAX_TILES = int(os.environ.get("CL4_TUNE_SPLIT_MAX_TILES", "256"))   # whole-launch K splits only for launches of fewer 128-row tiles than this
_FULL_SPLITS_MIN_N = int(os.environ.get("CL4_TUNE_FULL_SPLITS_MIN_N", "1000000"))   # experiment: launches of at least this many images may use whole-launch K splits


def _fills_the_chip(d):
    """A launch with at least one 128x128 tile per CU. The tuner times candidates ALONE, but the step runs two streams side by
    side (frozen old model || new model): there, what counts is the CU-time a plan consumes, not its latency in isolation, and a
    whole-launch K split (more prologues, slab traffic, a second pass) consumes more. Measured on the bench step: with the full
    splits tried everywhere the eager conv time is lowest (33.2 ms) but the overlapped step is slower (47.1 vs 46.8 ms) than with
    whole-launch splits reserved for the launches that cannot fill the chip (small batches, thin layers). Tail plans (splits < 0:
    only the last partial round is split) stay available everywhere."""
    tiles = -(-d.cout // 128) * -(-(d.n * d.hout * d.wout) // 128)
    if d.n >= _FULL_SPLITS_MIN_N:
        return False
    return d.precision != 0 and tiles >= SPLIT_MAX_TILES


def _autotune_conv(d, x, wp, scale, shift, pre, post, out, reps=6):
    """Measure-don't-guess: run each candidate plan of this shape a few times (HIP events on the launch stream) and pin
    the fastest through cl4_conv2d_fwd_set_plan. The output is simply rewritten, so tuning is side-effect free."""
    key = (d.n, d.cin, d.hin, d.win, d.cout, d.kh, d.stride, d.pad, d.dil, d.precision, x.device.index)
    if key in _tuned:
        return
    lib = _hip.load()
    K = d.cin * d.kh * d.kw
    best, best_ms = None, float("inf")
    seen = set()
    for impl, tile, splits in _CANDIDATES:
        if impl == 2 and (d.cout % 4 or d.cout < 64):
            continue
        if impl == 1 and d.precision != 0:
            continue
        if impl == 2 and (tile == 128 and d.cout < 128 or abs(splits) > max(1, K // 64)):
            continue
        if splits == -1 and tile != 128:
            continue
        if splits < 0 and d.precision == 0:
            continue
        if splits > 1 and _fills_the_chip(d):
            continue
        call("cl4_conv2d_fwd_set_plan", C.byref(d), impl, tile, splits)
        need = lib.cl4_conv2d_fwd_workspace_bytes(C.byref(d))
        sig = (impl, tile, need, splits < 0, splits == -1)
        if sig in seen or need > (1 << 30):
            continue
        seen.add(sig)
        ws = _conv_workspace(x.device, need) if need else None
        run = lambda: call("cl4_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(pre), ptr(post),
                           ptr(out), ptr(ws), need, stream())
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps + (SPLIT_MARGIN_MS if splits > 1 else TAIL_MARGIN_MS if splits < 0 else 0.0)
        if ms < best_ms:
            best, best_ms = (impl, tile, splits), ms
    call("cl4_conv2d_fwd_set_plan", C.byref(d), *best)
    _tuned[key] = (best, best_ms)


# A K-split plan is timed alone; inside the replayed step its second launch and its S-fold workgroups compete with the other chains of the
# graph, so it has to win by this margin (ms) to be pinned (experiments at small per-GPU batches: tools/ab_split_margin.sh)
SPLIT_MARGIN_MS = float(os.environ.get("CL4_TUNE_SPLIT_MARGIN_US", "0")) * 1e-3      # whole-launch K splits (splits > 1)
TAIL_MARGIN_MS = float(os.environ.get("CL4_TUNE_TAIL_MARGIN_US", "0")) * 1e-3        # main launch + K-split tail launch (splits < 0)
CONV_P = os.environ.get("CL4_CONV_P", "1") != "0"      # 0: never pick the persistent pipelined planes kernel (A/B measurements)
SMALL_TILE_BELOW = int(os.environ.get("CL4_SMALL_TILE_BELOW", "384"))    # planes launches with fewer 128 x 128 tiles than this also try 64-pixel tiles (0: never)
CONV_B = os.environ.get("CL4_CONV_B", "1") != "0"      # r03: the tuner may pick the big-tile LDS-DMA planes kernel (plan impl 4)
CONV_B_SPLIT = os.environ.get("CL4_CONV_B_SPLIT", "1") != "0"      # r05: ... and its K-split variant (plan impl 6 with splits > 1)
PLANES_TILES = [int(v) for v in os.environ.get("CL4_PLANES_TILES", "128,144").split(",")]   # pixel widths of the planes kernel's tiles the tuner may pick (64 is built too: three workgroups per CU, measured no faster — profiles/r03_conv_s_tile64.txt)


def _planes_key(d):
    return (d.n, d.cin, d.hin, d.win, d.cout, d.kh, d.stride, d.pad, d.dil, d.precision, torch.cuda.current_device())


def _planes_tuned(d):
    return _planes_key(d) in _tuned


def _autotune_planes(d, args, reps=6):
    """_autotune_conv for the planes-input entry point: 128 output channels x 128 or 144 pixels per tile (144: the 33 x 33 maps fill
    one round of resident workgroups, conv_s.hip), each unsplit, K-split, or main + split tail."""
    dev = torch.cuda.current_device()
    key = _planes_key(d)
    if key in _tuned:
        return
    lib = _hip.load()
    K = d.cin * d.kh * d.kw
    best, best_ms = None, float("inf")
    seen = set()
    npix = d.n * d.hout * d.wout
    tiles_ = list(PLANES_TILES)
    if SMALL_TILE_BELOW and 64 not in tiles_ and -(-d.cout // 128) * -(-npix // 128) < SMALL_TILE_BELOW:
        tiles_.append(64)       # launches that cannot fill the chip with 128-pixel tiles (small per-GPU batches): twice the workgroups, three per CU
    cands = [(2, tile, splits) for tile in ((144,) if d.precision == 4 else tiles_) for splits in _SPLITS]
    # the persistent pipelined kernel (csrc/conv_p.hip): short-K 1x1 layers that write planes only (and read a planes residual)
    pre, post_f, out_f32, out_pl = args[4], args[5], args[7], args[8]
    if (CONV_P and d.precision == 3 and d.kh == 1 and d.stride == 1 and d.pad == 0 and d.cout % 128 == 0 and K >= 128 and pre is None and post_f is None
            and out_f32 is None and out_pl is not None and -(-d.cout // 128) * -(-npix // 128) >= 512):
        cands.append((3, 128, 1))
    # the big-tile LDS-DMA kernel (csrc/conv_b.hip): 256 x 144 tiles, one workgroup per CU — long-K layers with >= 256 output channels
    if CONV_B and d.precision == 4 and d.cout % 8 == 0 and d.cout >= 128 and K >= 256:
        cands.append((6, 144, 1))           # three planes: the 128-row / eight-wave tile only
    if CONV_B and d.precision == 3 and d.cout % 8 == 0 and K >= 256:
        if d.cout >= 256:
            cands.append((4, 144, 1))
        if d.cout >= 128:
            cands.append((6, 144, 1))       # 128-row tiles of the same kernel, eight waves of 32 rows: launches with too few 256-row tiles to fill
                                            # the chip (impl 5 = the same tile as four waves of 64 rows is built and tested, but never faster)
            if CONV_B_SPLIT:                # r05: ... and K-split over workgroups (slabs + the split-K epilogue) where even those cannot (2-8 images per GPU)
                cands.extend((6, 144, sp) for sp in _SPLITS if sp > 1)
    for impl, tile, splits in cands:
        tiles = -(-d.cout // 128) * -(-npix // tile)
        if splits == -1 or abs(splits) > max(1, K // 64):
            continue
        if splits > 1 and tiles >= SPLIT_MAX_TILES and d.n < _FULL_SPLITS_MIN_N:      # whole-launch splits only where a launch cannot fill the chip (_fills_the_chip)
            continue
        call("cl4_conv2d_fwd_set_plan", C.byref(d), impl, tile, splits)
        need = lib.cl4_conv2d_fwd_workspace_bytes(C.byref(d))
        sig = (impl, tile, need, splits < 0)
        if sig in seen or need > (1 << 30):
            continue
        seen.add(sig)
        ws = _conv_workspace(torch.device("cuda", dev), need) if need else None
        run = lambda: call("cl4_conv2d_fwd_planes", C.byref(d), *args, ptr(ws), need, stream())
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps + (SPLIT_MARGIN_MS if splits > 1 else TAIL_MARGIN_MS if splits < 0 else 0.0)
        if ms < best_ms:
            best, best_ms = (impl, tile, splits), ms
    call("cl4_conv2d_fwd_set_plan", C.byref(d), *best)
    _tuned[key] = (best, best_ms)


def _conv_desc(x, cin, in_coff, cout, k, stride, pad, dil, out_ctot, out_coff, act1, slope1, act2, slope2, precision=0):
    n, in_ctot, hin, win = x.shape
    d = ConvDesc()
    d.n, d.cin, d.hin, d.win = n, cin, hin, win
    d.cout, d.kh, d.kw = cout, k, k
    d.stride, d.pad, d.dil = stride, pad, dil
    d.hout = conv_out_size(hin, k, stride, pad, dil)
    d.wout = conv_out_size(win, k, stride, pad, dil)
    d.in_ctot, d.in_coff = in_ctot, in_coff
    d.out_ctot, d.out_coff = out_ctot, out_coff
    d.act1, d.slope1, d.act2, d.slope2 = act1, slope1, act2, slope2
    d.precision = precision
    return d


_grad_depth = 0


def gradient_scope(fn):
    """Decorator of the hand-written BACKWARD passes: every convolution launched inside takes a gradient as its input — any magnitude —
    so an fp16x3 launch scales it dynamically (conv2d's `xdyn`) without each call site having to say so."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        global _grad_depth
        _grad_depth += 1
        try:
            return fn(*a, **k)
        finally:
            _grad_depth -= 1
    return wrapped


def conv2d(x, wp, k, stride=1, pad=0, dil=1, scale=None, shift=None, pre=None, post=None,
           act1=ACT_NONE, slope1=0.01, act2=ACT_NONE, slope2=0.01,
           out=None, out_coff=0, in_coff=0, cin=None, x6=False, xdyn=False, xs_cache=None, stats_out=None):
    """Fused conv (+affine +act +residual +act). `wp` comes from pack_conv_weight. `x6`: see X6_ROUTE. `xdyn`: `x` is a gradient (any
    magnitude): an fp16x3 launch then scales it by a power of two taken from its maximum (to_planes_f16(dynamic=True)). `xs_cache`: a
    dict the caller shares between convolutions that read the SAME input tensor (the ASPP branches, a block's conv1 and projection):
    the fp16x3 operand is converted once. `stats_out` (a list): a plain fp16x3 launch whose plan allows it also leaves the per-channel
    partial sums of its output — (partials, nparts) is appended — for the train-mode ABN behind it (abn_train_forward(partials=...)).

    `out`/`out_coff` write into a channel slice of a wider tensor (replaces torch.cat);
    `in_coff`/`cin` read a channel slice of `x`.
    """
    _f32(x, "conv2d.x")
    n, in_ctot, hin, win = x.shape
    if cin is None:
        cin = in_ctot - in_coff
    if not isinstance(wp, PackedWeight):      # a raw fp32 [K][cout] operand (e.g. the PyTorch weight of a 1x1 dgrad)
        wp = PackedWeight(wp, wp.shape[0], wp.shape[1], 0)
    K, cout = wp.K, wp.cout
    if K != cin * k * k:
        raise RuntimeError(f"conv2d: packed weight K={K} does not match cin*k*k={cin * k * k}")
    if wp.precision == 3 and not planes_f16():
        raise RuntimeError("conv2d: fp16x3 packed weights need an fp16x3 build of the library (cl4_planes_format() == 1)")
    if wp.precision == 3:
        # r05, fp16x3: the differentiated convolutions as three fp16 MFMA products per MAC (the accuracy of bf16x6's six) through an fp16
        # two-plane copy of the input and the LDS-DMA kernel (conv_b.hip, F16)
        global TIMER
        timer, e1 = TIMER, None
        if timer is not None:          # one span over the conversion AND the convolution: what the route costs against a direct bf16x6 launch
            hout_, wout_ = conv_out_size(hin, k, stride, pad, dil), conv_out_size(win, k, stride, pad, dil)
            e0, e1 = timer.span("conv_fwd_f16x3", 2.0 * n * hout_ * wout_ * cout * K, (n, cin, hin, cout, k, stride, dil),
                                4 * n * cout * hout_ * wout_ * ((pre is not None) + (post is not None)))
            e0.record()
            TIMER = None
        try:
            dyn = xdyn or _grad_depth > 0
            ckey = (x.data_ptr(), tuple(x.shape), in_coff, cin, dyn)
            xs = xs_cache.get(ckey) if xs_cache is not None else None
            if xs is None:
                # a forward activation: the planes format of the build itself (fp16 hi / lo of 16 x) — the frozen producer's twin when it
                # left one; a gradient: its own power-of-two scale, in device memory
                # (a gradient's twin carries its scale — `slot` — and comes from abn_train_backward(want_planes=True))
                xs = getattr(x, "_cl4_planes", None) if (in_coff == 0 and cin == in_ctot) else None
                if xs is None or xs.planes != 2 or (xs.slot is not None) != dyn:
                    xs = to_planes_f16(x, dynamic=True, in_coff=in_coff, c=cin) if dyn else to_planes(x, in_coff=in_coff, c=cin)
                if xs_cache is not None:
                    xs_cache[ckey] = xs
            if xs.slot is not None:
                y = conv2d_planes_f16(xs, wp, k, stride=stride, pad=pad, dil=dil, scale=scale, shift=shift, pre=pre, post=post, act1=act1, slope1=slope1,
                                      act2=act2, slope2=slope2, out=out, out_coff=out_coff, stats_out=stats_out)
            else:
                y = conv2d_planes(xs, wp, k, stride=stride, pad=pad, dil=dil, scale=scale, shift=shift, pre=pre, post=post, act1=act1, slope1=slope1,
                                  act2=act2, slope2=slope2, out=out, out_coff=out_coff, want_fp32=True, stats_out=stats_out)[0]
        finally:
            TIMER = timer
        if e1 is not None:
            e1.record()
        return y
    if (X6_ROUTE and (x6 or _x6_scope) and wp.precision == 2 and stride == 1 and in_coff == 0 and cin == in_ctot and cin % 32 == 0 and cout % 8 == 0
            and cout >= 128 and scale is None and shift is None and pre is None and post is None and act1 == ACT_NONE and act2 == ACT_NONE
            and x.is_contiguous()
            and ((k == 3 and K >= X6_ROUTE_MIN_K3) or (k == 1 and K >= X6_ROUTE_MIN_K1 and cout >= cin))):
        # r04: a plain bf16x6 convolution of the differentiated networks (phase 1 / step 0: body, DeepLabV3 head, PseudoLabeler, forward and
        # input gradient) whose K loop is long enough goes through a three-plane copy of its input and the LDS-DMA kernel (conv_b, FINDINGS.md
        # finding 32: 0.65 of the bf16x6 ceiling against 0.3-0.45 for the role-split kernel); same six products per MAC, fp32 accumulation.
        x3 = to_planes(x, planes=3)
        return conv2d_planes(x3, wp, k, stride=1, pad=pad, dil=dil, out=out, out_coff=out_coff, want_fp32=True)[0]
    hout = conv_out_size(hin, k, stride, pad, dil)
    wout = conv_out_size(win, k, stride, pad, dil)
    if out is None:
        out = torch.empty((n, cout, hout, wout), dtype=torch.float32, device=x.device)
        out_coff = 0
    elif tuple(out.shape[2:]) != (hout, wout) or out.shape[0] != n:
        raise RuntimeError(f"conv2d: out shape {tuple(out.shape)} incompatible with {(n, cout, hout, wout)}")
    for t, nm in ((pre, "pre"), (post, "post")):
        if t is not None and tuple(t.shape) != (n, cout, hout, wout):
            raise RuntimeError(f"conv2d: {nm} shape {tuple(t.shape)} != {(n, cout, hout, wout)}")
    for t, nm in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != cout:
            raise RuntimeError(f"conv2d: {nm} has {t.numel()} elements, expected {cout}")
    d = _conv_desc(x, cin, in_coff, cout, k, stride, pad, dil, out.shape[1], out_coff, act1, slope1, act2, slope2,
                   wp.precision)
    wp = wp.data
    if TIMER is not None:
        e0, e1 = TIMER.span(("conv_fwd_f32", "conv_fwd_bf16x3", "conv_fwd_bf16x6")[d.precision], 2.0 * n * hout * wout * cout * K,
                            (n, cin, hin, cout, k, stride, dil), 4 * n * cout * hout * wout * ((pre is not None) + (post is not None)))
        e0.record()
    if AUTOTUNE:
        _autotune_conv(d, x, wp, scale, shift, pre, post, out)
    need = _hip.load().cl4_conv2d_fwd_workspace_bytes(C.byref(d))
    ws = _conv_workspace(x.device, need) if need else None
    call("cl4_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(pre), ptr(post), ptr(out),
         ptr(ws), need, stream())
    if TIMER is not None:
        e1.record()
    return out


class Planes:
    """An activation tensor [n, c, h, w] kept as pre-split bf16 planes — [n][c/8][hi, lo][h*w][8 x bf16], 4 bytes per element
    (include/cl4wsis.h, "Planes" activations): what conv2d_planes reads and can write. `planes` = 3: [hi, mid, lo], 6 bytes per
    element — the operand of the bf16x6 launches (read only)."""
    __slots__ = ("data", "n", "c", "h", "w", "planes", "slot")

    def __init__(self, n, c, h, w, device, data=None, planes=2, slot=None):
        if c % 8:
            raise RuntimeError(f"Planes: channel count {c} is not a multiple of 8")
        self.n, self.c, self.h, self.w, self.planes = n, c, h, w, planes
        self.slot = slot       # fp16x3 operand (to_planes_f16): device floats, [0] = the power-of-two scale of the stored values; None: bf16 planes
        self.data = torch.empty(n * c * h * w * planes // 2, dtype=torch.int32, device=device) if data is None else data

    @property
    def shape(self):
        return (self.n, self.c, self.h, self.w)


def batch_slice(t, bs):
    """t[:bs] of an NCHW activation, keeping its planes twin (the first `bs` images of a Planes tensor are contiguous as well)."""
    v = t[:bs]
    ps = getattr(t, "_cl4_planes", None)
    if ps is not None and bs <= ps.n:
        v._cl4_planes = Planes(bs, ps.c, ps.h, ps.w, ps.data.device, data=ps.data[:bs * ps.c * ps.h * ps.w * ps.planes // 2], planes=ps.planes)
    return v


def to_planes(x, in_coff=0, c=None, out=None, planes=2):
    """fp32 NCHW (channel slice) -> Planes; planes = 3: the three-plane operand of the bf16x6 launches (cl4_to_planes3)."""
    _f32(x, "to_planes")
    n, ctot, h, w = x.shape
    c = ctot - in_coff if c is None else c
    ps = Planes(n, c, h, w, x.device, planes=planes) if out is None else out
    call("cl4_to_planes3" if ps.planes == 3 else "cl4_to_planes", ptr(x), ptr(ps.data), n, ctot, in_coff, c, h * w, stream())
    return ps


def to_planes_f16(x, dynamic=False, in_coff=0, c=None):
    """fp32 NCHW (channel slice) -> the fp16x3 operand of conv2d_planes_f16: two fp16 planes of x * s, s a power of two (2^4, or —
    `dynamic`, for gradients — chosen from max |x| by a reduction launch in front). The scale lives in device memory (`.slot[0]`)."""
    _f32(x, "to_planes_f16")
    n, ctot, h, w = x.shape
    c = ctot - in_coff if c is None else c
    slot = torch.empty(_hip.load().cl4_to_planes_f16_slot_floats(), dtype=torch.float32, device=x.device)
    ps = Planes(n, c, h, w, x.device, planes=2, slot=slot)
    call("cl4_to_planes_f16", ptr(x), ptr(ps.data), n, ctot, in_coff, c, h * w, int(bool(dynamic)), ptr(slot), stream())
    return ps


CONV_STATS = os.environ.get("CL4_CONV_STATS", "1") != "0"      # r05: a convolution in front of a train-mode ABN also leaves the ABN's partial sums


def _launch_with_stats(d, xs, wp_ptr, out, stats_out):
    """cl4_conv2d_fwd_planes_stats when the launch plan of this shape can produce the statistics: appends (partials, nparts) to `stats_out`
    and returns True; False: the caller launches the plain entry (and the ABN makes its own pass)."""
    parts = _hip.load().cl4_conv2d_fwd_planes_stats_parts(C.byref(d)) if CONV_STATS else 0
    if parts <= 0:
        return False
    partial = torch.empty(d.cout * parts * 2, dtype=torch.float64, device=out.device)
    need = _hip.load().cl4_conv2d_fwd_workspace_bytes(C.byref(d))
    ws = _conv_workspace(out.device, need) if need else None
    call("cl4_conv2d_fwd_planes_stats", C.byref(d), ptr(xs.data), ptr(xs.slot), wp_ptr, ptr(out), ptr(partial), parts, ptr(ws), need, stream())
    stats_out.append((partial, parts))
    return True


def conv2d_planes_f16(xs, wp, k, stride=1, pad=0, dil=1, scale=None, shift=None, pre=None, post=None,
                      act1=ACT_NONE, slope1=0.01, act2=ACT_NONE, slope2=0.01, out=None, out_coff=0, stats_out=None):
    """conv2d on an fp16x3 operand (to_planes_f16) with pack_conv_weight(w, 3) weights: three fp16 MFMA products per MAC, fp32
    accumulation, the accuracy of bf16x6 (csrc/conv_b.hip, F16). fp32 NCHW output (optionally a channel slice of `out`)."""
    if not isinstance(xs, Planes) or xs.slot is None:
        raise RuntimeError("conv2d_planes_f16: the input must come from ops.to_planes_f16")
    if not isinstance(wp, PackedWeight) or wp.precision != 3:
        raise RuntimeError("conv2d_planes_f16: fp16x3 packed weights (pack_conv_weight(w, 3)) expected")
    n, cin, hin, win = xs.shape
    K, cout = wp.K, wp.cout
    if K != cin * k * k:
        raise RuntimeError(f"conv2d_planes_f16: packed weight K={K} does not match cin*k*k={cin * k * k}")
    hout = conv_out_size(hin, k, stride, pad, dil)
    wout = conv_out_size(win, k, stride, pad, dil)
    dev = xs.data.device
    if out is None:
        out = torch.empty((n, cout, hout, wout), dtype=torch.float32, device=dev)
        out_coff = 0
    elif tuple(out.shape[2:]) != (hout, wout) or out.shape[0] != n:
        raise RuntimeError(f"conv2d_planes_f16: out shape {tuple(out.shape)} incompatible with {(n, cout, hout, wout)}")
    for t, nm in ((pre, "pre"), (post, "post")):
        if t is not None and tuple(t.shape) != (n, cout, hout, wout):
            raise RuntimeError(f"conv2d_planes_f16: {nm} shape {tuple(t.shape)} != {(n, cout, hout, wout)}")
    d = ConvDescFor(n, cin, hin, win, cin, 0, cout, k, stride, pad, dil, out.shape[1], out_coff, act1, slope1, act2, slope2, 5)
    if TIMER is not None:
        e0, e1 = TIMER.span("conv_fwd_f16x3", 2.0 * n * hout * wout * cout * K, (n, cin, hin, cout, k, stride, dil),
                            4 * n * cout * hout * wout * ((pre is not None) + (post is not None)))
        e0.record()
    args = (ptr(xs.data), ptr(xs.slot), ptr(wp.data), ptr(scale), ptr(shift), ptr(pre), ptr(post), ptr(out))
    if AUTOTUNE and not _planes_tuned(d):
        _autotune_f16(d, args)
    plain = scale is None and shift is None and pre is None and post is None and act1 == ACT_NONE and act2 == ACT_NONE
    if not (stats_out is not None and plain and out_coff == 0 and out.shape[1] == cout and _launch_with_stats(d, xs, ptr(wp.data), out, stats_out)):
        call("cl4_conv2d_fwd_planes_f16", C.byref(d), *args, None, 0, stream())
    if TIMER is not None:
        e1.record()
    return out


def _autotune_f16(d, args, reps=6):
    """fp16x3 launches run on the LDS-DMA kernel only: 256-row tiles (impl 4) against eight waves on 128-row tiles (impl 6)."""
    key = _planes_key(d)
    if key in _tuned:
        return
    cands = [(6, 144, 1)] + ([(4, 144, 1)] if d.cout % 256 == 0 else [])
    best, best_ms = cands[0], float("inf")
    for cand in cands if len(cands) > 1 else ():
        call("cl4_conv2d_fwd_set_plan", C.byref(d), *cand)
        run = lambda: call("cl4_conv2d_fwd_planes_f16", C.byref(d), *args, None, 0, stream())
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if ms < best_ms:
            best, best_ms = cand, ms
    call("cl4_conv2d_fwd_set_plan", C.byref(d), *best)
    _tuned[key] = (best, best_ms if best_ms < float("inf") else 0.0)


def from_planes(ps, out=None, out_coff=0):
    if ps.planes != 2:
        raise RuntimeError("from_planes: two-plane tensors only")
    if out is None:
        out = torch.empty(ps.shape, dtype=torch.float32, device=ps.data.device)
        out_coff = 0
    call("cl4_from_planes", ptr(ps.data), ptr(out), ps.n, ps.c, ps.h * ps.w, out.shape[1], out_coff, stream())
    return out


def conv2d_planes(xs, wp, k, stride=1, pad=0, dil=1, scale=None, shift=None, pre=None, post=None,
                  act1=ACT_NONE, slope1=0.01, act2=ACT_NONE, slope2=0.01, in_coff=0, cin=None,
                  out=None, out_coff=0, want_fp32=True, out_planes=None, want_planes=False, stats_out=None):
    """conv2d on a Planes input (csrc/conv_s.hip). `post` may be an fp32 tensor or a Planes tensor. Returns (y, ys): the fp32 NCHW
    output (None unless want_fp32 / out) and the Planes output (None unless want_planes / out_planes). `out` / `out_coff` /
    `out_planes` write channel slices of wider tensors (both outputs use the same slice offsets)."""
    if not isinstance(xs, Planes):
        raise RuntimeError("conv2d_planes: the input must be a Planes tensor (ops.to_planes)")
    n, in_ctot, hin, win = xs.shape
    if cin is None:
        cin = in_ctot - in_coff
    x6 = xs.planes == 3
    if not isinstance(wp, PackedWeight) or (wp.precision != 2 if x6 else wp.precision not in (1, 3)) or (not x6 and planes_f16() and wp.precision == 1 and wp.planes is None):
        raise RuntimeError("conv2d_planes: x3 packed weights (pack_conv_weight(w, 1), or (w, 3) in an fp16x3 build) expected for a two-plane input, bf16x6 "
                           "(pack_conv_weight(w, 2)) for a three-plane one")
    if not x6 and wp.precision == 3 and not planes_f16():
        raise RuntimeError("conv2d_planes: fp16x3 packed weights need an fp16x3 build of the library")
    if x6 and (isinstance(post, Planes) or out_planes is not None or want_planes):
        raise RuntimeError("conv2d_planes: a three-plane (bf16x6) launch writes fp32 only and takes no planes residual")
    K, cout = wp.K, wp.cout
    if K != cin * k * k:
        raise RuntimeError(f"conv2d_planes: packed weight K={K} does not match cin*k*k={cin * k * k}")
    hout = conv_out_size(hin, k, stride, pad, dil)
    wout = conv_out_size(win, k, stride, pad, dil)
    dev = xs.data.device
    if out is None and want_fp32:
        out = torch.empty((n, cout, hout, wout), dtype=torch.float32, device=dev)
        out_coff = 0
    if out_planes is None and want_planes:
        out_planes = Planes(n, cout if out is None else out.shape[1], hout, wout, dev)
    out_ctot = out.shape[1] if out is not None else out_planes.c
    for t in (out, out_planes):
        if t is not None and (tuple(t.shape[2:]) != (hout, wout) or t.shape[0] != n or t.shape[1] != out_ctot):
            raise RuntimeError(f"conv2d_planes: output shape {tuple(t.shape)} incompatible with {(n, out_ctot, hout, wout)}")
    post_f = post if (post is not None and not isinstance(post, Planes)) else None
    post_s = post if isinstance(post, Planes) else None
    for t, nm in ((pre, "pre"), (post, "post")):
        if t is not None and tuple(t.shape) != (n, cout, hout, wout):
            raise RuntimeError(f"conv2d_planes: {nm} shape {tuple(t.shape)} != {(n, cout, hout, wout)}")
    d = ConvDescFor(n, cin, hin, win, in_ctot, in_coff, cout, k, stride, pad, dil, out_ctot, out_coff, act1, slope1, act2, slope2, 4 if x6 else 3)
    if TIMER is not None:
        e0, e1 = TIMER.span("conv_fwd_bf16x6" if x6 else "conv_fwd_bf16x3", 2.0 * n * hout * wout * cout * K, (n, cin, hin, cout, k, stride, dil),
                            4 * n * cout * hout * wout * ((pre is not None) + (post is not None)))
        e0.record()
    args = (ptr(xs.data), ptr(wp.data if x6 else wp.for_planes()), ptr(scale), ptr(shift), ptr(pre), ptr(post_f), ptr(post_s.data) if post_s is not None else None,
            ptr(out), ptr(out_planes.data) if out_planes is not None else None)
    if AUTOTUNE and not _planes_tuned(d):
        targs, scratch = args, None
        if post_s is not None and out_planes is not None and out_planes.data.data_ptr() == post_s.data.data_ptr():
            # in-place residual (the output overwrites the residual operand): the tuner's repeated launches must not touch it. The scratch
            # output is held in a local until the tuner returns (a temporary's block goes back to the caching allocator at once
            # and `_conv_workspace` inside the tuner could be handed the same memory)
            scratch = Planes(out_planes.n, out_planes.c, out_planes.h, out_planes.w, dev)
            targs = args[:8] + (ptr(scratch.data),)
        _autotune_planes(d, targs)
        del scratch
    plain = (scale is None and shift is None and pre is None and post is None and act1 == ACT_NONE and act2 == ACT_NONE and not x6
             and out is not None and out_planes is None and out_coff == 0 and out.shape[1] == cout and xs.slot is None)
    if not (stats_out is not None and plain and _launch_with_stats(d, xs, ptr(wp.for_planes()), out, stats_out)):
        need = _hip.load().cl4_conv2d_fwd_workspace_bytes(C.byref(d))
        ws = _conv_workspace(dev, need) if need else None
        call("cl4_conv2d_fwd_planes", C.byref(d), *args, ptr(ws), need, stream())
    if TIMER is not None:
        e1.record()
    return out, out_planes


def ConvDescFor(n, cin, hin, win, in_ctot, in_coff, cout, k, stride, pad, dil, out_ctot, out_coff, act1, slope1, act2, slope2, precision):
    d = ConvDesc()
    d.n, d.cin, d.hin, d.win = n, cin, hin, win
    d.cout, d.kh, d.kw = cout, k, k
    d.stride, d.pad, d.dil = stride, pad, dil
    d.hout = conv_out_size(hin, k, stride, pad, dil)
    d.wout = conv_out_size(win, k, stride, pad, dil)
    d.in_ctot, d.in_coff = in_ctot, in_coff
    d.out_ctot, d.out_coff = out_ctot, out_coff
    d.act1, d.slope1, d.act2, d.slope2 = act1, slope1, act2, slope2
    d.precision = precision
    return d


# Arithmetic of the weight gradients: 1 = bf16x3 split products (phase 2's trainable branch: ~4e-6 relative, 1.4x faster),
# 0 = exact fp32 MFMA (step 0 / phase 1, where the whole network trains and heavily cancelling sums need fp32 products).
WGRAD_PRECISION = int(os.environ.get("CL4_WGRAD_PRECISION", "1"))


# The 3x3 weight gradients of wide layers as one K-major GEMM on the LDS-DMA planes kernel (csrc/wgrad_gemm.hip, r03): 0 = always the
# role-split conv_wgrad kernels.
WGRAD_GEMM = os.environ.get("CL4_WGRAD_GEMM", "1") != "0"


def _wgrad_gemm_ok(n, cin, hin, win, cout, k, stride, precision):
    kq = n * hin * win
    kpad = (kq + 31) // 32 * 32
    return (WGRAD_GEMM and k == 3 and stride == 1 and precision == 1 and cin >= 1024 and cin % 128 == 0 and cout % 16 == 0
            and 9 * cout >= 1024 and kq >= 4096 and kpad * max(cin, 9 * cout) * 4 < (1 << 31))


def _wgrad_gemm(x, dy, dw, n, cin, in_coff, cout, dy_coff, hin, win, hout, wout, pad, dil, kcache):
    """dw += the 3x3 weight gradient as Y[ci][(tap, co)] = xK . dyK^T (cl4_kpack x 2, cl4_gemm_kplanes, cl4_wgrad_scatter).
    `kcache`: a dict the caller keeps for as long as x's CONTENTS stay what they are (one backward pass): x's pack is made once."""
    lib = _hip.load()
    dev = x.device
    kq = n * hin * win
    kpad = (kq + 31) // 32 * 32
    key = ("xk", x.data_ptr(), tuple(x.shape), in_coff, cin)
    xk = kcache.get(key) if kcache is not None else None
    if xk is None:
        xk = torch.empty(lib.cl4_kpack_bytes(cin, kq) // 4, dtype=torch.int32, device=dev)
        call("cl4_kpack", ptr(x), ptr(xk), n, x.shape[1], in_coff, cin, hin, win, hin, win, 1, 1, 0, stream())
        if kcache is not None:
            kcache[key] = xk
    dyk = torch.empty(lib.cl4_kpack_bytes(9 * cout, kq) // 4, dtype=torch.int32, device=dev)
    call("cl4_kpack", ptr(dy), ptr(dyk), n, dy.shape[1], dy_coff, cout, hout, wout, hin, win, 3, dil, pad, stream())
    y = torch.empty((cin, 9 * cout), dtype=torch.float32, device=dev)
    call("cl4_gemm_kplanes", ptr(xk), ptr(dyk), ptr(y), cin, 9 * cout, kpad, stream())
    call("cl4_wgrad_scatter", ptr(y), ptr(dw), cin, cout, 9, stream())


class WgradSide:
    """Weight gradients on a second stream (r05): nothing in a backward pass reads a layer's dw before the update, so inside
    `with ops.WGRAD_SIDE.on(stream):` every conv2d_wgrad launch goes to `stream` behind an event of the launching stream, and the input
    gradients / ABN backward kernels of the following layers run beside it; `join()` (also the end of the `with` block) makes the
    launching stream wait for all of them. The operands of every forked launch are kept alive until the join (they were allocated on
    the launching stream: the allocator must not hand their memory out while the side stream still reads it). Works the same eagerly
    and under HIP-graph capture (fork and join are inside the capture)."""

    def __init__(self):
        self.stream = None
        self.keep = []
        self.launches = 0

    def on(self, stream):
        import contextlib

        @contextlib.contextmanager
        def scope():
            if stream is None or self.stream is not None:
                yield self
                return
            self.stream = stream
            try:
                yield self
            finally:
                try:
                    self.join()
                finally:
                    self.stream = None
        return scope()

    def join(self):
        if self.stream is not None and self.keep:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()


WGRAD_SIDE = WgradSide()
WGRAD_SIDE_ON = os.environ.get("CL4_WGRAD_SIDE", "0") != "0"       # (measured: no gain at 16 images per GPU, FINDINGS.md)


def conv2d_wgrad(x, dy, dw, k, stride=1, pad=0, dil=1, in_coff=0, cin=None, dy_coff=0, cout=None, precision=None, kcache=None):
    """dw[cout,cin,k,k] += conv weight gradient. dw is accumulated into (zero it first). `kcache` (optional dict, valid while x's
    contents do not change): lets several layers on one input share x's repacked copy on the GEMM path (_wgrad_gemm)."""
    side = WGRAD_SIDE.stream
    if side is not None and TIMER is None and x.is_cuda and torch.cuda.current_stream() != side:      # (per-layer timings: one stream)
        side.wait_stream(torch.cuda.current_stream())
        WGRAD_SIDE.keep.append((x, dy, dw))
        WGRAD_SIDE.launches += 1
        with torch.cuda.stream(side):
            return conv2d_wgrad(x, dy, dw, k, stride, pad, dil, in_coff, cin, dy_coff, cout, precision, kcache)
    n, in_ctot, hin, win = x.shape
    if cin is None:
        cin = in_ctot - in_coff
    if cout is None:
        cout = dy.shape[1] - dy_coff
    if tuple(dw.shape) != (cout, cin, k, k):
        raise RuntimeError(f"conv2d_wgrad: dw shape {tuple(dw.shape)} != {(cout, cin, k, k)}")
    d = _conv_desc(x, cin, in_coff, cout, k, stride, pad, dil, dy.shape[1], dy_coff, 0, 0.0, 0, 0.0,
                   WGRAD_PRECISION if precision is None else precision)
    if tuple(dy.shape[2:]) != (d.hout, d.wout) or dy.shape[0] != n:
        raise RuntimeError("conv2d_wgrad: dy spatial shape does not match the convolution geometry")
    if TIMER is not None:
        e0, e1 = TIMER.span("conv_wgrad", 2.0 * n * d.hout * d.wout * cout * cin * k * k, (n, cin, hin, cout, k, stride, dil))
        e0.record()
    if _wgrad_gemm_ok(n, cin, hin, win, cout, k, stride, d.precision) and x.is_contiguous() and dy.is_contiguous() and dw.is_contiguous():
        _wgrad_gemm(x, dy, dw, n, cin, in_coff, cout, dy_coff, hin, win, d.hout, d.wout, pad, dil, kcache)
    else:
        call("cl4_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), stream())
    if TIMER is not None:
        e1.record()
    return dw


# ---------------------------------------------------------------------------------------------
# spatial / streaming kernels
# ---------------------------------------------------------------------------------------------
def maxpool3x3s2(x):
    n, c, h, w = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    call("cl4_maxpool3x3s2", ptr(x), ptr(y), n * c, h, w, ho, wo, stream())
    return y


def pack_stem_weight(w):
    """[64, 3, 7, 7] -> the MFMA fragment image cl4_stem_fused reads (48 KB)."""
    _f32(w, "pack_stem_weight")
    if tuple(w.shape) != (64, 3, 7, 7):
        raise RuntimeError(f"pack_stem_weight: a [64, 3, 7, 7] weight expected, got {tuple(w.shape)}")
    wq = torch.empty(_hip.load().cl4_stem_weight_bytes(), dtype=torch.uint8, device=w.device)
    call("cl4_pack_stem_weight", ptr(w.contiguous()), ptr(wq), stream())
    return wq


FUSED_BLOCK = os.environ.get("CL4_FUSED_BLOCK", "1") != "0"    # r04: identity bottlenecks of res2 / res3 as one launch (csrc/conv_f.hip)
# Middle widths that run fused (built: 64 = res2, 128 = res3). Default 64 only: the 128-channel variant (one 8-wave workgroup per CU, 116 KB of
# y1 in LDS, 800 / 400 tiles = 3.1 / 1.6 rounds) is bit-exact but no faster than the three tuned launches — 385 vs 367 us at 32 images, 183 vs
# 182 us at 16 (tools/probes/probe_f.py, FINDINGS.md finding 38) — so res3 stays on three launches; CL4_FUSED_CM=64,128 selects it for experiments.
FUSED_CM = tuple(int(v) for v in os.environ.get("CL4_FUSED_CM", "64").split(",") if v)
FUSED_PROJ = os.environ.get("CL4_FUSED_PROJ", "1") != "0"      # the stage's first block (projection shortcut, width 64) fused as well


def bottleneck_fused(xs, w1, w2, w3, aff, slope1, slope2, slope_out, want_fp32=False, out_planes=None, f32_images=None, wp=None):
    """One ResidualBlock of a frozen network (modules/residual.py:84-97) as ONE launch (csrc/conv_f.hip). Identity block (`wp` None): `xs`
    planes input [n, 4 cm, h, w]; w1 / w2 / w3 bf16x3 packed weights of conv1 (1x1), conv2 (3x3, stride 1, dilation 1), conv3 (1x1); `aff`
    the folded ABN maps [scale1 | shift1 | scale2 | shift2 | scale3 | shift3]. Block with a projection shortcut (`wp`: its packed weight,
    cm -> 4 cm, stride 1): `xs` has cm channels, `aff` carries [scale_p | shift_p] behind the others. Returns (fp32 output or None, planes
    output). `f32_images`: the fp32 copy is written for the first so many images only (the rest of the returned tensor is uninitialised)."""
    if not isinstance(xs, Planes) or xs.planes != 2:
        raise RuntimeError("bottleneck_fused: two-plane Planes input expected")
    n, cin, h, w = xs.shape
    cm = cin if wp is not None else cin // 4
    c4 = 4 * cm
    for pw, K, co in ((w1, cin, cm), (w2, 9 * cm, cm), (w3, cm, c4)) + (((wp, cin, c4),) if wp is not None else ()):
        if not isinstance(pw, PackedWeight) or pw.precision != 1 or pw.K != K or pw.cout != co or (planes_f16() and pw.planes is None):
            raise RuntimeError("bottleneck_fused: bf16x3 packed weights of a [c_in -> cm -> cm (3x3) -> 4 cm] bottleneck (+ c_in -> 4 cm shortcut) expected")
    if aff.numel() != 4 * cm + 2 * c4 + (2 * c4 if wp is not None else 0):
        raise RuntimeError("bottleneck_fused: aff must hold 4 cm + 8 cm (+ 8 cm with a projection shortcut) floats")
    dev = xs.data.device
    ys = Planes(n, c4, h, w, dev) if out_planes is None else out_planes
    if ys.data.data_ptr() == xs.data.data_ptr():
        raise RuntimeError("bottleneck_fused: the output must not alias the input")
    y = torch.empty((n, c4, h, w), dtype=torch.float32, device=dev) if want_fp32 else None
    if TIMER is not None:     # the block's convolutions' algorithmic FLOPs, one span (shape key: k = 0 marks the fusion)
        e0, e1 = TIMER.span("conv_fwd_bf16x3", 2.0 * n * h * w * (cin * cm + 9 * cm * cm + cm * c4 + (cin * c4 if wp is not None else 0)),
                            (n, cin, h, c4, 0, 1, 1), 0)
        e0.record()
    f32n = -1 if f32_images is None else int(f32_images)
    if wp is None:
        call("cl4_bottleneck_fused", ptr(xs.data), ptr(ys.data), ptr(y), f32n, ptr(w1.for_planes()), ptr(w2.for_planes()), ptr(w3.for_planes()), ptr(aff), n, cm, h, w,
             C.c_float(slope1), C.c_float(slope2), C.c_float(slope_out), stream())
    else:
        call("cl4_bottleneck_fused_proj", ptr(xs.data), ptr(ys.data), ptr(y), f32n, ptr(w1.for_planes()), ptr(w2.for_planes()), ptr(w3.for_planes()), ptr(wp.for_planes()),
             ptr(aff), n, cm, h, w, C.c_float(slope1), C.c_float(slope2), C.c_float(slope_out), stream())
    if TIMER is not None:
        e1.record()
    return y, ys


def stem_fused(x, wq, scale, shift, slope=0.01, want_fp32=True, want_planes=True):
    """conv 7x7/2 (3 -> 64) + folded eval ABN + leaky-ReLU + max-pool 3x3/2 in one kernel (csrc/stem.hip). Returns (fp32 output or
    None, planes output or None)."""
    _f32(x, "stem_fused.x")
    n, c, h, w = x.shape
    if c != 3 or not x.is_contiguous():
        raise RuntimeError("stem_fused: contiguous [n, 3, h, w] input expected")
    hp, wp = conv_out_size(conv_out_size(h, 7, 2, 3, 1), 3, 2, 1, 1), conv_out_size(conv_out_size(w, 7, 2, 3, 1), 3, 2, 1, 1)
    y = torch.empty((n, 64, hp, wp), dtype=torch.float32, device=x.device) if want_fp32 else None
    ys = Planes(n, 64, hp, wp, x.device) if want_planes else None
    if TIMER is not None:     # the convolution's algorithmic FLOPs (147 real taps); the fused ABN / pooling are not counted
        hc, wc = conv_out_size(h, 7, 2, 3, 1), conv_out_size(w, 7, 2, 3, 1)
        e0, e1 = TIMER.span("conv_fwd_bf16x3", 2.0 * n * hc * wc * 64 * 147, (n, 3, h, 64, 7, 2, 1))
        e0.record()
    call("cl4_stem_fused", ptr(x), ptr(wq), ptr(scale), ptr(shift), C.c_float(slope), ptr(y), ptr(ys.data) if ys is not None else None,
         n, h, w, stream())
    if TIMER is not None:
        e1.record()
    return y, ys


def avgpool_s1(x, kh, kw):
    n, c, h, w = x.shape
    y = torch.empty((n, c, h - kh + 1, w - kw + 1), dtype=torch.float32, device=x.device)
    call("cl4_avgpool_s1", ptr(x), ptr(y), n * c, h, w, kh, kw, stream())
    return y


def replicate_pad(x, pad_l, pad_r, pad_t, pad_b):
    n, c, h, w = x.shape
    y = torch.empty((n, c, h + pad_t + pad_b, w + pad_l + pad_r), dtype=torch.float32, device=x.device)
    call("cl4_replicate_pad", ptr(x), ptr(y), n * c, h, w, pad_t, pad_l, y.shape[2], y.shape[3], stream())
    return y


def global_mean(x):
    n, c, h, w = x.shape
    y = torch.empty((n, c, 1, 1), dtype=torch.float32, device=x.device)
    call("cl4_global_mean", ptr(x), ptr(y), n * c, h * w, stream())
    return y


def plane_sum(x):
    """[n, c, h, w] -> [n, c, 1, 1] sums (fp64 partial sums)."""
    n, c, h, w = x.shape
    y = torch.empty((n, c, 1, 1), dtype=torch.float32, device=x.device)
    call("cl4_plane_sum", ptr(x), ptr(y), n * c, h * w, stream())
    return y


def resize_bilinear(x, size, align_corners, out=None, out_coff=0, in_coff=0, c=None):
    n, in_ctot, hin, win = x.shape
    if c is None:
        c = in_ctot - in_coff
    ho, wo = size
    if out is None:
        out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
        out_coff = 0
    call("cl4_resize_bilinear_fwd", ptr(x), ptr(out), n, c, hin, win, ho, wo, int(bool(align_corners)),
         in_ctot, in_coff, out.shape[1], out_coff, stream())
    return out


def resize_bilinear_bwd(dy, in_size, align_corners, dy_coff=0, c=None, dx=None, dx_coff=0, accumulate=False):
    n, dy_ctot, ho, wo = dy.shape
    if c is None:
        c = dy_ctot - dy_coff
    hin, win = in_size
    if dx is None:
        dx = torch.empty((n, c, hin, win), dtype=torch.float32, device=dy.device)
        dx_coff = 0
        accumulate = False
    call("cl4_resize_bilinear_bwd", ptr(dy), ptr(dx), n, c, hin, win, ho, wo, int(bool(align_corners)),
         dx.shape[1], dx_coff, dy_ctot, dy_coff, int(bool(accumulate)), stream())
    return dx


def flip_w(x, out=None):
    if out is None:
        out = torch.empty_like(x)
    w = x.shape[-1]
    call("cl4_flip_w", ptr(x), ptr(out), x.numel() // w, w, stream())
    return out


def axpy_(y, x, alpha=1.0):
    if y.numel() != x.numel():
        raise RuntimeError("axpy_: size mismatch")
    call("cl4_axpy", ptr(y), ptr(x), C.c_float(alpha), y.numel(), stream())
    return y


def scale_(y, alpha):
    """y *= alpha (cl4_axpy with x aliased to y: y += (alpha - 1) * y)."""
    return axpy_(y, y, float(alpha) - 1.0)


def copy_channels(x, out, c, in_coff=0, out_coff=0):
    n, in_ctot = x.shape[:2]
    hw = x.shape[2] * x.shape[3]
    call("cl4_copy_channels", ptr(x), ptr(out), n, c, hw, in_ctot, in_coff, out.shape[1], out_coff, stream())
    return out


# ---------------------------------------------------------------------------------------------
# train-mode BatchNorm / dropout / depthwise conv
# ---------------------------------------------------------------------------------------------
_misc_scratch = _Scratch(1 << 20)


def _workspace(device, nbytes):
    return _misc_scratch.get(device, nbytes)


def bn_workspace_bytes(c):
    return _hip.load().cl4_bn_workspace_bytes(c) + 2 * c * 4


def bn_train_stats(x, c=None, coff=0, running_mean=None, running_var=None, momentum=0.1):
    n, ctot, h, w = x.shape
    if c is None:
        c = ctot - coff
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    ws = _workspace(x.device, bn_workspace_bytes(c))
    call("cl4_bn_train_stats", ptr(x), n, c, h * w, ctot, coff, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
         C.c_float(momentum), ptr(ws), ws.numel(), stream())
    return mean, var


def bn_train_fwd(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, act=ACT_RELU, slope=0.01, keep=None,
                 keep_scale=1.0, c=None, x_coff=0, out=None, out_coff=0, partials=None):
    """bn_train_stats + bn_apply in two launches (cl4_bn_train_fwd): returns (y, mean, var). `partials` = (tensor, nparts) from the
    kernel that produced x (dwconv5_stats): one launch, x is not read for its statistics."""
    n, x_ctot, h, w = x.shape
    if c is None:
        c = x_ctot - x_coff
    if out is None:
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        out_coff = 0
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    if partials is not None:
        call("cl4_bn_train_fwd_partials", ptr(x), ptr(out), n, c, h * w, x_ctot, x_coff, out.shape[1], out_coff, ptr(mean), ptr(var),
             ptr(running_mean), ptr(running_var), C.c_float(momentum), ptr(gamma), ptr(beta), C.c_float(eps), act, C.c_float(slope),
             ptr(keep), C.c_float(keep_scale), ptr(partials[0]), partials[1], stream())
        return out, mean, var
    ws = _workspace(x.device, bn_workspace_bytes(c))
    call("cl4_bn_train_fwd", ptr(x), ptr(out), n, c, h * w, x_ctot, x_coff, out.shape[1], out_coff, ptr(mean), ptr(var),
         ptr(running_mean), ptr(running_var), C.c_float(momentum), ptr(gamma), ptr(beta), C.c_float(eps), act, C.c_float(slope),
         ptr(keep), C.c_float(keep_scale), ptr(ws), ws.numel(), stream())
    return out, mean, var


PRODUCER_PLANES = os.environ.get("CL4_PRODUCER_PLANES", "1") != "0"     # r05: train-mode ABN apply / residual add write the planes twin the next fp16x3 convolution reads
GRAD_PLANES = os.environ.get("CL4_GRAD_PLANES", "1") != "0"             # r05: ABN backward writes the fp16 planes twin of the gradient it hands on
ABN_ONE_RANK = os.environ.get("CL4_ABN_ONE_RANK", "1") != "0"           # r05: train-mode ABN without an exchange step when there is one rank (fewer launches)


def bn_apply(x, mean, var, gamma, beta, eps=1e-5, act=ACT_RELU, slope=0.01, keep=None, keep_scale=1.0,
             c=None, x_coff=0, out=None, out_coff=0, want_planes=False, post=None, act2=ACT_NONE, slope2=0.01):
    """`want_planes` (whole contiguous tensors, c % 8 == 0, fp16x3 build): also write the two-plane twin of the output and attach it as
    `y._cl4_planes` — what ops.conv2d's fp16x3 route multiplies instead of converting y in a pass of its own. `post` / `act2`:
    y = act2(act(bn(x)) + post) in the same pass (whole contiguous tensors, c % 8 == 0)."""
    n, x_ctot, h, w = x.shape
    if c is None:
        c = x_ctot - x_coff
    twin = want_planes and PRODUCER_PLANES and planes_f16()
    if post is not None or twin:
        whole = out is None and keep is None and x_coff == 0 and c == x_ctot and c % 8 == 0 and x.is_contiguous()
        if post is not None and not (whole and post.is_contiguous() and tuple(post.shape) == tuple(x.shape)):
            raise RuntimeError("bn_apply: the fused residual needs whole contiguous tensors with c % 8 == 0")
        if whole:
            out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
            ps = Planes(n, c, h, w, x.device) if twin else None
            call("cl4_bn_apply_planes", ptr(x), ptr(out), ptr(ps.data) if ps is not None else None, n, c, h * w, ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                 C.c_float(eps), act, C.c_float(slope), ptr(post), act2, C.c_float(slope2), stream())
            if ps is not None:
                out._cl4_planes = ps
            return out
    if out is None:
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        out_coff = 0
    call("cl4_bn_apply", ptr(x), ptr(out), n, c, h * w, x_ctot, x_coff, out.shape[1], out_coff, ptr(mean), ptr(var),
         ptr(gamma), ptr(beta), C.c_float(eps), act, C.c_float(slope), ptr(keep), C.c_float(keep_scale), stream())
    return out


def bn_train_bwd(x, dy, mean, var, gamma, beta, dgamma, dbeta, eps=1e-5, act=ACT_RELU, keep=None, keep_scale=1.0,
                 c=None, x_coff=0, dy_coff=0):
    """Returns dx with the layout of x (only the [x_coff, x_coff+c) slice is written when x is a slice)."""
    n, x_ctot, h, w = x.shape
    if c is None:
        c = x_ctot - x_coff
    dx = torch.empty_like(x)
    ws = _workspace(x.device, bn_workspace_bytes(c))
    call("cl4_bn_train_bwd", ptr(x), ptr(dy), ptr(dx), n, c, h * w, x_ctot, x_coff, dy.shape[1], dy_coff, ptr(mean), ptr(var),
         ptr(gamma), ptr(beta), C.c_float(eps), act, ptr(keep), C.c_float(keep_scale), ptr(dgamma), ptr(dbeta),
         ptr(ws), ws.numel(), stream())
    return dx


def _world(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class StatExchange:
    """The cross-rank exchange of InPlaceABNSync's batch statistics (segmentation_module.py:17-19: `group=WORLD`), one packed float64
    vector per layer and direction ([sum x | sum x^2] forward, [sum dy | sum dy * xhat] backward — a layer's statistics depend on the
    normalised output of the layer before it, so layers cannot share an exchange). r05: it goes through the library's own RCCL
    communicator (`cl4_allreduce_f64`, enqueue-only on the compute stream, no torch.distributed call on the data path) whenever
    GradBuckets has brought that communicator up; otherwise `dist.all_reduce` (gloo rehearsals, CL4_COMM=torch). `hook`, when set, is
    handed the exchange as a closure instead of running it: a segmented graph capture (train._SegmentedGraph) cuts the graph there and
    replays the closure between the segments. `count` = exchanges issued so far (Trainer.collectives_per_iteration prints the rate)."""

    def __init__(self):
        self.count = 0
        self.hook = None

    @staticmethod
    def _active(group):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        w = dist.get_world_size(group)
        return w if (w > 1 or os.environ.get("CL4_FORCE_COLLECTIVES") == "1") else 0

    def direct(self, t, group, world):
        return (group is None and t.is_cuda and os.environ.get("CL4_COMM", "native") != "torch"
                and _hip.load().cl4_comm_world() == world)

    def run(self, t, group):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        if self.direct(t, group, world):
            call("cl4_allreduce_f64", ptr(t), t.numel(), stream())
        elif world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    def allreduce(self, t, group):
        if not self._active(group):
            return t
        self.count += 1
        if self.hook is not None:
            self.hook(lambda: self.run(t, group))
        else:
            self.run(t, group)
        return t


STAT_EXCHANGE = StatExchange()


def _allreduce_sum(t, group):
    return STAT_EXCHANGE.allreduce(t, group)


def abn_train_forward(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, act=ACT_LEAKY, slope=0.01,
                      group=None, extra_sums=None, want_planes=False, post=None, act2=ACT_NONE, slope2=0.01, partials=None):
    """InPlaceABNSync in training mode (segmentation_module.py:17-29): batch statistics over every rank of
    `group`, then BN + leaky-ReLU / identity. Returns (y, mean, var, count). `extra_sums` ([c*2] doubles, count) adds the
    statistics of shards held elsewhere — used by the tests to emulate a second rank on one GPU."""
    n, c, h, w = x.shape
    if ABN_ONE_RANK and extra_sums is None and not STAT_EXCHANGE._active(group):
        # r05: nobody to exchange sums with — statistics straight from the reduction's partials (cl4_bn_train_stats: the same arithmetic
        # as fold + all-reduce + final, one launch less per layer)
        if partials is not None:      # the producing convolution left the partial sums (conv2d(stats_out=...)): no pass over x
            part, nparts = partials
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            var = torch.empty_like(mean)
            call("cl4_bn_stats_from_partials", ptr(part), int(nparts), n * h * w, c, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
                 C.c_float(momentum), stream())
        else:
            mean, var = bn_train_stats(x, running_mean=running_mean, running_var=running_var, momentum=momentum)
        y = bn_apply(x, mean, var, gamma, beta, eps=eps, act=act, slope=slope, want_planes=want_planes, post=post, act2=act2, slope2=slope2)
        return y, mean, var, float(n * h * w)
    # the count slot exists only when a caller brings extra shards (an uninitialised double must not go through the all-reduce)
    sums = torch.empty(2 * c + (1 if extra_sums is not None else 0), dtype=torch.float64, device=x.device)
    ws = _workspace(x.device, bn_workspace_bytes(c))
    if partials is not None:          # the producing convolution's partial sums (conv2d(stats_out=...)): no pass over x on this rank
        call("cl4_bn_sync_stats_from_partials", ptr(partials[0]), int(partials[1]), ptr(sums), c, stream())
    else:
        call("cl4_bn_sync_stats_local", ptr(x), n, c, h * w, c, 0, ptr(sums), ptr(ws), ws.numel(), stream())
    if extra_sums is not None:      # (a host scalar written into a device tensor is a synchronous copy: not inside a HIP-graph capture)
        sums[2 * c] = float(n * h * w)
        sums += extra_sums
    _allreduce_sum(sums, group)
    # equal shards on every rank (run.py:66 splits the batch evenly): the global count needs no host round trip
    count = float(sums[2 * c].item()) if extra_sums is not None else float(n * h * w) * _world(group)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty(c, dtype=torch.float32, device=x.device)
    call("cl4_bn_sync_stats_final", ptr(sums), C.c_double(count), ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
         C.c_float(momentum), c, stream())
    y = bn_apply(x, mean, var, gamma, beta, eps=eps, act=act, slope=slope, want_planes=want_planes, post=post, act2=act2, slope2=slope2)
    return y, mean, var, count


def abn_train_backward(x, dy, mean, var, gamma, beta, dgamma, dbeta, count, eps=1e-5, act=ACT_LEAKY, slope=0.01, group=None,
                       extra_sums=None, return_sums=False, want_planes=False, gamma_sign=None):
    """Backward of abn_train_forward: dx; this rank's share of the affine gradients is accumulated into dgamma / dbeta. `want_planes`
    (r05): dx also leaves as the fp16x3 operand of the input-gradient convolution in front (`dx._cl4_planes`, its power-of-two scale from
    a bound the reduction pass provides — cl4_bn_sync_bwd_apply_planes): no max |dx| pass, no conversion pass. `gamma_sign` (InPlaceABN:
    `gamma` is |weight| + eps): the raw weight — dgamma then receives d(weight) = sign(weight) d(gamma). With nobody to exchange sums with
    (one rank, no extra shards) the whole backward is cl4_bn_train_bwd_act: two launches, three with the twin."""
    n, c, h, w = x.shape
    want_planes = bool(want_planes) and c % 8 == 0 and planes_f16() and x.is_contiguous()
    dx = torch.empty_like(x)
    ps = None
    if want_planes:
        slot = torch.empty(_hip.load().cl4_to_planes_f16_slot_floats(), dtype=torch.float32, device=x.device)
        ps = Planes(n, c, h, w, x.device, planes=2, slot=slot)
    if ABN_ONE_RANK and extra_sums is None and not return_sums and not STAT_EXCHANGE._active(group) and count == float(n * h * w):
        ws = _workspace(x.device, _hip.load().cl4_bn_train_bwd_act_workspace_bytes(c))
        call("cl4_bn_train_bwd_act", ptr(x), ptr(dy), ptr(dx), ptr(ps.data) if ps is not None else None, ptr(ps.slot) if ps is not None else None,
             n, c, h * w, dy.shape[1], 0, ptr(mean), ptr(var), ptr(gamma), ptr(beta), C.c_float(eps), act, C.c_float(slope), ptr(dgamma), ptr(dbeta),
             ptr(gamma_sign), ptr(ws), ws.numel(), stream())
        if ps is not None:
            dx._cl4_planes = ps
        return dx
    sums = torch.empty(2 * c, dtype=torch.float64, device=x.device)
    ws = _workspace(x.device, bn_workspace_bytes(c))
    if want_planes:
        maxima = torch.empty(_hip.load().cl4_bn_bwd_maxima_floats(c), dtype=torch.float32, device=x.device)
        call("cl4_bn_sync_bwd_local_max", ptr(x), ptr(dy), n, c, h * w, dy.shape[1], 0, ptr(mean), ptr(var), ptr(gamma), ptr(beta),
             C.c_float(eps), act, C.c_float(slope), ptr(sums), ptr(dgamma), ptr(dbeta), ptr(gamma_sign), ptr(maxima), ptr(ws), ws.numel(), stream())
    else:
        call("cl4_bn_sync_bwd_local", ptr(x), ptr(dy), n, c, h * w, c, 0, dy.shape[1], 0, ptr(mean), ptr(var), ptr(gamma), ptr(beta),
             C.c_float(eps), act, C.c_float(slope), None, C.c_float(1.0), ptr(sums), ptr(dgamma), ptr(dbeta), ptr(gamma_sign), ptr(ws), ws.numel(),
             stream())
    local = sums.clone() if return_sums else None
    if extra_sums is not None:
        sums += extra_sums
    _allreduce_sum(sums, group)
    if want_planes:
        # (the bound uses this rank's maxima and the global sums: each rank scales its own dx)
        call("cl4_bn_sync_bwd_apply_planes", ptr(x), ptr(dy), ptr(dx), ptr(ps.data), ptr(ps.slot), n, c, h * w, dy.shape[1], 0, ptr(mean), ptr(var),
             ptr(gamma), ptr(beta), C.c_float(eps), act, C.c_float(slope), ptr(sums), C.c_double(count), ptr(maxima), ptr(ws), ws.numel(), stream())
        dx._cl4_planes = ps
    else:
        call("cl4_bn_sync_bwd_apply", ptr(x), ptr(dy), ptr(dx), n, c, h * w, c, 0, dy.shape[1], 0, ptr(mean), ptr(var), ptr(gamma),
             ptr(beta), C.c_float(eps), act, C.c_float(slope), None, C.c_float(1.0), ptr(sums), C.c_double(count), ptr(ws),
             ws.numel(), stream())
    return (dx, local) if return_sums else dx


def relu_bwd(y, dy):
    dx = torch.empty_like(dy)
    call("cl4_relu_bwd", ptr(y), ptr(dy), ptr(dx), y.numel(), stream())
    return dx


def bias_grad_(db, dy, c=None, coff=0):
    n, ctot, h, w = dy.shape
    if c is None:
        c = ctot - coff
    call("cl4_bias_grad", ptr(dy), ptr(db), n, c, h * w, ctot, coff, stream())
    return db


def dropout_mask(shape, p, seed, device):
    keep = torch.empty(shape, dtype=torch.uint8, device=device)
    call("cl4_dropout_mask", ptr(keep), keep.numel(), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_float(p), stream())
    return keep


def dropout_mask_dev(shape, p, seed_dev, advance, device):
    """Keep mask from a device-resident seed (uint64 tensor [1]) that the launch advances: graph-replay safe."""
    keep = torch.empty(shape, dtype=torch.uint8, device=device)
    call("cl4_dropout_mask_dev", ptr(keep), keep.numel(), ptr(seed_dev), C.c_uint64(advance), C.c_float(p), stream())
    return keep


def dwconv5(x, w, flip=False, c=None, x_coff=0, out=None, out_coff=0):
    n, x_ctot, h, wd = x.shape
    if c is None:
        c = x_ctot - x_coff
    if tuple(w.shape) != (c, 1, 5, 5):
        raise RuntimeError(f"dwconv5: weight shape {tuple(w.shape)} != {(c, 1, 5, 5)}")
    if out is None:
        out = torch.empty((n, c, h, wd), dtype=torch.float32, device=x.device)
        out_coff = 0
    call("cl4_dwconv5_fwd", ptr(x), ptr(w), ptr(out), n, c, h, wd, x_ctot, x_coff, out.shape[1], out_coff, int(flip), stream())
    return out


def dwconv5_stats(x, w):
    """dwconv5 (whole tensor) that also returns the per-channel partial sums of its output for the train-mode BatchNorm behind it:
    (y, partial, nparts) — hand (partial, nparts) to bn_train_fwd(partials=...)."""
    n, c, h, wd = x.shape
    if tuple(w.shape) != (c, 1, 5, 5):
        raise RuntimeError(f"dwconv5_stats: weight shape {tuple(w.shape)} != {(c, 1, 5, 5)}")
    out = torch.empty((n, c, h, wd), dtype=torch.float32, device=x.device)
    nparts = _hip.load().cl4_dwconv5_stats_parts(n, h, wd)
    partial = torch.empty(c * nparts * 2, dtype=torch.float64, device=x.device)
    call("cl4_dwconv5_fwd_stats", ptr(x), ptr(w), ptr(out), n, c, h, wd, c, 0, c, 0, ptr(partial), stream())
    return out, partial, nparts


def dwconv5_wgrad_(dw, x, dy, c=None, x_coff=0, dy_coff=0):
    n, x_ctot, h, wd = x.shape
    if c is None:
        c = x_ctot - x_coff
    call("cl4_dwconv5_wgrad", ptr(x), ptr(dy), ptr(dw), n, c, h, wd, x_ctot, x_coff, dy.shape[1], dy_coff, stream())
    return dw


# ---------------------------------------------------------------------------------------------
# label stage / losses / optimiser
# ---------------------------------------------------------------------------------------------
def cam_smooth(int_masks, pg_weight, pg_bias, label_new, alpha=0.7):
    """PeakGenerator eval + cam_normalize + smoothing; returns the new-class CAM [b, cnew, h, w]."""
    b, ctot, h, w = int_masks.shape
    cnew = pg_weight.shape[0]
    cam = torch.empty((b, cnew, h, w), dtype=torch.float32, device=int_masks.device)
    need = _hip.load().cl4_cam_smooth_workspace_bytes(b, cnew, h, w)
    ws = _workspace(int_masks.device, need) if need else None
    call("cl4_cam_smooth", ptr(int_masks), b, ctot, h, w, ptr(pg_weight.reshape(cnew, cnew)), ptr(pg_bias), ptr(label_new),
         cnew, C.c_float(alpha), ptr(cam), ptr(ws), ws.numel() if ws is not None else 0, stream())
    return cam


def cam_normalize(int_masks, pg_weight, pg_bias, label_new, alpha=0.7):
    """PeakGenerator eval CAM of the new classes WITHOUT smoothing: [b, cnew, h, w] (see cl4_cam_normalize)."""
    b, ctot, h, w = int_masks.shape
    cnew = label_new.shape[1]
    cam = torch.empty((b, cnew, h, w), dtype=torch.float32, device=int_masks.device)
    ws = _workspace(int_masks.device, _hip.load().cl4_cam_normalize_workspace_bytes(b, cnew, h, w))
    call("cl4_cam_normalize", ptr(int_masks), b, ctot, h, w, ptr(pg_weight.reshape(cnew, cnew)), ptr(pg_bias), ptr(label_new),
         cnew, C.c_float(alpha), ptr(cam), ptr(ws), ws.numel(), stream())
    return cam


def smoothing3x3(x):
    """`smoothing` of wss/utils.py:28-32 on [b, c, h, w]."""
    _f32c(x, "smoothing x")
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    call("cl4_smoothing3x3", ptr(x), b * c, h, w, ptr(out), stream())
    return out


def peak_extract(cam, label_new, size, kernel=15, K=25):
    """Upsample-to-`size` + NMS + top-K. Returns device tensors conf f32, ys i32, xs i32 of shape [b, c, K]."""
    b, c, h, w = cam.shape
    H, W = size
    conf = torch.empty((b, c, K), dtype=torch.float32, device=cam.device)
    ys = torch.empty((b, c, K), dtype=torch.int32, device=cam.device)
    xs = torch.empty_like(ys)
    nbytes = _hip.load().cl4_peak_workspace_bytes(b, c, H, W, K)
    ws = _workspace(cam.device, nbytes)
    call("cl4_peak_extract", ptr(cam), ptr(label_new), b, c, h, w, H, W, kernel, K, ptr(conf), ptr(ys), ptr(xs), ptr(ws),
         ws.numel(), stream())
    return conf, ys, xs


def seg_pseudo_gt(seg2b, l1h, old_classes, size, no_mask=False):
    n2, Cc, h, w = seg2b.shape
    bs = n2 // 2
    H, W = size
    dev = seg2b.device
    seg_gt = torch.empty((bs, H, W), dtype=torch.uint8, device=dev)
    old_w = torch.empty((bs, 1, H, W), dtype=torch.float32, device=dev)
    prob_win = torch.empty((bs, H, W), dtype=torch.float32, device=dev)
    call("cl4_seg_pseudo_gt", ptr(seg2b), ptr(l1h), bs, Cc, h, w, H, W, old_classes, int(no_mask), ptr(seg_gt), ptr(old_w),
         ptr(prob_win), stream())
    return seg_gt, old_w, prob_win


def ccl(cls_map, connectivity=8):
    b, H, W = cls_map.shape
    labels = torch.empty((b, H, W), dtype=torch.int32, device=cls_map.device)
    call("cl4_ccl", ptr(cls_map), b, H, W, connectivity, ptr(labels), stream())
    return labels


def cc_stats(labels):
    b, H, W = labels.shape
    area = torch.empty((b, H, W), dtype=torch.int32, device=labels.device)
    sumx = torch.empty_like(area)
    sumy = torch.empty_like(area)
    call("cl4_cc_stats", ptr(labels), b, H, W, ptr(area), ptr(sumx), ptr(sumy), stream())
    return area, sumx, sumy


_label_ws = {}


def label_generation(seg_gt, prob_win, label_new, peaks, center_lo, offset_lo, gauss, old_classes, sigma=6,
                     pseudo_thresh=0.7, refine_thresh=0.3, nms_kernel=41, beta=3.0, run_refine=True):
    """Batch pseudo_label_generation + refine_label_generation. Returns (pseudo, refined) dicts of device tensors
    with keys center [b,cnew,H,W], offset [b,2,H,W], weight [b,1,H,W]; refined is None when run_refine is False."""
    b, H, W = seg_gt.shape
    cnew = label_new.shape[1]
    dev = seg_gt.device
    conf, ys, xs = peaks
    new = lambda c: torch.empty((b, c, H, W), dtype=torch.float32, device=dev)
    pseudo = {"center": new(cnew), "offset": new(2), "weight": new(1)}
    refined = {"center": new(cnew), "offset": new(2), "weight": new(1)} if run_refine else None
    d = _hip.LabelIO()
    d.b, d.H, d.W, d.cnew, d.old_classes = b, H, W, cnew, old_classes
    d.hc, d.wc, d.center_ctot = center_lo.shape[2], center_lo.shape[3], center_lo.shape[1]
    d.K, d.sigma, d.nms_kernel, d.run_refine = conf.shape[2], sigma, nms_kernel, int(run_refine)
    d.pseudo_thresh, d.refine_thresh, d.beta = pseudo_thresh, refine_thresh, beta
    pv = lambda t: ptr(t).value if t is not None else None
    d.seg_gt, d.prob_win, d.label_new = pv(seg_gt), pv(prob_win), pv(label_new)
    d.peak_conf, d.peak_y, d.peak_x = pv(conf), pv(ys), pv(xs)
    d.center_lo, d.offset_lo, d.gauss = pv(center_lo), pv(offset_lo), pv(gauss)
    d.p_center, d.p_offset, d.p_weight = pv(pseudo["center"]), pv(pseudo["offset"]), pv(pseudo["weight"])
    if run_refine:
        d.r_center, d.r_offset, d.r_weight = pv(refined["center"]), pv(refined["offset"]), pv(refined["weight"])
    nbytes = _hip.load().cl4_label_workspace_bytes(b, H, W)
    # one buffer per (device, size), kept for the life of the process: the label kernels keep their counters in it between
    # launches, and a captured step addresses it on every replay — an eager call at another batch size (the short last batch of
    # an epoch) used to clear this table and free the buffer under the graph (r02)
    key = (dev.index, nbytes)
    ws = _label_ws.get(key)
    if ws is None:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _label_ws[key] = ws
    call("cl4_label_generation", C.byref(d), ptr(ws), ws.numel(), stream())
    return pseudo, refined


def label_counters(b, H, W, device):
    """{n_contour, n_psplat, n_center, n_cand, overflow} of the last label_generation (forces a sync)."""
    nbytes = _hip.load().cl4_label_workspace_bytes(b, H, W)
    ws = _label_ws[(device.index, nbytes)]
    out = torch.empty(5, dtype=torch.int32, device=device)
    call("cl4_label_counters", ptr(ws), b, H, W, ptr(out), stream())
    return dict(zip(("n_contour", "n_psplat", "n_center", "n_cand", "overflow"), out.cpu().tolist()))


def label_counters_accumulate_(acc5, b, H, W):
    """acc5 (int32[5], device): running max of {n_contour, n_psplat, n_center, n_cand} and OR of the overflow flag of the
    last label_generation. Enqueue-only."""
    nbytes = _hip.load().cl4_label_workspace_bytes(b, H, W)
    ws = _label_ws[(acc5.device.index, nbytes)]
    call("cl4_label_counters_accumulate", ptr(ws), b, H, W, ptr(acc5), stream())
    return acc5


LABEL_COUNTER_NAMES = ("n_contour", "n_psplat", "n_center", "n_cand", "overflow")


def p2_loss(center, offset, old_center, old_offset, old_w, pseudo, refined, size, center_weight=200.0,
            offset_weight=0.01, with_grad=True):
    """Returns (losses [4] device tensor: center_old, offset_old, center_new, offset_new; d_center; d_offset)."""
    b, ctot, hc, wc = center.shape
    cnew = pseudo["center"].shape[1]
    cold = ctot - cnew
    H, W = size
    dev = center.device
    losses = torch.empty(4, dtype=torch.float32, device=dev)
    dcen = torch.empty_like(center) if with_grad else None
    doff = torch.empty_like(offset) if with_grad else None
    d = _hip.P2LossIO()
    d.b, d.H, d.W, d.hc, d.wc, d.cold, d.cnew = b, H, W, hc, wc, cold, cnew
    pv = lambda t: ptr(t).value if t is not None else None
    d.center, d.offset, d.old_center, d.old_offset, d.old_w = pv(center), pv(offset), pv(old_center), pv(old_offset), pv(old_w)
    d.p_center, d.p_offset, d.p_weight = pv(pseudo["center"]), pv(pseudo["offset"]), pv(pseudo["weight"])
    if refined is not None:
        d.r_center, d.r_offset, d.r_weight = pv(refined["center"]), pv(refined["offset"]), pv(refined["weight"])
    d.center_weight, d.offset_weight = center_weight, offset_weight
    d.losses, d.d_center, d.d_offset = pv(losses), pv(dcen), pv(doff)
    nbytes = _hip.load().cl4_p2_loss_workspace_bytes(b, H, W, cold, cnew, int(with_grad))
    ws = _workspace(dev, nbytes)
    call("cl4_p2_loss", C.byref(d), ptr(ws), ws.numel(), stream())
    return losses, dcen, doff


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    n = p.numel()
    if not (g.numel() == n and m.numel() == n and v.numel() == n):
        raise RuntimeError("adam_step_: size mismatch")
    call("cl4_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), n, C.c_float(lr), C.c_float(beta1), C.c_float(beta2),
         C.c_float(eps), C.c_float(weight_decay), int(step), stream())


def adam_step_dev_(p, g, m, v, lr_dev, beta1, beta2, eps, weight_decay, step_dev, advance):
    call("cl4_adam_step_dev", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(lr_dev), C.c_float(beta1), C.c_float(beta2),
         C.c_float(eps), C.c_float(weight_decay), ptr(step_dev), int(advance), stream())


def sgd_step_(p, g, buf, lr, momentum=0.9, weight_decay=0.0, nesterov=True, lr_dev=None):
    """torch.optim.SGD step on flat fp32 segments (momentum buffer `buf` starts at zero)."""
    n = p.numel()
    if g.numel() != n or (buf is not None and buf.numel() != n):
        raise RuntimeError("sgd_step_: size mismatch")
    call("cl4_sgd_step", ptr(p), ptr(g), ptr(buf), n, C.c_float(lr), ptr(lr_dev), C.c_float(momentum),
         C.c_float(weight_decay), int(bool(nesterov)), stream())


def zero_(t):
    call("cl4_fill_zero", ptr(t), t.numel() * t.element_size(), stream())
    return t


# ---------------------------------------------------------------------------------------------------------------------
# Phase-1 / step-0 weakly-supervised loss stack — csrc/wss.hip
# ---------------------------------------------------------------------------------------------------------------------
def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{name}: contiguous float32 tensor expected")
    return t


def pamr(img, mask, num_iter=10, dilations=(1, 2, 4, 8, 12)):
    """PAMR.forward (wss/modules.py:133-152). img [b,3,H,W]; mask [b,c,h,w] is first resized to the image size
    (bilinear, align_corners=True) like the reference does; returns the refined mask [b,c,H,W]."""
    _f32c(img, "pamr img"), _f32c(mask, "pamr mask")
    b, _, h, w = img.shape
    if img.shape[1] != 3 or mask.shape[0] != b:
        raise RuntimeError("pamr: img [b,3,h,w] and mask [b,c,*,*] expected")
    if tuple(mask.shape[-2:]) != (h, w):
        mask = resize_bilinear(mask, (h, w), align_corners=True)
    c = mask.shape[1]
    out = torch.empty_like(mask)
    dil = (C.c_int * len(dilations))(*[int(d) for d in dilations])
    nbytes = _hip.load().cl4_pamr_workspace_bytes(b, c, h, w, len(dilations))
    ws = _workspace(img.device, nbytes)
    call("cl4_pamr", ptr(img), ptr(mask), ptr(out), b, c, h, w, int(num_iter), dil, len(dilations), ptr(ws), ws.numel(),
         stream())
    return out


def ngwp_focal_bce(x, labels, reduction="mean", lam=1e-2, with_grad=True):
    """bce_loss(x, labels, mode='ngwp', reduction) of utils/wss_loss.py:36-53. Returns (loss [1], ngwp [b,c], dx or None)."""
    _f32c(x, "ngwp x"), _f32c(labels, "ngwp labels")
    b, c, h, w = x.shape
    nl = labels.shape[1]
    if labels.shape[0] != b or nl > c or reduction not in ("mean", "sum"):
        raise RuntimeError("ngwp_focal_bce: labels [b, nl <= c] and reduction 'mean' | 'sum' expected")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    ngwp = torch.empty(b, c, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if with_grad else None
    nbytes = _hip.load().cl4_ngwp_workspace_bytes(b, c)
    ws = _workspace(x.device, nbytes)
    call("cl4_ngwp_focal_bce", ptr(x), ptr(labels), b, c, h, w, nl, int(reduction == "sum"), C.c_float(lam), ptr(ngwp),
         ptr(loss), ptr(dx) if with_grad else None, ptr(ws), ws.numel(), stream())
    return loss, ngwp, dx


def pseudo_gtmask(mask, labels=None, cutoff_top=0.6, cutoff_bkg=0.7, cutoff_low=0.2):
    """pseudo_gtmask(mask * labels, ambiguous=True, ...) of wss/single_stage.py:18-40 (label masking of train.py:381-382
    fused when `labels` [b,c-1] is given)."""
    _f32c(mask, "pseudo_gtmask mask")
    b, c, h, w = mask.shape
    if labels is not None and tuple(_f32c(labels, "pseudo_gtmask labels").shape) != (b, c - 1):
        raise RuntimeError("pseudo_gtmask: labels [b, c-1] expected")
    out = torch.empty_like(mask)
    thr = torch.empty(b, c, dtype=torch.float32, device=mask.device)
    call("cl4_pseudo_gtmask", ptr(mask), ptr(labels) if labels is not None else None, ptr(out), b, c, h, w,
         C.c_float(cutoff_top), C.c_float(cutoff_bkg), C.c_float(cutoff_low), ptr(thr), stream())
    return out


def binarize(x, labels=None):
    """binarize(x * labels) of utils/wss_loss.py:188-190."""
    _f32c(x, "binarize x")
    b, c, h, w = x.shape
    if labels is not None and tuple(_f32c(labels, "binarize labels").shape) != (b, c - 1):
        raise RuntimeError("binarize: labels [b, c-1] expected")
    out = torch.empty_like(x)
    call("cl4_binarize", ptr(x), ptr(labels) if labels is not None else None, ptr(out), b, c, h, w, stream())
    return out


def balanced_mask_loss_ce(mask, pseudo_gt, labels, with_grad=True):
    """balanced_mask_loss_ce of wss/single_stage.py:43-80. Returns (loss [1], dmask or None)."""
    _f32c(mask, "balanced_ce mask"), _f32c(pseudo_gt, "balanced_ce pseudo_gt"), _f32c(labels, "balanced_ce labels")
    if tuple(mask.shape[-2:]) != tuple(pseudo_gt.shape[-2:]):
        mask = resize_bilinear(mask, tuple(pseudo_gt.shape[-2:]), align_corners=True)
        if with_grad:
            raise RuntimeError("balanced_mask_loss_ce: gradient needs mask and pseudo_gt at the same size")
    b, c, h, w = pseudo_gt.shape
    if tuple(mask.shape) != (b, c, h, w) or tuple(labels.shape) != (b, c - 1):
        raise RuntimeError("balanced_mask_loss_ce: mask/pseudo_gt [b,c,h,w], labels [b,c-1] expected")
    loss = torch.empty(1, dtype=torch.float32, device=mask.device)
    dmask = torch.empty_like(mask) if with_grad else None
    nbytes = _hip.load().cl4_balanced_ce_workspace_bytes(b, c, h, w)
    ws = _workspace(mask.device, nbytes)
    call("cl4_balanced_mask_loss_ce", ptr(mask), ptr(pseudo_gt), ptr(labels), b, c, h, w, ptr(loss),
         ptr(dmask) if with_grad else None, ptr(ws), ws.numel(), stream())
    return loss, dmask


def bce_ignore_mean(x, target, ignore_index=255, with_grad=True):
    """BCEWithLogitsLossWithIgnoreIndex(reduction='none')(x, target).mean() — utils/loss.py:62-85, train.py:285-288."""
    _f32c(x, "bce_ignore x")
    b, c, h, w = x.shape
    if target.dtype != torch.int64 or not target.is_contiguous() or tuple(target.shape) != (b, h, w):
        raise RuntimeError("bce_ignore_mean: contiguous int64 target [b,h,w] expected")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if with_grad else None
    ws = _workspace(x.device, _hip.load().cl4_bce_ignore_workspace_bytes())
    call("cl4_bce_ignore_mean", ptr(x), ptr(target), b, c, h, w, int(ignore_index), ptr(loss), ptr(dx) if with_grad else None,
         ptr(ws), ws.numel(), stream())
    return loss, dx


# ---------------------------------------------------------------------------------------------------------------------
# Phase-1 training-path losses — csrc/p1loss.hip
# ---------------------------------------------------------------------------------------------------------------------
def _p1_ws(t, b, tot, h, w):
    return _workspace(t.device, _hip.load().cl4_p1loss_workspace_bytes(b, tot, h, w))


def peakgen_train_loss(x, weight, bias, labels_new, alpha=0.7, dweight=None, dbias=None, with_grad=True):
    """PeakGenerator (training) + peak loss. Returns (loss [1], logit [b,nc], dx or None); the weight / bias gradients are
    ACCUMULATED into `dweight` / `dbias` (allocated zeroed when None) and returned as a 5-tuple's last two items."""
    _f32c(x, "peakgen x"), _f32c(labels_new, "peakgen labels")
    b, ctot, h, w = x.shape
    nc = labels_new.shape[1]
    wt = _f32c(weight.reshape(nc, nc), "peakgen weight")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    logit = torch.empty(b, nc, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if with_grad else None
    if with_grad:
        dweight = torch.zeros(nc, nc, dtype=torch.float32, device=x.device) if dweight is None else dweight
        dbias = torch.zeros(nc, dtype=torch.float32, device=x.device) if dbias is None else dbias
    ws = _p1_ws(x, b, ctot, h, w)
    call("cl4_peakgen_train_loss", ptr(x), ptr(wt), ptr(bias), ptr(labels_new), b, ctot, h, w, nc, C.c_float(alpha), ptr(logit),
         ptr(loss), ptr(dx), ptr(dweight) if with_grad else None, ptr(dbias) if with_grad else None, ptr(ws), ws.numel(), stream())
    return loss, logit, dx, dweight, dbias


def peakgen_logits(x, weight, bias, nc, alpha=0.7):
    """PeakGenerator logits of the `nc` new classes (training-mode forward): [b, nc]."""
    _f32c(x, "peakgen x")
    b, ctot, h, w = x.shape
    logit = torch.empty(b, nc, dtype=torch.float32, device=x.device)
    ws = _p1_ws(x, b, ctot, h, w)
    call("cl4_peakgen_vjp", ptr(x), ptr(_f32c(weight.reshape(nc, nc), "peakgen weight")), ptr(bias), None, b, ctot, h, w, nc,
         C.c_float(alpha), ptr(logit), None, None, None, ptr(ws), ws.numel(), stream())
    return logit


def peakgen_vjp(x, weight, bias, dlogit, alpha, dweight, dbias):
    """d(logits)/d(x, weight, bias) contracted with `dlogit` [b, nc]: returns dx [b, ctot, h, w]; dweight / dbias accumulate."""
    _f32c(x, "peakgen x"), _f32c(dlogit, "peakgen dlogit")
    b, ctot, h, w = x.shape
    nc = dlogit.shape[1]
    logit = torch.empty(b, nc, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    ws = _p1_ws(x, b, ctot, h, w)
    call("cl4_peakgen_vjp", ptr(x), ptr(_f32c(weight.reshape(nc, nc), "peakgen weight")), ptr(bias), ptr(dlogit), b, ctot, h, w, nc,
         C.c_float(alpha), ptr(logit), ptr(dx), ptr(dweight), ptr(dbias), ptr(ws), ws.numel(), stream())
    return dx


def flac_loss(A, bs, old_classes, angle, with_grad=True):
    """FLAC consistency loss of train.py:314-333. Returns (loss [1], dA or None)."""
    _f32c(A, "flac A")
    n3, c, h, w = A.shape
    if n3 != 3 * bs:
        raise RuntimeError("flac_loss: A must hold 3*bs images")
    loss = torch.empty(1, dtype=torch.float32, device=A.device)
    dA = torch.empty_like(A) if with_grad else None
    ws = _p1_ws(A, bs, c, h, w)
    call("cl4_flac_loss", ptr(A), bs, c, int(old_classes), h, w, int(angle), ptr(loss), ptr(dA), ptr(ws), ws.numel(), stream())
    return loss, dA


def lloc_loss(x, old_seg, labels=None, with_grad=True):
    """l_loc of train.py:346-354. Returns (loss [1], dx or None; channels >= cold of dx are zero)."""
    _f32c(x, "lloc x"), _f32c(old_seg, "lloc old_seg")
    b, ctot, h, w = x.shape
    cold = old_seg.shape[1]
    if tuple(old_seg.shape) != (b, cold, h, w):
        raise RuntimeError("lloc_loss: old_seg must be [b, cold, h, w] at the size of x")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.zeros_like(x) if with_grad else None
    ws = _p1_ws(x, b, ctot, h, w)
    call("cl4_lloc_loss", ptr(x), ptr(old_seg), ptr(labels) if labels is not None else None, b, ctot, cold, h, w, ptr(loss),
         ptr(dx), ptr(ws), ws.numel(), stream())
    return loss, dx


def mse_loss(x, y, grad_scale=1.0, with_grad=True):
    """nn.MSELoss()(x, y) and grad_scale * d/dx. Returns (loss [1], dx or None)."""
    _f32c(x, "mse x"), _f32c(y, "mse y")
    if x.shape != y.shape:
        raise RuntimeError("mse_loss: shape mismatch")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if with_grad else None
    ws = _workspace(x.device, _hip.load().cl4_p1loss_workspace_bytes(1, 1, 1, 1))
    call("cl4_mse_loss", ptr(x), ptr(y), x.numel(), C.c_float(grad_scale), ptr(loss), ptr(dx), ptr(ws), ws.numel(), stream())
    return loss, dx


def p1_seg_loss(seg, old_seg, masks, labels, alpha=0.5, icarl_bkg=-1.0, weight=1.0, with_grad=True):
    """l_seg of train.py:374-409. Returns (loss [1], dseg or None)."""
    _f32c(seg, "l_seg seg"), _f32c(old_seg, "l_seg old_seg"), _f32c(masks, "l_seg masks"), _f32c(labels, "l_seg labels")
    b, tot, h, w = seg.shape
    cold = old_seg.shape[1]
    if tuple(masks.shape) != (b, tot, h, w) or tuple(old_seg.shape) != (b, cold, h, w) or tuple(labels.shape) != (b, tot - 1):
        raise RuntimeError("p1_seg_loss: seg/masks [b,tot,h,w], old_seg [b,cold,h,w], labels [b,tot-1] expected")
    loss = torch.empty(1, dtype=torch.float32, device=seg.device)
    dseg = torch.empty_like(seg) if with_grad else None
    ws = _p1_ws(seg, b, tot, h, w)
    call("cl4_p1_seg_loss", ptr(seg), ptr(old_seg), ptr(masks), ptr(labels), b, tot, cold, h, w, C.c_float(alpha),
         C.c_float(icarl_bkg), C.c_float(weight), ptr(loss), ptr(dseg), ptr(ws), ws.numel(), stream())
    return loss, dseg


# ---------------------------------------------------------------------------------------------------------------------
# Backward building blocks of the body / heads (phase 1 / step 0) — csrc/backward.hip
# ---------------------------------------------------------------------------------------------------------------------
def zero_insert(x, stride, size):
    n, c, h, w = x.shape
    H, W = size
    y = torch.empty((n, c, H, W), dtype=torch.float32, device=x.device)
    call("cl4_zero_insert", ptr(x), ptr(y), n * c, h, w, int(stride), H, W, stream())
    return y


def maxpool3x3s2_bwd(x, dy):
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    call("cl4_maxpool3x3s2_bwd", ptr(x), ptr(dy), ptr(dx), n * c, h, w, dy.shape[2], dy.shape[3], stream())
    return dx


def add_act(a, b, act=ACT_LEAKY, slope=0.01, want_planes=False):
    if a.shape != b.shape:
        raise RuntimeError("add_act: shape mismatch")
    y = torch.empty_like(a)
    if (want_planes and PRODUCER_PLANES and planes_f16() and a.dim() == 4 and a.shape[1] % 8 == 0 and a.is_contiguous() and b.is_contiguous()):
        n, c, h, w = a.shape
        ps = Planes(n, c, h, w, a.device)
        call("cl4_add_act_planes", ptr(a), ptr(b), ptr(y), ptr(ps.data), n, c, h * w, act, C.c_float(slope), stream())
        y._cl4_planes = ps
        return y
    call("cl4_add_act", ptr(a), ptr(b), ptr(y), a.numel(), act, C.c_float(slope), stream())
    return y


def act_bwd(y, dy, act=ACT_LEAKY, slope=0.01):
    dx = torch.empty_like(dy)
    call("cl4_act_bwd", ptr(y), ptr(dy), ptr(dx), y.numel(), act, C.c_float(slope), stream())
    return dx


def add_per_plane_(y, v, scale=1.0):
    """y [n,c,h,w] += scale * v [n,c] (broadcast over the plane)."""
    n, c, h, w = y.shape
    if v.numel() != n * c:
        raise RuntimeError("add_per_plane_: v must hold one value per (n, c) plane")
    call("cl4_add_per_plane", ptr(y), ptr(v), n * c, h * w, C.c_float(scale), stream())
    return y


def dropout2d(x, keep, p):
    """nn.Dropout2d forward / backward: planes with keep == 0 are zeroed, the others scaled by 1 / (1 - p). keep: uint8 [n, c]."""
    n, c, h, w = x.shape
    if keep.numel() != n * c or keep.dtype != torch.uint8:
        raise RuntimeError("dropout2d: keep must hold one byte per (n, c) plane")
    y = torch.empty_like(x)
    call("cl4_dropout2d", ptr(x), ptr(keep), ptr(y), n * c, h * w, C.c_float(p), stream())
    return y


def dgrad_weight(w, precision=0, force_f16=False):
    """[cout,cin,k,k] -> packed operand of the input-gradient convolution (a stride-1 conv over dy with cin' = cout). `force_f16`: see
    pack_conv_weight (the gradient arrives with its fp16 planes twin, so a channel-shrinking 1x1 is routed to fp16x3 as well)."""
    cout, cin, k, _ = w.shape
    wt = torch.empty((cin, cout, k, k), dtype=torch.float32, device=w.device)
    call("cl4_transpose_flip_weight", ptr(w), ptr(wt), cout, cin, k, stream())
    return pack_conv_weight(wt, precision if bf16x3_ok(cin) else 0, force_f16=force_f16)


def conv2d_dgrad(dy, wd, in_size, k, stride=1, pad=0, dil=1, x6=False, post=None):
    """Input gradient of conv2d(x, w, stride, pad, dil): `wd` = dgrad_weight(w); in_size = (hin, win) of x. `post` ([n, cin, hin, win]):
    returns the gradient PLUS post — the shortcut's gradient of a residual block rides in the convolution's epilogue (r05) instead of a
    separate axpy pass."""
    hin, win = in_size
    if k == 1 and pad == 0:
        if stride == 1:
            return conv2d(dy, wd, 1, x6=x6 and post is None, xdyn=True, post=post)
        dx = zero_insert(conv2d(dy, wd, 1, x6=x6, xdyn=True), stride, (hin, win))
        return dx if post is None else axpy_(dx, post)
    up = dy if stride == 1 else zero_insert(dy, stride, (hin + 2 * pad - dil * (k - 1), win + 2 * pad - dil * (k - 1)))
    return conv2d(up, wd, k, stride=1, pad=dil * (k - 1) - pad, dil=dil, x6=x6 and post is None, xdyn=True, post=post)


def rot90(x, k, out=None):
    """torch.rot90(x, k, dims=(-2, -1)) for square maps."""
    n, c, h, w = x.shape
    y = torch.empty_like(x) if out is None else out
    call("cl4_rot90", ptr(x), ptr(y), n * c, h, w, int(k), stream())
    return y


def softmax_channels(x):
    _f32c(x, "softmax_channels")
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    call("cl4_softmax_channels", ptr(x), ptr(y), b, c, h, w, stream())
    return y


def channel_affine(x, scale, shift):
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    call("cl4_channel_affine", ptr(x), ptr(y), n, c, h * w, ptr(scale), ptr(shift), stream())
    return y


def randrop_loss(x, masks, labels, neg_labels, cold, mask_labels=None, with_grad=True):
    """RandropLoss of utils/wss_loss.py:227-248 with caller-drawn random old classes (int32 [b,h,w])."""
    _f32c(x, "randrop x"), _f32c(masks, "randrop masks"), _f32c(labels, "randrop labels")
    b, tot, h, w = x.shape
    if neg_labels.dtype != torch.int32 or tuple(neg_labels.shape) != (b, h, w) or not neg_labels.is_contiguous():
        raise RuntimeError("randrop_loss: contiguous int32 neg_labels [b,h,w] expected")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if with_grad else None
    ws = _p1_ws(x, b, tot, h, w)
    call("cl4_randrop_loss", ptr(x), ptr(masks), ptr(labels), ptr(neg_labels), ptr(mask_labels) if mask_labels is not None else None,
         b, tot, int(cold), h, w, ptr(loss), ptr(dx), ptr(ws), ws.numel(), stream())
    return loss, dx


def random_labels(shape, modulus, seed, device):
    """int32 tensor of uniform draws in [0, modulus) from the counter-based generator of the Dropout masks."""
    out = torch.empty(shape, dtype=torch.int32, device=device)
    call("cl4_random_labels", ptr(out), out.numel(), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), int(modulus), stream())
    return out


def random_labels_dev(shape, modulus, seed_dev, advance, device):
    """random_labels from a device-resident seed (uint64 tensor [1]) that the launch advances: graph-replay safe."""
    out = torch.empty(shape, dtype=torch.int32, device=device)
    call("cl4_random_labels_dev", ptr(out), out.numel(), ptr(seed_dev), C.c_uint64(advance), int(modulus), stream())
    return out


def semseg_hist_(hist, logits, labels, n_classes, flip=False, zero_below=0, want_pred=False):
    """Confusion-matrix update of validation (train.py:695-745): hist (int64 [n,n] device tensor) += counts."""
    _f32c(logits, "semseg_hist logits")
    nb, c, H, W = logits.shape
    nimg = 1 if flip else nb
    if labels.dtype != torch.int64 or tuple(labels.shape) != (nimg, H, W) or not labels.is_contiguous():
        raise RuntimeError("semseg_hist_: contiguous int64 labels [nimg,H,W] expected")
    if hist.dtype != torch.int64 or hist.numel() != n_classes * n_classes:
        raise RuntimeError("semseg_hist_: int64 hist [n,n] expected")
    pred = torch.empty((nimg, H, W), dtype=torch.uint8, device=logits.device) if want_pred else None
    call("cl4_semseg_hist", ptr(logits), ptr(labels), ptr(hist), ptr(pred), nb, c, H, W, int(bool(flip)), int(n_classes),
         int(zero_below), stream())
    return pred


# ---------------------------------------------------------------------------------------------------------------------
# Validation post-processing (get_ins_map, dataset/utils.py:795-892) — label.hip / refine.hip
# ---------------------------------------------------------------------------------------------------------------------
VAL_MIN_MASK = 50        # MINIMUM_MASK_SIZE of dataset/utils.py:147 (training uses 20)


def val_seg(logits, flip=False, cls_label=None):
    """softmax -> flip average of the probabilities -> (val_clean mask) -> arg-max. Returns (seg_map u8 [H,W], prob [H,W])."""
    _f32c(logits, "val_seg logits")
    nb, c, H, W = logits.shape
    seg_map = torch.empty((H, W), dtype=torch.uint8, device=logits.device)
    prob = torch.empty((H, W), dtype=torch.float32, device=logits.device)
    call("cl4_val_seg", ptr(logits), ptr(cls_label) if cls_label is not None else None, ptr(seg_map), ptr(prob), nb, c, H, W,
         int(bool(flip)), stream())
    return seg_map, prob


def instance_map(seg_map, prob_win, center, offset, thresh=0.1, nms_kernel=41, beta=3.0, ignore=False, min_mask=VAL_MIN_MASK,
                 cap=4096):
    """get_ins_map's instance part for one image. center [C,H,W], offset [2,H,W]. Returns (ins_map int32 [H,W], table [n,4] =
    class / score / pixels / contour, device tensors); n is read back from the device (one small D2H copy)."""
    H, W = seg_map.shape
    _f32c(center, "instance_map center"), _f32c(offset, "instance_map offset"), _f32c(prob_win, "instance_map prob")
    if seg_map.dtype != torch.uint8 or tuple(center.shape[1:]) != (H, W) or tuple(offset.shape) != (2, H, W):
        raise RuntimeError("instance_map: seg_map u8 [H,W], center [C,H,W], offset [2,H,W] expected")
    dev = seg_map.device
    ins = torch.empty((H, W), dtype=torch.int32, device=dev)
    table = torch.zeros((cap, 4), dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    d = _hip.InsmapIO()
    d.H, d.W, d.classes, d.nms_kernel, d.min_mask, d.ignore, d.cap = H, W, center.shape[0], int(nms_kernel), int(min_mask), int(bool(ignore)), cap
    d.thresh, d.beta = float(thresh), float(beta)
    pv = lambda t: ptr(t).value
    d.seg_map, d.prob_win, d.center, d.offset = pv(seg_map), pv(prob_win), pv(center), pv(offset)
    d.ins_map, d.table, d.count = pv(ins), pv(table), pv(count)
    nbytes = _hip.load().cl4_label_workspace_bytes(1, H, W)
    ws = _workspace(dev, nbytes)
    call("cl4_instance_map", C.byref(d), ptr(ws), ws.numel(), stream())
    n = int(count.item())
    return ins, table[:n]


# ---------------------------------------------------------------------------------------------------------------------
# Input pipeline (dataset/__init__.py:14-27, dataset/dataset.py:202-221, dataset/utils.py:299-350) — input.hip
# ---------------------------------------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_input_scratch = _Scratch(1 << 20)


def input_batch(images, descs, out_hw, labels=None, lut=None, want_mask=True, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """Crop + Pillow-exact resample + flip + ToTensor + Normalize of a packed batch of decoded photos.

    images: uint8 device tensor, the photos packed back to back (H x W x 3 each); labels: the same for the H x W x 2
    (semantic id, instance id) labels, or None; descs: list of (img_off, lbl_off, H, W, ci, cj, ch, cw, flip);
    lut: uint8[256] device tensor (LabelTransform) or None. Returns (img f32 [b,3,oh,ow], seg int64 [b,oh,ow] | None,
    mask uint8 [b,oh,ow] | None)."""
    b = len(descs)
    oh, ow = out_hw
    if images.dtype != torch.uint8 or (labels is not None and labels.dtype != torch.uint8):
        raise RuntimeError("input_batch: uint8 packed buffers expected")
    arr = (_hip.SampleDesc * b)()
    lib = _hip.load()
    max_ch = ks = 0
    for k, d in enumerate(descs):
        io, lo, H, W, ci, cj, ch, cw, flip = (int(v) for v in d)
        arr[k] = _hip.SampleDesc(io, lo, H, W, ci, cj, ch, cw, flip, 0)
        max_ch = max(max_ch, ch)
        ks = max(ks, lib.cl4_input_ksize(cw, ow), lib.cl4_input_ksize(ch, oh))
    dev = images.device
    img = torch.empty((b, 3, oh, ow), dtype=torch.float32, device=dev)
    seg = torch.empty((b, oh, ow), dtype=torch.int64, device=dev) if labels is not None else None
    mask = torch.empty((b, oh, ow), dtype=torch.uint8, device=dev) if labels is not None and want_mask else None
    need = lib.cl4_input_workspace_bytes(b, oh, ow, max(max_ch, 1), max(ks, 1))
    ws = _input_scratch.get(dev, need)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    call("cl4_input_batch", ptr(images), images.numel(), ptr(labels), labels.numel() if labels is not None else 0, arr, b, oh, ow,
         m3, s3, ptr(lut), ptr(img), ptr(seg), ptr(mask), ptr(ws), ws.numel(), stream())
    return img, seg, mask


_gauss_tables = {}


def gaussian_table(sigma, device):
    """float32 device copy of the reference's float64 gaussian stamp (dataset/utils.py:182-192 == modules/utils.py:49-59)."""
    key = (int(sigma), device.index)
    if key not in _gauss_tables:
        import numpy as np
        size = 6 * sigma + 3
        x = np.arange(0, size, 1, float)
        y = x[:, np.newaxis]
        x0 = y0 = 3 * sigma + 1
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        _gauss_tables[key] = torch.from_numpy(g.astype(np.float32)).to(device).contiguous()
    return _gauss_tables[key]


def target_generation(seg_map, mask, num_classes, sigma=8):
    """label_generation of the supervised loader for a batch: (center [b,C,H,W], offset [b,2,H,W], weight [b,1,H,W])."""
    if seg_map.dtype != torch.int64 or mask.dtype != torch.uint8 or seg_map.shape != mask.shape or seg_map.dim() != 3:
        raise RuntimeError("target_generation: int64 seg_map and uint8 mask of the same [b,H,W] shape expected")
    b, H, W = seg_map.shape
    dev = seg_map.device
    center = torch.empty((b, num_classes, H, W), dtype=torch.float32, device=dev)
    offset = torch.empty((b, 2, H, W), dtype=torch.float32, device=dev)
    weight = torch.empty((b, 1, H, W), dtype=torch.float32, device=dev)
    need = _hip.load().cl4_target_generation_workspace_bytes(b)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    call("cl4_target_generation", ptr(seg_map), ptr(mask), b, H, W, int(num_classes), int(sigma), ptr(gaussian_table(sigma, dev)),
         ptr(center), ptr(offset), ptr(weight), ptr(ws), need, stream())
    return center, offset, weight
