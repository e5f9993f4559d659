"""Parameter holders and kernel-call helpers shared by the network modules.

The modules keep the reference's attribute names so that ``state_dict()`` keys are the
checkpoint contract of SURVEY.md Appendix B; their ``forward`` never runs a torch compute op —
it enqueues the HIP kernels of libcl4wsis_hip.so through :mod:`cl4wsis_amd.ops`.
"""
import os

import torch
import torch.nn as nn

from . import ops

EPS = 1e-5


class ABN(nn.Module):
    """Parameter holder with the contract of inplace_abn.ABN / InPlaceABN(Sync)
    (reference: segmentation_module.py:17-29; attribute use in modules/residual.py:52,65,82,92-93,
    modules/deeplab.py:39): weight, bias, running_mean, running_var, .activation, .activation_param.
    Eval mode is folded into the producing conv's epilogue (scale/shift + leaky-ReLU)."""
    abs_gamma = False           # plain ABN = functional.batch_norm + activation: scales with `weight` itself

    def __init__(self, num_features, eps=EPS, momentum=0.1, affine=True, activation="leaky_relu",
                 activation_param=0.01, group=None):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.activation = activation
        self.activation_param = activation_param
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def act_code(self):
        if self.activation == "leaky_relu":
            return ops.ACT_LEAKY, float(self.activation_param)
        if self.activation == "identity":
            return ops.ACT_NONE, 0.0
        raise NotImplementedError(f"ABN activation {self.activation!r} is not on the phase-2 path")

    def forward(self, x):  # pragma: no cover - never used on the product path
        raise RuntimeError("ABN is fused into the producing convolution; call the owning block instead")


class InPlaceABN(ABN):
    """inplace_abn.InPlaceABN: same parameters as ABN, but the fused kernels of the (absent, unpinned) package scale with
    `|weight| + eps` instead of `weight` and return d(weight) = sign(weight) * d(scale) — believed behaviour of
    mapillary/inplace_abn (its kernels read `gamma = abs(weight) + eps`). Irrelevant for positive scales up to 1e-5, decisive
    for a pretrained `*_iabn_sync` checkpoint that holds negative ones. CL4_IABN_PLAIN=1 restores the plain-BN reading."""
    abs_gamma = os.environ.get("CL4_IABN_PLAIN", "0") != "1"


class InPlaceABNSync(InPlaceABN):
    pass


def gamma_of(bn):
    """The scale the normalisation kernels must use: |weight| + eps for InPlaceABN(Sync), weight otherwise."""
    if getattr(bn, "abs_gamma", False):
        val = CACHE.get(bn, "absg", (bn.weight,), lambda: ops.abs_eps(bn.weight.detach(), bn.eps))
        if PackBatch.ACTIVE is not None:
            PackBatch.ACTIVE.register_scale(bn, val)
        return val
    return bn.weight.detach()


def gamma_grad(bn, grad_of):
    """(buffer the backward kernel accumulates the weight's gradient into, `gamma_sign` operand of ops.abn_train_backward or None). For
    InPlaceABN the kernels differentiate the effective scale |weight| + eps; handed the raw weight they apply the sign rule themselves
    (r05: no scratch gradient, zero fill and cl4_sign_axpy launch per layer)."""
    gw = grad_of(bn.weight)
    return gw, (bn.weight.detach() if getattr(bn, "abs_gamma", False) else None)


class _Cache:
    """Derived device tensors (packed weights, folded BN) are cached ON the owning module (attribute `_cl4_cache`),
    so they die with it, and are revalidated against the source tensors' (data_ptr, _version, external-write
    counter): load_state_dict / in-place torch ops bump _version; kernels that write a tensor behind PyTorch's back
    (fused Adam, BN running statistics) call `touch`."""

    def __init__(self):
        self._ext = {}      # data_ptr -> counter of external writes
        self._epoch = 0     # bumped by clear(): invalidates everything

    def touch(self, t):
        k = t.data_ptr()
        self._ext[k] = self._ext.get(k, 0) + 1

    def _sig(self, tensors):
        return (self._epoch,) + tuple((t.data_ptr(), t._version, self._ext.get(t.data_ptr(), 0), t.device.index)
                                      for t in tensors if t is not None)

    def get(self, owner, key, tensors, build):
        store = owner.__dict__.setdefault("_cl4_cache", {})
        sig = self._sig(tensors)
        hit = store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        val = build()
        store[key] = (sig, val)
        return val

    def clear(self):
        self._epoch += 1

    def snapshot(self, roots):
        """Strong references to every derived tensor currently cached under `roots` (modules). A captured HIP graph holds the
        ADDRESSES of the derived tensors of frozen weights it read (they are made eagerly, before the capture); whoever owns the
        graph keeps this list so that a later clear() + eager rebuild cannot hand that memory to another tensor under the graph."""
        keep = []
        for root in roots:
            if root is None:
                continue
            for m in root.modules():
                store = m.__dict__.get("_cl4_cache")
                if store:
                    keep.extend(v for _, v in store.values())
        return keep


CACHE = _Cache()


class PackBatch:
    """All fp16x3 weight packs (and InPlaceABN scales) of a training iteration in one launch each (r05, cl4_pack_conv_weights_f16x3_batched,
    cl4_abs_eps_batched). While a batch is the
    ACTIVE one (`with batch:`), every fp16x3 pack that packed_weight / body_train._dgrad_weight build lazily for a TRAINABLE convolution is
    registered with it — the PackedWeight it returned becomes that layer's persistent buffer. `refresh()` (the start of every later
    iteration) re-derives ALL registered packs in place with one launch and marks the cache entries current, so the lazy path finds
    nothing to do. The job table lives in device memory and is rebuilt only when a layer is added, never while a HIP graph is being
    captured (a captured iteration contains just the launch). In-place is safe because every reader of a pack runs on the stream that
    issues the refresh, behind it; a cache entry that somebody rebuilt lazily in between (validation after stale_trainable()) is
    simply replaced by the persistent buffer again at the next refresh."""

    ACTIVE = None
    ENABLED = os.environ.get("CL4_PACK_BATCH", "1") != "0"

    def __init__(self):
        self.entries = {}        # (id(conv), key) -> (conv, key, transposed, PackedWeight)
        self.scales = {}         # id(bn) -> (bn, |weight| + eps tensor): the InPlaceABN scales ride along (cl4_abs_eps_batched)
        self._table = None       # device job tables and what they were built for
        self._retired = []       # earlier tables (graphs captured with them keep replaying them)
        self.launches = 0

    def __enter__(self):
        self._prev, PackBatch.ACTIVE = PackBatch.ACTIVE, (self if PackBatch.ENABLED else None)
        return self

    def __exit__(self, *exc):
        PackBatch.ACTIVE = self._prev
        return False

    def register(self, conv, key, transposed, wp):
        if (isinstance(wp, ops.PackedWeight) and wp.precision == 3 and conv.weight.requires_grad and conv.weight.is_cuda
                and conv.weight.is_contiguous() and (id(conv), key) not in self.entries):
            self.entries[(id(conv), key)] = (conv, key, bool(transposed), wp)

    def register_scale(self, bn, val):
        if bn.weight.requires_grad and bn.weight.is_cuda and id(bn) not in self.scales:
            self.scales[id(bn)] = (bn, val)

    def _signature(self):
        return ([e[0].weight.data_ptr() for e in self.entries.values()], [e[0].weight.data_ptr() for e in self.scales.values()])

    def _build_table(self):
        from ._hip import AbsJob, PackJob

        def upload(jobs, dev):
            return torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).clone().to(dev)

        jobs = (PackJob * max(len(self.entries), 1))()
        first, dev = 0, None
        for j, (conv, key, transposed, wp) in zip(jobs, self.entries.values()):
            cout, cin, kh, kw = conv.weight.shape
            j.w, j.wq, j.first = conv.weight.data_ptr(), wp.data.data_ptr(), first
            j.cout, j.cin, j.khw, j.transposed = cout, cin, kh * kw, int(transposed)
            first += (kh * kw * (cout if transposed else cin) // 8) * (cin if transposed else cout)
            dev = conv.weight.device
        ajobs = (AbsJob * max(len(self.scales), 1))()
        afirst = 0
        for j, (bn, val) in zip(ajobs, self.scales.values()):
            j.w, j.out, j.first, j.n, j.eps = bn.weight.data_ptr(), val.data_ptr(), afirst, bn.weight.numel(), bn.eps
            afirst += bn.weight.numel()
            dev = bn.weight.device
        if self._table is not None:      # a captured graph may still read the table it was captured with: retired, never freed (as ops._Scratch)
            self._retired.append(self._table)
        self._table = {"packs": (upload(jobs, dev), len(self.entries), first), "scales": (upload(ajobs, dev), len(self.scales), afirst),
                       "sig": self._signature()}

    def refresh(self):
        """Re-derive every registered pack / scale from the current weights (one launch each) and mark the cache entries current."""
        if not self.entries and not self.scales:
            return False
        if self._table is None or self._table["sig"] != self._signature():
            if torch.cuda.is_current_stream_capturing():
                return False                                 # (no host-to-device copy inside a capture: the lazy path derives layer by layer)
            self._build_table()
        table, njobs, total = self._table["packs"]
        if njobs:
            ops.call("cl4_pack_conv_weights_f16x3_batched", ops.ptr(table), njobs, total, ops.stream())
            self.launches += 1
        table, njobs, total = self._table["scales"]
        if njobs:
            ops.call("cl4_abs_eps_batched", ops.ptr(table), njobs, total, ops.stream())
        for conv, key, _, wp in self.entries.values():
            conv.__dict__.setdefault("_cl4_cache", {})[key] = (CACHE._sig((conv.weight,)), wp)
        for bn, val in self.scales.values():
            bn.__dict__.setdefault("_cl4_cache", {})["absg"] = (CACHE._sig((bn.weight,)), val)
        return True


def packed_weight(conv, precision=None, force_f16=False):
    """The conv's weight in the operand layout of `precision` (default: ops.CONV_PRECISION), cached until it changes. `force_f16`
    (precision 3): the fp16x3 pack also for a 1x1 layer that shrinks the channel count — its input arrives with a planes twin made by the
    producer (ops.bn_apply / add_act want_planes), so there is no conversion pass to lose against."""
    w = conv.weight
    prec = ops.CONV_PRECISION if precision is None else precision
    key = "wp%d%s" % (prec, "f" if force_f16 else "")
    wp = CACHE.get(conv, key, (w,), lambda: ops.pack_conv_weight(w.detach(), prec, force_f16=force_f16))
    if PackBatch.ACTIVE is not None and prec == 3:
        PackBatch.ACTIVE.register(conv, key, False, wp)
    return wp


def folded_bn(bn, lo=0, hi=None):
    """(scale, shift) of an eval-mode ABN / BatchNorm2d, optionally for a channel slice."""
    ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)

    def build():
        return ops.bn_fold(gamma_of(bn), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)

    scale, shift = CACHE.get(bn, "bn", ts, build)
    if lo == 0 and hi is None:
        return scale, shift
    return scale[lo:hi], shift[lo:hi]


def conv_abn(x, conv, bn, post=None, act2=ops.ACT_NONE, slope2=0.01, out=None, out_coff=0, bn_lo=0, bn_hi=None,
             pre=None):
    """conv -> eval ABN (affine + its activation) [-> + post -> act2], one kernel."""
    scale, shift = folded_bn(bn, bn_lo, bn_hi)
    act1, slope1 = bn.act_code() if isinstance(bn, ABN) else (ops.ACT_RELU, 0.0)
    k = conv.kernel_size[0]
    return ops.conv2d(x, packed_weight(conv), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0],
                      scale=scale, shift=shift, pre=pre, post=post, act1=act1, slope1=slope1, act2=act2, slope2=slope2,
                      out=out, out_coff=out_coff)


# "Planes" activations between the dense convolutions of the frozen networks (include/cl4wsis.h; csrc/conv_s.hip): the producing
# convolution's epilogue emits the bf16 hi / lo split its consumers multiply, instead of every consumer re-splitting fp32 values in
# its K loop. CL4_CONV_PLANES=0 keeps every activation fp32 NCHW (A/B measurements).
CONV_PLANES = os.environ.get("CL4_CONV_PLANES", "1") != "0"
PLANES_MIN_COUT = int(os.environ.get("CL4_PLANES_MIN_COUT", "128"))   # 64 (res2 on the planes path too) measured neutral: 345.5 vs 345.3 images/s


def planes_ok(conv, min_cout=None):
    """Can this convolution READ a planes input (cl4_conv2d_fwd_planes)? bf16x3 arithmetic, 1x1 / 3x3, 32-channel K-tiles, at
    least PLANES_MIN_COUT output channels (one full 128-row M tile by default; `min_cout` overrides: the first block of a stage whose
    identity blocks run fused stays in planes down to 64 channels so that the fused blocks get a planes trunk)."""
    lim = PLANES_MIN_COUT if min_cout is None else min_cout
    return (CONV_PLANES and ops.CONV_PRECISION == 1 and conv.kernel_size[0] in (1, 3) and conv.kernel_size[0] == conv.kernel_size[1]
            and conv.in_channels % 32 == 0 and conv.out_channels >= lim and conv.out_channels % 8 == 0 and conv.groups == 1)


def conv_abn_planes(xs, conv, bn, post=None, act2=ops.ACT_NONE, slope2=0.01, pre=None, bn_lo=0, bn_hi=None, want_fp32=False,
                    want_planes=True, out=None, out_planes=None, out_coff=0):
    """conv_abn on a planes input; `post` may be fp32 or planes. Returns (fp32 output or None, planes output or None)."""
    scale, shift = folded_bn(bn, bn_lo, bn_hi)
    act1, slope1 = bn.act_code() if isinstance(bn, ABN) else (ops.ACT_RELU, 0.0)
    return ops.conv2d_planes(xs, packed_weight(conv, 1), conv.kernel_size[0], stride=conv.stride[0], pad=conv.padding[0],
                             dil=conv.dilation[0], scale=scale, shift=shift, pre=pre, post=post, act1=act1, slope1=slope1,
                             act2=act2, slope2=slope2, out=out, out_coff=out_coff, want_fp32=want_fp32, out_planes=out_planes,
                             want_planes=want_planes)


def planes_of(t):
    """The planes twin a producing layer attached to an fp32 activation (same values, pre-split), or None."""
    return getattr(t, "_cl4_planes", None)


def conv_bias(x, conv, out=None, out_coff=0, act=ops.ACT_NONE):
    """conv (+bias) with no normalisation (classifier heads)."""
    k = conv.kernel_size[0]
    shift = conv.bias.detach() if conv.bias is not None else None
    return ops.conv2d(x, packed_weight(conv), k, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0],
                      shift=shift, act1=act, out=out, out_coff=out_coff)
