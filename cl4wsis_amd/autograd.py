"""torch.autograd bridges over the hand-written forward / backward tapes, so that the modules can be called the way the
reference's training loop calls them (train.py:253-274, 304-337, 425-428) and `loss.backward()` (train.py:537) reaches the
HIP backward kernels.

`Trainer.step` does NOT go through these: it drives the same tapes directly, writes gradients into the flat data-parallel
buffer and overlaps the bucketed all-reduce. The bridges exist for drop-in use of the modules from foreign training code and
are tested against `Trainer.step` (tests/test_module_surface_gpu.py).

One Function class serves every module: `forward` runs `runner.fwd(*inputs) -> (outputs, tape)`, `backward` runs
`runner.bwd(tape, douts, grad_of, needs) -> input gradients`, where `grad_of(param)` hands out zero-initialised gradient
buffers that are returned to autograd as the gradients of the parameters (passed as Function inputs for exactly that purpose).
"""
import torch

from . import ops


class TapeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, n_in, *tensors):
        inputs, params = tensors[:n_in], tensors[n_in:]
        outs, tape = runner.fwd(*[t.detach() for t in inputs])
        ctx.runner, ctx.tape, ctx.params, ctx.n_in = runner, tape, params, n_in
        ctx.set_materialize_grads(False)
        single = torch.is_tensor(outs)
        ctx.single = single
        outs = (outs,) if single else tuple(outs)
        ctx.out_shapes = [tuple(o.shape) for o in outs]
        return outs[0] if single else outs

    @staticmethod
    def backward(ctx, *douts):
        if ctx.tape is None:
            raise RuntimeError("cl4wsis_amd: backward through the same forward twice (the tape is released after one backward)")
        dev = next(t.device for t in ctx.params) if ctx.params else douts[0].device
        douts = [None if d is None else d.to(dtype=torch.float32).contiguous() for d in douts]
        if not getattr(ctx.runner, "accepts_none", False):       # an output the loss never touched has no gradient
            douts = [torch.zeros(s, dtype=torch.float32, device=dev) if d is None else d for d, s in zip(douts, ctx.out_shapes)]
        grads = {}

        def grad_of(p):
            g = grads.get(id(p))
            if g is None:
                g = torch.zeros_like(p, dtype=torch.float32)
                grads[id(p)] = g
            return g

        grad_of.grads = grads               # InstanceBranch keeps its gradients in a dict keyed by id(param): the same one
        needs = ctx.needs_input_grad[2:2 + ctx.n_in]
        dins = ctx.runner.bwd(ctx.tape, douts[0] if ctx.single else douts, grad_of, needs)
        ctx.tape = None
        dins = tuple(dins) if isinstance(dins, (tuple, list)) else (dins,)
        dins = tuple(d if n else None for d, n in zip(dins, needs))
        dpar = tuple(grads.get(id(p)) if n else None for p, n in zip(ctx.params, ctx.needs_input_grad[2 + ctx.n_in:]))
        return (None, None) + dins + dpar


def apply(runner, inputs, params):
    """Run `runner` under autograd. With grad mode off (or nothing requiring grad) the tape is dropped right away."""
    params = [p for p in params]
    if not torch.is_grad_enabled() or not (any(p.requires_grad for p in params) or any(t.requires_grad for t in inputs)):
        outs, _ = runner.fwd(*[t.detach() for t in inputs])
        return outs
    return TapeFunction.apply(runner, len(inputs), *inputs, *params)


# ----------------------------------------------------------------------------------------------------------------------
# runners
# ----------------------------------------------------------------------------------------------------------------------
class ResizeRunner:
    """F.interpolate(mode='bilinear') with gradient (segmentation_module.py:428-441 `_upsample_predictions`)."""

    def __init__(self, size, align_corners):
        self.size, self.ac = tuple(size), align_corners

    def fwd(self, x):
        return ops.resize_bilinear(x.contiguous(), self.size, self.ac), tuple(x.shape[2:])

    def bwd(self, in_hw, dy, grad_of, needs):
        return ops.resize_bilinear_bwd(dy, in_hw, self.ac)


class BodyRunner:
    """ResNet body in training mode (cross-rank batch statistics): images -> the five module outputs."""
    accepts_none = True

    def __init__(self, body, group=None):
        self.body, self.group = body, group

    def fwd(self, x):
        from . import body_train
        outs, tape = body_train.body_forward(self.body, x.contiguous(), self.group)
        return tuple(outs), tape

    def bwd(self, tape, douts, grad_of, needs):
        from . import body_train
        body_train.body_backward(self.body, tape, {i: d for i, d in enumerate(douts) if d is not None}, grad_of, self.group)
        return (None,)                      # images need no gradient


class DeeplabRunner:
    def __init__(self, head, group=None):
        self.head, self.group = head, group

    def fwd(self, x):
        from . import heads_train
        return heads_train.deeplab_forward(self.head, x.contiguous(), self.group)

    def bwd(self, tape, dy, grad_of, needs):
        from . import heads_train
        return heads_train.deeplab_backward(self.head, tape, dy, grad_of, self.group)


class ClassifierRunner:
    def __init__(self, cls):
        self.cls = cls

    def fwd(self, x):
        from . import heads_train
        return heads_train.cls_forward(self.cls, x.contiguous())

    def bwd(self, x, dy, grad_of, needs):
        from . import heads_train
        return heads_train.cls_backward(self.cls, x, dy, grad_of, need_dx=needs[0])


class PseudoLabelerRunner:
    def __init__(self, pl, group=None):
        self.pl, self.group = pl, group

    def fwd(self, x):
        from . import heads_train
        return heads_train.pseudolabeler_forward(self.pl, x.contiguous(), self.group)

    def bwd(self, tape, dy, grad_of, needs):
        from . import heads_train
        return heads_train.pseudolabeler_backward(self.pl, tape, dy, grad_of, need_dx=needs[0], group=self.group)


class PeakGeneratorRunner:
    """PeakGenerator.forward in training mode (wss/modules.py:390-423): PAM -> 1x1 conv -> nGWP logit of the new classes."""

    def __init__(self, pg):
        self.pg = pg

    def fwd(self, x):
        pg = self.pg
        x = x.contiguous()
        logit = ops.peakgen_logits(x, pg.extra_conv4.weight.detach(), pg.extra_conv4.bias.detach(), pg.new_classes, pg.alpha)
        return logit, x

    def bwd(self, x, dlogit, grad_of, needs):
        pg = self.pg
        w, b = pg.extra_conv4.weight, pg.extra_conv4.bias
        return ops.peakgen_vjp(x, w.detach(), b.detach(), dlogit, pg.alpha, grad_of(w).view(pg.new_classes, -1), grad_of(b))


class DecoderRunner:
    """Instance decoder in training mode: (res2, res3, res4, res5) -> decoder features."""

    def __init__(self, branch, keys, dropout_keep=None, seed=0, detach=True):
        self.branch, self.keys, self.keep, self.seed, self.detach = branch, keys, dropout_keep, seed, detach

    def fwd(self, *feats):
        f = {k: t.contiguous() for k, t in zip(self.keys, feats)}
        return self.branch.decoder_forward_train(f, self.keep, self.seed)

    def bwd(self, tape, dy, grad_of, needs):
        fg = {} if (not self.detach and any(needs)) else None
        self.branch.decoder_backward(tape, dy, grad_of.grads, feature_grads=fg)
        return tuple((fg or {}).get(k) for k in self.keys)


class HeadRunner:
    def __init__(self, branch):
        self.branch = branch

    def fwd(self, y):
        pred, tape = self.branch.head_forward_train(y.contiguous())
        self.keys = list(pred.keys())
        return tuple(pred[k] for k in self.keys), tape

    def bwd(self, tape, douts, grad_of, needs):
        dy, _ = self.branch.head_backward(tape, dict(zip(self.keys, douts)), grad_of.grads)
        return dy
