"""Adam on flat buffers, one HIP kernel launch per parameter group (torch.optim.Adam semantics, train.py:181-182).

The optimizer is a torch.optim.Optimizer subclass so that LR schedulers and the checkpoint contract
(`optimizer_state`, run.py:24) keep working; `state_dict()` exposes per-parameter `step / exp_avg / exp_avg_sq` like
torch.optim.Adam (views of the flat moment buffers)."""
import torch

from . import ops
from .nn import CACHE
from .parallel import FlatBuffer


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        plist = [p for g in self.param_groups for p in g["params"]]
        if not plist:
            raise ValueError("FusedAdam: no parameters")
        dev = plist[0].device
        shapes = [p.shape for p in plist]
        self.params_fb = FlatBuffer(shapes, dev)
        self.grads_fb = FlatBuffer(shapes, dev)
        self.m_fb = FlatBuffer(shapes, dev)
        self.v_fb = FlatBuffer(shapes, dev)
        self.index = {}
        with torch.no_grad():
            for i, p in enumerate(plist):
                self.params_fb.views[i].copy_(p.data)
                p.data = self.params_fb.views[i]       # the parameter now aliases the flat buffer
                self.index[id(p)] = i
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.m_fb.views[i], "exp_avg_sq": self.v_fb.views[i]}
        # group -> contiguous [lo, hi) range of the flat buffers
        self._ranges, k = [], 0
        for g in self.param_groups:
            n = len(g["params"])
            self._ranges.append((self.params_fb.offsets[k], self.params_fb.offsets[k + n]))
            k += n
        self._step = 0
        # learning rates and the step counter also live on the device: the update kernels read them there, so a captured
        # HIP graph of the step stays valid while the schedule changes the learning rate
        self.lr_dev = torch.zeros(len(self.param_groups), dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._lr_host = None

    def grad_view(self, p):
        return self.grads_fb.views[self.index[id(p)]]

    def flat_params(self):
        """Parameters in the order of the flat buffers."""
        return [p for g in self.param_groups for p in g["params"]]

    def zero_grad(self, set_to_none=False):
        ops.zero_(self.grads_fb.flat)

    def upload_lr(self):
        """Push the groups' current learning rates to the device (no-op when unchanged). Call outside graph capture."""
        lrs = [float(g["lr"]) for g in self.param_groups]
        if lrs != self._lr_host:
            self.lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=False)
            self._lr_host = lrs

    @torch.no_grad()
    def step(self, closure=None, upload=True):
        if upload:
            self.upload_lr()
        self._step += 1
        last = max(i for i, (lo, hi) in enumerate(self._ranges) if hi > lo)
        for gi, (g, (lo, hi)) in enumerate(zip(self.param_groups, self._ranges)):
            if hi == lo:
                continue
            b1, b2 = g["betas"]
            ops.adam_step_dev_(self.params_fb.flat[lo:hi], self.grads_fb.flat[lo:hi], self.m_fb.flat[lo:hi],
                               self.v_fb.flat[lo:hi], self.lr_dev[gi:gi + 1], b1, b2, g["eps"], g["weight_decay"],
                               self.step_dev, 1 if gi == last else 0)
            for p in g["params"]:
                CACHE.touch(p)   # the kernel wrote the weights behind PyTorch's back: invalidate packed copies

    def note_replayed_step(self):
        """Bookkeeping after a graph replay that contained the update kernels."""
        self._step += 1
        self._opt_called = True                      # what lr_scheduler's "step() before optimizer.step()" check reads
        self._step_count = getattr(self, "_step_count", 0) + 1

    def state_dict(self):
        for g in self.param_groups:
            for p in g["params"]:
                self.state[p]["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():   # re-alias the moments into the flat buffers
            for p, i in ((p, self.index[id(p)]) for g in self.param_groups for p in g["params"]):
                st = self.state[p]
                self.m_fb.views[i].copy_(st["exp_avg"]); st["exp_avg"] = self.m_fb.views[i]
                self.v_fb.views[i].copy_(st["exp_avg_sq"]); st["exp_avg_sq"] = self.v_fb.views[i]
                self._step = max(self._step, int(st["step"]))
            self.step_dev.fill_(self._step)
            self._lr_host = None


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(momentum=0.9, nesterov=True) on flat buffers (train.py:179-180: phase 1 / step 0 with --optim sgd),
    one launch per parameter group (each group keeps its own lr / weight_decay, train.py:145-177). Same flat-buffer and
    device-resident-lr design as FusedAdam; `state_dict()` exposes `momentum_buffer` per parameter like torch."""

    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=0.0, nesterov=True):
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, dampening=0)
        super().__init__(params, defaults)
        plist = [p for g in self.param_groups for p in g["params"]]
        if not plist:
            raise ValueError("FusedSGD: no parameters")
        dev = plist[0].device
        shapes = [p.shape for p in plist]
        self.params_fb = FlatBuffer(shapes, dev)
        self.grads_fb = FlatBuffer(shapes, dev)
        self.buf_fb = FlatBuffer(shapes, dev)
        self.index = {}
        with torch.no_grad():
            for i, p in enumerate(plist):
                self.params_fb.views[i].copy_(p.data)
                p.data = self.params_fb.views[i]
                self.index[id(p)] = i
                self.state[p] = {"momentum_buffer": self.buf_fb.views[i]}
        self._ranges, k = [], 0
        for g in self.param_groups:
            n = len(g["params"])
            self._ranges.append((self.params_fb.offsets[k], self.params_fb.offsets[k + n]))
            k += n
        self.lr_dev = torch.zeros(len(self.param_groups), dtype=torch.float32, device=dev)
        self._lr_host = None

    def grad_view(self, p):
        return self.grads_fb.views[self.index[id(p)]]

    def zero_grad(self, set_to_none=False):
        ops.zero_(self.grads_fb.flat)

    def upload_lr(self):
        lrs = [float(g["lr"]) for g in self.param_groups]
        if lrs != self._lr_host:
            self.lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=False)
            self._lr_host = lrs

    @torch.no_grad()
    def step(self, closure=None, upload=True):
        if upload:
            self.upload_lr()
        for gi, (g, (lo, hi)) in enumerate(zip(self.param_groups, self._ranges)):
            if hi == lo:
                continue
            ops.sgd_step_(self.params_fb.flat[lo:hi], self.grads_fb.flat[lo:hi], self.buf_fb.flat[lo:hi], g["lr"],
                          g["momentum"], g["weight_decay"], g["nesterov"], lr_dev=self.lr_dev[gi:gi + 1])
            for p in g["params"]:
                CACHE.touch(p)

    def note_replayed_step(self):
        """Bookkeeping after a graph replay that contained the update kernels (SGD keeps no step counter)."""
        self._opt_called = True
        self._step_count = getattr(self, "_step_count", 0) + 1

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for p, i in ((p, self.index[id(p)]) for g in self.param_groups for p in g["params"]):
                st = self.state[p]
                if st.get("momentum_buffer") is not None:
                    self.buf_fb.views[i].copy_(st["momentum_buffer"])
                st["momentum_buffer"] = self.buf_fb.views[i]
            self._lr_host = None
