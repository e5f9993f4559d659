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

    def grad_view(self, p):
        return self.grads_fb.views[self.index[id(p)]]

    def zero_grad(self, set_to_none=False):
        ops.zero_(self.grads_fb.flat)

    @torch.no_grad()
    def step(self, closure=None):
        self._step += 1
        for g, (lo, hi) in zip(self.param_groups, self._ranges):
            if hi == lo:
                continue
            b1, b2 = g["betas"]
            ops.adam_step_(self.params_fb.flat[lo:hi], self.grads_fb.flat[lo:hi], self.m_fb.flat[lo:hi],
                           self.v_fb.flat[lo:hi], g["lr"], b1, b2, g["eps"], g["weight_decay"], self._step)
            for p in g["params"]:
                self.state[p]["step"] += 1
                CACHE.touch(p)   # the kernel wrote the weights behind PyTorch's back: invalidate packed copies

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():   # re-alias the moments into the flat buffers
            for p, i in ((p, self.index[id(p)]) for g in self.param_groups for p in g["params"]):
                st = self.state[p]
                self.m_fb.views[i].copy_(st["exp_avg"]); st["exp_avg"] = self.m_fb.views[i]
                self.v_fb.views[i].copy_(st["exp_avg_sq"]); st["exp_avg_sq"] = self.v_fb.views[i]
                self._step = max(self._step, int(st["step"]))
