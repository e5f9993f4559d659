"""LR schedules of the reference (utils/scheduler.py:5-51): poly, step, warm-up-then-poly, none."""
import torch
from torch.optim.lr_scheduler import _LRScheduler


class PolyLR(_LRScheduler):
    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1):
        self.power, self.max_iters = power, max_iters
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        f = (1 - self.last_epoch / self.max_iters) ** self.power
        return [b * f for b in self.base_lrs]


class NoScheduler(_LRScheduler):
    def get_lr(self):
        return self.base_lrs


class WarmUpPolyLR(_LRScheduler):
    """Constant until `start_decay` iterations, then polynomial decay (the `--lr_policy warmup` of scripts/run.sh)."""

    def __init__(self, optimizer, max_iters, power=0.9, start_decay=20, last_epoch=-1):
        self.power, self.max_iters, self.start_decay = power, max_iters, start_decay
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch < self.start_decay:
            return self.base_lrs
        f = (1 - self.last_epoch / self.max_iters) ** self.power
        return [b * f for b in self.base_lrs]


def get_scheduler(opts, optim):
    if opts.lr_policy == "poly":
        return PolyLR(optim, max_iters=opts.max_iters, power=opts.lr_power)
    if opts.lr_policy == "step":
        return torch.optim.lr_scheduler.StepLR(optim, step_size=opts.lr_decay_step, gamma=opts.lr_decay_factor)
    if opts.lr_policy == "warmup":
        return WarmUpPolyLR(optim, max_iters=opts.max_iters, power=opts.lr_power, start_decay=opts.start_decay)
    if opts.lr_policy == "none":
        return NoScheduler(optim)
    raise NotImplementedError
