"""Weak-supervision modules on the phase-2 path (reference: wss/modules.py:322-343, 346-434).

`PseudoLabeler` (two 3x3 convs + ABN, 1x1 classifier) and `PeakGenerator` (PAM + 1x1 conv + CAM normalisation)
keep the reference's parameter names (`conv1, norm1, conv2, norm2, cls`; `extra_conv4`) so that checkpoints
(`pseudolabeler` / `peakgenerator` entries, run.py:17-33) load unchanged. Phase 2 only evaluates them
(train.py:425-428); their training path belongs to phase 1.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .nn import conv_abn, conv_bias


class PseudoLabeler(nn.Module):
    def __init__(self, channels=2048, num_classes=21, norm=None):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, 256, kernel_size=3, stride=1, padding=1, bias=False)
        self.norm1 = norm(256)
        self.conv2 = nn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False)
        self.norm2 = norm(256)
        self.cls = nn.Conv2d(256, num_classes, kernel_size=1, stride=1)
        self.num_classes = num_classes

    @torch.no_grad()
    def forward(self, x):
        if self.training:
            raise NotImplementedError("PseudoLabeler training (phase 1, FLAC) is a later §8 row; phase 2 calls .eval()")
        y = conv_abn(x, self.conv1, self.norm1)
        y = conv_abn(y, self.conv2, self.norm2)
        return conv_bias(y, self.cls)


class PeakGenerator(nn.Module):
    def __init__(self, in_channel=2048, num_classes=20, alpha=0.7, init_weights=True, old_classes=15):
        super().__init__()
        self.num_classes = num_classes
        self.old_classes = old_classes
        self.new_classes = num_classes - old_classes
        self.alpha = alpha
        self.extra_conv4 = nn.Conv2d(self.new_classes, self.new_classes, kernel_size=1)
        if init_weights:
            n = self.new_classes  # kernel 1x1: n = k*k*out_channels (wss/modules.py:436-441)
            self.extra_conv4.weight.data.normal_(0, math.sqrt(2. / n))
            self.extra_conv4.bias.data.zero_()

    @torch.no_grad()
    def cam(self, int_masks, label_new):
        """Eval-mode CAM of the NEW classes after `smoothing` (wss/utils.py:28-32): [b, new_classes, h, w].
        The reference returns a 20-channel map whose old-class channels are identically zero
        (wss/modules.py:399-403); they are never materialised here."""
        return ops.cam_smooth(int_masks, self.extra_conv4.weight.detach(), self.extra_conv4.bias.detach(), label_new,
                              self.alpha)

    def forward(self, x, label=None, size=None):
        if self.training:
            raise NotImplementedError("PeakGenerator training (phase 1) is a later §8 row; phase 2 calls .eval()")
        raise RuntimeError("use PeakGenerator.cam(int_masks, label_new): the fused kernel also applies `smoothing`")
