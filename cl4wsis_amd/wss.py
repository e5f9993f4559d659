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
from .nn import conv_abn, conv_abn_planes, conv_bias, planes_of, planes_ok


class PseudoLabeler(nn.Module):
    def __init__(self, channels=2048, num_classes=21, norm=None):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, 256, kernel_size=3, stride=1, padding=1, bias=False)
        self.norm1 = norm(256)
        self.conv2 = nn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False)
        self.norm2 = norm(256)
        self.cls = nn.Conv2d(256, num_classes, kernel_size=1, stride=1)
        self.num_classes = num_classes

    def forward(self, x):
        """wss/modules.py:335-343. Training mode (phase 1, train.py:311-318): batch statistics over all ranks, differentiable
        through cl4wsis_amd/autograd.py; eval mode (train.py:304-309, 425-426): fused eval kernels, no autograd."""
        if self.training and torch.is_grad_enabled():
            from . import autograd as AG
            return AG.apply(AG.PseudoLabelerRunner(self), [x], list(self.parameters()))
        with torch.no_grad():
            if self.training:
                from . import heads_train
                return heads_train.pseudolabeler_forward(self, x.contiguous())[0]
            xp = planes_of(x)
            if xp is not None and planes_ok(self.conv1) and planes_ok(self.conv2):
                # the body handed its output out pre-split: both 3x3 convolutions run on the planes kernel (csrc/conv_s.hip)
                yp = conv_abn_planes(xp, self.conv1, self.norm1)[1]
                y = conv_abn_planes(yp, self.conv2, self.norm2, want_fp32=True, want_planes=False)[0]
            else:
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
        """wss/modules.py:390-410 — returns (logit [B, num_classes], map): the nGWP logits with zeros for the old classes and,
        in eval mode, the normalised CAM [B, num_classes, H, W] (old-class channels zero) that the caller smooths and upsamples
        (train.py:427-431; the step itself uses the fused `cam`). In training mode the second item of the reference is the raw
        conv map, which train.py:334-336 discards (`peak_logits, _ = ...`): None is returned instead of materialising it.
        `size` other than the input size is not on the path (cam_normalize is always called with the map's own size)."""
        B, _, H, W = x.shape
        if size is not None and tuple(size) != (H, W):
            raise NotImplementedError("PeakGenerator(size != input size) is never used by train.py")
        w, b = self.extra_conv4.weight, self.extra_conv4.bias
        if self.training and torch.is_grad_enabled():
            from . import autograd as AG
            new = AG.apply(AG.PeakGeneratorRunner(self), [x], [w, b])
        else:
            with torch.no_grad():
                new = ops.peakgen_logits(x.contiguous(), w.detach(), b.detach(), self.new_classes, self.alpha)
        logit = torch.cat([new.new_zeros(B, self.old_classes), new], 1) if self.old_classes > 0 else new   # tuple assembly
        if self.training:
            return logit, None
        if label is None:
            raise RuntimeError("PeakGenerator.forward in eval mode needs the image-level labels (wss/modules.py:408)")
        lab = label[:, -self.new_classes:].to(torch.float32).contiguous()
        with torch.no_grad():
            cam_new = ops.cam_normalize(x.contiguous(), w.detach(), b.detach(), lab, self.alpha)
            cam = x.new_zeros(B, self.num_classes, H, W)
            ops.copy_channels(cam_new, cam, self.new_classes, out_coff=self.old_classes)
        return logit, cam
