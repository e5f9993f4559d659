"""Model factory with the reference's plugin surface (segmentation_module.py:17-97, 289-451):
``get_norm(opts)``, ``get_body(opts, norm)``, ``make_model(opts, classes)`` and
``IncrementalInstanceSegmentationModule`` (``forward`` / ``forward_seg`` / attributes
``body, head, cls, decoder, instance_head, classes, tot_classes``; identical state_dict keys).
All compute runs in the HIP kernels of libcl4wsis_hip.so.
"""
from collections import OrderedDict
from functools import partial, reduce

import torch
import torch.nn as nn

from . import backbone, ops
from .heads import (DeeplabV3, IncrementalClassifier, IncrementalPanopticDeepLabDecoder,
                    IncrementalSinglePanopticDeepLabHead, InstanceBranch)
from .nn import ABN, InPlaceABN, InPlaceABNSync


def get_norm(opts):
    """segmentation_module.py:17-29. Every variant on the shipped scripts resolves to BN + leaky-ReLU(0.01);
    the cross-replica statistics of iabn_sync only matter in train mode (step 0 / phase 1)."""
    if opts.norm_act in ("iabn_sync", "iabn", "abn", "std"):
        cls = {"iabn_sync": InPlaceABNSync, "iabn": InPlaceABN}.get(opts.norm_act, ABN)
        return partial(cls, activation="leaky_relu", activation_param=.01)
    if opts.norm_act in ("abr", "iabr"):
        raise NotImplementedError("--norm_act abr/iabr (modules/custom_bn.py) is outside the hot-path scope (SURVEY.md §2 #12)")
    return partial(ABN, activation="leaky_relu", activation_param=.01)


def get_body(opts, norm):
    """segmentation_module.py:32-58."""
    ctor = getattr(backbone, f"net_{opts.backbone}")
    body = ctor(norm_act=norm, output_stride=opts.output_stride, keep_outputs=(opts.model == "PanopticDeepLab"))
    if not opts.no_pretrained:
        path = ("pretrained/wide_resnet38_ipabn_lr_256.pth.tar" if opts.backbone == "wider_resnet38_a2"
                else f"pretrained/{opts.backbone}_iabn_sync.pth.tar")
        pre = torch.load(path, map_location="cpu")
        state = {(k[7:] if "module" in k else k): v for k, v in pre["state_dict"].items()}
        state.pop("classifier.fc.weight", None)
        state.pop("classifier.fc.bias", None)
        body.load_state_dict(state)
    return body


def make_model(opts, classes=None, return_attn=False):
    """segmentation_module.py:61-97."""
    norm = get_norm(opts)
    body = get_body(opts, norm)
    if opts.model != "PanopticDeepLab":
        raise NotImplementedError("only --model PanopticDeepLab is selectable in the reference CLI (argparser.py:113-115)")
    head = None
    if opts.use_DeeplabV3_as_seg_branch:
        head = DeeplabV3(body.out_channels, 256, 256, norm_act=norm, out_stride=opts.output_stride,
                         pooling_size=opts.pooling)
    branch = {"all": ["seg", "ins"], "seg": ["seg"], "ins": ["ins"]}.get(opts.branch, [])
    detach_instance = (opts.step > 0 and opts.weakly and opts.pseudo is None) or opts.detach_instance
    if classes is None:
        raise NotImplementedError("non-incremental model (classes=None) is not reachable from run.py")
    return IncrementalInstanceSegmentationModule(body, classes=classes, branch=branch, seg_head=head,
                                                 detach_instance=detach_instance)


class IncrementalInstanceSegmentationModule(nn.Module):
    """segmentation_module.py:289-451."""

    def __init__(self, body, classes, branch, seg_head=None, detach_instance=False, low_level_channels=None):
        super().__init__()
        assert isinstance(classes, list), \
            "Classes must be a list where to every index correspond the num of classes for that task"
        self.body = body
        self.branch = branch
        oc = body.out_channels
        self.decoder = None
        if len(branch) > 0:
            self.decoder = IncrementalPanopticDeepLabDecoder(
                in_channels=oc, feature_key="res5",
                low_level_channels=(oc // 2, oc // 4, oc // 8) if low_level_channels is None else low_level_channels,
                low_level_key=["res4", "res3", "res2"], low_level_channels_project=(128, 64, 32),
                decoder_channels=256, atrous_rates=(3, 6, 9), branch=branch, detach_instance=detach_instance)
        self.head = None
        self.cls = None
        self.semantic_head = None
        self.instance_head = None
        if seg_head is None:
            raise NotImplementedError("Panoptic-DeepLab semantic head: the CLI always uses DeepLabV3 (argparser.py:19)")
        self.head = seg_head
        self.cls = IncrementalClassifier([nn.Conv2d(256, c, 1) for c in classes])
        if "ins" in self.branch:
            ins_classes = classes.copy()
            ins_classes[0] -= 1  # minus background
            self.instance_head = IncrementalSinglePanopticDeepLabHead(
                decoder_channels=128, head_channels=(128, 32), num_classes=(ins_classes, 2),
                class_key=["center", "offset"])
        self.classes = classes
        self.tot_classes = reduce(lambda a, b: a + b, self.classes)

    # ------------------------------------------------------------------------------------------
    def instance_branch(self):
        if self.decoder is None or self.instance_head is None:
            return None
        return InstanceBranch(self.decoder, self.instance_head)

    def _features(self, x):
        outs = self.body(x)
        return outs[-1], {f"res{i + 1}": o for i, o in enumerate(outs)}

    def _upsample_predictions(self, pred, size):
        return OrderedDict((k, ops.resize_bilinear(v, tuple(size), True)) for k, v in pred.items())

    @torch.no_grad()
    def forward_seg(self, x, as_feature_extractor=False, interpolate=True, scales=None, do_flip=False):
        """Body + DeepLabV3 + classifiers (segmentation_module.py:400-426). Always runs without autograd: the
        semantic sub-net is frozen wherever the reference calls this (train.py:260-261)."""
        x_b, feats = self._features(x)
        if as_feature_extractor:
            return {"body": x_b, "features": feats}
        pred = OrderedDict(seg=self.cls(self.head(x_b)))
        if interpolate:
            pred = self._upsample_predictions(pred, x.shape[-2:])
        return pred, {"body": x_b, "features": feats}

    def forward(self, x, as_feature_extractor=False, interpolate=True, scales=None, do_flip=False):
        """segmentation_module.py:360-398. Eval mode (the frozen old model of train.py:253-255, validation): fused eval kernels,
        no autograd. Training mode (`outputs, features = model(images, interpolate=False)`, train.py:274: step 0 and phase 1):
        every sub-module that is in training mode runs its batch-statistics forward under torch.autograd
        (cl4wsis_amd/autograd.py), so the reference's loss code + `loss.backward()` work unchanged; sub-modules switched to
        eval() (train.py:234-240 freezes body / head / cls in phase 2) run the eval kernels and receive no gradient."""
        if not (self.training and torch.is_grad_enabled()):
            with torch.no_grad():
                return self._forward_eval(x, as_feature_extractor, interpolate)
        from . import autograd as AG
        if self.body.training:
            outs = AG.apply(AG.BodyRunner(self.body), [x], list(self.body.parameters()))
        else:
            with torch.no_grad():
                outs = self.body(x)
        x_b = outs[-1]
        feats = {f"res{i + 1}": o for i, o in enumerate(outs)}
        if as_feature_extractor:
            return {"body": x_b, "features": feats}
        pred = OrderedDict()
        instance_l = None
        if self.decoder is not None:
            _, instance_l = self.decoder(feats)
        pred["seg"] = self.cls(self.head(x_b))
        if "ins" in self.branch:
            pred.update(self.instance_head(instance_l))
        if interpolate:
            pred = OrderedDict((k, AG.apply(AG.ResizeRunner(x.shape[-2:], True), [v], [])) for k, v in pred.items())
        return pred, {"body": x_b, "features": feats}

    def _forward_eval(self, x, as_feature_extractor=False, interpolate=True):
        x_b, feats = self._features(x)
        if as_feature_extractor:
            return {"body": x_b, "features": feats}
        pred = OrderedDict(seg=self.cls(self.head(x_b)))
        if "ins" in self.branch:
            pred.update(self.instance_branch().forward_eval(feats))
        if interpolate:
            pred = self._upsample_predictions(pred, x.shape[-2:])
        return pred, {"body": x_b, "features": feats}

    def init_new_classifier(self, device):
        """segmentation_module.py:342-358 (`--init_balanced`): the newest semantic and centre classifiers start from the
        background / first-channel weights with bias = bias_bkg - log(N_new + 1), and the background bias is lowered by the same
        amount. The reference's body addresses `self.segmentation_head['seg']['cls']`, an attribute its own constructor never
        creates with the DeepLabV3 semantic head (the flag therefore raises AttributeError upstream and no script sets it);
        this is the same arithmetic applied to the classifiers that do exist: `cls` and `instance_head.classifier.center.cls`."""
        import math
        groups = [self.cls]
        if self.instance_head is not None:
            groups.append(self.instance_head.classifier["center"]["cls"])
        with torch.no_grad():
            for all_cls in groups:
                new, first = all_cls[-1], all_cls[0]
                new_bias = first.bias[0] - math.log(self.classes[-1] + 1)
                new.weight.copy_(first.weight[0].expand_as(new.weight))
                new.bias.fill_(float(new_bias))
                first.bias[0] = new_bias

    def freeze(self):
        for par in self.parameters():
            par.requires_grad = False

    def fix_bn(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, ABN)):
                m.eval()
                m.weight.requires_grad = False
                m.bias.requires_grad = False
