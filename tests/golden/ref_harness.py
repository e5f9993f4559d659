"""Import harness for the upstream CL4WSIS reference (build container only — /root/reference never
travels to the GPU box). Registers the stand-in modules for third-party packages that are not
installed here and exposes the reference modules.

This file contains no reference source: it only arranges sys.modules so that the reference's own
files can be imported unmodified from /root/reference.

The stand-ins are NOT neutral plumbing in one place: the `inplace_abn` stub (absent PyPI package, version unpinned by the
reference) encodes this repository's BELIEF about that package's arithmetic — InPlaceABN / InPlaceABNSync normalise with
`|weight| + eps` and back-propagate `sign(weight)` into the weight, plain ABN uses the signed weight (DESIGN.md section 2,
"Unpinned third-party arithmetic"). Every fixture generated through this harness inherits that assumption; so do the OpenCV
connected-components stand-in (the oracle's own `connected_components`) and the chainercv / pycocotools restatements.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("CL4WSIS_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _install_stubs():
    sys.dont_write_bytecode = True
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.labels import connected_components

    # --- inplace_abn: BN + activation with the attribute contract the reference relies on --------------
    class ABN(nn.Module):
        def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                     activation_param=0.01, group=None):
            super().__init__()
            self.num_features, self.eps, self.momentum = num_features, eps, momentum
            self.activation, self.activation_param = activation, activation_param
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
            self.register_buffer("running_mean", torch.zeros(num_features))
            self.register_buffer("running_var", torch.ones(num_features))

        def forward(self, x):
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training,
                             self.momentum, self.eps)
            if self.activation == "leaky_relu":
                return F.leaky_relu(y, self.activation_param)
            if self.activation == "elu":
                return F.elu(y, self.activation_param)
            return y

    class InPlaceABN(ABN):
        """The in-place variants' fused kernels scale with |weight| + eps (believed behaviour of mapillary/inplace_abn,
        DESIGN.md §2); the plain ABN above goes through functional.batch_norm with the weight as it is."""

        def forward(self, x):
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight.abs() + self.eps, self.bias, self.training,
                             self.momentum, self.eps)
            if self.activation == "leaky_relu":
                return F.leaky_relu(y, self.activation_param)
            if self.activation == "elu":
                return F.elu(y, self.activation_param)
            return y

    abn = types.ModuleType("inplace_abn")
    abn.ABN = ABN
    abn.InPlaceABN = abn.InPlaceABNSync = InPlaceABN
    abn.functions = types.ModuleType("inplace_abn.functions")
    sys.modules["inplace_abn"] = abn
    sys.modules["inplace_abn.functions"] = abn.functions

    # --- cv2: connected components via the oracle's scipy restatement ---------------------------------------
    cv2 = types.ModuleType("cv2")
    cv2.CC_STAT_AREA = 4

    def ccws(mask, connectivity=8):
        n, lab, area, cent = connected_components(np.asarray(mask), connectivity)
        stats = np.zeros((n, 5), dtype=np.int32)
        stats[:, 4] = area
        return n, lab, stats, cent

    cv2.connectedComponentsWithStats = ccws
    cv2.norm = lambda *a, **k: None
    sys.modules["cv2"] = cv2

    # --- torchvision.transforms.functional (normalize / rotate by multiples of 90 degrees) ------------------
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.normalize = lambda t, mean, std: (t - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
    tvf.rotate = lambda t, angle: torch.rot90(t, int(round(angle / 90)) % 4, dims=(-2, -1))
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    for name in ("chainercv", "chainercv.utils", "chainercv.utils.mask", "chainercv.utils.mask.mask_iou",
                 "chainercv.evaluations", "wandb"):
        sys.modules.setdefault(name, types.ModuleType(name))


_loaded = {}


def load_reference():
    """Returns a dict of the reference modules needed by the golden generator."""
    if _loaded:
        return _loaded
    _install_stubs()
    sys.path.insert(0, REF)
    # a package named `tasks` / `argparser` also exists in cl4wsis_amd, but those are submodules -> no clash
    torch.Tensor.cuda = lambda self, *a, **k: self          # PeakGenerator hard-codes .cuda() (wss/modules.py:400-401)
    import models  # noqa: F401  (must precede `modules`: circular import models/resnet.py:7 <-> modules/deeplab.py:5)
    import segmentation_module
    import argparser
    import tasks
    from modules import utils as bestie
    from wss import modules as wss_modules
    from wss import utils as wss_utils

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    _loaded.update(dict(segmentation_module=segmentation_module, argparser=argparser, tasks=tasks, bestie=bestie,
                        wss_modules=wss_modules, wss_utils=wss_utils,
                        loss=by_path("ref_utils_loss", "utils/loss.py"),
                        wss_loss=by_path("ref_utils_wss_loss", "utils/wss_loss.py"),
                        dataset_utils=by_path("ref_dataset_utils", "dataset/utils.py")))
    return _loaded
