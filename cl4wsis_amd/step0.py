"""Step-0 (supervised) training iteration — train.py:241-300 with --step 0 --bce (BASELINE config 1): body, DeepLabV3,
classifier, instance decoder and heads all train (Adam), the instance branch back-propagates into the backbone features
(detach_instance is off at step 0, segmentation_module.py:92), loss = BCE-with-ignore on the upsampled semantic logits
+ 200 * Weighted_MSE(centre) + 0.01 * Weighted_L1(offset) against the loader's ground-truth maps."""
import torch

from . import body_train as BT
from . import body_train, heads_train, ops
from .nn import PackBatch


class Step0:
    def __init__(self, trainer):
        self.tr = trainer
        self._zero_w = None
        self.packs = PackBatch()

    def run(self, images, seg_maps, center_maps, offset_maps, weights, dropout_keep=None):
        with self.packs:
            self.packs.refresh()          # r05: every fp16x3 weight pack of the iteration in one launch (nn.PackBatch)
            return self._run(images, seg_maps, center_maps, offset_maps, weights, dropout_keep)

    def _run(self, images, seg_maps, center_maps, offset_maps, weights, dropout_keep):
        tr = self.tr
        bs, _, H, W = images.shape
        G = lambda p: tr._grad_map[id(p)]
        model = tr.model
        tr.optimizer.zero_grad()
        outs, btape = body_train.body_forward(model.body, images)
        feats = {f"res{i + 1}": o for i, o in enumerate(outs)}
        hy, htape = heads_train.deeplab_forward(model.head, outs[-1])
        seg, cx = heads_train.cls_forward(model.cls, hy)
        branch = tr._branch
        if dropout_keep is None:      # the trainer's device-resident seed, advanced by the launch: the same code path eager and under graph replay
            n5, _, h5, w5 = feats["res5"].shape
            drop = model.decoder.instance_decoder.aspp.project
            dropout_keep = ops.dropout_mask_dev((n5, drop[0].out_channels, h5, w5), drop[3].p, tr._seed_dev, 1, tr.device)
        pred, itape = branch.forward_train(feats, dropout_keep=dropout_keep)

        # semantic loss on the upsampled logits (train.py:281-282,296-299)
        seg_up = ops.resize_bilinear(seg, (H, W), False)
        seg_loss, dseg_up = ops.bce_ignore_mean(seg_up, seg_maps)
        dseg = ops.resize_bilinear_bwd(dseg_up, tuple(seg.shape[2:]), False)
        # centre / offset losses: the phase-2 loss kernel with no old classes, the ground-truth maps as the only target and
        # weights 2 x (200, 0.01) — its terms carry the 0.5 of the phase-2 blend (train.py:509-522)
        if self._zero_w is None or self._zero_w.shape != weights.shape:
            self._zero_w = torch.zeros_like(weights)
        pseudo = {"center": center_maps, "offset": offset_maps, "weight": weights}
        l4, dcen, doff = ops.p2_loss(pred["center"], pred["offset"], None, None, self._zero_w, pseudo, None, (H, W),
                                     center_weight=2 * tr.center_loss_weight, offset_weight=2 * tr.offset_loss_weight)
        # backward: the instance branch's weight gradients follow the arithmetic of the rest of the network (body_train)
        fg = {}
        saved, ops.WGRAD_PRECISION = ops.WGRAD_PRECISION, BT.WGRAD_PRECISION
        try:
            branch.backward(itape, {"center": dcen, "offset": doff}, grads=tr._grad_map, feature_grads=fg)
        finally:
            ops.WGRAD_PRECISION = saved
        # r05: the weight gradients of the heads and the body on the trainer's second stream, beside the input-gradient chain (ops.WgradSide)
        side = tr._side_stream if (ops.WGRAD_SIDE_ON and getattr(tr, "_side_stream", None) is not None and ops.STAT_EXCHANGE.hook is None) else None
        with ops.WGRAD_SIDE.on(side):
            dhy = heads_train.cls_backward(model.cls, cx, dseg, G)
            dres5 = heads_train.deeplab_backward(model.head, htape, dhy, G)
            ops.axpy_(dres5, fg["res5"])
            douts = {4: dres5, 3: fg.get("res4"), 2: fg.get("res3"), 1: fg.get("res2")}
            body_train.body_backward(model.body, btape, douts, G)
        return {"seg_loss": seg_loss, "center_loss": l4[2:3], "offset_loss": l4[3:4]}
