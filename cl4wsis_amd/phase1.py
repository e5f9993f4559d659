"""Phase-1 training iteration (train.py:241-566 with --phase 1; BASELINE config 2): the body, DeepLabV3 head, classifier,
PseudoLabeler and PeakGenerator train with SGD-Nesterov against the weakly-supervised loss stack

    l_cam_new (nGWP + focal BCE) + l_loc + FLAC + l_peak + lde            always
    + l_seg + l_cls (PAMR-refined pseudo masks) + RandDrop                from epoch `pseudo_ep` on

Everything runs on the HIP kernels of libcl4wsis_hip.so: train-mode forward with cross-rank batch statistics
(body_train / heads_train), the loss kernels of wss.hip / p1loss.hip (value + gradient), the hand-written backward, the
bucketed RCCL gradient average and the fused SGD. The frozen old model uses the eval path (bf16x3 convs).
"""
import os

import numpy as np
import torch

from . import body_train, heads_train, ops
from .nn import PackBatch

DENORM_MEAN = (0.485, 0.456, 0.4069)      # utils/utils.py:23 (sic)
DENORM_STD = (0.229, 0.224, 0.225)
P1_OVERLAP = os.environ.get("CL4_P1_OVERLAP", "0") != "0"      # r05 experiment: the frozen old model on the second stream (measured: no gain, FINDINGS.md)


class Phase1Step:
    def __init__(self, trainer):
        self.tr = trainer
        dev = trainer.device
        self._dn_scale = torch.tensor(DENORM_STD, dtype=torch.float32, device=dev)
        self._dn_shift = torch.tensor(DENORM_MEAN, dtype=torch.float32, device=dev)
        self._draws = 0
        self.packs = PackBatch()

    def run(self, images, l1h, cur_epoch, flac_angle=None, randrop_neg=None, debug=None):
        tr = self.tr
        # r05: two things of this iteration do not depend on the kernels around them and run on the trainer's second stream — the frozen old
        # model's forward (needed by the losses only) beside the new model's forward, and every weight gradient beside the input gradients
        # and ABN backward passes of the layers behind it (ops.WGRAD_SIDE). Not when the iteration is being captured as graph segments
        # (N > 1: a segment must not end with work forked off it).
        side = tr._side_stream if (P1_OVERLAP and getattr(tr, "_side_stream", None) is not None and ops.STAT_EXCHANGE.hook is None
                                     and ops.TIMER is None) else None
        main = torch.cuda.current_stream() if side is not None else None
        with self.packs, ops.WGRAD_SIDE.on(side if ops.WGRAD_SIDE_ON else None):
            self.packs.refresh()          # r05: every fp16x3 weight pack of the iteration (forward + input-gradient operands) in one launch
            return self._run(images, l1h, cur_epoch, flac_angle, randrop_neg, debug, side, main)

    def _run(self, images, l1h, cur_epoch, flac_angle, randrop_neg, debug, side, main):
        tr, opts = self.tr, self.tr.opts
        bs = images.shape[0]
        old, tot = tr.old_classes, tr.tot_classes
        G = lambda p: tr._grad_map[id(p)]
        model, pl, pg = tr.model, tr.pseudolabeler, tr.peakgenerator

        # train.py:253-255 — frozen old model (eval), low-resolution outputs
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                out_old, feats_old = tr.model_old(images, interpolate=False)
        else:
            out_old, feats_old = tr.model_old(images, interpolate=False)
        old_seg, res5_old = out_old["seg"], feats_old["body"]

        tr.optimizer.zero_grad()
        # train.py:272 — new model in training mode (branch 'none' in phase 1: body -> DeepLabV3 -> classifier)
        outs, btape = body_train.body_forward(model.body, images)
        res5 = outs[-1]
        hy, htape = heads_train.deeplab_forward(model.head, res5)
        seg, cx = heads_train.cls_forward(model.cls, hy)

        # train.py:304-311 — PseudoLabeler in eval mode on the same features, detached
        pl.eval()
        int_masks = pl(res5)
        pl.train()
        # train.py:312-318 — FLAC batch [x, flip(x), rot(x, angle)] through the PseudoLabeler in training mode
        if flac_angle is None:
            flac_angle = [90, 180, 270][np.random.randint(3)]
        k = flac_angle // 90
        n, c, h, w = res5.shape
        feat3 = torch.empty((3 * bs, c, h, w), dtype=torch.float32, device=res5.device)
        ops.copy_channels(res5, feat3[:bs], c)
        ops.flip_w(res5, out=feat3[bs:2 * bs])
        ops.rot90(res5, k, out=feat3[2 * bs:])
        A, ptape = heads_train.pseudolabeler_forward(pl, feat3)
        raw = A[:bs]
        lab_new = l1h[:, old - 1:].contiguous()

        if side is not None:
            main.wait_stream(side)                   # the old model's outputs enter the losses from here on
        losses = {}
        losses["flac"], dA = ops.flac_loss(A, bs, old, flac_angle)
        draw = dA[:bs]
        losses["l_cam_new"], _, g = ops.ngwp_focal_bce(raw, lab_new, reduction="mean")
        ops.axpy_(draw, g)
        losses["l_peak"], _, g, _, _ = ops.peakgen_train_loss(raw, pg.extra_conv4.weight.detach(), pg.extra_conv4.bias.detach(), lab_new,
                                                             alpha=pg.alpha, dweight=G(pg.extra_conv4.weight).view(lab_new.shape[1], -1),
                                                             dbias=G(pg.extra_conv4.bias))
        ops.axpy_(draw, g)
        losses["l_loc"], g = ops.lloc_loss(raw, old_seg, labels=l1h if opts.no_mask else None)
        ops.axpy_(draw, g)
        losses["lde"], dres5 = ops.mse_loss(res5, res5_old, grad_scale=tr.lde)
        dseg = None
        if cur_epoch >= tr.pseudo_epoch:
            # train.py:370-416 — PAMR-refined pseudo masks, l_seg, l_cls, RandDrop
            soft = ops.softmax_channels(int_masks)
            if tr.use_aff:
                img_raw = ops.channel_affine(images, self._dn_scale, self._dn_shift)
                im = ops.resize_bilinear(img_raw, (h, w), True)
                soft = ops.pamr(im, soft)
            pseudo_gt = ops.pseudo_gtmask(soft, l1h)
            losses["l_seg"], dseg = ops.p1_seg_loss(seg, old_seg, int_masks, l1h, alpha=opts.alpha, icarl_bkg=opts.icarl_bkg,
                                                    weight=opts.l_seg)
            losses["l_cls"], g = ops.balanced_mask_loss_ce(raw, pseudo_gt, l1h)
            ops.axpy_(draw, g)
            if opts.randrop:
                if randrop_neg is None:
                    # device-resident seed, advanced by the launch: the same code path eager and under HIP-graph replay
                    randrop_neg = ops.random_labels_dev((bs, h, w), old, tr._p1_seed_dev, 1, res5.device)
                losses["l_rand"], g = ops.randrop_loss(raw, int_masks, l1h, randrop_neg, old, mask_labels=l1h if opts.no_mask else None)
                ops.axpy_(draw, g)

        if debug is not None:
            debug.update(A=A.clone(), dA=dA.clone(), res5=res5.clone(), res5_old=res5_old.clone(), old_seg=old_seg.clone(),
                         seg=seg.clone(), dres5_lde=dres5.clone())
        # backward ------------------------------------------------------------------------------------------------
        dfeat3 = heads_train.pseudolabeler_backward(pl, ptape, dA, G)
        ops.axpy_(dres5, dfeat3[:bs])
        ops.axpy_(dres5, ops.flip_w(dfeat3[bs:2 * bs]))
        ops.axpy_(dres5, ops.rot90(dfeat3[2 * bs:], -k))
        if dseg is not None:
            dhy = heads_train.cls_backward(model.cls, cx, dseg, G)
            ops.axpy_(dres5, heads_train.deeplab_backward(model.head, htape, dhy, G))
        body_train.body_backward(model.body, btape, {4: dres5}, G)
        return losses
