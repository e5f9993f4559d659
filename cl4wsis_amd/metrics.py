"""Streaming semantic-segmentation metrics with the surface of the reference's `StreamSegMetrics`
(metrics/stream_metrics.py:34-131): confusion matrix, overall / mean accuracy, mean precision, mean IoU, per-class tables,
`synch` over the ranks. On the product path the confusion matrix is accumulated on the device by cl4_semseg_hist
(`update_device`); `update(label_trues, label_preds)` with host arrays keeps the reference's call working."""
import numpy as np
import torch


class StreamSegMetrics:
    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.confusion_matrix = np.zeros((n_classes, n_classes))
        self.total_samples = 0
        self._dev_hist = None

    # -- reference API (host arrays) -------------------------------------------------------------------------------------
    def update(self, label_trues, label_preds):
        for lt, lp in zip(label_trues, label_preds):
            lt, lp = np.asarray(lt).flatten(), np.asarray(lp).flatten()
            mask = (lt >= 0) & (lt < self.n_classes)
            self.confusion_matrix += np.bincount(self.n_classes * lt[mask].astype(int) + lp[mask],
                                                 minlength=self.n_classes ** 2).reshape(self.n_classes, self.n_classes)
        self.total_samples += len(label_trues)

    # -- device path ----------------------------------------------------------------------------------------------------------
    def device_hist(self, device):
        if self._dev_hist is None or self._dev_hist.device != device:
            self._dev_hist = torch.zeros(self.n_classes, self.n_classes, dtype=torch.int64, device=device)
        return self._dev_hist

    def flush_device(self, n_samples):
        """Fold the device-side counts into the host matrix (one D2H copy per validation run)."""
        if self._dev_hist is not None:
            self.confusion_matrix += self._dev_hist.cpu().numpy().astype(np.float64)
            self._dev_hist.zero_()
        self.total_samples += n_samples

    def reset(self):
        self.confusion_matrix = np.zeros((self.n_classes, self.n_classes))
        self.total_samples = 0
        if self._dev_hist is not None:
            self._dev_hist.zero_()

    def synch(self, device):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        cm = torch.tensor(self.confusion_matrix).to(device)
        n = torch.tensor(self.total_samples).to(device)
        dist.reduce(cm, dst=0)
        dist.reduce(n, dst=0)
        if dist.get_rank() == 0:
            self.confusion_matrix = cm.cpu().numpy()
            self.total_samples = int(n.cpu().numpy())

    def get_results(self):
        eps = 1e-6
        hist = self.confusion_matrix
        gt_sum = hist.sum(axis=1)
        mask = gt_sum != 0
        diag = np.diag(hist)
        acc = diag.sum() / hist.sum()
        acc_cls_c = diag / (gt_sum + eps)
        acc_cls = np.mean(acc_cls_c[mask])
        prec_c = diag / (hist.sum(axis=0) + eps)
        prec = np.mean(prec_c)
        iu = diag / (gt_sum + hist.sum(axis=0) - diag + eps)
        mean_iu = np.mean(iu[mask])
        tab = lambda v: dict(zip(range(self.n_classes), [v[i] if m else "X" for i, m in enumerate(mask)]))
        return {"Total samples": self.total_samples, "Overall Acc": acc, "Mean Acc": acc_cls, "Mean Prec": prec, "Mean IoU": mean_iu,
                "Class IoU": tab(iu), "Class Acc": tab(acc_cls_c), "Class Prec": tab(prec_c), "Agg": [mean_iu, acc_cls, prec],
                "Confusion Matrix": self.confusion_matrix.copy()}

    def to_str(self, results, verbose=True):
        skip = ["Class IoU", "Class Acc", "Class Prec", "Agg", "Confusion Matrix Pred", "Confusion Matrix", "Confusion Matrix Text"]
        s = "\n" + "".join("%s: %f\n" % (k, v) for k, v in results.items() if k not in skip)
        if verbose:
            s += "Class IoU:\n" + "".join("\tclass %d: %s\n" % (k, str(v)) for k, v in results["Class IoU"].items())
            for i, name in enumerate(["Class IoU", "Class Acc", "Class Prec"]):
                s += f"{name}:'\t: {results['Agg'][i]}\n"
        return s


# ---------------------------------------------------------------------------------------------------------------------
# Instance-segmentation AP (train.py:653-693 eval_detection_voc, metrics/voc_evaluation.py:11-140). chainercv's mask_iou and
# calc_detection_voc_ap are third-party code that is not part of the reference tree: restated from their documented
# behaviour (parity unpinned at that boundary).
# ---------------------------------------------------------------------------------------------------------------------
def mask_iou(a, b):
    a, b = a.reshape(len(a), -1).astype(np.float64), b.reshape(len(b), -1).astype(np.float64)
    inter = a @ b.T
    union = a.sum(1)[:, None] + b.sum(1)[None, :] - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-12), 0.0)


def calc_detection_voc_ap(prec, rec, use_07_metric=False):
    ap = np.empty(len(prec))
    for l in range(len(prec)):
        if prec[l] is None or rec[l] is None:
            ap[l] = np.nan
            continue
        if use_07_metric:
            ap[l] = 0
            for t in np.arange(0., 1.1, 0.1):
                p = 0 if np.sum(rec[l] >= t) == 0 else np.max(np.nan_to_num(prec[l])[rec[l] >= t])
                ap[l] += p / 11
        else:
            mpre = np.concatenate(([0], np.nan_to_num(prec[l]), [0]))
            mrec = np.concatenate(([0], rec[l], [1]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            i = np.where(mrec[1:] != mrec[:-1])[0]
            ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return ap


class DetectionAccumulator:
    """n_pos / score / match bookkeeping of Trainer.eval_detection_voc (train.py:653-693) for one IoU threshold."""

    def __init__(self):
        from collections import defaultdict
        self.n_pos, self.score, self.match = defaultdict(int), defaultdict(list), defaultdict(list)

    def update(self, gt_label, gt_mask, pred_label, pred_mask, pred_score, iou_thresh):
        for l in np.unique(np.concatenate((pred_label, gt_label)).astype(int)):
            keep = pred_label == l
            pm, ps = pred_mask[keep], pred_score[keep]
            order = ps.argsort()[::-1]
            pm, ps = pm[order], ps[order]
            gm = gt_mask[gt_label == l]
            self.n_pos[l] += int((gt_label == l).sum())
            self.score[l].extend(ps)
            if len(pm) == 0:
                continue
            if len(gm) == 0:
                self.match[l].extend((0,) * pm.shape[0])
                continue
            iou = mask_iou(pm, gm)
            gt_index = iou.argmax(axis=1)
            gt_index[iou.max(axis=1) < iou_thresh] = -1
            selec = np.zeros(gm.shape[0], dtype=bool)
            for g in gt_index:
                if g >= 0:
                    self.match[l].append(0 if selec[g] else 1)
                    selec[g] = True
                else:
                    self.match[l].append(0)

    def ap(self, use_07_metric=False):
        """metrics/voc_evaluation.py:11-140."""
        if not self.n_pos:
            return np.zeros(0)
        n_fg = max(self.n_pos.keys()) + 1
        prec, rec = [None] * n_fg, [None] * n_fg
        for l in self.n_pos.keys():
            score_l = np.array(self.score[l])
            match_l = np.array(self.match[l], dtype=np.int8)
            match_l = match_l[score_l.argsort()[::-1]]
            tp, fp = np.cumsum(match_l == 1), np.cumsum(match_l == 0)
            with np.errstate(all="ignore"):
                prec[l] = tp / (fp + tp)
            if self.n_pos[l] > 0:
                rec[l] = tp / self.n_pos[l]
        return calc_detection_voc_ap(prec, rec, use_07_metric=use_07_metric)
