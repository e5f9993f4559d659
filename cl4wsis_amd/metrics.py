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
