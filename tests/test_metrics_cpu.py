"""CPU: host-side metrics (cl4wsis_amd/metrics.py) — StreamSegMetrics against the formulae of
metrics/stream_metrics.py:72-131 on a hand-made confusion matrix, and the VOC instance-AP accumulation
(train.py:653-693, metrics/voc_evaluation.py) on a hand-computed case."""
import numpy as np

from cl4wsis_amd import metrics as MT


def test_stream_seg_metrics_known_answer():
    m = MT.StreamSegMetrics(3)
    lt = np.array([[0, 0, 1, 1, 2, 255]])
    lp = np.array([[0, 1, 1, 1, 0, 2]])
    m.update(lt, lp)
    r = m.get_results()
    hist = np.array([[1, 1, 0], [0, 2, 0], [1, 0, 0]], dtype=float)
    assert np.array_equal(r["Confusion Matrix"], hist) and r["Total samples"] == 1
    assert abs(r["Overall Acc"] - 3 / 5) < 1e-12
    iu = np.array([1 / 3, 2 / 3, 0.0])
    assert abs(r["Mean IoU"] - iu.mean()) < 1e-5
    assert abs(r["Mean Acc"] - np.mean([0.5, 1.0, 0.0])) < 1e-5
    assert "Mean IoU" in m.to_str(r)
    m.reset()
    assert m.confusion_matrix.sum() == 0


def test_detection_accumulator_known_answer():
    H = W = 8
    def box(y0, y1, x0, x1):
        m = np.zeros((H, W), bool); m[y0:y1, x0:x1] = True
        return m
    gt_mask = np.stack([box(0, 4, 0, 4), box(4, 8, 4, 8)])
    gt_label = np.array([2, 2])
    pred_mask = np.stack([box(0, 4, 0, 4), box(0, 4, 0, 3), box(4, 8, 5, 8), box(0, 2, 6, 8)])   # exact, duplicate, 0.75 IoU, miss
    pred_label = np.array([2, 2, 2, 2])
    pred_score = np.array([0.9, 0.8, 0.7, 0.6])
    a = MT.DetectionAccumulator()
    a.update(gt_label, gt_mask, pred_label, pred_mask, pred_score, 0.5)
    assert a.n_pos[2] == 2 and a.match[2] == [1, 0, 1, 0]
    ap = a.ap()
    assert np.isnan(ap[0]) and np.isnan(ap[1]) and abs(ap[2] - (0.5 * 1.0 + 0.5 * 2 / 3)) < 1e-12
    b = MT.DetectionAccumulator()
    b.update(gt_label, gt_mask, pred_label, pred_mask, pred_score, 0.8)       # the 0.75-IoU prediction no longer matches
    assert b.match[2] == [1, 0, 0, 0] and abs(b.ap()[2] - 0.5) < 1e-12
    iou = MT.mask_iou(pred_mask, gt_mask)
    assert abs(iou[2, 1] - 12 / 16) < 1e-12 and iou[3].max() == 0
