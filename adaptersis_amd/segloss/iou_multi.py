"""HIP-backed mirror of `segloss/iou_multi.py`: the multi-class soft-IoU loss (`:9-49`) runs in the fused loss kernels;
`ch_iou` / `isi_iou` (`:51-88`) keep the reference's numpy-in / float-out signature for host arrays and gain a
``*_from_counts`` form fed by the validation kernel's per-class pixel counts (``ops.ce_acc(..., counts=True)``), so
validation needs no B*H*W device->host copy."""
from __future__ import annotations

import numpy as np

from .. import ops
from .dice import seg_loss


def iou(y_true, y_pred):
    intersection = (y_true * y_pred).sum()
    union = y_true.sum() + y_pred.sum() - intersection
    return (intersection + 1e-6) / (union + 1e-6)


def iou_loss(preds, labels, smooth=1e-6, num_classes=8):
    """preds (B,C,H,W) (softmax is applied inside, as in the reference), labels (B,H,W) -> mean soft-IoU loss."""
    if preds.shape[1] != num_classes:
        raise ValueError(f"iou_loss: preds has {preds.shape[1]} channels, num_classes={num_classes}")
    return seg_loss(preds, labels, 1, ops.LOSS_IOU, float(smooth))


def _counts_of(y_true, y_pred, n):
    yt, yp = np.asarray(y_true).reshape(-1), np.asarray(y_pred).reshape(-1)
    c = np.zeros((n, 3), dtype=np.int64)
    for k in range(n):
        t, p = yt == k, yp == k
        c[k] = (t.sum(), p.sum(), (t & p).sum())
    return c


def _iou_c(row):
    inter = float(row[2])
    return (inter + 1e-6) / (float(row[0]) + float(row[1]) - inter + 1e-6)


def ch_iou_from_counts(counts) -> float:
    """counts [C,3] = #(true == c), #(pred == c), #(both): mean IoU over the non-background classes present in y_true."""
    c = np.asarray(counts.cpu() if hasattr(counts, "cpu") else counts)
    if c[1:, 0].sum() == 0:
        return 1 if c[1:, 1].sum() == 0 else 0
    res = [_iou_c(c[k]) for k in range(1, c.shape[0]) if c[k, 0] > 0]
    return float(np.mean(res))


def isi_iou_from_counts(counts, problem_type="instruments") -> float:
    type_number = {"binary": 2, "parts": 4, "instruments": 8}[problem_type]
    c = np.asarray(counts.cpu() if hasattr(counts, "cpu") else counts)
    if c[1:, 0].sum() == 0:
        return 1 if c[1:, 1].sum() == 0 else 0
    res = [_iou_c(c[k]) for k in range(1, min(type_number, c.shape[0])) if c[k, 0] != 0 or c[k, 1] != 0]
    return float(np.mean(res))


def ch_iou(y_true, y_pred):
    n = int(max(np.max(y_true), np.max(y_pred))) + 1
    return ch_iou_from_counts(_counts_of(y_true, y_pred, n))


def isi_iou(y_true, y_pred, problem_type="instruments"):
    n = int(max(np.max(y_true), np.max(y_pred), 7)) + 1
    return isi_iou_from_counts(_counts_of(y_true, y_pred, n), problem_type)
