"""Losses restated from reference utils/losses.py (test oracle, differentiable torch-CPU).

[TF-ext] Keras semantics per SURVEY.md A.7 (parity unpinned against TensorFlow itself).
"""
import torch

_EPS = 1e-7


def classification_loss(target_class_labels, pred_class_scores):
    """reference utils/losses.py:10-18 -> tf.keras.losses.CategoricalCrossentropy() on
    probabilities: p <- p / sum(p); clip to [1e-7, 1-1e-7]; -sum_c t_c log p_c; mean over rows."""
    p = pred_class_scores / pred_class_scores.sum(-1, keepdim=True)
    p = torch.clamp(p, _EPS, 1.0 - _EPS)
    per_row = -(target_class_labels * torch.log(p)).sum(-1)
    return per_row.mean()


def regression_loss(target_boxes_encoded, pred_boxes_encoded):
    """reference utils/losses.py:27-43: rows (b,s,c) with sum(target)!=0; Keras Huber(delta=1,
    reduction=NONE) = mean over the 4 coords; then SUM over rows (not a mean)."""
    keep = target_boxes_encoded.sum(-1) != 0.0
    t = target_boxes_encoded[keep]
    p = pred_boxes_encoded[keep]
    err = p - t
    abs_err = err.abs()
    hub = torch.where(abs_err <= 1.0, 0.5 * err * err, abs_err - 0.5)
    return hub.mean(-1).sum()
