"""Losses with the call surface of reference utils/losses.py (forward values; the training path
uses the fused loss+gradient kernel through models/faster_rcnn.py)."""
import torch

from .. import ops


def _losses(target_labels, pred_scores, target_boxes, pred_boxes):
    b, s, c1 = pred_scores.shape
    dev = pred_scores.device
    idx = torch.arange(s, dtype=torch.int32, device=dev).repeat(b, 1).contiguous()
    out = torch.empty(2, device=dev)
    ops.losses(pred_scores.contiguous(), pred_boxes.contiguous(), target_labels.contiguous(), target_boxes.contiguous(), idx, b, s, c1, s,
               1.0, 1.0, out)
    return out


class ClassificationLoss:
    """reference utils/losses.py:4-18 (Keras CategoricalCrossentropy on probabilities)."""

    def __call__(self, target_class_labels, pred_class_scores):
        c = pred_class_scores.shape[-1] - 1
        z = torch.zeros(*pred_class_scores.shape[:2], c, 4, device=pred_class_scores.device)
        return _losses(target_class_labels, pred_class_scores, z, z)[0]


class RegressionLoss:
    """reference utils/losses.py:21-43 (Huber on rows with non-zero target, summed)."""

    def __call__(self, target_boxes_encoded, pred_boxes_encoded):
        b, s, c, _ = pred_boxes_encoded.shape
        p = torch.full((b, s, c + 1), 1.0 / (c + 1), device=pred_boxes_encoded.device)
        return _losses(p, p, target_boxes_encoded, pred_boxes_encoded)[1]
