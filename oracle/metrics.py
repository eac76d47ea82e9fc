"""IoU restated from reference utils/metrics.py:136-208 (test oracle, fp32 torch-CPU).

The op order is kept literally (area = (x1-x0)*(y1-y0); union = (a1 + a2) - inter;
iou = inter / union; inter == 0 -> 0) so an fp32 implementation without FMA contraction
is bit-identical.
"""
import torch


def area(boxes):
    """reference utils/metrics.py:136-147."""
    return (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])


def intersection(boxes_1, boxes_2, pairwise=False):
    """reference utils/metrics.py:150-180."""
    if pairwise:
        b1 = boxes_1[:, None, :]
        b2 = boxes_2[None, :, :]
    else:
        b1, b2 = boxes_1, boxes_2
    dw = torch.minimum(b1[..., 2], b2[..., 2]) - torch.maximum(b1[..., 0], b2[..., 0])
    dh = torch.minimum(b1[..., 3], b2[..., 3]) - torch.maximum(b1[..., 1], b2[..., 1])
    zero = torch.zeros((), dtype=boxes_1.dtype)
    return torch.maximum(zero, dw) * torch.maximum(zero, dh)


def iou(boxes_1, boxes_2, pairwise=False):
    """reference utils/metrics.py:183-208."""
    inter = intersection(boxes_1, boxes_2, pairwise)
    a1, a2 = area(boxes_1), area(boxes_2)
    if pairwise:
        a1, a2 = a1[:, None], a2[None, :]
    unions = a1 + a2 - inter
    return torch.where(inter == 0.0, torch.zeros_like(inter), inter / unions)
