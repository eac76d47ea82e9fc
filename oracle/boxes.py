"""Box arithmetic restated from reference utils/boxes.py (test oracle, fp32 torch-CPU).

Box layout everywhere: [x_min, y_min, x_max, y_max].
"""
import torch


def clip_to_window(boxes, window):
    """reference utils/boxes.py:4-17.  `window` is read as [x_min, y_min, x_max, y_max]
    (the reference docstring says y-first, the code is x-first; the code wins)."""
    wx0, wy0, wx1, wy1 = [float(v) for v in window]
    hi = torch.tensor([wx1, wy1, wx1, wy1], dtype=torch.float32)
    lo = torch.tensor([wx0, wy0, wx0, wy0], dtype=torch.float32)
    return torch.maximum(torch.minimum(boxes, hi), lo)


def decode(boxes, reference_boxes):
    """reference utils/boxes.py:20-41."""
    mins_ref, maxs_ref = reference_boxes[..., :2], reference_boxes[..., 2:]
    centers_ref = (maxs_ref + mins_ref) / 2.0
    sizes_ref = maxs_ref - mins_ref
    centers, sizes = boxes[..., :2], boxes[..., 2:]
    centers = centers * sizes_ref + centers_ref
    sizes = torch.exp(sizes) * sizes_ref
    return torch.cat([centers - 0.5 * sizes, centers + 0.5 * sizes], dim=-1)


def encode(boxes, reference_boxes):
    """reference utils/boxes.py:44-73 (division by a zero-size reference is NOT guarded,
    as in the reference)."""
    mins_ref, maxs_ref = reference_boxes[..., :2], reference_boxes[..., 2:]
    centers_ref = (maxs_ref + mins_ref) / 2.0
    sizes_ref = maxs_ref - mins_ref
    mins, maxs = boxes[..., :2], boxes[..., 2:]
    centers = (maxs + mins) / 2.0
    sizes = maxs - mins
    centers = (centers - centers_ref) / sizes_ref
    sizes = torch.log(sizes / sizes_ref)
    return torch.cat([centers, sizes], dim=-1)


def _whwh(image_shape):
    h, w = image_shape[0], image_shape[1]
    return torch.tensor([w, h, w, h], dtype=torch.float32)


def to_absolute(boxes, image_shape):
    """reference utils/boxes.py:76-83."""
    return boxes * _whwh(image_shape)


def to_relative(boxes, image_shape):
    """reference utils/boxes.py:86-93."""
    return boxes / _whwh(image_shape)
