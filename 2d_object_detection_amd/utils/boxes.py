"""Box helpers with the call surface of reference utils/boxes.py, on HIP kernels
(fp32 CUDA tensors, [x_min, y_min, x_max, y_max])."""
import torch

from .. import ops


def _f32(t):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError("expected a float32 CUDA tensor (the HIP path has no CPU fallback)")
    return t.contiguous()


def clip_to_window(boxes, window):
    """reference utils/boxes.py:4-17 (window = [x_min, y_min, x_max, y_max], as the code reads it)."""
    boxes = _f32(boxes)
    out = torch.empty_like(boxes)
    ops.clip_to_window(boxes, out, window)
    return out


def to_absolute(boxes, image_shape):
    """reference utils/boxes.py:76-83."""
    boxes = _f32(boxes)
    out = torch.empty_like(boxes)
    ops.boxes_scale(boxes, out, float(image_shape[1]), float(image_shape[0]))
    return out


def decode_relative(pred_boxes, regions, image_shape):
    """decode (utils/boxes.py:20-41) of pred_boxes [B,R,C,4] against regions [R,4] or [B,R,4],
    followed by to_relative (utils/boxes.py:86-93), fused in one kernel."""
    pred_boxes, regions = _f32(pred_boxes), _f32(regions)
    b, r, c, _ = pred_boxes.shape
    out = torch.empty_like(pred_boxes)
    ops.decode_boxes(regions, pred_boxes, out, b, r, c, image_shape[1], image_shape[0])
    return out
