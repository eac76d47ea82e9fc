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


def to_relative(boxes, image_shape):
    """reference utils/boxes.py:86-93."""
    boxes = _f32(boxes)
    out = torch.empty_like(boxes)
    ops.boxes_divide(boxes, out, float(image_shape[1]), float(image_shape[0]))
    return out


def _as_brc(boxes, reference_boxes):
    """boxes [..., 4] and a reference broadcastable with it -> ([B,R,C,4] boxes, [R,4] or [B,R,4] references).  Supported
    broadcasts (those the reference uses, post_processing.py:39-44, training.py:69): equal shapes; reference [R,4] against
    boxes [B,R,4] / [B,R,C,4]; reference [B,R,1,4] or [B,R,4] against boxes [B,R,C,4]."""
    boxes, ref = _f32(boxes), _f32(reference_boxes)
    if boxes.dim() == 2:                                    # [R,4] x [R,4]
        if ref.shape != boxes.shape:
            raise ValueError("reference_boxes %s does not match boxes %s" % (tuple(ref.shape), tuple(boxes.shape)))
        return boxes.view(1, boxes.shape[0], 1, 4), ref
    if boxes.dim() == 3:                                    # [B,R,4]
        b, r, _ = boxes.shape
        if ref.dim() == 2 and ref.shape[0] == r:
            return boxes.view(b, r, 1, 4), ref
        if ref.shape == boxes.shape:
            return boxes.view(b, r, 1, 4), ref
    if boxes.dim() == 4:                                    # [B,R,C,4]
        b, r, c, _ = boxes.shape
        if ref.dim() == 2 and ref.shape[0] == r:
            return boxes, ref
        if ref.dim() == 3 and ref.shape[:2] == (b, r):
            return boxes, ref
        if ref.dim() == 4 and ref.shape[:2] == (b, r) and ref.shape[2] == 1:
            return boxes, ref.view(b, r, 4)
        if ref.dim() == 4 and ref.shape[0] == 1 and ref.shape[1] == r and ref.shape[2] == 1:
            return boxes, ref.view(r, 4)
    raise ValueError("unsupported broadcast: boxes %s, reference_boxes %s" % (tuple(boxes.shape), tuple(reference_boxes.shape)))


def decode(boxes, reference_boxes):
    """reference utils/boxes.py:20-41: reverse of encode; boxes [..., num_boxes, 4] are [tx, ty, tw, th]."""
    b4, ref = _as_brc(boxes, reference_boxes)
    out = torch.empty_like(b4)
    ops.decode_boxes(ref, b4, out, b4.shape[0], b4.shape[1], b4.shape[2], 1.0, 1.0)     # (x / 1.0f is exact)
    return out.view(boxes.shape)


def encode(boxes, reference_boxes):
    """reference utils/boxes.py:44-73."""
    b4, ref = _as_brc(boxes, reference_boxes)
    out = torch.empty_like(b4)
    ops.encode_boxes(b4, ref, out, b4.shape[0], b4.shape[1], b4.shape[2])
    return out.view(boxes.shape)
