"""postprocess_output with the signature of reference utils/post_processing.py:6-63.

decode -> normalise -> drop background column -> batched per-class NMS -> top-K merge -> clip ->
zero-pad, all on the GPU with no host synchronisation (the reference's
tf.image.combined_non_max_suppression is a CPU kernel and a sync point)."""
import torch

from .. import ops


class NmsBuffers:
    """Pre-allocated outputs + workspace for one (B, N, C) NMS geometry (graph-capture friendly)."""

    def __init__(self, b, n, c, max_output_size_per_class, max_total_size, device):
        t = int(max_total_size)
        self.boxes = torch.zeros(b, t, 4, device=device)
        self.scores = torch.zeros(b, t, device=device)
        self.classes = torch.zeros(b, t, dtype=torch.int32, device=device)
        self.valid = torch.zeros(b, dtype=torch.int32, device=device)
        self.decoded = torch.empty(b, n, c, 4, device=device)
        self.workspace = torch.zeros(ops.nms_workspace_bytes(b, n, c, int(max_output_size_per_class), t), dtype=torch.uint8,
                                     device=device)


def postprocess_plan(plan, image_shape, regions, pred_scores, pred_boxes, score_threshold, iou_threshold,
                     max_output_size_per_class, max_total_size, buffers=None, decoded_done=False, abs_boxes=None):
    """Append the launches to `plan`; returns the output dict (static buffers).  decoded_done: `buffers.decoded` already holds
    the decoded boxes (the head-post kernels write them: ops.rpn_head_post_decode, ops.rcnn_head_post_decode).
    abs_boxes [B,T,4]: also receives the kept boxes in absolute image coordinates (to_absolute, reference
    fast_rcnn_detector.py:67) from the NMS launch itself."""
    b, n, c, _ = pred_boxes.shape
    c1 = pred_scores.shape[-1]
    buf = buffers or NmsBuffers(b, n, c, max_output_size_per_class, max_total_size, pred_boxes.device)
    plan.hold(buf)
    if not decoded_done:
        plan.add(ops.decode_boxes, regions, pred_boxes, buf.decoded, b, n, c, image_shape[1], image_shape[0])
    # pred_scores[..., 1:] is expressed as (row stride C+1, column offset 1): no slice copy
    if abs_boxes is not None:
        plan.add(ops.nms_combined_abs, buf.decoded, pred_scores, b, n, c, c1 - 1, c1, 1, int(max_output_size_per_class), int(max_total_size),
                 float(iou_threshold), float(score_threshold), buf.boxes, buf.scores, buf.classes, buf.valid, buf.workspace, abs_boxes,
                 float(image_shape[1]), float(image_shape[0]))
    else:
        plan.add(ops.nms_combined, buf.decoded, pred_scores, b, n, c, c1 - 1, c1, 1, int(max_output_size_per_class), int(max_total_size),
                 float(iou_threshold), float(score_threshold), buf.boxes, buf.scores, buf.classes, buf.valid, buf.workspace)
    return {"pred_boxes": buf.boxes, "pred_scores": buf.scores, "pred_classes": buf.classes, "num_valid_detections": buf.valid}


def postprocess_output(image_shape, regions, pred_scores, pred_boxes, score_threshold, iou_threshold,
                       max_output_size_per_class, max_total_size):
    """reference utils/post_processing.py:6-63 (same argument names: the reference splats dicts
    into it).  Tensors: fp32 CUDA.  Returns pred_boxes [B,T,4] (relative, clipped, zero padded),
    pred_scores [B,T], pred_classes int32 [B,T], num_valid_detections int32 [B]; no gradient."""
    from ..runtime import Plan
    for t in (regions, pred_scores, pred_boxes):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise TypeError("postprocess_output expects float32 CUDA tensors (no CPU fallback)")
    if pred_boxes.dim() != 4 or pred_scores.dim() != 3 or pred_scores.shape[-1] != pred_boxes.shape[2] + 1 and pred_boxes.shape[2] != 1:
        raise ValueError("postprocess_output: pred_scores [B,N,C+1] / pred_boxes [B,N,C|1,4] expected")
    plan = Plan("postprocess")
    out = postprocess_plan(plan, image_shape, regions.contiguous(), pred_scores.contiguous(), pred_boxes.contiguous(), score_threshold,
                           iou_threshold, max_output_size_per_class, max_total_size)
    plan.run()
    return out
