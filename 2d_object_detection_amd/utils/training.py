"""Target assignment and sampling with the call surface of reference utils/training.py, batched
on the GPU (the reference maps these per image with tf.map_fn)."""
import torch

from .. import ops


def generate_targets(gt_labels, gt_boxes, regions, image_shape, foreground_iou_interval, background_iou_interval, objectness=False):
    """reference utils/training.py:7-77, per image as there (gt_labels [G,C+1], gt_boxes [G,4] relative, regions [R,4]
    absolute -> [R,C1], [R,C1-1,4]) or batched (the reference maps it per image with tf.map_fn, rpn_detector.py:143):
    gt_labels [B,G,C+1], gt_boxes [B,G,4], regions [R,4] or [B,R,4].  objectness=True applies rpn_detector.py:141 first.
    Returns target_labels [B,R,C1], target_boxes [B,R,C1-1,4]."""
    if gt_labels.dim() == 2:                       # the reference's own per-image form (utils/training.py:7: [G,C+1], [G,4], [R,4])
        tl, tb = generate_targets(gt_labels[None], gt_boxes[None], regions, image_shape, foreground_iou_interval, background_iou_interval,
                                  objectness)
        return tl[0], tb[0]
    b, g, c1g = gt_labels.shape
    r = regions.shape[-2]
    c1 = 2 if objectness else c1g
    dev = gt_labels.device
    tl = torch.empty(b, r, c1, device=dev)
    tb = torch.empty(b, r, c1 - 1, 4, device=dev)
    ops.assign_targets(regions.contiguous(), gt_labels.contiguous(), gt_boxes.contiguous(), b, r, g, c1g, objectness, image_shape[1],
                       image_shape[0], foreground_iou_interval, background_iou_interval, tl, tb)
    return tl, tb


def get_sample_indices(target_labels, num_samples, foreground_proportion, seed=0, step=None, stream_base=0):
    """Batched reference utils/training.py:80-120 with a counter-based RNG (Philox4x32-10).
    target_labels [B,R,C1] -> int32 [B,num_samples]; raises on an empty background set like the
    reference (this standalone form synchronises to read the status word)."""
    b, r, c1 = target_labels.shape
    dev = target_labels.device
    idx = torch.empty(b, num_samples, dtype=torch.int32, device=dev)
    ws = torch.empty(b, 2 * r, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    if step is None:
        step = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.sample_indices(target_labels.contiguous(), b, r, c1, num_samples, foreground_proportion, seed, step, stream_base, idx, ws, status)
    if int(status.item()) & 1:
        raise ValueError("get_sample_indices: empty background set (reference utils/training.py:115 fails with maxval=0)")
    return idx
