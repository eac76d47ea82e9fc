"""Target assignment + sampling restated from reference utils/training.py (test oracle)."""
import numpy as np
import torch

from .boxes import encode, to_absolute
from .metrics import iou
from .philox import rand_u32


def _get_labels_masks(max_iou_per_region, max_iou, foreground_iou_interval, background_iou_interval):
    """reference utils/training.py:123-143 (masks returned un-tiled, shape [R])."""
    min_f, max_f = foreground_iou_interval
    min_b, max_b = background_iou_interval
    background_mask = (max_iou_per_region >= min_b) & (max_iou_per_region < max_b)
    # first region whose max IoU equals the global max -- always forced to foreground (:137-138)
    max_iou_indice = int(torch.nonzero(max_iou_per_region == max_iou)[0, 0])
    foreground_mask = (max_iou_per_region >= min_f) & (max_iou_per_region < max_f)
    foreground_mask = foreground_mask.clone()
    foreground_mask[max_iou_indice] = True
    return background_mask, foreground_mask


def generate_targets(gt_labels, gt_boxes, regions, image_shape, foreground_iou_interval, background_iou_interval):
    """reference utils/training.py:7-77, one image.

    gt_labels [G, C+1], gt_boxes [G, 4] relative, regions [R, 4] absolute.
    Returns target_labels [R, C+1], target_boxes_encoded [R, C, 4].
    """
    num_classes = gt_labels.shape[1] - 1
    num_regions = regions.shape[0]

    keep = gt_labels.sum(-1) != 0.0                       # :43-45 padding filter
    gt_boxes = gt_boxes[keep]
    gt_labels = gt_labels[keep]
    abs_gt_boxes = to_absolute(gt_boxes, image_shape)     # :48

    ious = iou(regions, abs_gt_boxes, pairwise=True)      # :51
    max_iou_indices = torch.argmax(ious, dim=-1)          # first max, as tf.argmax
    max_iou_per_region = ious.max(dim=-1).values
    max_iou = max_iou_per_region.max()

    bg_mask, fg_mask = _get_labels_masks(max_iou_per_region, max_iou, foreground_iou_interval, background_iou_interval)
    background_labels = torch.zeros(num_regions, num_classes + 1)
    background_labels[:, 0] = 1.0
    foreground_labels = gt_labels[max_iou_indices]

    target_labels = torch.zeros(num_regions, num_classes + 1)
    target_labels = torch.where(bg_mask[:, None], background_labels, target_labels)
    target_labels = torch.where(fg_mask[:, None], foreground_labels, target_labels)   # fg overrides bg (:64-65)

    foreground_boxes = abs_gt_boxes[max_iou_indices]
    foreground_boxes_encoded = encode(foreground_boxes, regions)                       # :69
    foreground_boxes_encoded = foreground_boxes_encoded[:, None, :].expand(-1, num_classes, -1)
    fb_mask = target_labels[:, 1:].bool()[:, :, None].expand(-1, -1, 4)
    target_boxes_encoded = torch.where(fb_mask, foreground_boxes_encoded, torch.zeros(num_regions, num_classes, 4))
    return target_labels, target_boxes_encoded


def _round_half_even(x):
    return int(np.round(x))   # numpy rounds half to even, as tf.math.round


def split_fg_bg(target_labels):
    """reference utils/training.py:97-103: ordered foreground / background index lists."""
    s = target_labels.sum(-1) != 0.0
    fg = torch.nonzero(s & (target_labels[:, 0] == 0.0)).reshape(-1)
    bg = torch.nonzero(s & (target_labels[:, 0] == 1.0)).reshape(-1)
    return fg, bg


def get_sample_indices(target_labels, num_samples, foreground_proportion, image=0, step=0, seed=0, stream_base=0):
    """reference utils/training.py:80-120, one image, with a reproducible RNG.

    fg: partial Fisher-Yates shuffle of the ordered foreground list, first n_fg taken
        (== tf.random.shuffle(...)[:n_fg]: without replacement, uniform);
    bg: n_bg uniform draws WITH replacement from the ordered background list.
    Random words: Philox4x32-10, counter (i, image, step, stream) with stream = stream_base + 0 (fg)
    or + 1 (bg); the model uses stream_base 0 for the RPN and 2 for the RCNN head.
    Raises ValueError on an empty background set (reference: tf.random.uniform maxval=0 error).
    """
    fg, bg = split_fg_bg(target_labels)
    fg = fg.numpy().copy()
    bg = bg.numpy()
    n_fg = min(len(fg), _round_half_even(num_samples * foreground_proportion))
    n_bg = num_samples - n_fg
    if n_bg > 0 and len(bg) == 0:
        raise ValueError("get_sample_indices: empty background set (reference utils/training.py:115 maxval=0)")
    out = np.zeros(num_samples, dtype=np.int64)
    if n_fg:
        r = rand_u32(np.arange(n_fg), image, step, stream_base + 0, seed)
        for i in range(n_fg):
            j = i + int(r[i]) % (len(fg) - i)
            fg[i], fg[j] = fg[j], fg[i]
            out[i] = fg[i]
    if n_bg:
        r = rand_u32(np.arange(n_bg), image, step, stream_base + 1, seed)
        out[n_fg:] = bg[(r % np.uint32(len(bg))).astype(np.int64)]
    return torch.from_numpy(out)
