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


# ---------------------------------------------------------------------------------------------------------------------
# AP / mAP restated from reference utils/metrics.py:4-133 in plain loops (numpy float32), for tests only.
# Quirks kept: every prediction slot -- zero padding included -- counts as a positive (metrics.py:73,76); a prediction is
# a true positive when it is the arg-max prediction of at least one ground-truth box with IoU > threshold (:79-83);
# 11-point interpolation with the (precision 0, recall 2) sentinel (:33-39).
class AveragePrecisionOracle:
    def __init__(self, iou_threshold, num_points=11):
        self.iou_threshold, self.num_points = iou_threshold, num_points
        self.reset_states()

    def reset_states(self):
        self.true_count, self.pos_count, self.true_pos, self.scores = 0, 0, [], []

    def update_state(self, gt_boxes, pred_boxes, pred_scores):
        import numpy as np
        gt_boxes, pred_boxes, pred_scores = (np.asarray(a, dtype=np.float32) for a in (gt_boxes, pred_boxes, pred_scores))
        for b in range(pred_boxes.shape[0]):
            gts = [g for g in gt_boxes[b] if np.float32(g.sum(dtype=np.float32)) != 0.0]
            self.true_count += len(gts)
            self.pos_count += pred_boxes.shape[1]
            tp = [0] * pred_boxes.shape[1]
            for g in gts:
                best, best_iou = 0, np.float32(-1.0)
                for j, pb in enumerate(pred_boxes[b]):
                    v = float(iou(torch.from_numpy(g[None]), torch.from_numpy(pb[None]))[0])
                    if v > best_iou:                    # first maximum wins (tf.math.argmax)
                        best, best_iou = j, v
                if best_iou > self.iou_threshold:
                    tp[best] = 1
            self.true_pos += tp
            self.scores += [float(s) for s in pred_scores[b]]

    def result(self):
        import numpy as np
        order = sorted(range(len(self.scores)), key=lambda i: (-self.scores[i], i))
        ctp, precisions, recalls = 0, [], []
        for rank, i in enumerate(order):
            ctp += self.true_pos[i]
            precisions.append(np.float32(ctp) / np.float32(rank + 1))
            with np.errstate(divide="ignore", invalid="ignore"):
                recalls.append(np.float32(ctp) / np.float32(self.true_count))
        precisions.append(np.float32(0.0))
        recalls.append(np.float32(2.0))
        total = np.float32(0.0)
        for k in range(self.num_points):
            r = np.float32(k) / np.float32(self.num_points - 1)
            total += max(p for p, rc in zip(precisions, recalls) if rc >= r)
        return float(total / np.float32(self.num_points))


class MeanAveragePrecisionOracle:
    """reference utils/metrics.py:86-133."""

    def __init__(self, num_classes, iou_threshold, num_points=11):
        self.num_classes = num_classes
        self.aps = [AveragePrecisionOracle(iou_threshold, num_points) for _ in range(num_classes)]

    def reset_states(self):
        for ap in self.aps:
            ap.reset_states()

    def update_state(self, gt_boxes, gt_class_labels, pred_boxes, pred_scores, pred_classes):
        import numpy as np
        gt_boxes, gt_class_labels, pred_boxes, pred_scores, pred_classes = (np.asarray(a) for a in (gt_boxes, gt_class_labels, pred_boxes, pred_scores, pred_classes))
        for i, ap in enumerate(self.aps):
            g = np.where((gt_class_labels[:, :, i + 1] == 1.0)[..., None], gt_boxes, 0.0)
            pb = np.where((pred_classes == i)[..., None], pred_boxes, 0.0)
            ps = np.where(pred_classes == i, pred_scores, 0.0)
            ap.update_state(g, pb, ps)

    def result(self):
        return sum(ap.result() for ap in self.aps) / float(self.num_classes)
