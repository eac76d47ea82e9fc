"""Detection metrics with the call surface and semantics of the reference's utils/metrics.py: `area`, `intersection`,
`iou` (:136-208) and the streaming `AveragePrecision` / `MeanAveragePrecision` metrics (:4-133) used by the training
driver (train_faster_rcnn.py:84-98,137-143).

State lives on the device the predictions arrive on and `update_state` never synchronises with the host, so a metric
update per training step (as the reference driver does) does not stall the step's graph replay; `result()` is the only
call that reads back.  Reference quirks are kept (and tested against the loop oracle in oracle/metrics.py): every
prediction slot, zero padding included, counts as a positive; a prediction is a true positive when it is the arg-max
prediction of some ground-truth box with IoU above the threshold; 11-point interpolation with the (0, 2) sentinel."""
import torch


def area(boxes):
    """utils/metrics.py:136-147."""
    return (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])


def intersection(boxes_1, boxes_2, pairwise=False):
    """utils/metrics.py:150-180."""
    if pairwise:
        boxes_1, boxes_2 = boxes_1[..., :, None, :], boxes_2[..., None, :, :]
    dw = torch.minimum(boxes_1[..., 2], boxes_2[..., 2]) - torch.maximum(boxes_1[..., 0], boxes_2[..., 0])
    dh = torch.minimum(boxes_1[..., 3], boxes_2[..., 3]) - torch.maximum(boxes_1[..., 1], boxes_2[..., 1])
    return dw.clamp_min(0.0) * dh.clamp_min(0.0)


def iou(boxes_1, boxes_2, pairwise=False):
    """utils/metrics.py:183-208 (intersection == 0 -> 0, also for degenerate boxes)."""
    inter = intersection(boxes_1, boxes_2, pairwise)
    a1, a2 = area(boxes_1), area(boxes_2)
    if pairwise:
        a1, a2 = a1[..., :, None], a2[..., None, :]
    unions = a1 + a2 - inter
    return torch.where(inter == 0.0, torch.zeros_like(inter), inter / unions)


class AveragePrecision:
    """utils/metrics.py:4-83."""

    def __init__(self, iou_threshold, num_points=11, name=None):
        self.iou_threshold, self.num_points, self.name = iou_threshold, num_points, name
        self.reset_states()

    def reset_states(self):
        self._true_count, self._pos_count, self._true_pos, self._scores = [], 0, [], []

    def update_state(self, gt_boxes, pred_boxes, pred_scores):
        """gt_boxes [B,G,4] zero padded, pred_boxes [B,P,4] zero padded, pred_scores [B,P] (batched form of :56-83)."""
        gt_boxes, pred_boxes, pred_scores = gt_boxes.float(), pred_boxes.float(), pred_scores.float()
        b, p = pred_scores.shape
        real = gt_boxes.sum(-1) != 0.0                                     # :68 padding rows removed
        ious = iou(gt_boxes, pred_boxes, pairwise=True)                    # [B,G,P]
        best = ious.argmax(dim=2, keepdim=True)                            # :80 first maximum
        hit = (ious.gather(2, best) > self.iou_threshold) & real[..., None]   # [B,G,1]
        tp = torch.zeros(b, p, dtype=torch.int32, device=pred_scores.device)
        tp.scatter_reduce_(1, best[..., 0], hit[..., 0].to(torch.int32), reduce="amax")
        self._true_count.append(real.sum())
        self._pos_count += b * p                                           # :73,76 padding predictions count too
        self._true_pos.append(tp.reshape(-1))
        # clone: float32 contiguous inputs pass through .float() / .reshape() as VIEWS, and the training driver hands over the
        # train step's static prediction buffers, which the next step overwrites
        self._scores.append(pred_scores.reshape(-1).clone())

    def result(self):
        """:24-41."""
        if not self._scores:
            return 0.0
        scores, tp = torch.cat(self._scores), torch.cat(self._true_pos)
        true_count = torch.stack(self._true_count).sum().to(torch.float32)
        order = torch.sort(scores, descending=True, stable=True).indices
        ctp = tp[order].cumsum(0).to(torch.float32)
        precisions = ctp / torch.arange(1, self._pos_count + 1, dtype=torch.float32, device=ctp.device)
        recalls = ctp / true_count
        precisions = torch.cat([precisions, precisions.new_zeros(1)])
        recalls = torch.cat([recalls, recalls.new_full((1,), 2.0)])
        r = torch.arange(self.num_points, dtype=torch.float32, device=ctp.device) / float(self.num_points - 1)
        sel = recalls[None, :] >= r[:, None]                               # NaN recalls (no ground truth) never qualify
        interp = torch.where(sel, precisions[None, :], precisions.new_full((), -1.0)).max(dim=1).values
        return float(interp.mean())


class MeanAveragePrecision:
    """utils/metrics.py:86-133."""

    def __init__(self, num_classes, iou_threshold, num_points=11, name=None):
        self.num_classes, self.name = num_classes, name
        self.average_precisions = [AveragePrecision(iou_threshold, num_points) for _ in range(num_classes)]

    def reset_states(self):
        for ap in self.average_precisions:
            ap.reset_states()

    def update_state(self, gt_boxes, gt_class_labels, pred_boxes, pred_scores, pred_classes):
        """:107-130: class i's metric sees the ground-truth boxes labelled i and the predictions of class i, everything else zeroed.
        A zeroed box has no intersection with anything, so class i's IoU matrix is the full matrix with the other classes' rows and
        columns set to zero: ONE pass over [C,B,G,P] updates all the classes (the per-class form is 7 x 25 small launches per training
        step -- 2.1 ms of host dispatch beside a 3.7 ms step, tools/driver_rate.py; equality with it:
        tests/test_data_metrics.py::test_mean_average_precision_update_equals_the_per_class_form)."""
        aps = self.average_precisions
        gt_boxes, pred_boxes, pred_scores = gt_boxes.float(), pred_boxes.float(), pred_scores.float()
        b, p = pred_scores.shape
        cls = torch.arange(self.num_classes, device=pred_scores.device)
        of_class = (gt_class_labels[..., 1:] == 1.0).permute(2, 0, 1)              # [C,B,G]
        mine = pred_classes[None] == cls.view(-1, 1, 1).to(pred_classes.dtype)     # [C,B,P]
        real = of_class & (gt_boxes.sum(-1) != 0.0)[None]                          # :68 on the zeroed boxes
        ious = iou(gt_boxes, pred_boxes, pairwise=True)[None] * (of_class[..., :, None] & mine[..., None, :])   # [C,B,G,P]
        best = ious.argmax(dim=3, keepdim=True)
        hit = (ious.gather(3, best) > aps[0].iou_threshold) & real[..., None]
        tp = torch.zeros(self.num_classes, b, p, dtype=torch.int32, device=pred_scores.device)
        tp.scatter_reduce_(2, best[..., 0], hit[..., 0].to(torch.int32), reduce="amax")
        true_count = real.sum(dim=(1, 2))
        tp = tp.reshape(self.num_classes, -1)
        scores = torch.where(mine, pred_scores[None], pred_scores.new_zeros(())).reshape(self.num_classes, -1)
        for i, ap in enumerate(aps):
            ap._true_count.append(true_count[i])
            ap._pos_count += b * p
            ap._true_pos.append(tp[i])
            ap._scores.append(scores[i])

    def result(self):
        return sum(ap.result() for ap in self.average_precisions) / float(self.num_classes)
