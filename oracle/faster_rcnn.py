"""Faster-RCNN forward / train step / test step restated from reference models/faster_rcnn.py,
models/detectors/rpn_detector.py, models/detectors/fast_rcnn_detector.py and
utils/post_processing.py (test oracle; torch-CPU fp32 + autograd).

Parameters are a flat dict name -> tensor with Keras variable names and Keras layouts.
"""
import math

import torch
import torch.nn.functional as F

from . import resnet
from .boxes import clip_to_window, decode, to_absolute, to_relative
from .losses import classification_loss, regression_loss
from .nms import combined_nms
from .roi import roi_pooling
from .training import generate_targets, get_sample_indices

SGD_MOMENTUM = 0.9


# --------------------------------------------------------------------------- config
def default_config(image_shape=(375, 1242, 3)):
    """Same schema and values as the reference's config.json (image_shape defaults to the
    BASELINE.json KITTI size instead of the reference's 600x1987)."""
    return {
        "num_classes": 7,
        "image_shape": list(image_shape),
        "rpn": {
            "window_size": 3,
            "weight_decay": 0.0005,
            "anchors": {"scales": [0.25, 0.5, 1.0, 2.0], "aspect_ratios": [0.5, 1.0, 2.0], "base_anchor_shape": [256, 256]},
            "sampling": {"foreground_iou_interval": [0.7, 1.0], "background_iou_interval": [0.0, 0.3],
                         "num_samples": 256, "foreground_proportion": 0.5},
            "nms": {"score_threshold": 0.0, "iou_threshold": 0.7, "max_output_size_per_class": 300, "max_total_size": 300},
        },
        "rcnn": {
            "weight_decay": 0.0005,
            "roi_pooling": {"pooled_size": 7, "kernel_size": 2},
            "sampling": {"foreground_iou_interval": [0.5, 1.0], "background_iou_interval": [0.0, 0.5],
                         "num_samples": 64, "foreground_proportion": 0.25},
            "nms": {"score_threshold": 0.0, "iou_threshold": 0.6, "max_output_size_per_class": 100, "max_total_size": 300},
        },
    }


def feature_grid(image_shape):
    """Output grid of the truncated ResNet."""
    h, w = image_shape[0], image_shape[1]
    f = lambda n: ((n + 6 - 7) // 2 + 1)           # pad 3, 7x7/2 valid
    h, w = f(h), f(w)
    f = lambda n: ((n + 2 - 3) // 2 + 1)           # pad 1, 3x3/2 valid
    h, w = f(h), f(w)
    f = lambda n: (n - 1) // 2 + 1                 # 1x1 stride 2 valid
    return f(f(h)), f(f(w))


# --------------------------------------------------------------------------- anchors
def generate_anchors(grid_shape, scales, aspect_ratios, base_anchor_shape, stride_shape=(16, 16)):
    """reference rpn_detector.py:162-199 in fp32: row (y*gw+x)*A + k, k = ratio-major."""
    f32 = torch.float32
    scales_t = torch.tensor(scales, dtype=f32)
    ratios_t = torch.tensor(aspect_ratios, dtype=f32)
    s = scales_t[None, :].expand(len(aspect_ratios), -1).reshape(-1)      # tf.meshgrid(scales, ratios)
    r = ratios_t[:, None].expand(-1, len(scales)).reshape(-1)
    rs = torch.sqrt(r)
    heights = s / rs * float(base_anchor_shape[0])
    widths = s * rs * float(base_anchor_shape[1])
    xc = torch.arange(grid_shape[1], dtype=f32) * float(stride_shape[1])
    yc = torch.arange(grid_shape[0], dtype=f32) * float(stride_shape[0])
    xc = xc[None, :].expand(grid_shape[0], -1).reshape(-1)
    yc = yc[:, None].expand(-1, grid_shape[1]).reshape(-1)
    centers = torch.stack([xc[:, None].expand(-1, len(s)), yc[:, None].expand(-1, len(s))], dim=2).reshape(-1, 2)
    sizes = torch.stack([widths[None, :].expand(len(xc), -1), heights[None, :].expand(len(xc), -1)], dim=2).reshape(-1, 2)
    return torch.cat([centers - 0.5 * sizes, centers + 0.5 * sizes], dim=1)


def inside_indices(anchors, image_shape):
    """reference rpn_detector.py:216-225 (inclusive test)."""
    h, w = image_shape[0], image_shape[1]
    m = (anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) & (anchors[:, 2] <= w) & (anchors[:, 3] <= h)
    return torch.nonzero(m).reshape(-1)


# --------------------------------------------------------------------------- parameters
def head_param_shapes(config, feat_channels=1024):
    na = len(config["rpn"]["anchors"]["scales"]) * len(config["rpn"]["anchors"]["aspect_ratios"])
    ws = config["rpn"]["window_size"]
    ps = config["rcnn"]["roi_pooling"]["pooled_size"]
    nc = config["num_classes"]
    flat = ps * ps * feat_channels
    return {
        "rpn_intermediate_layer/kernel": (ws, ws, feat_channels, 256),
        "rpn_intermediate_layer/bias": (256,),
        "rpn_classification_head/kernel": (1, 1, 256, 2 * na),
        "rpn_classification_head/bias": (2 * na,),
        "rpn_regression_head/kernel": (1, 1, 256, 4 * na),
        "rpn_regression_head/bias": (4 * na,),
        "fast_rcnn_classification_head/kernel": (flat, nc + 1),
        "fast_rcnn_classification_head/bias": (nc + 1,),
        "fast_rcnn_regression_head/kernel": (flat, 4 * nc),
        "fast_rcnn_regression_head/bias": (4 * nc,),
    }


REGULARIZED = ("rpn_intermediate_layer/kernel", "rpn_classification_head/kernel", "rpn_regression_head/kernel",
               "fast_rcnn_classification_head/kernel", "fast_rcnn_regression_head/kernel")


def init_params(config, depth=50, seed=0, randomize_affine=False):
    """Seeded synthetic init (no ImageNet download possible): He-normal backbone,
    TruncatedNormal(0, .01) RPN (rpn_detector.py:24), Glorot-uniform heads
    (fast_rcnn_detector.py:20)."""
    p = resnet.init_params(depth, seed, randomize_affine)
    g = torch.Generator().manual_seed(seed + 1)
    for name, shape in head_param_shapes(config).items():
        if name.endswith("/bias"):
            p[name] = torch.zeros(shape)
        elif name.startswith("rpn"):
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=0.01, a=-0.02, b=0.02, generator=g)
            p[name] = t
        else:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            p[name] = (torch.rand(shape, generator=g) * 2 - 1) * lim
    return p


def trainable_names(params):
    return [n for n in params if not (n.endswith("moving_mean") or n.endswith("moving_variance"))]


# --------------------------------------------------------------------------- model pieces
def rpn_forward(p, feature_maps, anchors, image_shape, training, quant=None):
    """reference RPNDetector.call (rpn_detector.py:60-96).  feature_maps NHWC."""
    x = feature_maps.permute(0, 3, 1, 2)
    w = p["rpn_intermediate_layer/kernel"].permute(3, 2, 0, 1)
    pad = w.shape[-1] // 2
    f = F.relu(F.conv2d(x, w, p["rpn_intermediate_layer/bias"], padding=pad))
    if quant is not None:
        f = quant(f)
    cls = F.conv2d(f, p["rpn_classification_head/kernel"].permute(3, 2, 0, 1), p["rpn_classification_head/bias"])
    reg = F.conv2d(f, p["rpn_regression_head/kernel"].permute(3, 2, 0, 1), p["rpn_regression_head/bias"])
    B = x.shape[0]
    pred_scores = torch.softmax(cls.permute(0, 2, 3, 1).reshape(B, -1, 2), dim=-1)
    pred_boxes = reg.permute(0, 2, 3, 1).reshape(B, -1, 1, 4)
    if training:
        keep = inside_indices(anchors, image_shape)
        regions = anchors[keep]
        pred_scores = pred_scores[:, keep]
        pred_boxes = pred_boxes[:, keep]
    else:
        regions = clip_to_window(anchors, [0, 0, image_shape[1], image_shape[0]])
    return {"regions": regions, "pred_scores": pred_scores, "pred_boxes": pred_boxes}


def rcnn_forward(p, feature_maps, rois, image_shape, config, quant=None):
    """reference FastRCNNDetector.call (fast_rcnn_detector.py:43-69)."""
    rp = config["rcnn"]["roi_pooling"]
    flat = roi_pooling(feature_maps, rois, rp["pooled_size"], rp["kernel_size"])
    if quant is not None:
        flat = quant(flat)
    logits = flat @ p["fast_rcnn_classification_head/kernel"] + p["fast_rcnn_classification_head/bias"]
    pred_scores = torch.softmax(logits, dim=-1)
    reg = flat @ p["fast_rcnn_regression_head/kernel"] + p["fast_rcnn_regression_head/bias"]
    pred_boxes = reg.reshape(reg.shape[0], reg.shape[1], config["num_classes"], 4)
    return {"regions": to_absolute(rois, image_shape), "pred_scores": pred_scores, "pred_boxes": pred_boxes}


def postprocess_output(image_shape, regions, pred_scores, pred_boxes, score_threshold, iou_threshold,
                       max_output_size_per_class, max_total_size):
    """reference utils/post_processing.py:6-63."""
    num_classes = pred_boxes.shape[2]
    regions = regions.unsqueeze(-2)
    regions = regions.expand(*regions.shape[:-2], num_classes, 4)
    boxes = decode(pred_boxes.detach(), regions)
    boxes = to_relative(boxes, image_shape)
    scores = pred_scores.detach()[..., 1:]
    nb, ns, nc, nv = combined_nms(boxes.contiguous(), scores.contiguous(), max_output_size_per_class,
                                  max_total_size, iou_threshold, score_threshold)
    return {"pred_boxes": nb, "pred_scores": ns, "pred_classes": nc, "num_valid_detections": nv}


def _training_samples(gt_labels, gt_boxes, regions, pred_scores, pred_boxes, image_shape, sampling, step, seed,
                      stream_base, sample_indices=None):
    """reference get_training_samples of both detectors (rpn_detector.py:98-160,
    fast_rcnn_detector.py:71-130).  regions [R,4] shared or [B,R,4] per image."""
    B = gt_labels.shape[0]
    tl, tb, idx = [], [], []
    for b in range(B):
        reg_b = regions if regions.dim() == 2 else regions[b]
        l, t = generate_targets(gt_labels[b], gt_boxes[b], reg_b, image_shape,
                                sampling["foreground_iou_interval"], sampling["background_iou_interval"])
        tl.append(l)
        tb.append(t)
        if sample_indices is not None:
            idx.append(sample_indices[b].long())
        else:
            idx.append(get_sample_indices(l, sampling["num_samples"], sampling["foreground_proportion"],
                                          image=b, step=step, seed=seed, stream_base=stream_base))
    tl, tb, idx = torch.stack(tl), torch.stack(tb), torch.stack(idx)
    ar = torch.arange(B)[:, None]
    return {"target_labels": tl[ar, idx], "pred_scores": pred_scores[ar, idx],
            "target_boxes": tb[ar, idx], "pred_boxes": pred_boxes[ar, idx],
            "sample_indices": idx, "all_target_labels": tl, "all_target_boxes": tb}


def forward(p, config, images, training, depth=50, taps=None, quant=None):
    """reference FasterRCNN.call (faster_rcnn.py:39-57)."""
    image_shape = config["image_shape"]
    feature_maps, new_stats = resnet.forward(p, images, training, depth, taps, quant)
    anchors = generate_anchors(feature_maps.shape[1:3], **config["rpn"]["anchors"])
    rpn_out = rpn_forward(p, feature_maps, anchors, image_shape, training, quant)
    nmsed_rpn = postprocess_output(image_shape, **rpn_out, **config["rpn"]["nms"])
    rcnn_out = rcnn_forward(p, feature_maps, nmsed_rpn["pred_boxes"], image_shape, config, quant)
    if taps is not None:
        taps["feature_maps"] = feature_maps
    return rpn_out, rcnn_out, nmsed_rpn, new_stats


def compute_losses(p, config, images, gt_labels, gt_boxes, training, step=0, seed=0, depth=50,
                   rpn_sample_indices=None, rcnn_sample_indices=None, taps=None, quant=None):
    image_shape = config["image_shape"]
    rpn_out, rcnn_out, nmsed_rpn, new_stats = forward(p, config, images, training, depth, taps, quant)
    gt_obj = F.one_hot(gt_labels.sum(-1).to(torch.int64), 2).to(torch.float32)       # rpn_detector.py:141
    rs = _training_samples(gt_obj, gt_boxes, **rpn_out, image_shape=image_shape, sampling=config["rpn"]["sampling"],
                           step=step, seed=seed, stream_base=0, sample_indices=rpn_sample_indices)
    cs = _training_samples(gt_labels, gt_boxes, **rcnn_out, image_shape=image_shape, sampling=config["rcnn"]["sampling"],
                           step=step, seed=seed, stream_base=2, sample_indices=rcnn_sample_indices)
    losses = {
        "rpn_cls": classification_loss(rs["target_labels"], rs["pred_scores"]),
        "rpn_reg": regression_loss(rs["target_boxes"], rs["pred_boxes"]),
        "rcnn_cls": classification_loss(cs["target_labels"], cs["pred_scores"]),
        "rcnn_reg": regression_loss(cs["target_boxes"], cs["pred_boxes"]),
    }
    nmsed_rcnn = postprocess_output(image_shape, **rcnn_out, **config["rcnn"]["nms"])
    preds = {"rpn_boxes": nmsed_rpn["pred_boxes"], "rpn_scores": nmsed_rpn["pred_scores"],
             "rcnn_boxes": nmsed_rcnn["pred_boxes"], "rcnn_scores": nmsed_rcnn["pred_scores"],
             "rcnn_classes": nmsed_rcnn["pred_classes"]}
    aux = {"rpn_out": rpn_out, "rcnn_out": rcnn_out, "rpn_samples": rs, "rcnn_samples": cs,
           "new_stats": new_stats, "nmsed_rpn": nmsed_rpn, "nmsed_rcnn": nmsed_rcnn}
    return losses, preds, aux


def train_step(p, velocity, config, images, gt_labels, gt_boxes, lr, step=0, seed=0, depth=50,
               rpn_sample_indices=None, rcnn_sample_indices=None, taps=None, quant=None):
    """reference FasterRCNN.train_step (faster_rcnn.py:59-117) + Keras SGD(momentum=.9)
    (train_faster_rcnn.py:109-112).  Updates `p` and `velocity` in place.  Returns
    (losses, preds, grads, aux)."""
    names = trainable_names(p)
    for n in names:
        p[n].requires_grad_(True)
    losses, preds, aux = compute_losses(p, config, images, gt_labels, gt_boxes, True, step, seed, depth,
                                        rpn_sample_indices, rcnn_sample_indices, taps, quant)
    wd = {n: (config["rpn"]["weight_decay"] if n.startswith("rpn") else config["rcnn"]["weight_decay"]) for n in REGULARIZED}
    reg = sum(wd[n] * (p[n] ** 2).sum() for n in REGULARIZED)                  # faster_rcnn.py:101 self.losses
    total = sum(losses.values()) + reg
    grads = torch.autograd.grad(total, [p[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(p[n])) for n, g in zip(names, grads)}
    with torch.no_grad():
        for n in names:
            p[n].requires_grad_(False)
            v = velocity.setdefault(n, torch.zeros_like(p[n]))
            v.mul_(SGD_MOMENTUM).sub_(lr * grads[n])
            p[n].add_(v)
        for n, v in aux["new_stats"].items():
            p[n] = v
    losses = {k: v.detach() for k, v in losses.items()}
    return losses, preds, grads, aux


def test_step(p, config, images, gt_labels, gt_boxes, step=0, seed=0, depth=50, quant=None):
    """reference FasterRCNN.test_step (faster_rcnn.py:119-169)."""
    with torch.no_grad():
        losses, preds, aux = compute_losses(p, config, images, gt_labels, gt_boxes, False, step, seed, depth, quant=quant)
    return losses, preds, aux


def synthetic_batch(batch, image_shape, num_classes=7, seed=1234, max_objects=100):
    """Synthetic KITTI-like inputs per SURVEY.md 8(d) (output contract of
    data/input_pipeline.py:83-130)."""
    g = torch.Generator().manual_seed(seed)
    H, W = image_shape[0], image_shape[1]
    images = torch.randint(0, 256, (batch, H, W, 3), generator=g, dtype=torch.uint8)
    gt_boxes = torch.zeros(batch, max_objects, 4)
    gt_labels = torch.zeros(batch, max_objects, num_classes + 1)
    for b in range(batch):
        n = int(torch.randint(1, 16, (1,), generator=g))
        w = 0.03 + torch.rand(n, generator=g) * 0.32
        h = 0.08 + torch.rand(n, generator=g) * 0.52
        x0 = torch.rand(n, generator=g) * (1 - w)
        y0 = torch.rand(n, generator=g) * (1 - h)
        gt_boxes[b, :n] = torch.stack([x0, y0, x0 + w, y0 + h], 1)
        cls = torch.randint(1, num_classes + 1, (n,), generator=g)
        gt_labels[b, torch.arange(n), cls] = 1.0
    return images, gt_labels, gt_boxes
