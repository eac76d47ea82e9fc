"""TEST ORACLE ONLY (see oracle/__init__.py): Feature-Pyramid-Network variant of the Faster-RCNN path -- BASELINE.json configs[4]
("ResNet-50 FPN Faster-RCNN").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.

PARITY UNPINNED, and more so than the rest of the oracle: the reference has NO FPN (models/faster_rcnn.py:25-34 wires the single
conv4_block6 map into RPNDetector and FastRCNNDetector; models/feature_extractor.py:8-9 truncates ResNet-50 there).  What follows
restates the published construction -- Lin, Dollar, Girshick, He, Hariharan, Belongie, "Feature Pyramid Networks for Object
Detection", CVPR 2017 -- on top of the reference's own pieces (its anchor formula, decode, combined NMS, crop_and_resize RoI pooling,
target assignment, sampling and losses, all reused unchanged from this package), with the choices the paper leaves open written down:

  sec. 3   lateral 1x1 convolution to d = 256 channels on every backbone stage output; top-down pathway: nearest-neighbour
           upsampling of the coarser merged map to the finer map's size, element-wise addition; a 3x3 convolution on every merged
           map; no non-linearities in these extra layers.
           Here: stages C2, C3, C4 = conv2_block3_out, conv3_block4_out, conv4_block6_out (strides 4, 8, 16: the reference's
           backbone ends at conv4).  Upsampling to a size that is not exactly 2x (47 rows from 24): source index = (dst * in) // out.
  sec. 4.1 the RPN head (3x3 conv + two sibling 1x1 convs) is attached to every level with SHARED parameters; anchors of a single
           scale per level and three aspect ratios; one extra coarser level by stride-2 subsampling (the paper's P6 from P5).
           Here: levels P2..P5 (P5 = P4[::2, ::2]) carry the reference config's four scales 0.25, 0.5, 1, 2 (x base 256 px) in
           that order, strides 4, 8, 16, 32 with the reference's centre convention (x * stride, no half-pixel offset,
           rpn_detector.py:162-199).  Regions of all levels are concatenated in level order P2..P5 and go through ONE combined NMS
           (the reference's postprocess_output); no per-level pre-NMS top-k.
  sec. 4.2 an RoI of width w and height h (input pixels) is pooled from level k = floor(k0 + log2(sqrt(w h) / 224)), k0 = 4, clamped
           to the levels with an output convolution (2..4).  Restated without logarithms (bit-exact on every device):
           k = 2 + [w h >= 112^2] + [w h >= 224^2].  The box head is the reference's two Dense layers on the flattened 7x7x256 crop.
"""
import math

import torch
import torch.nn.functional as F

from . import resnet
from .boxes import to_absolute
from .faster_rcnn import (REGULARIZED, SGD_MOMENTUM, _training_samples, generate_anchors, inside_indices, postprocess_output)
from .boxes import clip_to_window
from .losses import classification_loss, regression_loss
from .roi import roi_pooling

FPN_DIM = 256
LEVELS = (2, 3, 4)                 # levels with a lateral + output convolution
RPN_LEVELS = (2, 3, 4, 5)          # + the subsampled level
STAGE_OUT = {2: "conv2_block3_out", 3: "conv3_block4_out", 4: "conv4_block6_out"}
STAGE_CH = {2: 256, 3: 512, 4: 1024}
STRIDE = {2: 4, 3: 8, 4: 16, 5: 32}
FPN_REGULARIZED = REGULARIZED + tuple("fpn_lateral%d/kernel" % l for l in LEVELS) + tuple("fpn_output%d/kernel" % l for l in LEVELS)


def stage_out_name(level, depth=50):
    last = {2: 3, 3: 4, 4: {50: 6, 101: 23}[depth]}[level]
    return "conv%d_block%d_out" % (level, last)


def param_shapes(config):
    nc = config["num_classes"]
    ps = config["rcnn"]["roi_pooling"]["pooled_size"]
    ws = config["rpn"]["window_size"]
    na = len(config["rpn"]["anchors"]["aspect_ratios"])           # one scale per level
    flat = ps * ps * FPN_DIM
    s = {}
    for l in LEVELS:
        s["fpn_lateral%d/kernel" % l] = (1, 1, STAGE_CH[l], FPN_DIM)
        s["fpn_lateral%d/bias" % l] = (FPN_DIM,)
        s["fpn_output%d/kernel" % l] = (3, 3, FPN_DIM, FPN_DIM)
        s["fpn_output%d/bias" % l] = (FPN_DIM,)
    s.update({
        "rpn_intermediate_layer/kernel": (ws, ws, FPN_DIM, 256), "rpn_intermediate_layer/bias": (256,),
        "rpn_classification_head/kernel": (1, 1, 256, 2 * na), "rpn_classification_head/bias": (2 * na,),
        "rpn_regression_head/kernel": (1, 1, 256, 4 * na), "rpn_regression_head/bias": (4 * na,),
        "fast_rcnn_classification_head/kernel": (flat, nc + 1), "fast_rcnn_classification_head/bias": (nc + 1,),
        "fast_rcnn_regression_head/kernel": (flat, 4 * nc), "fast_rcnn_regression_head/bias": (4 * nc,),
    })
    return s


def init_params(config, depth=50, seed=0, randomize_affine=False):
    """Backbone as oracle.faster_rcnn.init_params; FPN layers Glorot-uniform; RPN TruncatedNormal(0, .01); heads Glorot-uniform."""
    p = resnet.init_params(depth, seed, randomize_affine)
    g = torch.Generator().manual_seed(seed + 1)
    for name, shape in param_shapes(config).items():
        if name.endswith("/bias"):
            p[name] = torch.zeros(shape)
        elif name.startswith("rpn"):
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=0.01, a=-0.02, b=0.02, generator=g)
            p[name] = t
        else:
            fan_in = shape[0] if len(shape) == 2 else shape[0] * shape[1] * shape[2]
            fan_out = shape[-1] if len(shape) == 2 else shape[0] * shape[1] * shape[3]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            p[name] = (torch.rand(shape, generator=g) * 2 - 1) * lim
    return p


# --------------------------------------------------------------------------- neck
def upsample_nearest(x_nchw, size):
    """nearest-neighbour resize with integer index arithmetic: src = (dst * in) // out"""
    h, w = size
    ys = (torch.arange(h) * x_nchw.shape[2]) // h
    xs = (torch.arange(w) * x_nchw.shape[3]) // w
    return x_nchw[:, :, ys][:, :, :, xs]


def neck(p, stage_maps, quant=None):
    """stage_maps: {2: C2, 3: C3, 4: C4} NHWC -> {2: P2, 3: P3, 4: P4, 5: P5} NHWC (Lin et al. sec. 3; P5 = P4[::2, ::2], sec. 4.1).
    `quant` (e.g. resnet.bf16_storage) is applied wherever the HIP path stores a tensor."""
    Q = quant if quant is not None else (lambda t: t)
    lat = {}
    for l in LEVELS:
        x = stage_maps[l].permute(0, 3, 1, 2)
        lat[l] = Q(F.conv2d(x, p["fpn_lateral%d/kernel" % l].permute(3, 2, 0, 1), p["fpn_lateral%d/bias" % l]))
    merged = {4: lat[4]}
    for l in (3, 2):
        merged[l] = Q(lat[l] + upsample_nearest(merged[l + 1], lat[l].shape[2:]))
    out = {}
    for l in LEVELS:
        y = Q(F.conv2d(merged[l], p["fpn_output%d/kernel" % l].permute(3, 2, 0, 1), p["fpn_output%d/bias" % l], padding=1))
        out[l] = y.permute(0, 2, 3, 1).contiguous()
    out[5] = out[4][:, ::2, ::2].contiguous()
    return out


# --------------------------------------------------------------------------- RPN over the pyramid
def level_anchors(config, grids):
    """{level: anchors [gh*gw*3, 4]} -- the reference's formula (rpn_detector.py:162-199) with ONE scale per level and the
    level's stride; row (y*gw+x)*3 + ratio index."""
    a = config["rpn"]["anchors"]
    assert len(a["scales"]) == len(RPN_LEVELS), "one anchor scale per pyramid level"
    return {l: generate_anchors(grids[l], [a["scales"][i]], a["aspect_ratios"], a["base_anchor_shape"], (STRIDE[l], STRIDE[l]))
            for i, l in enumerate(RPN_LEVELS)}


def rpn_forward(p, pyramid, config, image_shape, training, quant=None):
    """The reference's RPNDetector.call (rpn_detector.py:60-96) on every level with shared parameters; outputs concatenated in
    level order (per image: all kept anchors of P2, then P3, P4, P5)."""
    grids = {l: tuple(pyramid[l].shape[1:3]) for l in RPN_LEVELS}
    anchors = level_anchors(config, grids)
    regs, scores, boxes = [], [], []
    w = p["rpn_intermediate_layer/kernel"].permute(3, 2, 0, 1)
    for l in RPN_LEVELS:
        x = pyramid[l].permute(0, 3, 1, 2)
        f = F.relu(F.conv2d(x, w, p["rpn_intermediate_layer/bias"], padding=w.shape[-1] // 2))
        if quant is not None:
            f = quant(f)
        cls = F.conv2d(f, p["rpn_classification_head/kernel"].permute(3, 2, 0, 1), p["rpn_classification_head/bias"])
        reg = F.conv2d(f, p["rpn_regression_head/kernel"].permute(3, 2, 0, 1), p["rpn_regression_head/bias"])
        B = x.shape[0]
        sc = torch.softmax(cls.permute(0, 2, 3, 1).reshape(B, -1, 2), dim=-1)
        bx = reg.permute(0, 2, 3, 1).reshape(B, -1, 1, 4)
        if training:
            keep = inside_indices(anchors[l], image_shape)
            regs.append(anchors[l][keep])
            scores.append(sc[:, keep])
            boxes.append(bx[:, keep])
        else:
            regs.append(clip_to_window(anchors[l], [0, 0, image_shape[1], image_shape[0]]))
            scores.append(sc)
            boxes.append(bx)
    return {"regions": torch.cat(regs, 0), "pred_scores": torch.cat(scores, 1), "pred_boxes": torch.cat(boxes, 1)}


# --------------------------------------------------------------------------- Fast-RCNN heads over the pyramid
def roi_levels(rois_rel, image_shape):
    """Lin et al. eq. (1) with k0 = 4, clamped to [2, 4], in its logarithm-free form; rois_rel [..., 4] = [x1, y1, x2, y2] in [0, 1]."""
    w = (rois_rel[..., 2] - rois_rel[..., 0]) * float(image_shape[1])
    h = (rois_rel[..., 3] - rois_rel[..., 1]) * float(image_shape[0])
    area = w * h
    return 2 + (area >= 112.0 * 112.0).to(torch.int32) + (area >= 224.0 * 224.0).to(torch.int32)


def rcnn_forward(p, pyramid, rois, image_shape, config, quant=None):
    """The reference's FastRCNNDetector.call (fast_rcnn_detector.py:43-69) with every RoI pooled from its assigned level."""
    rp = config["rcnn"]["roi_pooling"]
    lv = roi_levels(rois, image_shape)
    flat = None
    for l in LEVELS:
        pooled = roi_pooling(pyramid[l], rois, rp["pooled_size"], rp["kernel_size"])          # [B, P, 49*256]
        flat = pooled * (lv == l).unsqueeze(-1) if flat is None else flat + pooled * (lv == l).unsqueeze(-1)
    if quant is not None:
        flat = quant(flat)
    logits = flat @ p["fast_rcnn_classification_head/kernel"] + p["fast_rcnn_classification_head/bias"]
    reg = flat @ p["fast_rcnn_regression_head/kernel"] + p["fast_rcnn_regression_head/bias"]
    return {"regions": to_absolute(rois, image_shape), "pred_scores": torch.softmax(logits, dim=-1),
            "pred_boxes": reg.reshape(reg.shape[0], reg.shape[1], config["num_classes"], 4), "levels": lv, "pooled": flat}


# --------------------------------------------------------------------------- model
def forward(p, config, images, training, depth=50, quant=None, taps=None, rois=None):
    """rois (optional): proposals to feed the heads instead of this forward pass's own NMS output (teacher forcing: the proposal
    list is a discrete function of near-tied scores at random init)."""
    image_shape = config["image_shape"]
    taps = {} if taps is None else taps
    _, new_stats = resnet.forward(p, images, training, depth, taps, quant)
    stage_maps = {l: taps[stage_out_name(l, depth)] for l in LEVELS}
    pyramid = neck(p, stage_maps, quant)
    rpn_out = rpn_forward(p, pyramid, config, image_shape, training, quant)
    nmsed_rpn = postprocess_output(image_shape, **rpn_out, **config["rpn"]["nms"])
    rcnn_out = rcnn_forward(p, pyramid, nmsed_rpn["pred_boxes"] if rois is None else rois, image_shape, config, quant)
    taps["pyramid"] = pyramid
    return rpn_out, rcnn_out, nmsed_rpn, new_stats


def compute_losses(p, config, images, gt_labels, gt_boxes, training, step=0, seed=0, depth=50, rpn_sample_indices=None,
                   rcnn_sample_indices=None, quant=None, taps=None, rois=None):
    """oracle.faster_rcnn.compute_losses on the pyramid (same targets, sampling, losses, post-processing)."""
    image_shape = config["image_shape"]
    rpn_out, rcnn_out, nmsed_rpn, new_stats = forward(p, config, images, training, depth, quant, taps, rois)
    gt_obj = F.one_hot(gt_labels.sum(-1).to(torch.int64), 2).to(torch.float32)
    head = {k: rcnn_out[k] for k in ("regions", "pred_scores", "pred_boxes")}
    rs = _training_samples(gt_obj, gt_boxes, **rpn_out, image_shape=image_shape, sampling=config["rpn"]["sampling"], step=step, seed=seed,
                           stream_base=0, sample_indices=rpn_sample_indices)
    cs = _training_samples(gt_labels, gt_boxes, **head, image_shape=image_shape, sampling=config["rcnn"]["sampling"], step=step, seed=seed,
                           stream_base=2, sample_indices=rcnn_sample_indices)
    losses = {"rpn_cls": classification_loss(rs["target_labels"], rs["pred_scores"]),
              "rpn_reg": regression_loss(rs["target_boxes"], rs["pred_boxes"]),
              "rcnn_cls": classification_loss(cs["target_labels"], cs["pred_scores"]),
              "rcnn_reg": regression_loss(cs["target_boxes"], cs["pred_boxes"])}
    nmsed_rcnn = postprocess_output(image_shape, **head, **config["rcnn"]["nms"])
    preds = {"rpn_boxes": nmsed_rpn["pred_boxes"], "rpn_scores": nmsed_rpn["pred_scores"], "rcnn_boxes": nmsed_rcnn["pred_boxes"],
             "rcnn_scores": nmsed_rcnn["pred_scores"], "rcnn_classes": nmsed_rcnn["pred_classes"]}
    aux = {"rpn_out": rpn_out, "rcnn_out": rcnn_out, "rpn_samples": rs, "rcnn_samples": cs, "new_stats": new_stats, "nmsed_rpn": nmsed_rpn,
           "nmsed_rcnn": nmsed_rcnn}
    return losses, preds, aux


def train_step(p, velocity, config, images, gt_labels, gt_boxes, lr, step=0, seed=0, depth=50, rpn_sample_indices=None,
               rcnn_sample_indices=None, quant=None, taps=None):
    """oracle.faster_rcnn.train_step on the pyramid: the L2 regulariser also covers the FPN convolution kernels (they are
    Conv2D layers like the RPN's: the reference regularises every head-side kernel, rpn_detector.py:25-55)."""
    names = [n for n in p if not (n.endswith("moving_mean") or n.endswith("moving_variance"))]
    for n in names:
        p[n].requires_grad_(True)
    losses, preds, aux = compute_losses(p, config, images, gt_labels, gt_boxes, True, step, seed, depth, rpn_sample_indices,
                                        rcnn_sample_indices, quant, taps)
    wd = lambda n: config["rcnn"]["weight_decay"] if n.startswith("fast_rcnn") else config["rpn"]["weight_decay"]
    reg = sum(wd(n) * (p[n] ** 2).sum() for n in FPN_REGULARIZED)
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
    return {k: v.detach() for k, v in losses.items()}, preds, grads, aux
