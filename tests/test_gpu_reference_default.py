"""The reference's OWN configuration on the GPU: /root/reference config.json:3 `image_shape [600, 1987, 3]` with the batch of 2 that
/root/reference train_faster_rcnn.py:52-54 defaults to -- the file INTEGRATION.md tells a maintainer to load unchanged.  Every other
full-size test and every bench line runs BASELINE.json's 375 x 1242; this geometry has a 38 x 125 feature grid (57 000 anchors),
odd extents at every stage (300 x 994 -> 150 x 497 -> 75 x 249 -> 38 x 125) and takes other workgroup-count branches of the 3x3
dispatch (tests/test_conv_dispatch.py::test_reference_default_plan_instantiations_have_parity_cases names them on the CPU).

* the train step, stage by stage on the HIP path's own upstream tensors against the oracle (anchors, RPN, proposal NMS, RoI pooling +
  heads, targets, sample indices, losses, detection NMS: discrete stages exact, the four losses 1e-4):
  tests/test_gpu_fullsize_stages.py::test_head_stages_at_benchmark_size[reference_default_r50_b2_p300_600x1987];
* the backbone's feature maps layer by layer against the bf16-storage oracle forward:
  tests/test_gpu_backbone_layers.py::test_backbone_teacher_forced[True-600x1987-b2];
* HERE: the anchor counts against a closed form derived by hand from rpn_detector.py:162-230 (not from the oracle's generator);
  hipGraph replay == eager on the first step; `test_step` at batch 1 with all 57 000 anchors clipped to the image."""
import importlib
import math

import pytest
import torch

from oracle import faster_rcnn as O
from oracle import resnet as oresnet

pytestmark = pytest.mark.gpu
SHAPE = (600, 1987, 3)
BATCH = 2


def _rel(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


def closed_form_anchor_counts(shape, cfg):
    """(grid h, grid w, anchors, per-k in-image counts) from the reference's definitions alone: centres (x * 16, y * 16) for x < gw,
    y < gh (rpn_detector.py:186-187); k = ratio-major, scale-minor with h = s / sqrt(r) * base_h, w = s * sqrt(r) * base_w (:178-184);
    inside <=> xmin >= 0, ymin >= 0, xmax <= W, ymax <= H (:218-221).  Per axis the centres that fit form a run
    ceil(half / 16) .. floor((extent - half) / 16): the count is a product of two run lengths."""
    H, W = shape[0], shape[1]
    f1 = lambda n: (n + 6 - 7) // 2 + 1            # 7x7 / 2, pad 3
    f2 = lambda n: (n + 2 - 3) // 2 + 1            # 3x3 / 2 max pool, pad 1
    f3 = lambda n: (n - 1) // 2 + 1                # 1x1 / 2
    gh, gw = f3(f3(f2(f1(H)))), f3(f3(f2(f1(W))))
    a = cfg["rpn"]["anchors"]
    bh, bw = a["base_anchor_shape"]
    per_k = []
    for r in a["aspect_ratios"]:
        for s in a["scales"]:
            # the reference computes in float32 (tf.sqrt on float32 tensors): follow it so that a box touching the border falls the same way
            h = float(torch.tensor(s, dtype=torch.float32) / torch.sqrt(torch.tensor(r, dtype=torch.float32)) * bh)
            w = float(torch.tensor(s, dtype=torch.float32) * torch.sqrt(torch.tensor(r, dtype=torch.float32)) * bw)

            def run(half, extent, n):
                lo, hi = math.ceil(half / 16.0), math.floor((extent - half) / 16.0)
                return max(0, min(hi, n - 1) - lo + 1)
            per_k.append(run(w / 2, W, gw) * run(h / 2, H, gh))
    return gh, gw, gh * gw * len(per_k), per_k


def test_anchor_counts_against_the_closed_form():
    """375 x 1242 -> 24 x 78 grid, 22 464 anchors, 8 768 inside (per k: 1350, 864, 66, 0, 1480, 1120, 496, 0, 1512, 1188,
    660, 32); the reference's own 600 x 1987 -> 38 x 125, 57 000 anchors, 30 833 inside.  The closed form reproduces both sets, and the
    HIP detector's `regions` (training: the in-image anchors, in anchor order) has exactly that many rows per k."""
    RPN = importlib.import_module("2d_object_detection_amd.models.detectors.rpn_detector")
    cfg0 = O.default_config((375, 1242, 3))
    gh, gw, total, per_k = closed_form_anchor_counts((375, 1242, 3), cfg0)
    assert (gh, gw, total, sum(per_k)) == (24, 78, 22464, 8768)
    assert per_k == [1350, 864, 66, 0, 1480, 1120, 496, 0, 1512, 1188, 660, 32]
    cfg = O.default_config(SHAPE)
    gh, gw, total, per_k = closed_form_anchor_counts(SHAPE, cfg)
    assert (gh, gw, total, sum(per_k)) == (38, 125, 57000, 30833), (gh, gw, total, sum(per_k), per_k)
    assert per_k == [3872, 3094, 1695, 0, 4114, 3510, 2398, 558, 4165, 3616, 2626, 1185]
    det = RPN.RPNDetector(SHAPE, (None, gh, gw, 1024), cfg["rpn"])
    det.setup(1, True)
    inside = det.regions(True).cpu()
    assert inside.shape == (30833, 4)
    # per k: recover k of every in-image row from its (w, h) -- the twelve shapes are distinct -- and count
    wh = torch.stack([inside[:, 2] - inside[:, 0], inside[:, 3] - inside[:, 1]], 1)
    anchors = O.generate_anchors((gh, gw), **cfg["rpn"]["anchors"])
    shapes = torch.stack([anchors[:12, 2] - anchors[:12, 0], anchors[:12, 3] - anchors[:12, 1]], 1)
    k_of = ((wh[:, None, :] - shapes[None]).abs().sum(-1)).argmin(1)
    assert torch.bincount(k_of, minlength=12).tolist() == per_k
    det.setup(1, False)
    assert det.regions(False).shape == (57000, 4)


def test_graph_replay_equals_eager_at_the_reference_shape():
    """600 x 1987, batch 2: the first step of a hipGraph-replayed model against the eager one -- same weights, inputs and sampler seed;
    the forward pass is reproducible (f64 BatchNorm statistics), so proposals and samples are equal and the losses agree to 1e-5; the
    weights after the update differ by the order of the float-atomic sums only."""
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    OPT = importlib.import_module("2d_object_detection_amd.optimizers")
    cfg = O.default_config(SHAPE)
    images, gl, gb = O.synthetic_batch(BATCH, SHAPE, seed=5)
    dimg, dgl, dgb = images.cuda(), gl.cuda(), gb.cuda()
    out = []
    for graphs in (False, True):
        m = M.FasterRCNN(cfg, sampling_seed=11)
        m.use_graphs = graphs
        m.init_weights(seed=4)
        w0 = m.store.w.clone()
        losses, preds = m.train_step(dimg, dgl, dgb, OPT.SGD(learning_rate=1e-3, momentum=0.9))
        torch.cuda.synchronize()
        assert m._train_plan["plan"].captured == graphs and int(m.status[0].item()) == 0
        aux = m._train_plan["aux"]
        out.append(dict(losses={k: float(v) for k, v in losses.items()}, rois=aux["nms_rpn"]["pred_boxes"].clone(),
                        rcnn_idx=aux["targets"]["rcnn_idx"].clone(), rpn_idx=aux["targets"]["rpn_idx"].clone(), w=m.store.w.clone(), w0=w0,
                        classes=preds["rcnn_classes"].clone(), launches=m._train_plan["plan"].num_launches))
        del m
    e, g = out
    assert e["launches"] == g["launches"]
    assert torch.equal(e["rois"], g["rois"]) and torch.equal(e["rpn_idx"], g["rpn_idx"]) and torch.equal(e["rcnn_idx"], g["rcnn_idx"])
    for k, v in e["losses"].items():
        assert v == v and abs(g["losses"][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, v, g["losses"][k])
    # the UPDATE (the step's gradient): two runs of one step differ by the order of their float-atomic sums (RoI backward, BatchNorm
    # partials), amplified on the way down a random-init backbone -- 1-2 % of the gradient in bf16; measured here 1.3e-4 of
    # the weights at the reference's learning rate of 1e-3
    assert torch.equal(e["w0"], g["w0"])
    upd = _rel(g["w"] - g["w0"], e["w"] - e["w0"])
    assert upd < 0.06 and _rel(g["w"], e["w"]) < 4e-4, (upd, _rel(g["w"], e["w"]))
    print("600x1987 batch 2: %d launches, losses %s, update rel %.2e" % (e["launches"], {k: round(v, 5) for k, v in e["losses"].items()}, upd))


def test_test_step_batch_one_all_anchors_clipped():
    """reference faster_rcnn.py:119-169 at the reference's shape, batch 1: BatchNorm on moving statistics, ALL 57 000 anchors clipped to the
    image (rpn_detector.py:92-94), proposal NMS over 57 000 candidates, heads, losses, detection NMS -- stage by stage against the oracle
    on the HIP path's own upstream tensors; discrete stages exact."""
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    cfg = O.default_config(SHAPE)
    images, gl, gb = O.synthetic_batch(1, SHAPE, seed=6)
    m = M.FasterRCNN(cfg, sampling_seed=11)
    m.init_weights(seed=4)
    # moving statistics of a trained net are not 0 / 1: give the eval-mode BatchNorm a variance that keeps activations O(1) through 16 blocks
    w = m.get_weights()
    for k in w:
        if k.endswith("_3_bn/gamma"):
            w[k] = w[k] * 0.25
    m.set_weights(w)
    p = {k: torch.as_tensor(v).float() for k, v in m.get_weights().items()}
    losses, preds = m.test_step(images.cuda(), gl.cuda(), gb.cuda())
    torch.cuda.synchronize()
    aux = m._eval_plan["aux"]
    feat = aux["feature_maps"].float().cpu()
    assert feat.shape == (1, 38, 125, 1024) and bool(torch.isfinite(feat).all())
    pq = {k: (v.to(torch.bfloat16).float() if k.endswith("/kernel") else v) for k, v in p.items()}
    feat_ref, _ = oresnet.forward(pq, images, False, quant=oresnet.bf16_storage)
    assert _rel(feat, feat_ref) < 0.03, _rel(feat, feat_ref)
    anchors = O.generate_anchors((38, 125), **cfg["rpn"]["anchors"])
    rpn_ref = O.rpn_forward(pq, feat, anchors, SHAPE, False, quant=oresnet.bf16_storage)
    hip_rpn = {k: v.cpu() for k, v in aux["rpn_out"].items()}
    assert hip_rpn["regions"].shape == (57000, 4) and torch.equal(hip_rpn["regions"], rpn_ref["regions"])
    assert float(hip_rpn["regions"].min()) >= 0 and float(hip_rpn["regions"][:, 2].max()) <= 1987 and float(hip_rpn["regions"][:, 3].max()) <= 600
    assert _rel(hip_rpn["pred_boxes"], rpn_ref["pred_boxes"]) < 0.03
    nms_ref = O.postprocess_output(SHAPE, **hip_rpn, **cfg["rpn"]["nms"])
    assert torch.equal(aux["nms_rpn"]["num_valid_detections"].cpu(), nms_ref["num_valid_detections"])
    assert torch.equal(aux["nms_rpn"]["pred_scores"].cpu(), nms_ref["pred_scores"])
    assert (aux["nms_rpn"]["pred_boxes"].cpu() - nms_ref["pred_boxes"]).abs().max() < 1e-5
    hip_rcnn = {k: v.cpu() for k, v in aux["rcnn_out"].items()}
    rcnn_ref = O.rcnn_forward(pq, feat, aux["nms_rpn"]["pred_boxes"].cpu(), SHAPE, cfg, quant=oresnet.bf16_storage)
    assert _rel(hip_rcnn["pred_boxes"], rcnn_ref["pred_boxes"]) < 0.03
    nms2 = O.postprocess_output(SHAPE, **hip_rcnn, **cfg["rcnn"]["nms"])
    assert torch.equal(preds["rcnn_classes"].cpu(), nms2["pred_classes"]) and torch.equal(preds["rcnn_scores"].cpu(), nms2["pred_scores"])
    assert preds["rpn_boxes"].shape == (1, 300, 4) and all(bool(torch.isfinite(v).all()) for v in losses.values())
