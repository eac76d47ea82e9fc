"""CPU: the oracle against the hand-derived known answers (tests/golden/known_answers.json) and its
own regression pin (oracle_small_step.json, restatement-derived)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import boxes, faster_rcnn as O, losses, metrics, nms, roi, training
from oracle.philox import philox4x32

HERE = os.path.dirname(os.path.abspath(__file__))
K = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))
T = lambda x: torch.tensor(x, dtype=torch.float32)


def test_anchors_known_answers():
    a = K["anchors_375x1242"]
    cfg = O.default_config()
    assert O.feature_grid(cfg["image_shape"]) == (24, 78)
    anc = O.generate_anchors((24, 78), **cfg["rpn"]["anchors"])
    assert anc.shape[0] == a["count"]
    inside = O.inside_indices(anc, cfg["image_shape"])
    assert inside.numel() == a["inside_count"]
    assert [int(((inside % 12) == k).sum()) for k in range(12)] == a["inside_per_k"]
    wh = torch.stack([anc[:12, 2] - anc[:12, 0], anc[:12, 3] - anc[:12, 1]], 1)
    assert torch.allclose(wh, T(a["wh_per_k"]), rtol=1e-6)
    assert torch.allclose(anc[0], T(a["row0"]), rtol=1e-6)
    r = a["row_y1_x2_k5"]
    assert torch.allclose(anc[r["index"]], T(r["box"]))


def test_box_math_known_answers():
    d = K["decode"]
    assert torch.allclose(boxes.decode(T(d["deltas"]), T(d["reference_box"])), T(d["decoded"]), atol=1e-5)
    e = K["encode"]
    assert torch.allclose(boxes.encode(T(e["box"]), T(e["reference_box"])), T(e["encoded"]), atol=1e-6)
    c = K["clip_to_window"]
    assert torch.equal(boxes.clip_to_window(T(c["box"]), c["window"]), T(c["clipped"]))
    g = torch.Generator().manual_seed(0)
    ref = torch.rand(50, 4, generator=g)
    ref[:, 2:] += ref[:, :2] + 0.1
    b = torch.rand(50, 4, generator=g)
    b[:, 2:] += b[:, :2] + 0.1
    assert torch.allclose(boxes.decode(boxes.encode(b, ref), ref), b, atol=1e-5)      # encode o decode identity
    assert torch.allclose(boxes.to_relative(boxes.to_absolute(b, (375, 1242, 3)), (375, 1242, 3)), b, atol=1e-6)


def test_iou_known_answers():
    for c in K["iou"]["cases"]:
        a, b = T([c["a"]]), T([c["b"]])
        assert abs(float(metrics.iou(a, b, pairwise=True)[0, 0]) - c["iou"]) < 1e-6, c["why"]
        assert abs(float(metrics.iou(a, b)[0]) - c["iou"]) < 1e-6, c["why"]


def test_target_assignment_truth_table():
    t = K["target_assignment"]
    tl, tb = training.generate_targets(T(t["gt_labels"]), T(t["gt_boxes"]), T(t["regions"]), t["image_shape"], t["fg_interval"],
                                       t["bg_interval"])
    assert torch.equal(tl, T(t["target_labels"]))
    assert torch.allclose(tb[1, 1], T(t["target_box_r1_class2"]), atol=1e-6)       # class 2 -> slot 1
    assert torch.allclose(tb[2, 0], T(t["target_box_r2_class1"]), atol=1e-6)
    assert float(tb[3].abs().sum()) == 0 and float(tb[4].abs().sum()) == 0 and float(tb[1, 0].abs().sum()) == 0
    assert torch.allclose(tb[0, 1], torch.zeros(4), atol=1e-6)                     # forced fg with IoU 1: zero offsets
    q = K["rpn_objectness_padding_quirk"]
    obj = torch.nn.functional.one_hot(torch.tensor(q["gt_label_sums"]), 2).float()
    assert obj.tolist() == q["objectness"] and bool((obj.sum(-1) != 0).all())


def test_losses_known_answers():
    c = K["classification_loss"]
    assert abs(float(losses.classification_loss(T(c["target"]), T(c["pred"]))) - c["loss"]) < 1e-6
    r = K["regression_loss"]
    assert abs(float(losses.regression_loss(T(r["target"]), T(r["pred"]))) - r["loss"]) < 1e-6


def test_sampling_counts_and_contract():
    for c in K["sampling_counts"]["cases"]:
        assert training._round_half_even(c["S"] * c["p"]) == c["n_fg_max"]
    tl = torch.zeros(1000, 8)
    tl[:300, 0] = 1.0
    tl[300:340, 3] = 1.0
    idx = training.get_sample_indices(tl, 64, 0.25, image=1, step=2, seed=3)
    assert len(set(idx[:16].tolist())) == 16 and all(300 <= i < 340 for i in idx[:16].tolist())
    assert all(i < 300 for i in idx[16:].tolist())
    assert not torch.equal(idx, training.get_sample_indices(tl, 64, 0.25, image=1, step=3, seed=3))
    assert torch.equal(idx, training.get_sample_indices(tl, 64, 0.25, image=1, step=2, seed=3))
    with pytest.raises(ValueError):
        training.get_sample_indices(tl[300:340], 64, 0.25)


def test_combined_nms_known_answer_and_c_vs_python():
    n = K["combined_nms"]
    b, s = T(n["boxes"])[None], T(n["scores"])[None]
    for fn in (nms.combined_nms, nms.combined_nms_py):
        ob, os_, oc, ov = fn(b, s, n["max_per_class"], n["max_total"], n["iou_threshold"], n["score_threshold"])
        assert torch.allclose(ob[0], T(n["out_boxes"])) and torch.allclose(os_[0], T(n["out_scores"])) and int(ov[0]) == n["num_valid"]
    g = torch.Generator().manual_seed(1)
    for (B, N, q, C) in ((2, 60, 1, 1), (1, 40, 3, 3)):
        ctr, sz = torch.rand(B, N, q, 2, generator=g), torch.rand(B, N, q, 2, generator=g) * 0.3 + 0.02
        bx = torch.cat([ctr - sz / 2, ctr + sz / 2], -1)
        sc = torch.rand(B, N, C, generator=g)
        a = nms.combined_nms(bx, sc, 10, 15, 0.5, 0.1)
        p = nms.combined_nms_py(bx, sc, 10, 15, 0.5, 0.1)
        assert all(torch.equal(x, y) for x, y in zip(a, p))
        # invariants: sorted scores, caps, zero padding
        for bi in range(B):
            v = int(a[3][bi])
            assert v <= 15 and bool((a[1][bi, :v][:-1] >= a[1][bi, :v][1:]).all()) and float(a[1][bi, v:].abs().sum()) == 0


def test_crop_and_resize_known_answer():
    c = K["crop_and_resize"]
    img = T(c["image"]).reshape(1, 2, 2, 1)
    out = roi.crop_and_resize(img, T([[0, 0, 1, 1]]), torch.tensor([0]), (3, 3))
    assert torch.allclose(out[0, :, :, 0], T(c["full_box_3x3"]))
    out = roi.crop_and_resize(img, T([[0, 0, 2, 2]]), torch.tensor([0]), (3, 3))
    assert torch.allclose(out[0, :, :, 0], T(c["overshoot_box_3x3"]))


def test_philox_known_answers():
    for c in K["philox4x32_10"]["cases"]:
        assert philox4x32(np.array(c["ctr"], dtype=np.uint32), c["key"]).tolist() == c["out"]


def test_parameter_count():
    p = O.init_params(O.default_config(), seed=0)
    assert sum(v.numel() for k, v in p.items() if "moving" not in k) == K["work_per_image"]["trainable_params"]


def test_oracle_small_step_regression():
    pin = json.load(open(os.path.join(HERE, "golden", "oracle_small_step.json")))
    cfg = O.default_config((96, 160, 3))
    cfg["rpn"]["anchors"]["base_anchor_shape"] = [32, 32]
    p = O.init_params(cfg, seed=0)
    images, gl, gb = O.synthetic_batch(2, cfg["image_shape"], seed=1)
    ls, preds, grads, aux = O.train_step(p, {}, cfg, images, gl, gb, lr=1e-3, step=0, seed=7)
    for k, v in pin["losses"].items():
        assert abs(float(ls[k]) - v) <= 2e-3 * max(1.0, abs(v)), k
    assert aux["nmsed_rpn"]["num_valid_detections"].tolist() == pin["num_valid_rpn"]
    assert aux["rpn_samples"]["sample_indices"][:, :8].tolist() == pin["rpn_sample_indices_first8"]


# ------------------------------------------------------------------ hardening of the (parity-unpinned) [TF-ext] restatements
def test_keras_resnet50_topology_known_answers():
    """oracle/resnet.py against the layer table tf.keras.applications.ResNet50 prints (model.summary(), public): parameter
    count per layer name, blocks per stage, total up to conv4_block6_out, and the stage output shapes at 224 x 224."""
    from oracle import resnet as R
    k = K["keras_resnet50_v1_summary"]
    shapes = R.param_shapes(50)
    count = lambda n: int(np.prod(shapes[n]))
    for layer, want in k["layer_params"].items():
        if layer.endswith("_conv"):
            got = count(layer + "/kernel") + count(layer + "/bias")
        else:
            got = sum(count(layer + "/" + s) for s in ("gamma", "beta", "moving_mean", "moving_variance"))
        assert got == want, (layer, got, want)
    for stage, nb in k["blocks_per_stage"].items():
        assert sorted({n.split("_")[1] for n in shapes if n.startswith(stage + "_")}) == ["block%d" % (i + 1) for i in range(nb)]
    assert not any(n.startswith("conv5") for n in shapes)
    assert sum(count(n) for n in shapes) == k["total_params_to_conv4_block6_out"]
    assert sum(count(n) for n in shapes if "moving" in n) == k["non_trainable_to_conv4_block6_out"]
    taps = {}
    p = R.init_params(50, seed=0)
    out, _ = R.forward(p, torch.zeros(1, 224, 224, 3, dtype=torch.uint8), False, taps=taps)
    assert [out.shape[1], out.shape[2], out.shape[3]] == k["output_shapes_224"]["conv4_block6_out"]
    checked = 0
    for name, shp in k["output_shapes_224"].items():
        if name in taps:
            t = taps[name]
            assert sorted(t.shape[1:]) == sorted(shp), (name, tuple(t.shape))
            checked += 1
    assert checked >= 2, sorted(taps)


def _random_boxes(g, n, lo=-0.2, hi=1.2):
    """normalised corner boxes incl. out-of-range coordinates, inverted corners and zero-area boxes"""
    a, b = torch.rand(n, 2, generator=g) * (hi - lo) + lo, torch.rand(n, 2, generator=g) * (hi - lo) + lo
    bx = torch.cat([a, b], 1)
    bx[::7, 2:] = bx[::7, :2]                         # zero area
    bx[3::11] = 0.0                                   # NMS zero padding
    return bx


def test_crop_and_resize_two_independent_restatements_agree():
    """oracle/roi.py (vectorised torch) vs oracle/indep.c (scalar loops written from the published contract): bit-equal forward
    on random boxes incl. out-of-range, inverted, zero-area and all-zero (padding) boxes, odd and 1-wide crops."""
    from oracle import indep
    g = torch.Generator().manual_seed(2)
    for (B, H, W, C, n, cs) in ((2, 7, 9, 3, 40, (14, 14)), (1, 24, 78, 8, 30, (14, 14)), (3, 5, 4, 2, 25, (3, 5)), (1, 6, 6, 1, 10, (1, 1))):
        img = torch.randn(B, H, W, C, generator=g)
        bx = _random_boxes(g, n)
        bi = torch.randint(0, B, (n,), generator=g)
        a = roi.crop_and_resize(img, bx, bi, cs)
        b = indep.crop_and_resize(img, bx, bi, cs)
        assert torch.equal(a, b), float((a - b).abs().max())
        # the all-zero padding proposal samples pixel (0, 0) everywhere
        z = roi.crop_and_resize(img, torch.zeros(1, 4), torch.tensor([0]), cs)
        assert torch.equal(z, img[0, 0, 0].expand_as(z))


def test_crop_and_resize_gradient_gradcheck_and_scatter_form():
    """fp64 finite-difference check of the autograd gradient the oracle's RoI backward relies on, and agreement with the
    independent 4-tap scatter (CropAndResizeGradImage form) in oracle/indep.c."""
    from oracle import indep
    g = torch.Generator().manual_seed(3)
    img64 = torch.randn(2, 5, 6, 2, generator=g, dtype=torch.float64, requires_grad=True)
    bx = torch.tensor([[0.1, 0.05, 0.9, 0.8], [0.0, 0.0, 1.0, 1.0], [-0.3, 0.2, 0.7, 1.4], [0.5, 0.5, 0.5, 0.5], [0.0, 0.0, 0.0, 0.0]])
    bi = torch.tensor([0, 1, 1, 0, 1])
    assert torch.autograd.gradcheck(lambda im: roi.crop_and_resize(im, bx, bi, (4, 3)), (img64,), eps=1e-6, atol=1e-6)
    for (B, H, W, C, n, cs) in ((2, 7, 9, 3, 40, (14, 14)), (1, 24, 30, 4, 64, (14, 14))):
        img = torch.randn(B, H, W, C, generator=g, requires_grad=True)
        bxs, bis = _random_boxes(g, n), torch.randint(0, B, (n,), generator=g)
        out = roi.crop_and_resize(img, bxs, bis, cs)
        up = torch.randn(out.shape, generator=g)
        out.backward(up)
        ref = indep.crop_and_resize_grad_image(up, (B, H, W, C), bxs, bis)
        err = float((img.grad - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
        assert err < 1e-5, err
    # through the 2x2 max-pool: every pooled element routes its gradient to exactly one crop sample, whose four bilinear
    # weights sum to one -> for in-range RoIs the image gradient of sum(pooled) sums to the number of pooled elements
    feat = torch.randn(1, 6, 7, 3, generator=g, requires_grad=True)
    rois = torch.tensor([[[0.1, 0.2, 0.8, 0.9], [0.3, 0.1, 0.6, 0.7]]])
    pooled = roi.roi_pooling(feat, rois, pooled_size=2, kernel_size=2)
    pooled.sum().backward()
    assert float(feat.grad.sum()) == pytest.approx(float(pooled.numel()), rel=1e-5)


def test_combined_nms_three_restatements_agree():
    """oracle/nms.c (sorted walk against a kept list), its Python twin and oracle/indep.c (textbook take-best / kill-overlaps,
    top-k by repeated arg-max): identical outputs on random inputs with out-of-range, inverted and zero-area boxes, score
    thresholds that cut candidates, and per-class / total caps that bind."""
    from oracle import indep
    g = torch.Generator().manual_seed(4)
    for (B, N, q, C, mpc, mt, thr, sthr) in ((2, 80, 1, 1, 10, 15, 0.5, 0.1), (1, 60, 3, 3, 4, 7, 0.3, 0.3), (2, 200, 1, 1, 300, 300, 0.7, 0.0),
                                              (1, 50, 7, 7, 10, 30, 0.5, 0.0), (1, 30, 1, 2, 50, 100, 0.5, 0.95)):
        bx = _random_boxes(g, B * N * q).reshape(B, N, q, 4)
        sc = torch.rand(B, N, C, generator=g)
        a = nms.combined_nms(bx, sc, mpc, mt, thr, sthr)
        b = indep.combined_nms(bx, sc, mpc, mt, thr, sthr)
        for x, y, what in zip(a, b, ("boxes", "scores", "classes", "valid")):
            assert torch.equal(x, y), (what, (B, N, q, C))
        assert float(a[0].min()) >= 0.0 and float(a[0].max()) <= 1.0            # clip_boxes
    bx = _random_boxes(g, 40).reshape(1, 40, 1, 4)
    sc = torch.rand(1, 40, 1, generator=g)
    assert all(torch.equal(x, y) for x, y in zip(nms.combined_nms_py(bx, sc, 5, 8, 0.4, 0.2), indep.combined_nms(bx, sc, 5, 8, 0.4, 0.2)))
