"""Writes tests/golden/known_answers.json and tests/golden/oracle_small_step.json.

known_answers.json -- HAND-DERIVED from the reference source text (no TensorFlow available, the
reference has no tests/fixtures).  Each entry carries its derivation.  These pin the oracle's
pure box/target/loss logic.

oracle_small_step.json -- RESTATEMENT-DERIVED (produced by oracle/ itself, NOT captured from
TensorFlow): a regression pin of the oracle on a tiny seeded train step.

Run:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

S2 = math.sqrt(2.0)
KNOWN = {
    "_provenance": "hand-derived from /root/reference source text (file:line in each entry); not TensorFlow-captured",
    "anchors_375x1242": {
        "derivation": "rpn_detector.py:162-199 with config.json:7-11; grid 24x78 (SURVEY A.1); k = ratio-major; "
                      "h = s/sqrt(r)*256, w = s*sqrt(r)*256; centres x*16, y*16; inclusive inside test :218-221",
        "count": 24 * 78 * 12,
        "inside_count": 8768,
        "inside_per_k": [1350, 864, 66, 0, 1480, 1120, 496, 0, 1512, 1188, 660, 32],
        "wh_per_k": [[64 * 0.5 * S2, 64 * S2], [128 * 0.5 * S2, 128 * S2], [256 * 0.5 * S2, 256 * S2], [512 * 0.5 * S2, 512 * S2],
                     [64, 64], [128, 128], [256, 256], [512, 512],
                     [64 * S2, 64 * 0.5 * S2], [128 * S2, 128 * 0.5 * S2], [256 * S2, 256 * 0.5 * S2], [512 * S2, 512 * 0.5 * S2]],
        "row0": [-32 * 0.5 * S2, -32 * S2, 32 * 0.5 * S2, 32 * S2],
        "row_y1_x2_k5": {"index": (1 * 78 + 2) * 12 + 5, "box": [32 - 64, 16 - 64, 32 + 64, 16 + 64]},
    },
    "decode": {
        "derivation": "utils/boxes.py:20-41: ref [0,0,10,20] -> centre (5,10), size (10,20); deltas [.1,.2,ln2,ln.5] -> "
                      "centre (6,14), size (20,10) -> [-4,9,16,19]",
        "reference_box": [0, 0, 10, 20], "deltas": [0.1, 0.2, math.log(2.0), math.log(0.5)], "decoded": [-4, 9, 16, 19],
    },
    "encode": {
        "derivation": "utils/boxes.py:44-73: inverse of the decode case",
        "box": [-4, 9, 16, 19], "reference_box": [0, 0, 10, 20], "encoded": [0.1, 0.2, math.log(2.0), math.log(0.5)],
    },
    "clip_to_window": {
        "derivation": "utils/boxes.py:4-17, window read x-first: [-5,-5,2000,400] in [0,0,1242,375] -> [0,0,1242,375]",
        "box": [-5, -5, 2000, 400], "window": [0, 0, 1242, 375], "clipped": [0, 0, 1242, 375],
    },
    "iou": {
        "derivation": "utils/metrics.py:136-208",
        "cases": [
            {"a": [0, 0, 2, 2], "b": [0, 0, 2, 2], "iou": 1.0, "why": "identical"},
            {"a": [0, 0, 2, 2], "b": [1, 0, 3, 2], "iou": 1.0 / 3.0, "why": "inter 2, union 4+4-2"},
            {"a": [0, 0, 2, 2], "b": [2, 0, 4, 2], "iou": 0.0, "why": "touching edge: width 0"},
            {"a": [0, 0, 2, 2], "b": [5, 5, 6, 6], "iou": 0.0, "why": "disjoint; inter==0 -> 0 (:208)"},
            {"a": [0, 0, 2, 2], "b": [0, 0, 0, 0], "iou": 0.0, "why": "zero-area padding box"},
            {"a": [0, 0, 4, 4], "b": [1, 1, 3, 3], "iou": 0.25, "why": "contained: 4/16"},
        ],
    },
    "target_assignment": {
        "derivation": "utils/training.py:7-77,123-143 on a 100x100 image; all coordinates are exact binary fractions so that fp32 "
                      "IoUs hit the thresholds exactly; fg [0.5,1), bg [0,0.3). gt0 = [12.5,12.5,50,50] class 2 (area 1406.25), "
                      "gt1 = [50,50,100,100] class 1, one padding row. regions: r0 == gt0 (IoU 1.0: NOT in [0.5,1) but it is the "
                      "global max -> forced fg, :137-138); r1 = [12.5,12.5,50,40] (1031.25/1406.25 = .7333 -> fg class 2); "
                      "r2 = [50,50,100,75] (1250/2500 = .5 exactly -> fg class 1, interval closed at .5); r3 = [0,0,20,20] "
                      "(56.25/1750 -> bg); r4 = [12.5,12.5,50,27.5] (562.5/1406.25 = .4: ignored); r5 = [200,200,210,210] "
                      "(IoU 0 -> bg, interval closed at 0)",
        "image_shape": [100, 100, 3],
        "gt_boxes": [[0.125, 0.125, 0.5, 0.5], [0.5, 0.5, 1.0, 1.0], [0, 0, 0, 0]],
        "gt_labels": [[0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 0]],
        "regions": [[12.5, 12.5, 50, 50], [12.5, 12.5, 50, 40], [50, 50, 100, 75], [0, 0, 20, 20], [12.5, 12.5, 50, 27.5],
                    [200, 200, 210, 210]],
        "fg_interval": [0.5, 1.0], "bg_interval": [0.0, 0.3],
        "target_labels": [[0, 0, 1, 0], [0, 0, 1, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0], [1, 0, 0, 0]],
        "target_box_r1_class2": [0.0, (31.25 - 26.25) / 27.5, 0.0, math.log(37.5 / 27.5)],
        "target_box_r2_class1": [0.0, (75 - 62.5) / 25.0, 0.0, math.log(50.0 / 25.0)],
    },
    "rpn_objectness_padding_quirk": {
        "derivation": "rpn_detector.py:141 + training.py:43-45 (SURVEY A.6): one_hot(int(sum(labels)),2) maps padding rows to [1,0], "
                      "whose sum != 0, so the RPN path keeps all gt rows (padding boxes have zero area -> IoU 0)",
        "gt_label_sums": [1, 0], "objectness": [[0, 1], [1, 0]],
    },
    "classification_loss": {
        "derivation": "utils/losses.py:10-18 + Keras CCE on probabilities (SURVEY A.7): rows p=[.25,.75],t=[0,1] and p=[.5,.5],t=[1,0]; "
                      "mean(-ln .75, -ln .5)",
        "pred": [[[0.25, 0.75], [0.5, 0.5]]], "target": [[[0, 1], [1, 0]]],
        "loss": 0.5 * (-math.log(0.75) - math.log(0.5)),
    },
    "regression_loss": {
        "derivation": "utils/losses.py:27-43: row0 target [1,1,1,1] pred [1.5,3,-2,1]: d=[.5,2,-3,0] -> huber [.125,1.5,2.5,0], mean 1.03125; "
                      "row1 target all zero -> dropped; row2 target [1,-1,0,0] sums to 0 -> dropped (quirk of :35); SUM over rows",
        "target": [[[[1, 1, 1, 1]], [[0, 0, 0, 0]], [[1, -1, 0, 0]]]], "pred": [[[[1.5, 3, -2, 1]], [[9, 9, 9, 9]], [[5, 5, 5, 5]]]],
        "loss": 1.03125,
    },
    "sampling_counts": {
        "derivation": "utils/training.py:104-107: n_fg = min(#fg, round_half_even(S*p)); 256*.5=128, 64*.25=16; round(2.5)=2",
        "cases": [{"S": 256, "p": 0.5, "n_fg_max": 128}, {"S": 64, "p": 0.25, "n_fg_max": 16}, {"S": 5, "p": 0.5, "n_fg_max": 2}],
    },
    "combined_nms": {
        "derivation": "SURVEY A.5 / post_processing.py:53-55: 1 image, 1 class, iou_thr .5: boxes b0=[0,0,.4,.4] s=.9, b1=[0,0,.4,.3] s=.8 "
                      "(IoU .75 with b0 -> suppressed), b2=[.5,.5,1.2,.9] s=.7 (kept, clipped to 1.0), b3 s=0 (not > threshold 0); "
                      "max_total 3 -> padded with zeros",
        "boxes": [[[0, 0, 0.4, 0.4]], [[0, 0, 0.4, 0.3]], [[0.5, 0.5, 1.2, 0.9]], [[0.1, 0.1, 0.2, 0.2]]],
        "scores": [[0.9], [0.8], [0.7], [0.0]], "iou_threshold": 0.5, "score_threshold": 0.0, "max_per_class": 3, "max_total": 3,
        "out_boxes": [[0, 0, 0.4, 0.4], [0.5, 0.5, 1.0, 0.9], [0, 0, 0, 0]], "out_scores": [0.9, 0.7, 0.0], "num_valid": 2,
    },
    "crop_and_resize": {
        "derivation": "SURVEY A.4: 2x2 single-channel image [[0,1],[2,3]], box [0,0,1,1], crop 3x3 -> samples at 0,.5,1 -> bilinear grid; "
                      "box [0,0,2,2] -> samples at 0,1,2: coordinate 2 > H-1 -> extrapolation 0",
        "image": [[0, 1], [2, 3]], "full_box_3x3": [[0, 0.5, 1], [1, 1.5, 2], [2, 2.5, 3]],
        "overshoot_box_3x3": [[0, 1, 0], [2, 3, 0], [0, 0, 0]],
    },
    "philox4x32_10": {
        "derivation": "Random123 known-answer vectors (Salmon et al. SC'11)",
        "cases": [{"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
                  {"ctr": [0xffffffff] * 4, "key": [0xffffffff, 0xffffffff], "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
                  {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0],
                   "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]}],
    },
    "keras_resnet50_v1_summary": {
        "derivation": "tf.keras.applications.ResNet50 as printed by model.summary() in the public Keras documentation / every tutorial "
                      "that shows it (input 224x224x3): parameter count per layer (Conv2D: kh*kw*cin*cout + cout -- every conv has a "
                      "bias; BatchNormalization: 4*C incl. the two moving statistics) and output shapes of the stage outputs.  Total of "
                      "include_top=False is the published 23,587,712; minus conv5_x (block1 6,054,912 + 2 x 4,471,808 = 14,998,528) "
                      "leaves 8,589,184 up to conv4_block6_out, the tensor models/feature_extractor.py:9 taps.  Strides: the FIRST 1x1 "
                      "of block1 and the shortcut carry the stride (v1), so conv3_block1_1_conv maps 56x56 -> 28x28.",
        "layer_params": {"conv1_conv": 9472, "conv1_bn": 256,
                         "conv2_block1_0_conv": 16640, "conv2_block1_1_conv": 4160, "conv2_block1_2_conv": 36928, "conv2_block1_3_conv": 16640,
                         "conv2_block1_3_bn": 1024, "conv2_block2_1_conv": 16448,
                         "conv3_block1_0_conv": 131584, "conv3_block1_1_conv": 32896, "conv3_block1_2_conv": 147584, "conv3_block1_3_conv": 66048,
                         "conv3_block2_1_conv": 65664,
                         "conv4_block1_0_conv": 525312, "conv4_block1_1_conv": 131328, "conv4_block1_2_conv": 590080, "conv4_block1_3_conv": 263168,
                         "conv4_block2_1_conv": 262400, "conv4_block6_3_bn": 4096},
        "blocks_per_stage": {"conv2": 3, "conv3": 4, "conv4": 6},
        "total_params_to_conv4_block6_out": 8589184,
        "non_trainable_to_conv4_block6_out": 30592,       # published 53,120 minus conv5_x BN statistics 2 * (5120 + 2 * 3072)
        "output_shapes_224": {"conv1_conv": [112, 112, 64], "pool1_pool": [56, 56, 64], "conv2_block3_out": [56, 56, 256],
                              "conv3_block1_1_conv": [28, 28, 128], "conv3_block4_out": [28, 28, 512], "conv4_block6_out": [14, 14, 1024]},
    },
    "work_per_image": {
        "derivation": "SURVEY A.1/A.2: trainable parameter count of R50-C4 + RPN + heads",
        "trainable_params": 12743020,
    },
}


def small_step():
    import torch
    from oracle import faster_rcnn as O
    cfg = O.default_config((96, 160, 3))
    cfg["rpn"]["anchors"]["base_anchor_shape"] = [32, 32]
    p = O.init_params(cfg, seed=0)
    images, gl, gb = O.synthetic_batch(2, cfg["image_shape"], seed=1)
    losses, preds, grads, aux = O.train_step(p, {}, cfg, images, gl, gb, lr=1e-3, step=0, seed=7)
    return {
        "_provenance": "restatement-derived: produced by oracle/faster_rcnn.py (torch %s), NOT captured from TensorFlow" % torch.__version__,
        "config": "default_config((96,160,3)), base_anchor_shape [32,32], init seed 0, batch seed 1, sampling seed 7, lr 1e-3",
        "losses": {k: float(v) for k, v in losses.items()},
        "num_valid_rpn": aux["nmsed_rpn"]["num_valid_detections"].tolist(),
        "rpn_sample_indices_first8": aux["rpn_samples"]["sample_indices"][:, :8].tolist(),
        "grad_norm_conv1": float(grads["conv1_conv/kernel"].norm()),
        "grad_norm_rpn": float(grads["rpn_intermediate_layer/kernel"].norm()),
    }


if __name__ == "__main__":
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(KNOWN, f, indent=1)
    with open(os.path.join(HERE, "oracle_small_step.json"), "w") as f:
        json.dump(small_step(), f, indent=1)
    print("golden files written")
