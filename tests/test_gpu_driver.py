"""GPU: the training driver end to end on tiny inputs -- synthetic records and a KITTI-like TFRecord data set through the
input pipeline -- with epoch validation, scalar logs, checkpoint every 5 epochs, restore, and the final weight file
(reference train_faster_rcnn.py:109-132,145-244)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _config(tmp_path):
    C = importlib.import_module("2d_object_detection_amd.config")
    cfg = C.default_config((128, 192, 3))
    cfg["rpn"]["anchors"]["base_anchor_shape"] = [32, 32]
    cfg["rpn"]["nms"]["max_total_size"] = cfg["rpn"]["nms"]["max_output_size_per_class"] = 40
    cfg["rpn"]["sampling"]["num_samples"] = 32
    cfg["rcnn"]["sampling"]["num_samples"] = 16
    cfg["rcnn"]["nms"]["max_total_size"] = 30
    cfg["rcnn"]["nms"]["max_output_size_per_class"] = 10
    path = str(tmp_path / "config.json")
    json.dump(cfg, open(path, "w"))
    return path


def _scalars(logs, split):
    runs = os.listdir(logs)
    out = []
    for r in runs:
        f = os.path.join(logs, r, "faster-rcnn", split, "scalars.jsonl")
        if os.path.exists(f):
            out += [json.loads(line) for line in open(f)]
    return out


def test_driver_synthetic_checkpoint_and_restore(tmp_path):
    sys.path.insert(0, ROOT)
    drv = importlib.import_module("train_faster_rcnn")
    common = ["--synthetic", "6", "--config-file", _config(tmp_path), "--logs-dir", str(tmp_path / "logs"), "--save-dir", str(tmp_path / "saved"),
              "--checkpoints-dir", str(tmp_path / "ckpt"), "--num-steps-per-epoch", "1", "--batch-size", "2",
              "--learning-rates", "1e-4", "1e-5", "--decay-steps", "4"]
    assert drv.main(common + ["--num-steps", "5"]) == 0
    assert os.listdir(tmp_path / "ckpt" / "faster-rcnn") == ["ckpt-5.pt"]                  # epoch 5: first checkpoint
    w5 = torch.load(tmp_path / "saved" / "faster-rcnn" / "weights")
    assert "conv1_conv/kernel" in w5 and all(torch.isfinite(torch.as_tensor(v)).all() for v in w5.values())
    tr, va = _scalars(tmp_path / "logs", "train"), _scalars(tmp_path / "logs", "valid")
    tags = {"Losses/Faster-RCNN/classification_loss", "Losses/Faster-RCNN/regression_loss", "Metrics/Faster-RCNN/mAP@IoU=.50",
            "Losses/RPN/classification_loss", "Losses/RPN/regression_loss", "Metrics/RPN/AP@IoU=.50"}
    assert {s["tag"] for s in tr} == tags and {s["tag"] for s in va} == tags
    assert sorted({s["step"] for s in tr}) == [1, 2, 3, 4, 5]
    assert all(np.isfinite(s["value"]) for s in tr + va)
    # the validation pass's image summaries (reference train_faster_rcnn.py:169-195): ground truth at epoch 1, detections at epoch 5
    import glob
    EV = importlib.import_module("2d_object_detection_amd.data.tfevents")
    imgs = [r for f in glob.glob(str(tmp_path / "logs" / "*" / "faster-rcnn" / "valid" / "events.out.tfevents.*")) for r in EV.read_images(f)]
    assert sorted((s_, t_) for s_, t_, _, _, _ in imgs) == [(0, "Ground-truth"), (5, "Predictions/pred@score=.50")]
    assert all(png[:4] == b"\x89PNG" and h > 0 and w_ > 0 for _, _, h, w_, png in imgs)
    # second run: restores step 5 (weights, momentum, schedule position) and stops at 7
    ck = torch.load(tmp_path / "ckpt" / "faster-rcnn" / "ckpt-5.pt")
    assert ck["step"] == 5 and ck["optimizer"]["iterations"] == 5
    assert torch.equal(torch.as_tensor(ck["model"]["conv1_conv/kernel"]), torch.as_tensor(w5["conv1_conv/kernel"]))
    assert drv.main(common + ["--num-steps", "7"]) == 0
    steps = sorted({s["step"] for s in _scalars(tmp_path / "logs", "train")})
    assert steps == [1, 2, 3, 4, 5, 6, 7]
    w7 = torch.load(tmp_path / "saved" / "faster-rcnn" / "weights")
    assert not torch.equal(torch.as_tensor(w7["conv1_conv/kernel"]), torch.as_tensor(w5["conv1_conv/kernel"]))


def test_driver_on_tfrecords(tmp_path):
    """KITTI-like PNG + label files -> build_records -> the driver's sharded input pipeline -> two training steps."""
    sys.path.insert(0, ROOT)
    drv = importlib.import_module("train_faster_rcnn")
    BR = importlib.import_module("2d_object_detection_amd.data.build_records")
    os.makedirs(tmp_path / "image_2")
    os.makedirs(tmp_path / "label_2")
    rng = np.random.default_rng(1)
    for i in range(6):
        Image.fromarray(rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)).save(tmp_path / "image_2" / ("%06d.png" % i))
        with open(tmp_path / "label_2" / ("%06d.txt" % i), "w") as fh:
            fh.write("Car 0 0 0 %d 20 %d 90 1 1 1 1 1 1 0\n" % (20 + 5 * i, 110 + 5 * i))
            fh.write("Pedestrian 0 0 0 120 30 150 100 1 1 1 1 1 1 0\n")
    BR.main(["--images-dir", str(tmp_path / "image_2"), "--labels-dir", str(tmp_path / "label_2"), "--output-dir", str(tmp_path / "rec"),
             "--validation-set-size", "2"])
    rc = drv.main(["--train-data-path", str(tmp_path / "rec" / "train.tfrecord"), "--valid-data-path", str(tmp_path / "rec" / "valid.tfrecord"),
                   "--config-file", _config(tmp_path), "--logs-dir", str(tmp_path / "logs"), "--save-dir", str(tmp_path / "saved"),
                   "--checkpoints-dir", str(tmp_path / "ckpt"), "--num-steps", "2", "--num-steps-per-epoch", "2", "--batch-size", "2",
                   "--learning-rates", "1e-4", "--decay-steps"])
    assert rc == 0
    va = _scalars(tmp_path / "logs", "valid")
    assert len(va) == 6 and all(np.isfinite(s["value"]) for s in va)
    assert os.path.exists(tmp_path / "saved" / "faster-rcnn" / "weights")


def test_metrics_on_their_own_stream_equal_the_metrics_on_the_training_stream():
    """train_faster_rcnn.MetricStream: the AP / mAP updates run on a side stream on clones of their inputs.  The inputs here live in
    REUSED buffers that are overwritten right after every update (as the step's static prediction buffers are by the next replay),
    with enough queued work on the training stream that a missing clone or a missing stream dependency would read the next update's
    values: state and result must equal the same metrics updated in place on the training stream from private copies."""
    sys.path.insert(0, ROOT)
    T = importlib.import_module("train_faster_rcnn")
    MET = importlib.import_module("2d_object_detection_amd.utils.metrics")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, G, P, C = 2, 100, 300, 7
    ms = T.MetricStream(dev)
    side_map, side_ap = ms.wrap(MET.MeanAveragePrecision(C, 0.5)), ms.wrap(MET.AveragePrecision(0.5))
    main_map, main_ap = MET.MeanAveragePrecision(C, 0.5), MET.AveragePrecision(0.5)
    bufs = None
    ballast = torch.zeros(64 << 20, device=dev)
    for _ in range(6):
        gt = torch.rand(B, G, 2, generator=g) * 0.5
        gt = torch.cat([gt, gt + 0.05 + torch.rand(B, G, 2, generator=g) * 0.3], -1)
        gt[:, 40:] = 0.0
        labels = torch.zeros(B, G, C + 1)
        labels.scatter_(2, torch.randint(1, C + 1, (B, G, 1), generator=g), 1.0)
        labels[:, 40:] = 0.0
        pb = torch.rand(B, P, 2, generator=g) * 0.5
        pb = torch.cat([pb, pb + 0.05 + torch.rand(B, P, 2, generator=g) * 0.3], -1)
        pb[:, :40] = gt[:, :40] + (torch.rand(B, 40, 4, generator=g) - 0.5) * 0.04
        ps = torch.rand(B, P, generator=g)
        pc = torch.randint(0, C, (B, P), generator=g).float()
        fresh = [t.to(dev) for t in (gt, labels, pb, ps, pc)]
        main_map.update_state(*[t.clone() for t in fresh])
        main_ap.update_state(fresh[0].clone(), fresh[2].clone(), fresh[3].clone())
        if bufs is None:
            bufs = [torch.empty_like(t) for t in fresh]
        for b, t in zip(bufs, fresh):
            b.copy_(t)
        ballast.add_(1.0)                                  # (work queued ahead of the updates on the training stream)
        side_map.update_state(*bufs)
        side_ap.update_state(bufs[0], bufs[2], bufs[3])
        for b in bufs:                                     # the "next step" overwrites the buffers at once
            b.fill_(0.25)
    torch.cuda.synchronize()
    for a, b in zip(side_map.metric.average_precisions + [side_ap.metric], main_map.average_precisions + [main_ap]):
        with ms.reading():
            assert a._pos_count == b._pos_count
            assert all(torch.equal(x, y) for x, y in zip(a._true_pos, b._true_pos))
            assert all(torch.equal(x, y) for x, y in zip(a._scores, b._scores))
            assert [int(x) for x in a._true_count] == [int(y) for y in b._true_count]
    assert side_map.result() == main_map.result() and side_ap.result() == main_ap.result()
    assert 0.0 < main_ap.result() <= 1.0
    side_map.reset_states()
    assert side_map.result() == 0.0
