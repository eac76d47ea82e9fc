"""Trainability check of the fp8 path: the same resident synthetic batch trained for N steps in bf16 and in fp8 (forward convolutions,
data gradients AND weight gradients on fp8 operands), same seed, same schedule (learning rate 1e-5 as in bench.py: the reference's 1e-3
presumes pretrained weights) -- the two loss curves must fall together.  Not a parity test (the two runs sample different RoIs as soon
as a score moves); a measurement, kept under profiles/.
usage (GPU box): python tools/overfit_check.py [steps] [--fpn] [--batch N]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(precision, topology, steps, batch):
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    OPT = importlib.import_module("2d_object_detection_amd.optimizers")
    C = importlib.import_module("2d_object_detection_amd.config")
    DATA = importlib.import_module("2d_object_detection_amd.data")
    cfg = C.default_config()
    dev = torch.device("cuda", 0)
    model = M.FasterRCNN(cfg, depth=50, device=dev, seed=0, sampling_seed=0, precision=precision, topology=topology)
    opt = OPT.SGD(learning_rate=1e-5, momentum=0.9)
    images, gl, gb = DATA.synthetic_batch(batch, cfg["image_shape"], seed=1234, device=dev)
    curve = []
    acc = None
    for i in range(steps):
        losses, _ = model.train_step(images, gl, gb, opt)
        vals = torch.stack([losses[k] for k in sorted(losses)]).float()
        acc = vals.clone() if acc is None else acc + vals
        if (i + 1) % 50 == 0:
            torch.cuda.synchronize()
            curve.append((i + 1, (acc / 50).cpu().tolist()))
            acc = None
    return sorted(losses), curve


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 400
    topology = "fpn" if "--fpn" in sys.argv else "c4"
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
    out = {}
    for precision in ("bf16", "fp8"):
        names, out[precision] = run(precision, topology, steps, batch)
    print("mean losses over windows of 50 steps, %s topology, batch %d, lr 1e-5, momentum 0.9; columns: %s" % (topology, batch, ", ".join(names)))
    for (s, a), (_, b) in zip(out["bf16"], out["fp8"]):
        print("steps %4d-%4d   bf16 %s (sum %.4f)   fp8 %s (sum %.4f)" % (s - 49, s, " ".join("%.4f" % v for v in a), sum(a), " ".join("%.4f" % v for v in b), sum(b)))
        assert all(v == v for v in a + b), "non-finite loss"


if __name__ == "__main__":
    main()
