"""The bench step fed from HOST buffers: the PCIe-inclusive rate beside bench.py's `value` (inputs resident in HBM).

The hot path's boundary takes device tensors (FasterRCNN.train_step: CUDA uint8 images, fp32 labels / boxes); the training driver's
input pipeline (data/input_pipeline.py) hands over pinned host batches.  Three feeds of BASELINE.json configs[1], same windows as
bench.py (20 steps, state restored before each window, synchronize on both sides):

  resident   the batches already on the device (bench.py's timed region)
  serial     every step copies its pinned batch on the training stream, then steps
  prefetch   batch s+1 is copied on a side stream while step s runs (two device slots, event-ordered: the input pipeline's form)

    python tools/pcie_inclusive.py [--steps 20] [--windows 5] [--batch 4]   ->  one JSON line
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    OPT = importlib.import_module("2d_object_detection_amd.optimizers")
    C = importlib.import_module("2d_object_detection_amd.config")
    DATA = importlib.import_module("2d_object_detection_amd.data")
    dev = torch.device("cuda:0")
    cfg = C.default_config()
    B = args.batch
    model = M.FasterRCNN(cfg, depth=50, device=dev, seed=0, sampling_seed=0)
    opt = OPT.SGD(learning_rate=OPT.PiecewiseConstantDecay([40000, 80000], [1e-5, 1e-6, 1e-7]), momentum=0.9)
    NB = 4
    resident = [DATA.synthetic_batch(B, cfg["image_shape"], seed=1234 + 100 * i, device=dev) for i in range(NB)]
    pinned = [tuple(x.cpu().pin_memory() for x in b) for b in resident]
    bytes_per_step = sum(x.numel() * x.element_size() for x in pinned[0])
    for s in range(args.warmup):
        model.train_step(*resident[s % NB], opt)
    torch.cuda.synchronize()
    state0 = model._snapshot(opt)
    main_stream = torch.cuda.current_stream(dev)
    copy_stream = torch.cuda.Stream(dev)
    slots = [tuple(torch.empty_like(x) for x in resident[0]) for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def feed_resident(s):
        model.train_step(*resident[s % NB], opt)

    def feed_serial(s):
        slot = slots[0]
        for d, h in zip(slot, pinned[s % NB]):
            d.copy_(h, non_blocking=True)
        model.train_step(*slot, opt)

    def issue_copy(s):
        k = s % 2
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[k])              # the step that read this slot has taken its inputs
            for d, h in zip(slots[k], pinned[s % NB]):
                d.copy_(h, non_blocking=True)
            copied[k].record(copy_stream)

    def feed_prefetch(s):
        k = s % 2
        if s == 0:
            issue_copy(0)
        if s + 1 < args.steps:
            issue_copy(s + 1)
        main_stream.wait_event(copied[k])
        model.train_step(*slots[k], opt)
        consumed[k].record(main_stream)                       # (train_step's first launch copies the inputs into the plan's buffers)

    out = {"workload": "configs[1]: ResNet-50 C4, bf16, batch %d, 375x1242" % B, "steps": args.steps, "host_bytes_per_step": bytes_per_step}
    for name, feed in (("resident", feed_resident), ("serial", feed_serial), ("prefetch", feed_prefetch), ("resident_again", feed_resident)):
        ms = []
        for _ in range(args.windows):
            model._restore(state0, opt)
            for e in consumed:
                e.record(main_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(args.steps):
                feed(s)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) / args.steps * 1e3)
        med = sorted(ms)[len(ms) // 2]
        out[name] = {"ms_per_step": [round(m, 4) for m in ms], "median_ms": round(med, 4), "images_per_s": round(B / med * 1e3, 1)}
    # the copy alone, for the link rate it reaches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(50):
        for d, h in zip(slots[0], pinned[s % NB]):
            d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    out["copy_alone"] = {"us": round(dt * 1e6, 1), "GBs": round(bytes_per_step / dt / 1e9, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
