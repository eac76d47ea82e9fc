"""Time frcnn_nms_combined alone (kernel-development aid): RPN-like (N = 8768, 1 class, 300 kept) and detection-like
(N = 300, 7 classes) geometries, uniform and clustered scores."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")


def run(name, B, N, q, C, mpc, mt, thr, scores, boxes):
    ob, os_ = torch.zeros(B, mt, 4, device="cuda"), torch.zeros(B, mt, device="cuda")
    oc, ov = torch.zeros(B, mt, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")
    ws = torch.zeros(ops.nms_workspace_bytes(B, N, C, mpc, mt), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        ops.nms_combined(boxes, scores, B, N, q, C, C, 0, mpc, mt, thr, 0.0, ob, os_, oc, ov, ws)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.nms_combined(boxes, scores, B, N, q, C, C, 0, mpc, mt, thr, 0.0, ob, os_, oc, ov, ws)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("%-44s median %.1f us  min %.1f us   valid %s" % (name, ts[len(ts) // 2], ts[0], ov.tolist()), flush=True)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    B, N = 4, 8768
    ctr = torch.rand(B, N, 1, 2, device="cuda", generator=g)
    sz = torch.rand(B, N, 1, 2, device="cuda", generator=g) * 0.3 + 0.02
    boxes = torch.cat([ctr - sz / 2, ctr + sz / 2], -1).contiguous()
    run("rpn 8768 -> 300, uniform scores", B, N, 1, 1, 300, 300, 0.7, torch.rand(B, N, 1, device="cuda", generator=g), boxes)
    run("rpn 8768 -> 8 (select + sort + one chunk)", B, N, 1, 1, 8, 8, 0.7, torch.rand(B, N, 1, device="cuda", generator=g), boxes)
    run("rpn 8768 -> 300, scores 0.5 +- 0.01", B, N, 1, 1, 300, 300, 0.7, 0.5 + 0.01 * torch.randn(B, N, 1, device="cuda", generator=g), boxes)
    run("rpn 8768 -> 300, all scores equal", B, N, 1, 1, 300, 300, 0.7, torch.full((B, N, 1), 0.5, device="cuda"), boxes)
    run("rpn 22464 -> 300 (eval), uniform", B, 22464, 1, 1, 300, 300, 0.7, torch.rand(B, 22464, 1, device="cuda", generator=g),
        torch.cat([torch.rand(B, 22464, 1, 2, device="cuda", generator=g) * 0.7, torch.rand(B, 22464, 1, 2, device="cuda", generator=g) * 0.3 + 0.7], -1).contiguous())
    N2, C = 300, 7
    ctr = torch.rand(B, N2, C, 2, device="cuda", generator=g)
    sz = torch.rand(B, N2, C, 2, device="cuda", generator=g) * 0.3 + 0.02
    b2 = torch.cat([ctr - sz / 2, ctr + sz / 2], -1).contiguous()
    run("detections 300 x 7 classes, uniform scores", B, N2, C, C, 300, 300, 0.5, torch.rand(B, N2, C, device="cuda", generator=g), b2)
    run("detections 300 x 7 classes -> 8 per class", B, N2, C, C, 8, 56, 0.5, torch.rand(B, N2, C, device="cuda", generator=g), b2)
    run("detections 300 x 7 classes, softmax-like 1/8", B, N2, C, C, 300, 300, 0.5, 0.125 + 0.01 * torch.randn(B, N2, C, device="cuda", generator=g), b2)


if __name__ == "__main__":
    main()
