"""RoI crop + pool forward alone at the benchmark's shapes: kernel-development aid.
usage (GPU box):  FRCNN_LIB=lib2dod_hip_sweep.so python tools/roi_bench.py       (FRCNN_ROI_FWD_OLD=1: the round-3 per-item kernel)
Box populations: `rpn` -- what the proposal NMS hands over early in training (anchor-sized boxes, 32..512 px); `small` -- 2-4 feature
cells wide (every bin's samples share their taps); `large` -- most of the image (no sharing)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16


def boxes(kind, B, P, g):
    ctr = torch.rand(B, P, 2, generator=g)
    if kind == "small":
        wh = torch.rand(B, P, 2, generator=g) * 0.03 + 0.03
    elif kind == "large":
        wh = torch.rand(B, P, 2, generator=g) * 0.4 + 0.5
    else:
        side = torch.tensor([32.0, 64, 128, 256, 512])[torch.randint(0, 5, (B, P), generator=g)]
        ratio = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (B, P), generator=g)]
        wh = torch.stack([side * ratio.sqrt() / 1242, side / ratio.sqrt() / 375], -1)
    b = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).clamp(0, 1)
    return b


def main():
    g = torch.Generator().manual_seed(0)
    for (B, P, Hf, Wf, C) in ((4, 300, 24, 78, 1024), (2, 1000, 24, 78, 1024), (8, 300, 94, 311, 256)):
        feat = torch.randn(B, Hf, Wf, C, generator=g).to(BF).cuda()
        pooled = torch.empty(B * P, 49 * C, dtype=BF, device="cuda")
        am = torch.empty(B * P, 49 * C, dtype=torch.uint8, device="cuda")
        for kind in ("rpn", "small", "large"):
            rois = boxes(kind, B, P, g).cuda()
            for old in ("1", "0"):
                os.environ["FRCNN_ROI_FWD_OLD"] = old
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        ops.roi_crop_pool_fwd(feat, rois, B, P, Hf, Wf, C, 7, 2, pooled, am)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 5)
                ref = pooled.clone() if old == "1" else None
                if old == "1":
                    keep = ref
                else:
                    same = torch.equal(keep.view(torch.int16), pooled.view(torch.int16))
                print("B=%d P=%d %dx%dx%d %-5s %s kernel: %7.1f us%s" % (B, P, Hf, Wf, C, kind, "round-3" if old == "1" else "round-4", sorted(ts)[3],
                                                                        "" if old == "1" else "   (bit-identical: %s)" % same), flush=True)


if __name__ == "__main__":
    main()
