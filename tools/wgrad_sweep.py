"""Sweep tile / split configurations of the weight-gradient kernel over the layer shapes of the R50-C4 step
(kernel development aid).  usage: python tools/wgrad_sweep.py [iters]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16

SHAPES = [  # n, h, w, cin, cout, k, stride, pad
    (4, 94, 311, 64, 256, 1, 1, 0), (4, 94, 311, 64, 64, 1, 1, 0), (4, 94, 311, 64, 64, 3, 1, 1), (4, 94, 311, 256, 64, 1, 1, 0),
    (4, 94, 311, 256, 512, 1, 2, 0), (4, 94, 311, 256, 128, 1, 2, 0),
    (4, 47, 156, 128, 128, 3, 1, 1), (4, 47, 156, 128, 512, 1, 1, 0), (4, 47, 156, 512, 128, 1, 1, 0),
    (4, 47, 156, 512, 1024, 1, 2, 0), (4, 47, 156, 512, 256, 1, 2, 0),
    (4, 24, 78, 256, 256, 3, 1, 1), (4, 24, 78, 256, 1024, 1, 1, 0), (4, 24, 78, 1024, 256, 1, 1, 0), (4, 24, 78, 1024, 256, 3, 1, 1),
    (4, 24, 78, 256, 128, 1, 1, 0),
]


FPN_SHAPES = [  # the feature pyramid's layers at batch 8 (BASELINE.json configs[4]): laterals, output / RPN 3x3 convolutions per level
    (8, 94, 311, 256, 256, 3, 1, 1), (8, 47, 156, 256, 256, 3, 1, 1), (8, 24, 78, 256, 256, 3, 1, 1), (8, 12, 39, 256, 256, 3, 1, 1),
    (8, 94, 311, 256, 256, 1, 1, 0), (8, 47, 156, 512, 256, 1, 1, 0), (8, 24, 78, 1024, 256, 1, 1, 0),
]


F8 = "fp8" in sys.argv


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    global SHAPES
    if len(sys.argv) > 2 and sys.argv[2] == "fpn":
        SHAPES = FPN_SHAPES
    g = torch.Generator(device="cuda").manual_seed(0)
    for (n, h, w, cin, cout, k, s, p) in SHAPES:
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        m = n * ho * wo
        x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
        dz = torch.randn(m, cout, device="cuda", generator=g).to(BF)
        if F8:                                      # fp8 twins (e4m3 / e5m2 bytes) and unit scales
            x, dz = x.to(torch.float8_e4m3fn).view(torch.uint8), dz.to(torch.float8_e5m2).view(torch.uint8)
            one = torch.ones(1, device="cuda")
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        d = ops.conv_desc(n, h, w, cin, k, k, s, p, p, ho, wo, cout)
        run = (lambda d_, x_, z_, w_: ops.conv2d_wgrad_fp8(d_, x_, z_, one, one, w_)) if F8 else ops.conv2d_wgrad
        res = []
        for bm in (128, 64):
            for bn in (128, 64):
                if (bm == 128 and cout < 128) or (bn == 128 and cin < 128):
                    continue
                if F8 and (cin % 64 or cout % 64):
                    continue
                for st in (2, 3):
                    for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64):
                        tiles = ((cout + bm - 1) // bm) * ((cin + bn - 1) // bn) * k * k
                        if tiles * sp < 128 or tiles * sp > 4096 or sp > (m + 63) // 64:
                            continue
                        os.environ["FRCNN_WGRAD"] = "%d,%d,%d,%d" % (bm, bn, st, sp)
                        try:
                            for _ in range(2):
                                run(d, x, dz, dw)
                            torch.cuda.synchronize()
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(iters):
                                run(d, x, dz, dw)
                            e1.record()
                            torch.cuda.synchronize()
                            res.append((e0.elapsed_time(e1) * 1e3 / iters, (bm, bn, st, sp)))
                        except Exception:  # noqa: BLE001
                            pass
        res.sort()
        fl = 2.0 * m * cout * k * k * cin
        print("M=%6d cin=%5d cout=%5d k=%d s=%d : " % (m, cin, cout, k, s) + "  ".join("%s %.1fus(%.0fTF)" % (",".join(map(str, c)), t, fl / t / 1e6) for t, c in res[:6]), flush=True)
    os.environ.pop("FRCNN_WGRAD", None)


if __name__ == "__main__":
    main()
