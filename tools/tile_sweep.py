"""Sweep the tile configurations of the one-tile conv kernel over the layer shapes of the R50-C4 step
(kernel development aid).  Needs a sweep build of the library: FRCNN_SWEEP=1 python 2d_object_detection_amd/csrc/build.py --force
(production builds carry only the instantiations the dispatcher selects and read no environment variables).
usage: python tools/tile_sweep.py [iters]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16

SHAPES = [  # n, h, w, cin, cout, k, stride, pad
    (4, 188, 621, 32, 64, 0, 0, 0),       # placeholder for the stem (skipped: packed row-gather geometry)
    (4, 94, 311, 64, 256, 1, 1, 0), (4, 94, 311, 64, 64, 1, 1, 0), (4, 94, 311, 64, 64, 3, 1, 1), (4, 94, 311, 256, 64, 1, 1, 0),
    (4, 94, 311, 256, 512, 1, 2, 0), (4, 94, 311, 256, 128, 1, 2, 0),
    (4, 47, 156, 128, 128, 3, 1, 1), (4, 47, 156, 128, 512, 1, 1, 0), (4, 47, 156, 512, 128, 1, 1, 0), (4, 47, 156, 128, 256, 1, 1, 0),
    (4, 47, 156, 512, 256, 1, 1, 0), (4, 47, 156, 512, 1024, 1, 2, 0), (4, 47, 156, 512, 256, 1, 2, 0),
    (4, 24, 78, 256, 256, 3, 1, 1), (4, 24, 78, 256, 1024, 1, 1, 0), (4, 24, 78, 1024, 256, 1, 1, 0), (4, 24, 78, 1024, 256, 3, 1, 1),
    (4, 24, 78, 256, 1024, 3, 1, 1), (4, 24, 78, 256, 512, 1, 1, 0), (4, 24, 78, 1024, 512, 1, 1, 0), (4, 24, 78, 256, 128, 1, 1, 0),
    (4, 24, 78, 128, 256, 1, 1, 0), (1, 16, 16, 64, 50176, 1, 1, 0),
]
CONFIGS = [(bm, bn, bk, s, 1) for bk in (64, 128) for bm in (128, 64) for bn in (128, 64) for s in (2, 3) if not (bk == 128 and s == 3)]
CONFIGS += [(128, 64, 64, 4, 1), (128, 64, 64, 6, 1), (128, 128, 64, 4, 1), (64, 64, 64, 6, 1)]               # deep rings
CONFIGS += [(256, 128, 64, 2, 1), (256, 64, 64, 2, 1)]                                                         # 256-row tiles
CONFIGS += [(bm, bn, 64, 2, t) for (bm, bn) in ((128, 64), (128, 128), (64, 64)) for t in (2, 4, 8, 16)]      # tile runs


COLD = "--cold" in sys.argv
F8 = "--fp8" in sys.argv                     # e4m3 operands (frcnn_conv2d_fprop_fp8), no statistics
FPN = "--fpn" in sys.argv                    # the feature pyramid's 3x3 layers at batch 8
FPN_SHAPES = [None, (8, 94, 311, 256, 256, 3, 1, 1), (8, 47, 156, 256, 256, 3, 1, 1), (8, 24, 78, 256, 256, 3, 1, 1)]
FLUSH = None


def main():
    global FLUSH
    iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
    if COLD:
        FLUSH = torch.zeros(160 * 1024 * 1024, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    only = [a for a in sys.argv[1:] if a.startswith("--m=")]
    batch = [int(a[8:]) for a in sys.argv[1:] if a.startswith("--batch=")]          # (--batch=2: ResNet-101's configuration, BASELINE configs[3])
    for (n, h, w, cin, cout, k, s, p) in (FPN_SHAPES if FPN else SHAPES)[1:]:
        n = batch[0] if batch and n == 4 else n
        if only and str(n * ((h + 2 * p - k) // s + 1) * ((w + 2 * p - k) // s + 1)) != only[0][4:]:
            continue
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        m = n * ho * wo
        x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
        wt = (torch.randn(cout, k, k, cin, device="cuda", generator=g) / (cin * k * k) ** 0.5).to(BF)
        bias = torch.zeros(cout, device="cuda")
        y = torch.empty(m, cout, dtype=BF, device="cuda")
        stats = torch.zeros(16, 2, cout, dtype=torch.float64, device="cuda")
        d = ops.conv_desc(n, h, w, cin, k, k, s, p, p, ho, wo, cout, flags=ops.CONV_BIAS | (0 if F8 else ops.CONV_STATS))
        if F8:
            if cin % 128:
                continue
            x8, w8 = x.to(torch.float8_e4m3fn).view(torch.uint8), wt.to(torch.float8_e4m3fn).view(torch.uint8)
            one, ones = torch.ones(1, device="cuda"), torch.ones(cout, device="cuda")
            run = lambda: ops.conv2d_fprop_fp8(d, x8, w8, one, ones, y, bias=bias)
        else:
            run = lambda: ops.conv2d_fprop(d, x, wt, y, bias=bias, stats=stats)
        res = []
        for cfg in [None, "kws"] + CONFIGS:
            if cfg is None or cfg == "kws":                    # the dispatcher's own choice / kw sharing forced on
                os.environ.pop("FRCNN_TILE", None)
                os.environ.pop("FRCNN_KWS", None)
                if cfg == "kws":
                    os.environ["FRCNN_KWS"] = "1"
                try:
                    for _ in range(2):
                        run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    res.append((e0.elapsed_time(e1) * 1e3 / iters, ("default",) if cfg is None else ("kws",)))
                except Exception:  # noqa: BLE001
                    pass
                os.environ.pop("FRCNN_KWS", None)
                continue
            if cin % cfg[2] != 0 or (cfg[1] == 128 and cout < 128):
                continue
            os.environ["FRCNN_TILE"] = ",".join(str(v) for v in cfg)
            try:
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                if COLD:
                    # one call at a time behind a 640 MB write that evicts L2 and the Infinity Cache: in the training step
                    # every layer finds its operands in HBM, not in the caches a back-to-back loop keeps warm
                    tot = 0.0
                    for _ in range(iters):
                        FLUSH.add_(1.0)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        run()
                        e1.record()
                        torch.cuda.synchronize()
                        tot += e0.elapsed_time(e1) * 1e3
                    res.append((tot / iters, cfg))
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res.append((e0.elapsed_time(e1) * 1e3 / iters, cfg))
            except Exception as ex:  # noqa: BLE001
                res.append((1e9, cfg))
        res.sort(key=lambda e: e[0])
        fl = 2.0 * m * cout * k * k * cin
        print("M=%6d cin=%5d cout=%5d k=%d s=%d : " % (m, cin, cout, k, s) + "  ".join("%s %.1fus(%.0fTF)" % (",".join(map(str, c)), t, fl / t / 1e6) for t, c in res[:int(os.environ.get("SWEEP_TOP", "5"))]), flush=True)
    os.environ.pop("FRCNN_TILE", None)


if __name__ == "__main__":
    main()
