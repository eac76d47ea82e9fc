"""conv3x3_patch_kernel (the input patch of a spatial tile resident in LDS) against the tile kernel, layer by layer, on the 3x3 shapes of
the train step at 375x1242 batch 4.  Timed as hipGraph replays of 10 launches (no host launch floor), warm.  Needs the sweep library:
    FRCNN_SWEEP=1 python 2d_object_detection_amd/csrc/build.py;  FRCNN_LIB=lib2dod_hip_sweep.so python tools/patch_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
ALL = [("c4 3x3 256->256 stats", 4, 24, 78, 256, 256, "stats"), ("c4 dgrad 3x3 256->256 red", 4, 24, 78, 256, 256, "red"),
          ("rpn 3x3 1024->256 relu", 4, 24, 78, 1024, 256, "plain"), ("rpn dgrad 3x3 256->1024", 4, 24, 78, 256, 1024, "none"),
          ("c3 3x3 128->128 stats", 4, 47, 156, 128, 128, "stats"), ("c3 dgrad 3x3 128->128 red", 4, 47, 156, 128, 128, "red"),
          ("c2 3x3 64->64 stats", 4, 94, 311, 64, 64, "stats"), ("c2 dgrad 3x3 64->64 red", 4, 94, 311, 64, 64, "red"),
          ("r101 c4 3x3 b2 stats", 2, 24, 78, 256, 256, "stats"), ("r101 c3 3x3 b2 stats", 2, 47, 156, 128, 128, "stats"),
          ("b8 c3 3x3 stats", 8, 47, 156, 128, 128, "stats"), ("b8 c3 dgrad red", 8, 47, 156, 128, 128, "red"), ("b8 fpn p4 256->256", 8, 24, 78, 256, 256, "plain"),
          ("b8 c4 3x3 stats", 8, 24, 78, 256, 256, "stats")]
LAYERS = [l for l in ALL if not os.environ.get("PATCH_LAYERS") or any(t in l[0] for t in os.environ["PATCH_LAYERS"].split(","))]
VARIANTS = [v for v in os.environ.get("PATCH_VARIANTS", "0 1:4 1:3").split()]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, n, h, w, cin, cout, mode in LAYERS:
        m = n * h * w
        x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
        wt = (torch.randn(cout, 3, 3, cin, device="cuda", generator=g) / (cin * 9) ** 0.5).to(BF)
        bias = torch.randn(cout, device="cuda", generator=g)
        flags = (ops.CONV_BIAS | ops.CONV_STATS) if mode == "stats" else (ops.CONV_BIAS | ops.CONV_RELU) if mode == "plain" else 0
        z = torch.randn(m, cout, device="cuda", generator=g).to(BF)
        mask = torch.randint(0, 256, (m, cout // 8), device="cuda", generator=g, dtype=torch.uint8)
        mean, invstd = torch.randn(cout, device="cuda", generator=g), torch.rand(cout, device="cuda", generator=g) + 0.5
        out, ys = [], []
        for var in VARIANTS:
            on, sb, lw, bn = (var.split(":") + ["", "", ""])[:4]
            os.environ["FRCNN_PATCH_BN"] = bn or "64"        # (128: 128 output channels per workgroup)
            os.environ["FRCNN_PATCH_LW"] = lw or "0"         # (4: dedicated loader waves)
            os.environ["FRCNN_PATCH"] = on
            os.environ["FRCNN_WRES"] = on                    # (64-channel layers: the weights-resident form)
            os.environ["FRCNN_PATCH_SB"] = sb or "4"
            d = ops.conv_desc(n, h, w, cin, 3, 3, 1, 1, 1, h, w, cout, flags=flags)
            ws = ops.conv_attach_workspace(d, "cuda")
            y = torch.zeros(m, cout, dtype=BF, device="cuda")
            stats = torch.zeros(16, 2, cout, dtype=torch.float64, device="cuda")
            part = torch.zeros(16, 2, cout, device="cuda")
            red = ops.bn_reduce_args(z, mask, mean, invstd, part)

            def launch():
                if mode == "red":
                    ops.conv2d_dgrad_bnreduce(d, x, wt, y, red)
                else:
                    ops.conv2d_fprop(d, x, wt, y, bias=bias if flags & ops.CONV_BIAS else None, stats=stats if mode == "stats" else None)

            launch()
            torch.cuda.synchronize()
            inst = ops.last_conv_instantiation()
            ys.append((y.float().clone(), stats.sum(0).clone(), part.sum(0).clone()))
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(10):
                        launch()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 100.0)
            out.append("%s %5.1f us [%s]" % (var, sorted(ts)[len(ts) // 2], inst.split(" grid")[0][:34]))
            del graph
        y0, s0, p0 = ys[0]
        diffs = []
        for y1, s1, p1 in ys[1:]:
            diffs.append("dy %.2e ds %.1e dp %.1e" % (float((y1 - y0).abs().max() / (y0.abs().max() + 1e-9)),
                                                      float((s1 - s0).abs().max() / (s0.abs().max() + 1e-9)), float((p1 - p0).abs().max() / (p0.abs().max() + 1e-9))))
        print("%-28s %s | %s" % (name, "  ".join(out), "; ".join(diffs)), flush=True)


if __name__ == "__main__":
    main()
