"""Split-K fix-up form of the conv kernel (conv_tile.hip, FIX) against the one-workgroup-per-tile form, layer by layer, on the
M = 7,488 shapes.  Needs the sweep library (FRCNN_FIX=0 disables the form at dispatch):
    FRCNN_SWEEP=1 python 2d_object_detection_amd/csrc/build.py;  FRCNN_LIB=lib2dod_hip_sweep.so python tools/fix_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
LAYERS = [("c4 1x1 1024->256 stats", 1024, 256, 1, "stats"), ("c4 3x3 256->256 stats", 256, 256, 3, "stats"),
          ("rpn 3x3 1024->256 relu", 1024, 256, 3, "plain"), ("c4 dgrad 1x1 1024->256 red", 1024, 256, 1, "red"),
          ("c4 dgrad 3x3 256->256 red", 256, 256, 3, "red"), ("c4 1x1 2048->512 stats", 2048, 512, 1, "stats")]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    n, h, w = 4, 24, 78
    m = n * h * w
    flush = torch.zeros(96 * 1024 * 1024, device="cuda")
    for name, cin, cout, k, mode in LAYERS:
        x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
        wt = (torch.randn(cout, k, k, cin, device="cuda", generator=g) / (cin * k * k) ** 0.5).to(BF)
        bias = torch.zeros(cout, device="cuda")
        y = torch.empty(m, cout, dtype=BF, device="cuda")
        stats = torch.zeros(16, 2, cout, dtype=torch.float64, device="cuda")
        flags = (ops.CONV_BIAS | ops.CONV_STATS) if mode == "stats" else (ops.CONV_BIAS | ops.CONV_RELU) if mode == "plain" else 0
        z = torch.randn(m, cout, device="cuda", generator=g).to(BF)
        mask = torch.randint(0, 256, (m, cout // 8), device="cuda", generator=g, dtype=torch.uint8)
        mean, invstd = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        part = torch.zeros(16, 2, cout, device="cuda")
        red = ops.bn_reduce_args(z, mask, mean, invstd, part)
        out = []
        for fix in ("0", "1"):
            os.environ["FRCNN_FIX"] = fix
            d = ops.conv_desc(n, h, w, cin, k, k, 1, k // 2, k // 2, h, w, cout, flags=flags)
            ws = ops.conv_attach_workspace(d, "cuda")

            def launch():
                if mode == "red":
                    ops.conv2d_dgrad_bnreduce(d, x, wt, y, red)
                else:
                    ops.conv2d_fprop(d, x, wt, y, bias=bias, stats=stats if mode == "stats" else None)

            launch()
            torch.cuda.synchronize()
            inst = ops.last_conv_instantiation()
            res = {}
            for cold in (True, False):
                ts = []
                for _ in range(7):
                    if cold:
                        flush.add_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                res[cold] = sorted(ts)[len(ts) // 2]
            out.append("FIX=%s cold %5.1f warm %5.1f us (%s, ws %s)" % (fix, res[True], res[False], "FIX" if "FIX=1" in inst else "plain",
                                                                     "yes" if ws is not None else "no"))
        print("%-30s %s" % (name, "  |  ".join(out)), flush=True)


if __name__ == "__main__":
    main()
