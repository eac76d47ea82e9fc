"""All-taps 3x3 weight-gradient kernel (wgrad3_body) against the per-tap kernel, layer by layer (hipGraph replays of 5 launches, warm).
    FRCNN_SWEEP=1 python 2d_object_detection_amd/csrc/build.py;  FRCNN_LIB=lib2dod_hip_sweep.so python tools/wgrad3_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
LAYERS = [("c4 256->256", 4, 24, 78, 256, 256), ("c3 128->128", 4, 47, 156, 128, 128), ("c2 64->64", 4, 94, 311, 64, 64), ("rpn 1024->256", 4, 24, 78, 1024, 256)]
VARIANTS = os.environ.get("W3_VARIANTS", "0 1").split()


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, n, h, w, cin, cout in LAYERS:
        m = n * h * w
        x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
        dz = torch.randn(m, cout, device="cuda", generator=g).to(BF)
        out, ref = [], None
        for var in VARIANTS:
            on, model, dev = (var.split(":") + ["", ""])[:3]
            os.environ["FRCNN_WGRAD3"] = on
            os.environ["FRCNN_WGRAD3_MODEL"] = model or "1.1,6,1.3e6"
            os.environ["FRCNN_W3_DEV"] = dev or "0"
            d = ops.conv_desc(n, h, w, cin, 3, 3, 1, 1, 1, h, w, cout)
            dw = torch.zeros(cout, 3, 3, cin, device="cuda")
            ops.conv2d_wgrad(d, x, dz, dw)
            torch.cuda.synchronize()
            inst = ops.last_conv_instantiation()
            if ref is None:
                ref = dw.clone()
            err = float((dw - ref).abs().max() / ref.abs().max())
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(5):
                        ops.conv2d_wgrad(d, x, dz, dw)
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 200.0)
            out.append("%s %6.1f us [%s] err %.1e" % (var, sorted(ts)[3], inst[:40], err))
            del graph
        print("%-16s %s" % (name, "  ".join(out)), flush=True)


if __name__ == "__main__":
    main()
