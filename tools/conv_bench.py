"""Micro-benchmark of one implicit-GEMM conv shape through the C ABI (kernel development aid).
usage: python tools/conv_bench.py N H W CIN COUT K STRIDE PAD [fprop|wgrad] [iters]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16


def main():
    n, h, w, cin, cout, k, s, p = (int(x) for x in sys.argv[1:9])
    mode = sys.argv[9] if len(sys.argv) > 9 else "fprop"
    iters = int(sys.argv[10]) if len(sys.argv) > 10 else 20
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
    wt = (torch.randn(cout, k, k, cin, device="cuda", generator=g) / (cin * k * k) ** 0.5).to(BF)
    bias = torch.zeros(cout, device="cuda")
    m = n * ho * wo
    flops = 2.0 * m * cout * k * k * cin
    if mode == "fprop":
        d = ops.conv_desc(n, h, w, cin, k, k, s, p, p, ho, wo, cout, flags=ops.CONV_BIAS | ops.CONV_STATS)
        y = torch.empty(m, cout, dtype=BF, device="cuda")
        stats = torch.zeros(16, 2, cout, dtype=torch.float64, device="cuda")
        fn = lambda: ops.conv2d_fprop(d, x, wt, y, bias=bias, stats=stats)
    else:
        d = ops.conv_desc(n, h, w, cin, k, k, s, p, p, ho, wo, cout)
        dz = torch.randn(m, cout, device="cuda", generator=g).to(BF)
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        fn = lambda: ops.conv2d_wgrad(d, x, dz, dw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%s M=%d cin=%d cout=%d k=%d s=%d: %.1f us  %.1f TFLOP/s" % (mode, m, cin, cout, k, s, us, flops / us / 1e6))


if __name__ == "__main__":
    main()
