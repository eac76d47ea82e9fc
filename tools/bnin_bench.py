"""Micro-benchmark of the BatchNorm-carrying convolutions against their parts (kernel development aid): for each shape the time of
bn_train_apply, of the plain convolution on its output, and of conv2d_fprop_bnin on the raw input -- graph replays of 20 launches.
usage: [FRCNN_LIB=lib2dod_hip_<tag>.so] python tools/bnin_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
SHAPES = [("c2 1x1 64->256", 4, 94, 311, 64, 256, 1), ("c3 1x1 128->512", 4, 47, 156, 128, 512, 1), ("c4 1x1 256->1024", 4, 24, 78, 256, 1024, 1),
          ("c2 3x3 64->64", 4, 94, 311, 64, 64, 3), ("c3 3x3 128->128", 4, 47, 156, 128, 128, 3), ("c4 3x3 256->256", 4, 24, 78, 256, 256, 3)]


def timed(fn, iters=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    gen = torch.Generator().manual_seed(0)
    print("library:", os.environ.get("FRCNN_LIB", "lib2dod_hip.so"))
    for name, n, h, w, cin, cout, k in SHAPES:
        m = n * h * w
        z = (torch.randn(m, cin, generator=gen) * 1.3 + 0.2).to(BF).cuda()
        zf = z.double()
        zstats = torch.zeros(16, 2, cin, dtype=torch.float64, device="cuda")
        for s_ in range(16):
            rows = slice(s_ * m // 16, (s_ + 1) * m // 16)
            zstats[s_, 0], zstats[s_, 1] = zf[rows].sum(0), (zf[rows] * zf[rows]).sum(0)
        gamma, beta = (torch.rand(cin, generator=gen) + 0.5).cuda(), (torch.randn(cin, generator=gen) * 0.2).cuda()
        wt = (torch.randn(cout, k, k, cin, generator=gen) / (8.0 * k)).to(BF).cuda()
        bias = torch.randn(cout, generator=gen).cuda()
        d = ops.conv_desc(n, h, w, cin, k, k, 1, k // 2, k // 2, h, w, cout, flags=ops.CONV_BIAS | ops.CONV_STATS)
        act, mask = torch.zeros(m, cin, dtype=BF, device="cuda"), torch.zeros(m, cin // 8, dtype=torch.uint8, device="cuda")
        mean, invstd, mm, mv = (torch.zeros(cin, device="cuda") for _ in range(4))
        y, ystats = torch.zeros(m, cout, dtype=BF, device="cuda"), torch.zeros(16, 2, cout, dtype=torch.float64, device="cuda")
        t_bn = timed(lambda: ops.bn_train_apply(z, zstats, 16, m, gamma, beta, mm, mv, 0.99, 1.001e-5, act, mean, invstd, m, cin, relu=True, relu_mask=mask))
        t_conv = timed(lambda: ops.conv2d_fprop(d, act, wt, y, bias=bias, stats=ystats))
        plain = ops.last_conv_instantiation().split(" grid")[0]
        t_both = timed(lambda: (ops.bn_train_apply(z, zstats, 16, m, gamma, beta, mm, mv, 0.99, 1.001e-5, act, mean, invstd, m, cin, relu=True, relu_mask=mask),
                                ops.conv2d_fprop(d, act, wt, y, bias=bias, stats=ystats)))
        line = "%-18s bn %5.1f us + conv %5.1f us (back to back %5.1f)" % (name, t_bn, t_conv, t_both)
        if ops.conv2d_bnin_supported(d):
            bn = ops.bn_in_args(zstats, gamma, beta, mm, mv, 0.99, 1.001e-5, m, act, mask, mean, invstd)
            t_f = timed(lambda: ops.conv2d_fprop_bnin(d, z, wt, y, bn, bias=bias, stats=ystats))
            line += " | fused %5.1f us  (%+.1f)  %s" % (t_f, t_f - t_both, ops.last_conv_instantiation().split(" grid")[0])
        else:
            line += " | " + plain
        print(line, flush=True)


if __name__ == "__main__":
    main()
