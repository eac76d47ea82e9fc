"""Achieved HBM bandwidth of the BatchNorm apply kernels on the R50-C4 layer shapes (kernel development aid).
usage: python tools/bn_bench.py   -- prints us and GB/s, L2/MALL-warm and behind a cache flush"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
SHAPES = [(466992, 64), (116936, 64), (116936, 256), (29328, 128), (29328, 512), (7488, 256), (7488, 1024)]


def timed(fn, flush, iters=10):
    out = []
    for cold in (False, True):
        ts = []
        for _ in range(iters):
            if cold:
                flush.add_(1.0)
            else:
                torch.cuda._sleep(100000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out.append(sorted(ts)[len(ts) // 2])
    return out


def main():
    flush = torch.zeros(160 * 1024 * 1024, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ov = []
    for _ in range(50):
        e0.record(); e1.record(); torch.cuda.synchronize(); ov.append(e0.elapsed_time(e1) * 1e3)
    ov = sorted(ov)[25]
    print("event pair overhead %.1f us (subtracted)" % ov)
    for (m, c) in SHAPES:
        z = torch.randn(m, c, device="cuda").to(BF)
        res = torch.randn(m, c, device="cuda").to(BF)
        out = torch.empty_like(z)
        mask = torch.empty(m * c // 8, dtype=torch.uint8, device="cuda")
        stats = torch.zeros(ops.STAT_SLOTS, 2, c, dtype=torch.float64, device="cuda")
        stats[0, 0] = z.float().sum(0).double()
        stats[0, 1] = (z.float() ** 2).sum(0).double()
        gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        mm, mv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        mean, invstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        g = torch.randn(m, c, device="cuda").to(BF)
        dz = torch.empty_like(z)
        partial = torch.zeros(ops.STAT_SLOTS, 2, c, device="cuda")
        dgamma, dbeta = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        for name, fn, byts in (
            ("fwd", lambda: ops.bn_train_apply(z, stats, ops.STAT_SLOTS, m, gamma, beta, mm, mv, 0.99, 1e-5, out, mean, invstd, m, c, relu=True, relu_mask=mask), m * c * 4.125),
            ("fwd+res", lambda: ops.bn_train_apply(z, stats, ops.STAT_SLOTS, m, gamma, beta, mm, mv, 0.99, 1e-5, out, mean, invstd, m, c, res=res, relu=True, relu_mask=mask), m * c * 6.125),
            ("bwd_reduce", lambda: ops.bn_bwd_reduce(g, None, z, mean, invstd, partial, m, c, relu_mask=mask), m * c * 4.125),
            ("bwd_apply", lambda: ops.bn_bwd_apply_fused(g, None, z, mean, invstd, gamma, partial, ops.STAT_SLOTS, dgamma, dbeta, dz, None, m, c, relu_mask=mask), m * c * 6.125),
        ):
            fn()
            torch.cuda.synchronize()
            w, cd = timed(fn, flush)
            w, cd = max(w - ov, 0.1), max(cd - ov, 0.1)
            print("M=%6d C=%4d %-10s %6.1f MB  warm %6.1f us %5.0f GB/s   cold %6.1f us %5.0f GB/s" % (m, c, name, byts / 1e6, w, byts / w / 1e3, cd, byts / cd / 1e3), flush=True)


if __name__ == "__main__":
    main()
