"""BatchNorm forward (bn_train_apply) alone, on the backbone's shapes: where do its microseconds go?  Kernel-development aid.
Needs the sweep library for the FRCNN_BN_VAR experiments:  FRCNN_SWEEP=1 python 2d_object_detection_amd/csrc/build.py
usage (GPU box):  FRCNN_LIB=lib2dod_hip_sweep.so python tools/bn_bench.py
  FRCNN_BN_VAR  0 production | 4 round-1 form (2-D placement, cached loads)
Each shape: cold (384 MB flush before) and warm-z (z rewritten by a copy just before, as the conv kernel leaves it)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
SHAPES = [("conv2 C=256 +res", 4 * 94 * 311, 256, True), ("conv2 C=256", 4 * 94 * 311, 256, False), ("conv2 C=64", 4 * 94 * 311, 64, False),
          ("conv3 C=512 +res", 4 * 47 * 156, 512, True), ("conv3 C=128", 4 * 47 * 156, 128, False),
          ("conv4 C=1024 +res", 4 * 24 * 78, 1024, True), ("conv4 C=256", 4 * 24 * 78, 256, False)]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    flush = torch.zeros(96 * 1024 * 1024, device="cuda")
    for var in os.environ.get("BN_VARS", "0 4").split():
        os.environ["FRCNN_BN_VAR"] = var
        for name, m, c, with_res in SHAPES:
            z = torch.randn(m, c, device="cuda", generator=g).to(BF)
            zsrc = z.clone()
            res = torch.randn(m, c, device="cuda", generator=g).to(BF) if with_res else None
            stats = torch.zeros(16, 2, c, dtype=torch.float64, device="cuda")
            stats[0, 0] = z.double().sum(0)
            stats[0, 1] = (z.double() ** 2).sum(0)
            gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
            mm, mv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
            out = torch.empty_like(z)
            mask = torch.empty(m, c // 8, dtype=torch.uint8, device="cuda")
            mean, invstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
            f8 = None
            if os.environ.get("BN_F8"):                      # with the fp8 twin of the output (frcnn_fp8_out)
                out8 = torch.empty(m, c, dtype=torch.uint8, device="cuda")
                qs, amax = torch.ones(1, device="cuda"), torch.zeros(ops.FP8_AMAX_SLOTS, device="cuda")
                f8 = ops.fp8_out(out8, qs, amax if os.environ["BN_F8"] != "noamax" else None)
            res_t = {}
            for mode in ("cold", "warm-z"):
                ts = []
                for _ in range(5):
                    if mode == "cold":
                        flush.add_(1.0)
                    else:
                        flush.add_(1.0)
                        z.copy_(zsrc)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.bn_train_apply(z, stats, 16, m, gamma, beta, mm, mv, 0.99, 1.001e-5, out, mean, invstd, m, c, res=res, relu=True,
                                       relu_mask=mask, f8=f8)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                res_t[mode] = sorted(ts)[len(ts) // 2]
            mb = (m * c * 2 * (3 if with_res else 2) + m * c // 8 + (m * c if f8 is not None else 0)) / 1e6
            print("var %s  %-18s %6.1f MB  cold %6.1f us (%.2f TB/s)   warm-z %6.1f us (%.2f TB/s)" % (
                var, name, mb, res_t["cold"], mb / res_t["cold"], res_t["warm-z"], mb / res_t["warm-z"]), flush=True)


if __name__ == "__main__":
    main()
