"""Does a buffer that was just WRITTEN by one kernel come out of the 256 MB memory-side cache when the next kernel reads it?
(kernel-development probe: sizes the head GEMM's 60 us in the step against its 30 us back to back, tools/head_gemm_bench.py)
Times the Dense-head GEMM ([1200 x 50176] x [64 x 50176]^T, 120 MB of A): back to back (hot), right after a kernel wrote A (+ 60 MB next
to it, as the RoI kernel does), after 1 GiB of unrelated fill, after 1 GiB of unrelated reads."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
m, k, n = 1200, 50176, 64
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(m, k, device="cuda", generator=g).to(BF)
x2 = x.clone()
w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).to(BF)
y = torch.zeros(m, n, device="cuda")
side = torch.empty(60 * 1024 * 1024, dtype=torch.uint8, device="cuda")
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
big.fill_(1)
d = ops.conv_desc(1, 1, m, k, 1, 1, 1, 0, 0, 1, m, n, flags=ops.CONV_SPLITK_ATOMIC, split_k=16)


def timed():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d_fprop(d, x, w, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def med(prep, n=9):
    v = []
    for _ in range(n):
        prep()
        v.append(timed())
    return sorted(v)[n // 2]


print("GEMM, hot (run just before)            : %6.1f us" % med(lambda: ops.conv2d_fprop(d, x, w, y)))
print("GEMM, right after A was written (copy) : %6.1f us" % med(lambda: (side.fill_(3), x.copy_(x2))))
print("GEMM, right after A was written (fill) : %6.1f us" % med(lambda: (side.fill_(3), x.fill_(0.5))))
print("GEMM, after a 1 GiB fill (cold)        : %6.1f us" % med(lambda: big.fill_(1)))
print("GEMM, after a 1 GiB read (cold)        : %6.1f us" % med(lambda: big.sum()))

# the step's own sequence: RoI crop + pool writes A (pooled) and the arg-max bytes, the GEMM follows
B, P, Hf, Wf, C = 4, 300, 24, 78, 1024
gg = torch.Generator().manual_seed(0)
feat = torch.randn(B, Hf, Wf, C, generator=gg).to(BF).cuda()
ctr = torch.rand(B, P, 2, generator=gg)
wh = torch.rand(B, P, 2, generator=gg) * 0.2 + 0.05
rois = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).clamp(0, 1).cuda()
am = torch.empty(B * P, 49 * C, dtype=torch.uint8, device="cuda")
print("GEMM, right after the RoI kernel       : %6.1f us" % med(lambda: ops.roi_crop_pool_fwd(feat, rois, B, P, Hf, Wf, C, 7, 2, x, am)))
print("GEMM, 1 GiB fill, RoI kernel, GEMM     : %6.1f us" % med(lambda: (big.fill_(1), ops.roi_crop_pool_fwd(feat, rois, B, P, Hf, Wf, C, 7, 2, x, am))))
for split in (8, 16, 32, 49):
    d = ops.conv_desc(1, 1, m, k, 1, 1, 1, 0, 0, 1, m, n, flags=ops.CONV_SPLITK_ATOMIC, split_k=split)
    print("  split %2d, 1 GiB fill, RoI, GEMM      : %6.1f us" % (split, med(lambda: (big.fill_(1), ops.roi_crop_pool_fwd(feat, rois, B, P, Hf, Wf, C, 7, 2, x, am)))))
