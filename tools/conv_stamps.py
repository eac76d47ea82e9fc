"""Where a conv workgroup spends its cycles: per-workgroup phase stamps (s_memtime) of conv_tile_kernel on the R50-C4 layer
shapes of the benchmark step (kernel-development aid, cdna_hip_programming.md section 7 'In-kernel stamps').

Needs the stamps variant of the library:   FRCNN_STAMPS=1 python 2d_object_detection_amd/csrc/build.py
usage (GPU box):                           FRCNN_LIB=lib2dod_hip_stamps.so python tools/conv_stamps.py

Per layer: the kernel's span, workgroups, average resident workgroups per CU, the shader clock, and median / p90 of the phases
  setup   entry -> first K slice issued (index math, descriptors, prologue DMA issue)
  kloop   K loop of the (last) tile incl. the wait for the first slice
  epi     convert + staging + stores issued (+ fused reduce loads)
  tail    statistics / reduce flush, exit
"""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRCNN_LIB", "lib2dod_hip_stamps.so")
ops = importlib.import_module("2d_object_detection_amd.ops")
_lib = importlib.import_module("2d_object_detection_amd._lib")
BF = torch.bfloat16

# name, n, h, w, cin, cout, k, stride, mode ("stats": forward with BN statistics, "red": data gradient with fused reduce [+ residual])
LAYERS = [
    ("c2 3x3 64->64 fwd (KWS)", 4, 94, 311, 64, 64, 3, 1, "stats"),
    ("c2 1x1 64->256 fwd (run4)", 4, 94, 311, 64, 256, 1, 1, "stats"),
    ("c2 1x1 64->256 dgrad+red+res (run4)", 4, 94, 311, 64, 256, 1, 1, "redres"),
    ("c2 1x1 256->64 fwd", 4, 94, 311, 256, 64, 1, 1, "stats"),
    ("c2 1x1 256->64 dgrad+red", 4, 94, 311, 256, 64, 1, 1, "red"),
    ("c3 3x3 128->128 fwd (KWS)", 4, 47, 156, 128, 128, 3, 1, "stats"),
    ("c3 1x1 128->512 fwd (run2)", 4, 47, 156, 128, 512, 1, 1, "stats"),
    ("c3 1x1 512->128 fwd (128x128)", 4, 47, 156, 512, 128, 1, 1, "stats"),
    ("c4 1x1 256->1024 fwd (128x128)", 4, 24, 78, 256, 1024, 1, 1, "stats"),
    ("c4 1x1 1024->256 fwd (S=3)", 4, 24, 78, 1024, 256, 1, 1, "stats"),
    ("c4 3x3 256->256 fwd", 4, 24, 78, 256, 256, 3, 1, "stats"),
    ("rpn 3x3 1024->256 fwd", 4, 24, 78, 1024, 256, 3, 1, "plain"),
    ("rpn 3x3 256->1024 dgrad+red+res", 4, 24, 78, 256, 1024, 3, 1, "redres"),
]


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


RAW = {}


def main():
    lib = _lib.load()
    lib.frcnn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.frcnn_debug_set_stamp_buffer.restype = None
    g = torch.Generator(device="cuda").manual_seed(0)
    flush = torch.zeros(96 * 1024 * 1024, device="cuda")                 # 384 MB: evicts L2 + Infinity Cache between launches
    for (name, n, h, w, cin, cout, k, s, mode) in LAYERS:
        p = k // 2
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        m = n * ho * wo
        x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
        wt = (torch.randn(cout, k, k, cin, device="cuda", generator=g) / (cin * k * k) ** 0.5).to(BF)
        bias = torch.zeros(cout, device="cuda")
        y = torch.empty(m, cout, dtype=BF, device="cuda")
        stats = torch.zeros(16, 2, cout, dtype=torch.float64, device="cuda")
        flags = (ops.CONV_BIAS | ops.CONV_STATS) if mode == "stats" else (ops.CONV_BIAS | ops.CONV_RELU) if mode == "plain" else \
            (ops.CONV_ADD_RES if mode == "redres" else 0)
        d = ops.conv_desc(n, h, w, cin, k, k, s, p, p, ho, wo, cout, flags=flags)
        z = torch.randn(m, cout, device="cuda", generator=g).to(BF)
        mask = torch.randint(0, 256, (m, cout // 8), device="cuda", generator=g, dtype=torch.uint8)
        mean, invstd = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        part = torch.zeros(16, 2, cout, device="cuda")
        res = torch.randn(m, cout, device="cuda", generator=g).to(BF) if mode == "redres" else None
        red = ops.bn_reduce_args(z, mask, mean, invstd, part)

        def launch():
            if mode in ("red", "redres"):
                ops.conv2d_dgrad_bnreduce(d, x, wt, y, red, res=res, res_mask=mask if mode == "redres" and k == 1 else None)
            else:
                ops.conv2d_fprop(d, x, wt, y, bias=bias, stats=stats if mode == "stats" else None)

        launch()
        torch.cuda.synchronize()
        inst = ops.last_conv_instantiation()
        grid = int(inst.split("grid=")[1].split("x")[0])
        for cold in (True, False):
            dbg = torch.zeros(grid * 24, dtype=torch.int64, device="cuda")
            if cold:
                flush.add_(1.0)
            lib.frcnn_debug_set_stamp_buffer(ctypes.c_void_p(dbg.data_ptr()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            torch.cuda.synchronize()
            lib.frcnn_debug_set_stamp_buffer(None)
            t = dbg.view(grid, 24).cpu()
            RAW["%s|%s" % (name, "cold" if cold else "warm")] = t.clone()
            if os.environ.get("STAMPS_DEBUG"):
                print(t[:3].tolist())
            base = t[:, 0].min()                                             # (64-bit counters: subtract in integers first)
            t0, t1, t2, t3, t4 = ((t[:, i] - base).double() for i in range(5))
            rt = (t[:, 6] - t[:, 6].min()).double()
            xcc = t[:, 7] & 0xF
            span_ticks = max(float(t4[xcc == x].max() - t0[xcc == x].min()) for x in xcc.unique().tolist())   # (per XCD: own counter)
            # shader clock from ONE XCD's workgroups (s_memtime counters of different XCDs are offset against each other): ticks
            # between its first and last workgroup start over the same interval in 100 MHz realtime ticks
            x0 = (t[:, 7] & 0xF) == (t[0, 7] & 0xF)
            span_rt = float(rt[x0].max() - rt[x0].min())
            start_span = float(t[x0, 0].max() - t[x0, 0].min())
            clock_ghz = start_span / span_rt * 0.1 if span_rt > 0 else float("nan")
            if not (1.0 < clock_ghz < 3.0):
                clock_ghz = 2.1
            us = lambda ticks: ticks / (clock_ghz * 1e3) if clock_ghz == clock_ghz else float("nan")
            hw = t[:, 5]
            cu = ((t[:, 7] & 0xF) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
            ncu = int(torch.unique(cu).numel())
            resident = float((t4 - t0).sum()) / (span_ticks * max(ncu, 1))
            ph = {"setup": t1 - t0, "kloop": t2 - t1, "epi": t3 - t2, "tail": t4 - t3, "total": t4 - t0}
            line = "  ".join("%s %.2f/%.2f" % (k_, us(pct(v.tolist(), 0.5)), us(pct(v.tolist(), 0.9))) for k_, v in ph.items())
            fl = 2.0 * m * cout * k * k * cin
            print("%-38s %-5s event %.1f us | span %.1f us, %d WGs on %d CUs, %.2f resident/CU, clock %.2f GHz | med/p90 us: %s | %s" % (
                name, "cold" if cold else "warm", e0.elapsed_time(e1) * 1e3, us(span_ticks), grid, ncu, resident, clock_ghz, line,
                inst.split(">")[0].replace("conv_tile<", "") if cold else "%.0f TF/s" % (fl / (e0.elapsed_time(e1) * 1e-3) / 1e12)), flush=True)
            # K-loop slices 4 and 5 of wave 0 (shader-clock cycles; s_memtime ticks at 100 MHz x 1 on this part? -> reported as ticks
            # converted with the same clock): wait for the slice's DMA | barrier | DMA issue | fragment reads + MFMA issue | whole step
            ks = t[:, 8:18]
            ok = (ks > 0).all(1)
            if int(ok.sum()) > 0:
                kz = ks[ok].double()
                parts = []
                for sl in range(2):
                    b = kz[:, sl * 5:sl * 5 + 5]
                    dl = [b[:, j + 1] - b[:, j] for j in range(4)] + [b[:, 4] - b[:, 0]]
                    parts.append("slice %d: " % (4 + sl) + " ".join("%s %.0f/%.0f" % (nm, us(pct(v.tolist(), 0.5)) * 1e3, us(pct(v.tolist(), 0.9)) * 1e3)
                                                                    for nm, v in zip(("wait", "barrier", "issue", "mfma", "step"), dl)))
                gap = kz[:, 5] - kz[:, 4]
                print("      ns med/p90, %d WGs | %s | %s | between %.0f" % (int(ok.sum()), parts[0], parts[1], us(pct(gap.tolist(), 0.5)) * 1e3), flush=True)


if __name__ == "__main__":
    main()
    if os.path.isdir("gpurun_out"):
        torch.save(RAW, "gpurun_out/stamps_raw.pt")
