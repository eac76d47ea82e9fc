"""Driver of tools/kloop_bench.hip (K-loop structure experiment): correctness against torch, then time per launch.
build (container):  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -std=c++17 -o tools/libkloop_bench.so tools/kloop_bench.hip
run (GPU box):      python tools/kloop_bench.py"""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libkloop_bench.so"))
lib.kloop_bench.restype = ctypes.c_int
lib.kloop_bench.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
VARIANTS = [(0, "sync  S=3"), (4, "sync  S=2"), (1, "roles S=3 D=1"), (2, "roles S=4 D=2"), (3, "roles S=5 D=3")]
SHAPES = [(7488, 256, 1024), (7488, 256, 2304), (7488, 256, 9216), (29328, 128, 512), (29328, 128, 1152), (116936, 64, 576)]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    flush = torch.zeros(96 * 1024 * 1024, device="cuda")
    err = torch.zeros(4, dtype=torch.int32, device="cuda")
    for (M, N, K) in SHAPES:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        ref = A[:2048].float() @ B.float().t()
        line = []
        for v, name in VARIANTS:
            C = torch.full((M, N), float("nan"), device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            rc = lib.kloop_bench(v, A.data_ptr(), B.data_ptr(), C.data_ptr(), err.data_ptr(), M, N, K, st)
            torch.cuda.synchronize()
            ok = rc == 0 and int(err[0]) == 0 and bool(torch.isfinite(C).all()) and float((C[:2048] - ref).abs().max()) < 2e-2 * float(ref.abs().max())
            ts = {}
            for cold in (True, False):
                t = []
                for _ in range(7):
                    if cold:
                        flush.add_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    lib.kloop_bench(v, A.data_ptr(), B.data_ptr(), C.data_ptr(), err.data_ptr(), M, N, K, st)
                    e1.record()
                    torch.cuda.synchronize()
                    t.append(e0.elapsed_time(e1) * 1e3)
                ts[cold] = sorted(t)[len(t) // 2]
            tiles = ((M + 127) // 128) * ((N + 63) // 64)
            fill_gb = tiles * (K // 64) * 24576 / 1e9
            line.append("%s: %s cold %6.1f warm %6.1f us (%4.0f TF/s, fill %5.1f GB/s/CU)" % (
                name, "ok " if ok else "BAD rc=%d err=%d" % (rc, int(err[0])), ts[True], ts[False], 2.0 * M * N * K / ts[False] / 1e6,
                fill_gb / (ts[False] * 1e-6) / min(256, tiles)))
            err.zero_()
        print("M=%d N=%d K=%d (%d tiles)\n   " % (M, N, K, tiles) + "\n   ".join(line), flush=True)


if __name__ == "__main__":
    main()
