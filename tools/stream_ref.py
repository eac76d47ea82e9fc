"""What does the chip give a plain streaming kernel on the BatchNorm kernels' tensor shapes?  torch's own elementwise kernels (copy, a + b)
as the reference, cold (384 MB written in between) and warm.  Kernel-development aid: python tools/stream_ref.py"""
import torch

BF = torch.bfloat16
SHAPES = [("conv2 C=256", 4 * 94 * 311, 256), ("conv2 C=64", 4 * 94 * 311, 64), ("conv3 C=512", 4 * 47 * 156, 512), ("conv4 C=1024", 4 * 24 * 78, 1024),
          ("conv4 C=256", 4 * 24 * 78, 256)]


def timed(fn, flush):
    ts = []
    for _ in range(9):
        if flush is not None:
            flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[4]


def main():
    flush = torch.zeros(96 * 1024 * 1024, device="cuda")
    for name, m, c in SHAPES:
        a = torch.randn(m, c, device="cuda").to(BF)
        b = torch.randn(m, c, device="cuda").to(BF)
        o = torch.empty_like(a)
        mb = m * c * 2 / 1e6
        for label, fn, units in (("copy", lambda: o.copy_(a), 2), ("a + b", lambda: torch.add(a, b, out=o), 3), ("relu(a*s+b)", lambda: torch.relu_(torch.addcmul(b, a, a, out=o)), 3)):
            cold, warm = timed(fn, flush), timed(fn, None)
            print("%-14s %-12s %6.1f MB moved: cold %6.1f us (%.2f TB/s)   warm %6.1f us (%.2f TB/s)" % (
                name, label, mb * units, cold, mb * units / cold / 1e6 * 1e6 / 1e6, warm, mb * units / warm / 1e6 * 1e6 / 1e6), flush=True)


if __name__ == "__main__":
    main()
