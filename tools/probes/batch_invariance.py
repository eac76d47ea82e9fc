"""Is a convolution's output for image 0 the same bits whether the launch holds one image or two?  (The two-rank == one-process tests
rely on it: the accumulation order of an output element must not depend on the batch.)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("2d_object_detection_amd.ops")
BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
for (h, w, cin, cout, k) in [(16, 24, 128, 128, 3), (8, 12, 256, 256, 3), (32, 48, 64, 64, 3), (16, 24, 512, 128, 1), (8, 12, 1024, 256, 3), (8, 12, 1024, 256, 1), (32, 48, 64, 256, 1)]:
    x2 = torch.randn(2, h, w, cin, device="cuda", generator=g).to(BF)
    wt = (torch.randn(cout, k, k, cin, device="cuda", generator=g) / (cin * k * k) ** 0.5).to(BF)
    bias = torch.randn(cout, device="cuda", generator=g)
    res = []
    for n in (1, 2):
        d = ops.conv_desc(n, h, w, cin, k, k, 1, k // 2, k // 2, h, w, cout, flags=ops.CONV_BIAS | ops.CONV_STATS)
        y = torch.zeros(n * h * w, cout, dtype=BF, device="cuda")
        st = torch.zeros(16, 2, cout, dtype=torch.float64, device="cuda")
        ops.conv2d_fprop(d, x2[:n].contiguous(), wt, y, bias=bias, stats=st)
        torch.cuda.synchronize()
        res.append((y[:h * w].clone(), ops.last_conv_instantiation().split(" grid")[0]))
    same = torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16))
    print(h, w, cin, cout, k, "same bits:", same, "|", res[0][1], "|", res[1][1], flush=True)
