"""GPU parity (through the C ABI) of the non-GEMM kernels against the CPU oracle.
Integer / index / discrete outputs: bit-exact.  Float outputs: tolerance stated per test."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import conv_cases
from oracle import boxes as oboxes
from oracle import faster_rcnn as O
from oracle import losses as olosses
from oracle import nms as onms
from oracle import roi as oroi
from oracle import training as otraining
from oracle.philox import rand_u32

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rt(t):
    return t.to(BF).to(torch.float32)


def _close(a, b, rtol, atol, what):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bad = int((err > atol + rtol * b.abs()).sum())
    assert bad == 0, "%s: %d/%d mismatches, max err %g" % (what, bad, a.numel(), float(err.max()))


# ------------------------------------------------------------------ BN / pool / SGD
def test_bn_forward_backward(ops):
    g = torch.Generator().manual_seed(0)
    m, c = 1000, 64
    z = _rt(torch.randn(m, c, generator=g) * 2 + 0.5)
    res = _rt(torch.randn(m, c, generator=g))
    gamma = 1 + 0.1 * torch.randn(c, generator=g)
    beta = 0.1 * torch.randn(c, generator=g)
    zz = z.clone().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    out_ref = F.relu(F.batch_norm(zz, rm, rv, gm, bt, training=True, momentum=0.01, eps=1.001e-5) + res)
    gout = _rt(torch.randn(m, c, generator=g))
    out_ref.backward(gout)

    dev = "cuda"
    zd = z.to(BF).to(dev)
    # stats partials as the conv epilogue would write them: 4 row blocks
    parts = torch.zeros(4, 2, c, dtype=torch.float64, device=dev)
    for i, blk in enumerate(z.chunk(4)):
        parts[i, 0] = blk.sum(0).to(dev)
        parts[i, 1] = (blk * blk).sum(0).to(dev)
    mm, mv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    scale, shift, mean, invstd = (torch.empty(c, device=dev) for _ in range(4))
    ops.bn_finalize_train(parts, 4, c, m, gamma.to(dev), beta.to(dev), mm, mv, 0.99, 1.001e-5, scale, shift, mean, invstd)
    out = torch.empty(m, c, dtype=BF, device=dev)
    ops.bn_apply(zd, scale, shift, out, m, c, res=res.to(BF).to(dev), relu=True)
    torch.cuda.synchronize()
    _close(mm, rm, 1e-4, 1e-5, "moving mean")
    _close(mv, rv, 1e-4, 1e-5, "moving variance")
    _close(out, out_ref.detach(), 2 ** -7, 1e-2, "bn apply")
    # backward (mask from the ORACLE's activation so both sides use the same ReLU mask)
    act = out_ref.detach().to(BF).to(dev)
    nb = ops.bn_bwd_blocks(m)
    partial = torch.zeros(nb, 2, c, device=dev)          # slots are accumulated with atomics: pre-zeroed
    gd = gout.to(BF).to(dev)
    ops.bn_bwd_reduce(gd, act, zd, mean, invstd, partial, m, c)
    dgamma, dbeta, c1, c2 = (torch.empty(c, device=dev) for _ in range(4))
    ops.bn_bwd_finalize(partial, nb, c, m, dgamma, dbeta, c1, c2)
    dz = torch.empty(m, c, dtype=BF, device=dev)
    gpre = torch.empty(m, c, dtype=BF, device=dev)
    ops.bn_bwd_apply(gd, act, zd, mean, invstd, gamma.to(dev), c1, c2, dz, gpre, m, c)
    torch.cuda.synchronize()
    _close(dgamma, gm.grad, 1e-3, 1e-2, "dgamma")
    _close(dbeta, bt.grad, 1e-3, 1e-2, "dbeta")
    _close(dz, zz.grad, 2 ** -6, 2e-3, "dz")
    _close(gpre, gout * (out_ref.detach() > 0), 0, 0, "gpre")
    # fused forms (one launch each): same arithmetic, every workgroup reduces the partial sums of its own 64 channels
    mm2, mv2 = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    mean2, invstd2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    out2 = torch.empty(m, c, dtype=BF, device=dev)
    ops.bn_train_apply(zd, parts, 4, m, gamma.to(dev), beta.to(dev), mm2, mv2, 0.99, 1.001e-5, out2, mean2, invstd2, m, c,
                       res=res.to(BF).to(dev), relu=True)
    dgamma2, dbeta2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    dz2, gpre2 = torch.empty(m, c, dtype=BF, device=dev), torch.empty(m, c, dtype=BF, device=dev)
    ops.bn_bwd_apply_fused(gd, act, zd, mean, invstd, gamma.to(dev), partial, nb, dgamma2, dbeta2, dz2, gpre2, m, c)
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and torch.equal(mean2, mean) and torch.equal(invstd2, invstd), "fused train apply"
    assert torch.equal(mm2, mm) and torch.equal(mv2, mv), "fused moving statistics"
    assert torch.equal(gpre2, gpre) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta), "fused bwd apply"
    # dz of the fused kernel is rounded with first-order error feedback along each thread's rows: the column sums of dz (zero in
    # exact arithmetic) stay exact to an ulp per chain instead of collecting the correlated rounding errors of a bf16-valued
    # gradient; an element is off by at most half an ulp of itself plus half an ulp of its predecessor in the chain
    col_max = zz.grad.abs().max(0).values
    bound = 2.0 ** -8 * (zz.grad.abs() + col_max) * 1.05 + 2e-3
    assert bool(((dz2.float().cpu() - zz.grad).abs() <= bound).all()), "dz (fused): beyond the error-feedback rounding bound"
    s_fb, s_rne = dz2.double().sum(0).abs().max(), dz.double().sum(0).abs().max()
    assert float(s_fb) <= float(s_rne) + 1e-6, (float(s_fb), float(s_rne))
    # ReLU bit mask: written by the forward kernel, read by the backward kernels instead of the activation tensor
    rmask = torch.zeros(m, c // 8, dtype=torch.uint8, device=dev)
    out3 = torch.empty(m, c, dtype=BF, device=dev)
    ops.bn_train_apply(zd, parts, 4, m, gamma.to(dev), beta.to(dev), mm2.clone(), mv2.clone(), 0.99, 1.001e-5, out3, mean2, invstd2, m, c,
                       res=res.to(BF).to(dev), relu=True, relu_mask=rmask)
    torch.cuda.synchronize()
    bits = ((rmask.cpu()[:, :, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(m, c).bool()
    assert torch.equal(out3, out2) and torch.equal(bits, out3.float().cpu() > 0), "relu bit mask"
    pa, pb = torch.zeros(nb, 2, c, device=dev), torch.zeros(nb, 2, c, device=dev)
    ops.bn_bwd_reduce(gd, out3, zd, mean, invstd, pa, m, c)
    ops.bn_bwd_reduce(gd, None, zd, mean, invstd, pb, m, c, relu_mask=rmask)
    res_a = [torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(m, c, dtype=BF, device=dev), torch.empty(m, c, dtype=BF, device=dev)]
    res_b = [torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(m, c, dtype=BF, device=dev), torch.empty(m, c, dtype=BF, device=dev)]
    ops.bn_bwd_apply_fused(gd, out3, zd, mean, invstd, gamma.to(dev), pa, nb, res_a[0], res_a[1], res_a[2], res_a[3], m, c)
    ops.bn_bwd_apply_fused(gd, None, zd, mean, invstd, gamma.to(dev), pb, nb, res_b[0], res_b[1], res_b[2], res_b[3], m, c, relu_mask=rmask)
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and all(torch.equal(x, y) for x, y in zip(res_a, res_b)), "bit mask == activation mask"


@pytest.mark.parametrize("m,c", [(7488, 1024), (29328, 512), (5000, 256), (203, 64)])
def test_bn_bwd_apply_red2_equals_apply_plus_reduce(ops, m, c):
    """frcnn_bn_bwd_apply_fused_red2 == frcnn_bn_bwd_apply_fused (ReLU bit mask form) + frcnn_bn_bwd_reduce of a second BatchNorm on the same
    masked gradient (the shortcut BatchNorm of a stage's first block: conv<N>_block1_0_bn beside _3_bn): dz, dgamma, dbeta bit for bit; the
    second layer's slot partials up to the order of their float atomics (the workgroups and their reduction trees are the same: with one
    row chunk per strip -- the small case -- bit for bit).  Shapes: conv4 / conv3 at the benchmark's batch, ragged row counts."""
    g = torch.Generator().manual_seed(m + c)
    dev = "cuda"
    nb = ops.bn_bwd_blocks(m)
    gout, z, z2 = (torch.randn(m, c, generator=g).to(BF).to(dev) for _ in range(3))
    rmask = torch.randint(0, 256, (m, c // 8), generator=g, dtype=torch.uint8).to(dev)
    mean, mean2 = torch.randn(c, generator=g).to(dev) * 0.1, torch.randn(c, generator=g).to(dev) * 0.1
    invstd, invstd2 = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.rand(c, generator=g) + 0.5).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
    part = torch.zeros(nb, 2, c, device=dev)
    ops.bn_bwd_reduce(gout, None, z, mean, invstd, part, m, c, relu_mask=rmask)

    def outs():
        return dict(dg=torch.empty(c, device=dev), db=torch.empty(c, device=dev), dz=torch.empty(m, c, dtype=BF, device=dev),
                    p2=torch.full((nb, 2, c), 0.25, device=dev))
    a, b = outs(), outs()
    ops.bn_bwd_apply_fused(gout, None, z, mean, invstd, gamma, part, nb, a["dg"], a["db"], a["dz"], None, m, c, relu_mask=rmask)
    ops.bn_bwd_reduce(gout, None, z2, mean2, invstd2, a["p2"], m, c, relu_mask=rmask)
    red2 = ops.bn_reduce_args(z2, None, mean2, invstd2, b["p2"])
    ops.bn_bwd_apply_fused_red2(gout, z, mean, invstd, gamma, part, nb, b["dg"], b["db"], b["dz"], m, c, rmask, red2)
    torch.cuda.synchronize()
    assert torch.equal(a["dz"].view(torch.int16), b["dz"].view(torch.int16)) and torch.equal(a["dg"], b["dg"]) and torch.equal(a["db"], b["db"])
    sa, sb = a["p2"].double().sum(0), b["p2"].double().sum(0)
    assert float(sa.abs().max()) > 1.0 and float(((sa - sb).abs() / (sa.abs() + 1.0)).max()) < 1e-4, "second layer's backward sums"      # (fp32 slot atomics: a few ulp of sums of magnitude 10 - 300)
    if m <= 300:
        assert torch.equal(a["p2"], b["p2"])
    with pytest.raises(RuntimeError):
        ops.bn_bwd_apply_fused_red2(gout[:, :72].contiguous(), z[:, :72].contiguous(), mean[:72], invstd[:72], gamma[:72], part[:, :, :72].contiguous(), nb,
                                    b["dg"][:72], b["db"][:72], b["dz"][:, :72].contiguous(), m, 72, rmask[:, :9].contiguous(), red2)       # c % 64 != 0


def test_bn_wide_channels_and_eval(ops):
    g = torch.Generator().manual_seed(1)
    m, c = 300, 1024
    z = _rt(torch.randn(m, c, generator=g))
    gout = _rt(torch.randn(m, c, generator=g))
    mean, invstd = torch.randn(c, generator=g) * 0.1, 1 + 0.1 * torch.rand(c, generator=g)
    dev = "cuda"
    nb = ops.bn_bwd_blocks(m)
    partial = torch.zeros(nb, 2, c, device=dev)
    ops.bn_bwd_reduce(gout.to(BF).to(dev), None, z.to(BF).to(dev), mean.to(dev), invstd.to(dev), partial, m, c)
    colsum = torch.full((c,), 1.0, device=dev)                           # colsum ADDS into its output
    ops.colsum_bf16(gout.to(BF).to(dev), m, c, c, colsum)
    torch.cuda.synchronize()
    _close(colsum, gout.sum(0) + 1.0, 1e-4, 1e-3, "colsum")
    xh = (z - mean) * invstd
    _close(partial[:, 0].sum(0), gout.sum(0), 1e-4, 1e-3, "sum g")
    _close(partial[:, 1].sum(0), (gout * xh).sum(0), 1e-4, 1e-3, "sum g*xhat")
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    scale, shift = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ops.bn_finalize_eval(c, gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), 1.001e-5, scale, shift)
    out = torch.empty(m, c, dtype=BF, device=dev)
    ops.bn_apply(z.to(BF).to(dev), scale, shift, out, m, c, relu=False)
    torch.cuda.synchronize()
    _close(out, F.batch_norm(z, rm, rv, gamma, beta, training=False, eps=1.001e-5), 2 ** -7, 1e-2, "bn eval")


def test_maxpool(ops):
    g = torch.Generator().manual_seed(2)
    n, h, w, c = 2, 19, 23, 64
    x = F.relu(_rt(torch.randn(n, h, w, c, generator=g)))
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.max_pool2d(F.pad(xr, (1, 1, 1, 1)), 3, 2)
    ho, wo = yr.shape[2], yr.shape[3]
    gy = _rt(torch.randn(n, ho, wo, c, generator=g))
    yr.backward(gy.permute(0, 3, 1, 2))
    dev = "cuda"
    y = torch.empty(n, ho, wo, c, dtype=BF, device=dev)
    am = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=dev)
    ops.maxpool_fwd(x.to(BF).to(dev), y, am, n, h, w, c, ho, wo)
    gx = torch.empty(n, h, w, c, dtype=BF, device=dev)
    ops.maxpool_bwd(gy.to(BF).to(dev), am, gx, n, h, w, c, ho, wo)
    torch.cuda.synchronize()
    _close(y, yr.detach().permute(0, 2, 3, 1), 0, 0, "maxpool fwd")
    # gradient routing can differ only where a window has tied maxima (zeros after ReLU); compare where x > 0
    ref = xr.grad.permute(0, 2, 3, 1)
    mask = x > 0
    _close(gx.float().cpu()[mask], ref[mask], 2 ** -7, 1e-2, "maxpool bwd")


@pytest.mark.parametrize("n,h,w", [(2, 19, 23), (4, 188, 621), (1, 5, 7)])
def test_maxpool_bwd_bnreduce_equals_two_launches(ops, n, h, w):
    """frcnn_maxpool3x3s2_bwd_bnreduce == frcnn_maxpool3x3s2_bwd (the activation gradient: bit for bit) + frcnn_bn_bwd_reduce on it
    (the stem BatchNorm's masked backward sums: equal up to the order of the float additions; (4, 188, 621) is the benchmark's stem)."""
    g = torch.Generator().manual_seed(n * 100 + h)
    c, dev = 64, "cuda"
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    z = torch.randn(n * h * w, c, generator=g).to(BF).to(dev)
    act = torch.relu(z.float()).to(BF)
    mask_bits = (act.float() > 0).view(n * h * w, c // 8, 8).to(torch.int32)
    relu_mask = (mask_bits * (2 ** torch.arange(8, device=dev, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous()
    y = torch.empty(n, ho, wo, c, dtype=BF, device=dev)
    am = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=dev)
    ops.maxpool_fwd(act.view(n, h, w, c), y, am, n, h, w, c, ho, wo)
    gy = torch.randn(n, ho, wo, c, generator=g).to(BF).to(dev)
    mean, invstd = (torch.randn(c, generator=g) * 0.1).to(dev), (torch.rand(c, generator=g) + 0.5).to(dev)
    gx_a, gx_b = torch.empty(n, h, w, c, dtype=BF, device=dev), torch.empty(n, h, w, c, dtype=BF, device=dev)
    part_a, part_b = torch.zeros(16, 2, c, device=dev), torch.zeros(16, 2, c, device=dev)
    ops.maxpool_bwd(gy, am, gx_a, n, h, w, c, ho, wo)
    ops.bn_bwd_reduce(gx_a, None, z, mean, invstd, part_a, n * h * w, c, relu_mask=relu_mask)
    red = ops.bn_reduce_args(z, relu_mask, mean, invstd, part_b)
    ops.maxpool_bwd_bnreduce(gy, am, gx_b, n, h, w, c, ho, wo, red)
    torch.cuda.synchronize()
    assert torch.equal(gx_a.view(torch.int16), gx_b.view(torch.int16)), "activation gradient"
    sa, sb = part_a.sum(0).cpu(), part_b.sum(0).cpu()
    assert float(sa.abs().max()) > 0
    scale = float(gx_a.float().view(-1, c).abs().sum(0).max().cpu())       # the sums are compared relative to the sum of magnitudes they cancel from
    assert float((sa - sb).abs().max()) <= 2e-6 * scale + 1e-6, (float((sa - sb).abs().max()), scale)
    m = (act.float() > 0).float()
    xhat = (z.float() - mean) * invstd
    ref = torch.stack([(gx_a.float().view(-1, c) * m).double().sum(0), (gx_a.float().view(-1, c) * m * xhat).double().sum(0)]).float().cpu()
    assert float((sb - ref).abs().max()) <= 1e-5 * scale * 4 + 1e-5, float((sb - ref).abs().max())


@pytest.mark.parametrize("n,h,w,c", [(2, 21, 30, 64), (1, 8, 9, 64), (3, 13, 16, 128), (1, 7, 5, 8)])
def test_stem_bn_relu_maxpool_fused(ops, n, h, w, c):
    """frcnn_bn_train_apply_maxpool == bn_train_apply(ReLU, bit mask) + maxpool_fwd bit for bit: pooled values, arg-max bytes, ReLU
    bit mask, mean / invstd / moving statistics -- without the activation tensor."""
    g = torch.Generator().manual_seed(31 + h)
    dev = "cuda"
    m = n * h * w
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    z = (torch.randn(m, c, generator=g) * 1.5 + 0.3).to(BF).to(dev)
    zf = z.double()
    stats = torch.zeros(16, 2, c, dtype=torch.float64, device=dev)
    stats[3, 0], stats[3, 1] = zf.sum(0), (zf * zf).sum(0)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.2).to(dev)

    def fresh():
        return torch.full((c,), 0.25, device=dev), torch.full((c,), 0.75, device=dev)

    mm_a, mv_a = fresh()
    act = torch.empty(m, c, dtype=BF, device=dev)
    mask_a = torch.empty(m, c // 8, dtype=torch.uint8, device=dev)
    mean_a, inv_a = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ops.bn_train_apply(z, stats, 16, m, gamma, beta, mm_a, mv_a, 0.99, 1.001e-5, act, mean_a, inv_a, m, c, relu=True, relu_mask=mask_a)
    pool_a = torch.empty(n * ho * wo, c, dtype=BF, device=dev)
    arg_a = torch.empty(n * ho * wo, c, dtype=torch.uint8, device=dev)
    ops.maxpool_fwd(act, pool_a, arg_a, n, h, w, c, ho, wo)
    mm_b, mv_b = fresh()
    pool_b = torch.full((n * ho * wo, c), 7.0, dtype=BF, device=dev)
    arg_b = torch.full((n * ho * wo, c), 77, dtype=torch.uint8, device=dev)
    mask_b = torch.full((m, c // 8), 0xA5, dtype=torch.uint8, device=dev)
    mean_b, inv_b = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ops.bn_train_apply_maxpool(z, stats, 16, m, gamma, beta, mm_b, mv_b, 0.99, 1.001e-5, pool_b, arg_b, mask_b, mean_b, inv_b, n, h, w, c, ho, wo)
    torch.cuda.synchronize()
    assert torch.equal(pool_a.view(torch.int16), pool_b.view(torch.int16)), "pooled values"
    assert torch.equal(arg_a, arg_b), "arg-max bytes"
    assert torch.equal(mask_a, mask_b), "ReLU bit mask (every pixel written once, by the cell that owns it)"
    for x, y in ((mean_a, mean_b), (inv_a, inv_b), (mm_a, mm_b), (mv_a, mv_b)):
        assert torch.equal(x, y)
    assert 0.2 < float((act > 0).float().mean()) < 0.9


@pytest.mark.parametrize("m,c,relu", [(1500, 256, True), (333, 64, True), (700, 1024, False), (64, 8, True)])
def test_bn_dual_equals_two_launches(ops, m, c, relu):
    """frcnn_bn_train_apply_dual == bn_train_apply(z2 -> tmp, no ReLU) followed by bn_train_apply(z, res = tmp): output, ReLU mask and
    both layers' mean / invstd / moving statistics bit for bit -- without tmp."""
    g = torch.Generator().manual_seed(51 + c)
    dev = "cuda"
    z1 = (torch.randn(m, c, generator=g) * 1.2 + 0.1).to(BF).to(dev)
    z2 = (torch.randn(m, c, generator=g) * 0.7 - 0.2).to(BF).to(dev)

    def stats_of(z):
        st = torch.zeros(16, 2, c, dtype=torch.float64, device=dev)
        st[5, 0], st[5, 1] = z.double().sum(0), (z.double() ** 2).sum(0)
        return st

    s1, s2 = stats_of(z1), stats_of(z2)
    prm = [(torch.rand(c, generator=g) + 0.5).to(dev) for _ in range(2)] + [(torch.randn(c, generator=g) * 0.2).to(dev) for _ in range(2)]
    ga1, ga2, be1, be2 = prm

    def state():
        return [torch.full((c,), v, device=dev) for v in (0.1, 0.9, 0.2, 0.8)] + [torch.empty(c, device=dev) for _ in range(4)]

    mm1a, mv1a, mm2a, mv2a, me1a, iv1a, me2a, iv2a = state()
    tmp = torch.empty(m, c, dtype=BF, device=dev)
    out_a = torch.empty(m, c, dtype=BF, device=dev)
    mask_a = torch.zeros(m, c // 8, dtype=torch.uint8, device=dev)
    ops.bn_train_apply(z2, s2, 16, m, ga2, be2, mm2a, mv2a, 0.99, 1.001e-5, tmp, me2a, iv2a, m, c, relu=False)
    ops.bn_train_apply(z1, s1, 16, m, ga1, be1, mm1a, mv1a, 0.99, 1.001e-5, out_a, me1a, iv1a, m, c, res=tmp, relu=relu,
                       relu_mask=mask_a if relu else None)
    mm1b, mv1b, mm2b, mv2b, me1b, iv1b, me2b, iv2b = state()
    out_b = torch.full((m, c), 3.0, dtype=BF, device=dev)
    mask_b = torch.zeros(m, c // 8, dtype=torch.uint8, device=dev)
    ops.bn_train_apply_dual(z1, s1, ga1, be1, mm1b, mv1b, me1b, iv1b, z2, s2, ga2, be2, mm2b, mv2b, me2b, iv2b, 16, m, 0.99, 1.001e-5,
                            out_b, m, c, relu=relu, relu_mask=mask_b if relu else None)
    torch.cuda.synchronize()
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16)), "output"
    assert torch.equal(mask_a, mask_b), "ReLU bit mask"
    for x, y in ((mm1a, mm1b), (mv1a, mv1b), (mm2a, mm2b), (mv2a, mv2b), (me1a, me1b), (iv1a, iv1b), (me2a, me2b), (iv2a, iv2b)):
        assert torch.equal(x, y)
    ref = torch.nn.functional.batch_norm(z1.float().cpu(), None, None, ga1.cpu(), be1.cpu(), training=True, eps=1.001e-5) + \
        torch.nn.functional.batch_norm(z2.float().cpu(), None, None, ga2.cpu(), be2.cpu(), training=True, eps=1.001e-5)
    _close(out_b, torch.relu(ref) if relu else ref, 2 ** -7, 2e-2, "dual BatchNorm against torch")


def test_sgd_and_lr_schedule(ops):
    g = torch.Generator().manual_seed(3)
    n = 10007
    w, gr, v = torch.randn(n, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    dev = "cuda"
    wd, gd, vd = w.to(dev), gr.to(dev), v.to(dev)
    wb = torch.empty(n, dtype=BF, device=dev)
    bounds = torch.tensor([40000, 80000], dtype=torch.int64, device=dev)
    values = torch.tensor([1e-3, 1e-4, 1e-5], device=dev)
    for step_val, lr in ((0, 1e-3), (39999, 1e-3), (40000, 1e-3), (40001, 1e-4), (80000, 1e-4), (80001, 1e-5)):    # Keras: values[i] while step <= boundaries[i]
        step = torch.tensor([step_val], dtype=torch.int64, device=dev)
        w0, v0 = wd.clone().cpu(), vd.clone().cpu()
        ops.sgd_momentum(wd, gd, vd, wb, n, 0.9, 0.0005, 1.0, step, bounds, values, 2)
        torch.cuda.synchronize()
        gp = gr + 2 * 0.0005 * w0
        v1 = 0.9 * v0 - np.float32(lr) * gp
        _close(vd, v1, 1e-5, 1e-7, "velocity @%d" % step_val)
        _close(wd, w0 + v1, 1e-5, 1e-7, "weights @%d" % step_val)
        _close(wb, (w0 + v1), 2 ** -8, 0, "bf16 copy")
    step = torch.tensor([5], dtype=torch.int64, device=dev)
    ops.step_increment(step)
    assert int(step.item()) == 6


# ------------------------------------------------------------------ boxes / NMS
def test_anchors_and_decode(ops):
    cfg = O.default_config()
    gh, gw = O.feature_grid(cfg["image_shape"])
    ref = O.generate_anchors((gh, gw), **cfg["rpn"]["anchors"])
    a = cfg["rpn"]["anchors"]
    out = torch.empty(gh * gw * 12, 4, device="cuda")
    ops.anchors_generate(out, gh, gw, a["scales"], a["aspect_ratios"], a["base_anchor_shape"][0], a["base_anchor_shape"][1])
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref), "anchors must be bit-exact"
    g = torch.Generator().manual_seed(4)
    B, R, C = 2, 500, 7
    regions = ref[1000:1000 + R]
    deltas = torch.randn(B, R, C, 4, generator=g) * 0.3
    exp = oboxes.to_relative(oboxes.decode(deltas, regions[None, :, None, :]), cfg["image_shape"])
    got = torch.empty(B, R, C, 4, device="cuda")
    ops.decode_boxes(regions.cuda(), deltas.cuda(), got, B, R, C, 1242, 375)
    clipped = torch.empty(R, 4, device="cuda")
    ops.clip_to_window(regions.cuda(), clipped, [0, 0, 1242, 375])
    torch.cuda.synchronize()
    _close(got, exp, 1e-5, 1e-6, "decode")
    assert torch.equal(clipped.cpu(), oboxes.clip_to_window(regions, [0, 0, 1242, 375]))


def _random_boxes(g, B, N, q):
    ctr = torch.rand(B, N, q, 2, generator=g)
    sz = torch.rand(B, N, q, 2, generator=g) * 0.3 + 0.02
    return torch.cat([ctr - sz / 2, ctr + sz / 2], -1)


@pytest.mark.parametrize("B,N,q,C,mpc,mt,thr", [
    (2, 300, 7, 7, 100, 300, 0.6),       # RCNN configuration
    (2, 8768, 1, 1, 300, 300, 0.7),      # RPN train configuration (LDS sort, 16384 pad)
    (1, 22464, 1, 1, 300, 300, 0.7),     # RPN eval configuration (global-scratch sort)
    (3, 1000, 1, 1, 50, 20, 0.3),        # max_total < kept, heavy suppression
    (1, 70, 3, 3, 100, 300, 0.5),        # fewer candidates than any cap
    (2, 8768, 1, 1, 1000, 1000, 0.7),    # BASELINE config 4: 1000-proposal stress (RPN)
    (2, 1000, 7, 7, 100, 300, 0.6),      # BASELINE config 4: RCNN NMS over 1000 proposals
    (3, 81929, 1, 1, 300, 300, 0.7),     # BASELINE config 4: the pyramid's in-image anchors (team of workgroups, candidates split over it)
    (2, 57000, 1, 1, 2000, 2000, 0.7),   # 600 x 1987 (the reference's config.json), more kept than the team round holds: rounds after it
    (2, 600, 1, 1, 300, 300, 0.7),       # single class, one workgroup's round: a team without a select
    (2, 700, 1, 1, 25, 7, 0.5),          # odd max_per_class (the LDS arrays behind the kept lists must stay 16-byte aligned), team
    (2, 300, 3, 3, 7, 10, 0.5),          # ... and without a team
])
def test_nms_combined_bit_exact(ops, B, N, q, C, mpc, mt, thr):
    g = torch.Generator().manual_seed(N + C)
    boxes = _random_boxes(g, B, N, q)
    # distinct scores (ties are implementation-defined in TF), with a few <= threshold
    perm = torch.stack([torch.randperm(N * C, generator=g) for _ in range(B)]).reshape(B, N, C)
    scores = (perm.float() + 1) / (N * C + 1)
    scores[:, ::17, :] = 0.0
    exp = onms.combined_nms(boxes, scores, mpc, mt, thr, 0.0)
    dev = "cuda"
    ob = torch.full((B, mt, 4), -1.0, device=dev)
    os_ = torch.full((B, mt), -1.0, device=dev)
    oc = torch.full((B, mt), -1, dtype=torch.int32, device=dev)
    ov = torch.full((B,), -1, dtype=torch.int32, device=dev)
    ws = torch.zeros(ops.nms_workspace_bytes(B, N, C, mpc, mt), dtype=torch.uint8, device=dev)
    ops.nms_combined(boxes.to(dev), scores.to(dev), B, N, q, C, C, 0, mpc, mt, thr, 0.0, ob, os_, oc, ov, ws)
    torch.cuda.synchronize()
    assert torch.equal(ov.cpu(), exp[3]), "num_valid_detections"
    assert torch.equal(os_.cpu(), exp[1]), "scores"
    assert torch.equal(oc.cpu(), exp[2]), "classes"
    assert torch.equal(ob.cpu(), exp[0]), "boxes"


@pytest.mark.parametrize("N,levels,thr,mpc", [
    (8768, 16, 0.5, 300),        # many equal scores: the radix select of a round runs deep into the index bytes; IoUs of exactly 1/2
    (8768, 4096, 1.0 / 3.0, 300),
    (3000, 3, 0.5, 1000),        # almost all scores equal, more kept than one round holds
    (22464, 64, 0.25, 300),      # score keys split over the team's workgroups (no whole-list LDS staging)
    (57000, 8, 0.5, 300),        # ... with runs of equal scores longer than a member's slice
])
def test_nms_equal_scores_and_ious_on_the_threshold(ops, N, levels, thr, mpc):
    """Boxes on a coarse grid (corners k/32, a few sizes): many pairs have an IoU that is EXACTLY a small rational -- 1/2, 1/3, 1/4 --
    i.e. exactly on the threshold or one rounding away from it (the kernel decides inter / union > thr without the division unless the
    quotient could round to either side; this is the case where it must divide), and scores quantised to a few levels (ties are
    resolved by index, as in the oracle: the rounds of the kernel's select then split inside a run of equal scores)."""
    g = torch.Generator().manual_seed(N + levels)
    B = 2
    y0 = torch.randint(0, 24, (B, N, 1, 1), generator=g).float()
    x0 = torch.randint(0, 24, (B, N, 1, 1), generator=g).float()
    h = torch.randint(1, 5, (B, N, 1, 1), generator=g).float()
    w = torch.randint(1, 5, (B, N, 1, 1), generator=g).float()
    boxes = torch.cat([y0, x0, y0 + h, x0 + w], -1) / 32.0
    scores = (torch.randint(0, levels, (B, N, 1), generator=g).float() + 1) / (levels + 1)
    scores[:, ::29, :] = 0.0
    exp = onms.combined_nms(boxes, scores, mpc, mpc, thr, 0.0)
    dev = "cuda"
    ob = torch.full((B, mpc, 4), -1.0, device=dev)
    os_ = torch.full((B, mpc), -1.0, device=dev)
    oc = torch.full((B, mpc), -1, dtype=torch.int32, device=dev)
    ov = torch.full((B,), -1, dtype=torch.int32, device=dev)
    ws = torch.zeros(ops.nms_workspace_bytes(B, N, 1, mpc, mpc), dtype=torch.uint8, device=dev)
    ops.nms_combined(boxes.to(dev), scores.to(dev), B, N, 1, 1, 1, 0, mpc, mpc, thr, 0.0, ob, os_, oc, ov, ws)
    torch.cuda.synchronize()
    assert torch.equal(ov.cpu(), exp[3]), "num_valid_detections"
    assert torch.equal(os_.cpu(), exp[1]), "scores"
    assert torch.equal(ob.cpu(), exp[0]), "boxes"


def test_nms_with_background_column_and_threshold(ops):
    """scores given WITH the background column (stride C+1, offset 1), threshold > 0."""
    g = torch.Generator().manual_seed(9)
    B, N, C = 2, 400, 4
    boxes = _random_boxes(g, B, N, C)
    full = torch.rand(B, N, C + 1, generator=g)
    exp = onms.combined_nms(boxes, full[..., 1:].contiguous(), 30, 50, 0.5, 0.4)
    dev = "cuda"
    ob, os_ = torch.empty(B, 50, 4, device=dev), torch.empty(B, 50, device=dev)
    oc, ov = torch.empty(B, 50, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.zeros(ops.nms_workspace_bytes(B, N, C, 30, 50), dtype=torch.uint8, device=dev)
    ops.nms_combined(boxes.to(dev), full.to(dev), B, N, C, C, C + 1, 1, 30, 50, 0.5, 0.4, ob, os_, oc, ov, ws)
    torch.cuda.synchronize()
    for got, e, name in ((ov, exp[3], "valid"), (os_, exp[1], "scores"), (oc, exp[2], "classes"), (ob, exp[0], "boxes")):
        assert torch.equal(got.cpu(), e), name


def test_rpn_head_post(ops):
    g = torch.Generator().manual_seed(5)
    B, locs, apl, ld = 2, 30, 12, 128
    head = torch.randn(B * locs, ld, generator=g)
    keep = torch.randperm(locs * apl, generator=g)[:100].sort().values.to(torch.int32)
    cls = head[:, :24].reshape(B, locs * apl, 2)
    reg = head[:, 24:72].reshape(B, locs * apl, 4)
    exp_s = torch.softmax(cls, -1)[:, keep.long()]
    exp_d = reg[:, keep.long()]
    s, d = torch.empty(B, 100, 2, device="cuda"), torch.empty(B, 100, 4, device="cuda")
    ops.rpn_head_post(head.cuda(), ld, B, locs * apl, apl, keep.cuda(), 100, s, d)
    torch.cuda.synchronize()
    _close(s, exp_s, 1e-5, 1e-6, "rpn scores")
    assert torch.equal(d.cpu(), exp_d)


def test_rpn_head_post_decode_equals_two_launches(ops):
    """the head-post kernel can also emit the decoded proposals (first launch of proposal NMS): same bits as decode_boxes"""
    g = torch.Generator().manual_seed(6)
    B, locs, apl, ld, n = 2, 30, 12, 128, 100
    head = torch.randn(B * locs, ld, generator=g).cuda()
    keep = torch.randperm(locs * apl, generator=g)[:n].sort().values.to(torch.int32).cuda()
    ctr, sz = torch.rand(n, 2, generator=g) * 300 + 50, torch.rand(n, 2, generator=g) * 80 + 4
    regions = torch.cat([ctr - sz / 2, ctr + sz / 2], -1).cuda()
    s_a, d_a = torch.empty(B, n, 2, device="cuda"), torch.empty(B, n, 1, 4, device="cuda")
    s_b, d_b = torch.empty_like(s_a), torch.empty_like(d_a)
    dec_a, dec_b = torch.empty(B, n, 1, 4, device="cuda"), torch.empty(B, n, 1, 4, device="cuda")
    ops.rpn_head_post(head, ld, B, locs * apl, apl, keep, n, s_a, d_a)
    ops.decode_boxes(regions, d_a, dec_a, B, n, 1, 400, 375)
    ops.rpn_head_post_decode(head, ld, B, locs * apl, apl, keep, n, s_b, d_b, regions, dec_b, 400, 375)
    torch.cuda.synchronize()
    assert torch.equal(s_a, s_b) and torch.equal(d_a, d_b)
    assert torch.equal(dec_a, dec_b) and bool(torch.isfinite(dec_b).all())


@pytest.mark.parametrize("m,c,ld", [(7488, 64, 64), (1000, 256, 256), (256, 64, 64), (300, 40, 64), (77, 36, 37), (5, 8, 8)])
def test_colsum_vector_and_scalar_paths(ops, m, c, ld):
    """bias gradients: column sums of the first c columns of a bf16 [m, ld] matrix, ADDED into the output"""
    g = torch.Generator().manual_seed(m + c)
    x = torch.randn(m, ld, generator=g).to(BF)
    out = torch.full((c,), 2.0, device="cuda")
    ops.colsum_bf16(x.cuda(), m, c, ld, out)
    torch.cuda.synchronize()
    _close(out, x[:, :c].double().sum(0).float() + 2.0, 1e-5, 1e-4, "colsum m=%d c=%d ld=%d" % (m, c, ld))


def test_copy_bytes_multi(ops):
    """one launch for the step's three input copies: different sizes, byte tails, an empty slot"""
    g = torch.Generator().manual_seed(3)
    sizes = (4 * 375 * 1242 * 3, 4 * 100 * 8 * 4, 1607)
    srcs = [torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g).cuda() for n in sizes]
    dsts = [torch.full((n + 32,), 9, dtype=torch.uint8, device="cuda") for n in sizes]
    ops.copy_bytes_multi([(s, d[:s.numel()]) for s, d in zip(srcs, dsts)])
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(d[:s.numel()], s) and bool((d[s.numel():] == 9).all())
    one = torch.zeros(48, dtype=torch.uint8, device="cuda")
    ops.copy_bytes_multi([(srcs[2][:48].clone(), one)])
    torch.cuda.synchronize()
    assert torch.equal(one, srcs[2][:48])


def test_copy_bytes(ops):
    """the full-width device copy that feeds the plan's static image buffer: whole 16-byte words plus a byte tail"""
    g = torch.Generator().manual_seed(2)
    for n in (16, 4 * 375 * 1242 * 3, 1000 * 16 + 7, 5):
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g).cuda()
        dst = torch.full((n + 32,), 9, dtype=torch.uint8, device="cuda")
        ops.copy_bytes(src, dst[:n])
        torch.cuda.synchronize()
        assert torch.equal(dst[:n], src) and bool((dst[n:] == 9).all())


# ------------------------------------------------------------------ RoI
def test_roi_crop_pool_fwd_bwd(ops):
    g = torch.Generator().manual_seed(6)
    B, P, Hf, Wf, C = 2, 20, 24, 78, 64
    feat = _rt(torch.randn(B, Hf, Wf, C, generator=g))
    x0, y0 = torch.rand(B, P, generator=g) * 0.8, torch.rand(B, P, generator=g) * 0.8
    rois = torch.stack([x0, y0, x0 + torch.rand(B, P, generator=g) * 0.4, y0 + torch.rand(B, P, generator=g) * 0.4], -1)
    rois[0, 0] = 0.0                                    # NMS padding row: samples pixel (0,0) everywhere
    rois[0, 1] = torch.tensor([0.0, 0.0, 1.0, 1.0])     # full image
    rois[1, 0] = torch.tensor([0.9, 0.9, 1.3, 1.2])     # partly outside -> extrapolation 0
    fr = feat.clone().requires_grad_(True)
    exp = oroi.roi_pooling(fr, rois, 7, 2)              # [B,P,49*C]
    dev = "cuda"
    pooled = torch.empty(B * P, 49 * C, dtype=BF, device=dev)
    am = torch.empty(B * P, 49 * C, dtype=torch.uint8, device=dev)
    ops.roi_crop_pool_fwd(feat.to(BF).to(dev), rois.to(dev), B, P, Hf, Wf, C, 7, 2, pooled, am)
    torch.cuda.synchronize()
    _close(pooled.view(B, P, -1), exp.detach(), 2 ** -7, 2e-2, "roi pooled")
    # backward on a sampled subset with duplicates
    rows = torch.tensor([0, 1, 5, 5, 20, 21, 39, 39, 39], dtype=torch.int32)
    gp = _rt(torch.randn(len(rows), 49 * C, generator=g))
    gfull = torch.zeros(B * P, 49 * C)
    gfull.index_add_(0, rows.long(), gp)
    exp.backward(gfull.view(B, P, -1))
    gfeat = torch.zeros(B, Hf, Wf, C, device=dev)
    ops.roi_crop_pool_bwd(gp.to(BF).to(dev), am, rois.to(dev), rows.to(dev), len(rows), P, Hf, Wf, C, 7, 2, gfeat)
    torch.cuda.synchronize()
    # argmax ties between the oracle (fp32) and the kernel can differ only where samples are equal
    _close(gfeat, fr.grad, 1e-3, 5e-2, "roi grad")
    # gather form (no global atomics): complete bf16 gradient, every element written once -- also over garbage
    gbf = torch.full((B, Hf, Wf, C), 7.0, dtype=BF, device=dev)
    ops.roi_crop_pool_bwd_bf16(gp.to(BF).to(dev), am, rois.to(dev), rows.to(dev), len(rows), B, P, Hf, Wf, C, 7, 2, gbf)
    torch.cuda.synchronize()
    _close(gbf, gfeat.to(BF), 2 ** -7, 1e-3, "roi grad (gather) vs atomic kernel")
    _close(gbf, fr.grad, 2 ** -6, 5e-2, "roi grad (gather) vs oracle")


@pytest.mark.parametrize("B,C", [(8, 512), (2, 1024), (4, 1024), (1, 1024)])
def test_roi_fwd_wave_uniform_form_equals_generic_kernel(ops, B, C):
    """The round-4 form of the fused crop + pool forward (a wave owns whole bins: scalar tap offsets / weights, taps shared between the
    samples of a bin that fall into the same cells, packed-pair interpolation) against (a) the oracle and (b) the per-item kernel it
    replaces, which still serves maps of fewer than 64 channel vectors: 64-channel slices of the same map go through that kernel
    and must give the same bits.  Slices of 64 vectors ((8, 512): one slice; (4, 1024): two; (2, 1024) / (1, 1024): two slices with
    fewer (image, slice) pairs than XCD lanes).  Boxes: padding rows, the whole image, partly and wholly outside, a few cells wide
    (shared taps in both axes), one cell wide and many cells wide (no shared taps)."""
    g = torch.Generator().manual_seed(B * 1000 + C)
    P, Hf, Wf = 24, 24, 78
    feat = _rt(torch.randn(B, Hf, Wf, C, generator=g))
    x0, y0 = torch.rand(B, P, generator=g) * 0.8, torch.rand(B, P, generator=g) * 0.8
    wh = torch.rand(B, P, 2, generator=g) * torch.tensor([0.02, 0.05, 0.1, 0.2, 0.5, 0.9]).repeat(4)[None, :, None]
    rois = torch.stack([x0, y0, x0 + wh[..., 0] + 0.004, y0 + wh[..., 1] + 0.01], -1)
    rois[0, 0] = 0.0
    rois[0, 1] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    rois[-1, 3] = torch.tensor([0.9, 0.9, 1.3, 1.2])
    rois[-1, 4] = torch.tensor([1.1, 0.2, 1.4, 0.6])
    rois[-1, 5] = torch.tensor([-0.2, -0.1, 0.3, 0.4])
    rois[0, 2] = torch.tensor([0.25, 0.5, 0.25 + 1.0 / 77, 0.5 + 1.0 / 23])           # exactly one cell: integer sample coordinates at the corners
    exp = oroi.roi_pooling(feat.clone(), rois, 7, 2)
    dev = "cuda"
    fd, rd = feat.to(BF).to(dev), rois.to(dev)
    pooled = torch.empty(B * P, 49 * C, dtype=BF, device=dev)
    am = torch.empty(B * P, 49 * C, dtype=torch.uint8, device=dev)
    ops.roi_crop_pool_fwd(fd, rd, B, P, Hf, Wf, C, 7, 2, pooled, am)
    torch.cuda.synchronize()
    _close(pooled.view(B, P, -1), exp, 2 ** -7, 2e-2, "roi pooled (round-4 form)")
    for c0 in (0, C - 64):
        sl = fd[..., c0:c0 + 64].contiguous()
        p64 = torch.empty(B * P, 49 * 64, dtype=BF, device=dev)
        a64 = torch.empty(B * P, 49 * 64, dtype=torch.uint8, device=dev)
        ops.roi_crop_pool_fwd(sl, rd, B, P, Hf, Wf, 64, 7, 2, p64, a64)
        torch.cuda.synchronize()
        assert torch.equal(pooled.view(B * P, 49, C)[:, :, c0:c0 + 64].contiguous().view(torch.int16), p64.view(B * P, 49, 64).view(torch.int16)), "pooled values"
        assert torch.equal(am.view(B * P, 49, C)[:, :, c0:c0 + 64].contiguous(), a64.view(B * P, 49, 64)), "arg-max bytes"


def test_roi_bwd_with_mostly_padding_rois(ops):
    """Regression case of the gather-form backward: more than half of the sampled RoI rows are boxes of zero extent -- the zero
    padding of an NMS output (all at pixel (0,0)) and degenerate boxes at fractional positions.  All ps*ps bins of such a box
    hit the same <= 4 pixels; the kernel sums them in registers (one LDS atomic per tap) instead of ps*ps atomics on one
    address in the one workgroup that owns the row (which made the launch time data-dependent: 48 -> 376 us in round 2)."""
    g = torch.Generator().manual_seed(16)
    B, P, Hf, Wf, C = 2, 64, 24, 78, 128
    feat = _rt(torch.randn(B, Hf, Wf, C, generator=g))
    x0, y0 = torch.rand(B, P, generator=g) * 0.7, torch.rand(B, P, generator=g) * 0.7
    rois = torch.stack([x0, y0, x0 + torch.rand(B, P, generator=g) * 0.3 + 0.02, y0 + torch.rand(B, P, generator=g) * 0.3 + 0.02], -1)
    rois[:, :36] = 0.0                                   # 56 % padding rows
    rois[0, 36] = torch.tensor([0.31, 0.42, 0.31, 0.42])   # degenerate boxes off the pixel grid (4 taps) ...
    rois[1, 37] = torch.tensor([0.5, 1.0, 0.5, 1.0])       # ... on the last feature row ...
    rois[1, 38] = torch.tensor([1.2, 0.3, 1.2, 0.3])       # ... and outside the image (extrapolation: no gradient)
    fr = feat.clone().requires_grad_(True)
    exp = oroi.roi_pooling(fr, rois, 7, 2)
    dev = "cuda"
    pooled = torch.empty(B * P, 49 * C, dtype=BF, device=dev)
    am = torch.empty(B * P, 49 * C, dtype=torch.uint8, device=dev)
    ops.roi_crop_pool_fwd(feat.to(BF).to(dev), rois.to(dev), B, P, Hf, Wf, C, 7, 2, pooled, am)
    torch.cuda.synchronize()
    _close(pooled.view(B, P, -1), exp.detach(), 2 ** -7, 2e-2, "roi pooled")
    rows = torch.cat([torch.arange(0, 40), torch.arange(64, 64 + 44)]).to(torch.int32)       # 40 + 44 sampled rows, 72 of them degenerate
    gp = _rt(torch.randn(len(rows), 49 * C, generator=g))
    gfull = torch.zeros(B * P, 49 * C)
    gfull.index_add_(0, rows.long(), gp)
    exp.backward(gfull.view(B, P, -1))
    gbf = torch.full((B, Hf, Wf, C), 7.0, dtype=BF, device=dev)
    ops.roi_crop_pool_bwd_bf16(gp.to(BF).to(dev), am, rois.to(dev), rows.to(dev), len(rows), B, P, Hf, Wf, C, 7, 2, gbf)
    torch.cuda.synchronize()
    # pixel (0,0) collects 36 boxes x 49 bins per image: a large sum, compared relative to its own size
    _close(gbf, fr.grad, 2 ** -6, 2e-2 * float(fr.grad.abs().max()) / 8, "roi grad (gather) with padding rows")
    assert float(fr.grad[0, 0, 0].abs().max()) > 20.0


def test_roi_bwd_add_form_with_fused_reduce(ops):
    """frcnn_roi_crop_pool_bwd_bf16_add: gfeat = bf16(existing + RoI-branch gradient) -- against the plain gather form added in fp32 --
    and, with `red`, the BatchNorm-backward sums of the layer gfeat arrives at, against frcnn_bn_bwd_reduce on the stored result."""
    g = torch.Generator().manual_seed(26)
    B, P, Hf, Wf, C, dev = 2, 24, 24, 78, 128, "cuda"
    feat = _rt(torch.randn(B, Hf, Wf, C, generator=g))
    x0, y0 = torch.rand(B, P, generator=g) * 0.7, torch.rand(B, P, generator=g) * 0.7
    rois = torch.stack([x0, y0, x0 + torch.rand(B, P, generator=g) * 0.3 + 0.02, y0 + torch.rand(B, P, generator=g) * 0.3 + 0.02], -1)
    rois[0, :3] = 0.0
    pooled = torch.empty(B * P, 49 * C, dtype=BF, device=dev)
    am = torch.empty(B * P, 49 * C, dtype=torch.uint8, device=dev)
    ops.roi_crop_pool_fwd(feat.to(BF).to(dev), rois.to(dev), B, P, Hf, Wf, C, 7, 2, pooled, am)
    rows = torch.cat([torch.arange(0, 16), torch.arange(24, 24 + 20)]).to(torch.int32).to(dev)
    gp = _rt(torch.randn(len(rows), 49 * C, generator=g)).to(BF).to(dev)
    existing = (torch.randn(B, Hf, Wf, C, generator=g) * 0.5).to(BF).to(dev)
    plain = torch.empty(B, Hf, Wf, C, dtype=BF, device=dev)
    ops.roi_crop_pool_bwd_bf16(gp, am, rois.to(dev), rows, len(rows), B, P, Hf, Wf, C, 7, 2, plain)
    z = torch.randn(B * Hf * Wf, C, generator=g).to(BF).to(dev)
    relu_mask = torch.randint(0, 256, (B * Hf * Wf, C // 8), generator=g, dtype=torch.uint8).to(dev)
    mean, invstd = (torch.randn(C, generator=g) * 0.1).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
    part_a, part_b = torch.zeros(16, 2, C, device=dev), torch.zeros(16, 2, C, device=dev)
    out = existing.clone()
    red = ops.bn_reduce_args(z, relu_mask, mean, invstd, part_b)
    ops.roi_crop_pool_bwd_bf16_add(gp, am, rois.to(dev), rows, len(rows), B, P, Hf, Wf, C, 7, 2, out, red=red)
    ops.bn_bwd_reduce(out, None, z, mean, invstd, part_a, B * Hf * Wf, C, relu_mask=relu_mask)
    torch.cuda.synchronize()
    # the add form rounds existing + fp32 row sum once; the reference here adds two bf16 tensors: one bf16 ulp of the larger term apart
    ref = existing.float() + plain.float()
    _close(out, ref.cpu(), 2 ** -7, 2e-2, "RoI backward, add form")
    sa, sb = part_a.sum(0).cpu(), part_b.sum(0).cpu()
    scale = float(out.float().view(-1, C).abs().sum(0).max().cpu())
    assert float(sa.abs().max()) > 0 and float((sa - sb).abs().max()) <= 4e-6 * scale + 1e-5, (float((sa - sb).abs().max()), scale)
    # without `red` nothing but gfeat is touched
    out2 = existing.clone()
    ops.roi_crop_pool_bwd_bf16_add(gp, am, rois.to(dev), rows, len(rows), B, P, Hf, Wf, C, 7, 2, out2)
    torch.cuda.synchronize()
    assert torch.equal(out2.view(torch.int16), out.view(torch.int16))


# ------------------------------------------------------------------ targets / sampling / losses
def _targets_case(seed, B, R, rpn):
    g = torch.Generator().manual_seed(seed)
    cfg = O.default_config()
    _, gl, gb = O.synthetic_batch(B, (8, 8, 3), seed=seed)
    if rpn:
        anchors = O.generate_anchors((24, 78), **cfg["rpn"]["anchors"])
        regions = anchors[O.inside_indices(anchors, cfg["image_shape"])][:R]
    else:
        ctr = torch.rand(B, R, 2, generator=g) * torch.tensor([1242.0, 375.0])
        sz = torch.rand(B, R, 2, generator=g) * torch.tensor([300.0, 200.0]) + 5
        regions = torch.cat([ctr - sz / 2, ctr + sz / 2], -1)
        regions[:, -3:] = 0.0                           # zero-size NMS padding rows
        regions[0, 0] = oboxes.to_absolute(gb[0, 0], cfg["image_shape"])   # an exact match: IoU == 1.0
    return cfg, gl, gb, regions


@pytest.mark.parametrize("rpn", [True, False])
def test_assign_targets(ops, rpn):
    B, R = 3, 8768 if rpn else 300
    cfg, gl, gb, regions = _targets_case(11, B, R, rpn)
    samp = cfg["rpn" if rpn else "rcnn"]["sampling"]
    c1 = 2 if rpn else 8
    exp_l, exp_b = [], []
    for b in range(B):
        labels = F.one_hot(gl[b].sum(-1).long(), 2).float() if rpn else gl[b]
        l, t = otraining.generate_targets(labels, gb[b], regions if rpn else regions[b], cfg["image_shape"],
                                          samp["foreground_iou_interval"], samp["background_iou_interval"])
        exp_l.append(l)
        exp_b.append(t)
    exp_l, exp_b = torch.stack(exp_l), torch.stack(exp_b)
    dev = "cuda"
    tl = torch.full((B, R, c1), -7.0, device=dev)
    tb = torch.full((B, R, c1 - 1, 4), -7.0, device=dev)
    ops.assign_targets(regions.to(dev), gl.to(dev), gb.to(dev), B, R, 100, 8, rpn, 1242, 375,
                       samp["foreground_iou_interval"], samp["background_iou_interval"], tl, tb)
    torch.cuda.synchronize()
    assert torch.equal(tl.cpu(), exp_l), "target labels must be bit-exact"
    fin = torch.isfinite(exp_b)
    assert torch.equal(torch.isfinite(tb.cpu()), fin)
    _close(tb.cpu()[fin], exp_b[fin], 1e-5, 1e-5, "target boxes")   # logf ulp differences only


def test_sample_indices_matches_oracle_and_contract(ops):
    B, R, c1, S, prop = 4, 5000, 8, 64, 0.25
    g = torch.Generator().manual_seed(12)
    tl = torch.zeros(B, R, c1)
    kind = torch.rand(B, R, generator=g)
    tl[..., 0][kind < 0.3] = 1.0                                         # background
    fgm = kind > 0.995
    tl[..., 3][fgm] = 1.0                                                # foreground (class 3)
    tl[3] = 0
    tl[3, :, 0] = 1.0
    tl[3, 5, 0], tl[3, 5, 2] = 0.0, 1.0                                  # a single foreground
    dev = "cuda"
    for step_val, base, seed in ((0, 0, 0), (7, 2, 0x123456789ABCDEF)):
        idx = torch.empty(B, S, dtype=torch.int32, device=dev)
        ws = torch.empty(B, 2 * R, dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        step = torch.tensor([step_val], dtype=torch.int64, device=dev)
        ops.sample_indices(tl.to(dev), B, R, c1, S, prop, seed, step, base, idx, ws, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        for b in range(B):
            exp = otraining.get_sample_indices(tl[b], S, prop, image=b, step=step_val, seed=seed, stream_base=base)
            assert torch.equal(idx[b].cpu().long(), exp), "sample indices image %d" % b
            fg, bg = otraining.split_fg_bg(tl[b])
            n_fg = min(len(fg), 16)
            got = idx[b].cpu().long()
            assert set(got[:n_fg].tolist()) <= set(fg.tolist()) and len(set(got[:n_fg].tolist())) == n_fg   # w/o replacement
            assert set(got[n_fg:].tolist()) <= set(bg.tolist())
    # empty background set -> status flag (the reference raises)
    tl2 = torch.zeros(1, 100, c1)
    tl2[0, :, 2] = 1.0
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.sample_indices(tl2.to(dev), 1, 100, c1, S, prop, 0, torch.zeros(1, dtype=torch.int64, device=dev), 0,
                       torch.empty(1, S, dtype=torch.int32, device=dev), torch.empty(1, 200, dtype=torch.int32, device=dev), status)
    torch.cuda.synchronize()
    assert int(status.item()) == 1


@pytest.mark.parametrize("c1,C,B,S", [(2, 1, 3, 64), (8, 7, 3, 64), (2, 1, 5, 256)])      # (the last: more rows than threads)
def test_losses_and_gradients(ops, c1, C, B, S):
    g = torch.Generator().manual_seed(13 + c1)
    R = 400
    logits = torch.randn(B, R, c1, generator=g).requires_grad_(True)
    deltas = (torch.randn(B, R, C, 4, generator=g) * 1.5).requires_grad_(True)
    scores = torch.softmax(logits, -1)
    tl = F.one_hot(torch.randint(0, c1, (B, R), generator=g), c1).float()
    tb = torch.zeros(B, R, C, 4)
    fg = tl[..., 0] == 0
    cls = tl[..., 1:].argmax(-1)
    tb[fg, cls[fg]] = torch.randn(int(fg.sum()), 4, generator=g)
    idx = torch.randint(0, R, (B, S), generator=g)
    idx[:, 1] = idx[:, 0]                                                 # duplicates
    ar = torch.arange(B)[:, None]
    lc = olosses.classification_loss(tl[ar, idx], scores[ar, idx])
    lr = olosses.regression_loss(tb[ar, idx], deltas[ar, idx])
    (0.5 * lc + 2.0 * lr).backward()
    dev = "cuda"
    out = torch.empty(2, device=dev)
    dl = torch.empty(B, S, c1, device=dev)
    dd = torch.empty(B, S, C, 4, device=dev)
    ops.losses(scores.detach().to(dev), deltas.detach().to(dev), tl.to(dev), tb.to(dev), idx.to(torch.int32).to(dev),
               B, R, c1, S, 0.5, 2.0, out, dl, dd)
    torch.cuda.synchronize()
    _close(out[0], lc.detach(), 1e-5, 1e-6, "cls loss")
    _close(out[1], lr.detach(), 1e-5, 1e-5, "reg loss")
    dense_l = torch.zeros(B, R, c1)
    dense_d = torch.zeros(B, R, C, 4)
    for b in range(B):
        dense_l[b].index_add_(0, idx[b], dl[b].cpu())
        dense_d[b].index_add_(0, idx[b], dd[b].cpu())
    _close(dense_l, logits.grad, 1e-4, 1e-7, "dlogits")
    _close(dense_d, deltas.grad, 1e-5, 1e-7, "ddeltas")


def test_head_grad_scatter_gather(ops):
    g = torch.Generator().manual_seed(14)
    dev = "cuda"
    B, S, locs, apl, ld = 2, 32, 40, 12, 128
    A = locs * apl
    keep = torch.randperm(A, generator=g)[:200].sort().values.to(torch.int32)
    idx = torch.randint(0, 200, (B, S), generator=g, dtype=torch.int32)
    idx[:, 3] = idx[:, 2]
    dl, dd = torch.randn(B, S, 2, generator=g), torch.randn(B, S, 4, generator=g)
    exp = torch.zeros(B * locs, ld)
    for b in range(B):
        for s in range(S):
            a = int(keep[int(idx[b, s])])
            loc, k = a // apl, a % apl
            exp[b * locs + loc, 2 * k:2 * k + 2] += dl[b, s]
            exp[b * locs + loc, 24 + 4 * k:24 + 4 * k + 4] += dd[b, s]
    dhead = torch.zeros(B * locs, ld, device=dev)
    ops.rpn_head_grad(dl.to(dev), dd.to(dev), idx.to(dev), keep.to(dev), B, S, A, apl, dhead, ld)
    torch.cuda.synchronize()
    _close(dhead, exp, 1e-6, 1e-6, "rpn head grad scatter")
    R, c1 = 300, 8
    idx2 = torch.randint(0, R, (B, S), generator=g, dtype=torch.int32)
    dl2, dd2 = torch.randn(B, S, c1, generator=g), torch.randn(B, S, 7, 4, generator=g)
    out = torch.empty(B * S, 64, dtype=BF, device=dev)
    rows = torch.empty(B * S, dtype=torch.int32, device=dev)
    ops.rcnn_head_grad(dl2.to(dev), dd2.to(dev), idx2.to(dev), B, R, c1, S, out, 64, rows)
    torch.cuda.synchronize()
    exp2 = torch.zeros(B * S, 64)
    exp2[:, :8] = dl2.reshape(-1, 8)
    exp2[:, 8:36] = dd2.reshape(-1, 28)
    _close(out, exp2, 2 ** -8, 0, "rcnn head grad rows")
    assert torch.equal(rows.cpu().long(), (torch.arange(B)[:, None] * R + idx2.long()).reshape(-1))


def test_losses_head_grad_fused_equals_two_launches(ops):
    """frcnn_losses_head_grad == frcnn_losses followed by frcnn_rcnn_head_grad, bit for bit (the train plan uses the fused form)."""
    g = torch.Generator().manual_seed(21)
    dev = "cuda"
    B, R, S, c1, C, ld = 4, 300, 64, 8, 7, 64
    scores = torch.softmax(torch.randn(B, R, c1, generator=g), -1).to(dev)
    deltas = (torch.randn(B, R, C, 4, generator=g) * 1.5).to(dev)
    tl = F.one_hot(torch.randint(0, c1, (B, R), generator=g), c1).float()
    tb = torch.zeros(B, R, C, 4)
    fg = tl[..., 0] == 0
    tb[fg, tl[..., 1:].argmax(-1)[fg]] = torch.randn(int(fg.sum()), 4, generator=g)
    idx = torch.randint(0, R, (B, S), generator=g, dtype=torch.int32).to(dev)
    tl, tb = tl.to(dev), tb.to(dev)
    out_a, out_b = torch.empty(2, device=dev), torch.empty(2, device=dev)
    dl_a, dd_a = torch.empty(B, S, c1, device=dev), torch.empty(B, S, C, 4, device=dev)
    dl_b, dd_b = torch.empty_like(dl_a), torch.empty_like(dd_a)
    rows_a, rows_b = torch.full((B * S,), -1, dtype=torch.int32, device=dev), torch.full((B * S,), -1, dtype=torch.int32, device=dev)
    h_a = torch.full((B * S, ld), 7.0, dtype=BF, device=dev)
    h_b = torch.full((B * S, ld), 7.0, dtype=BF, device=dev)
    ops.losses(scores, deltas, tl, tb, idx, B, R, c1, S, 0.25, 1.0, out_a, dl_a, dd_a)
    ops.rcnn_head_grad(dl_a, dd_a, idx, B, R, c1, S, h_a, ld, rows_a)
    ops.losses_head_grad(scores, deltas, tl, tb, idx, B, R, c1, S, 0.25, 1.0, out_b, dl_b, dd_b, h_b, ld, rows_b)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b) and torch.equal(dl_a, dl_b) and torch.equal(dd_a, dd_b)
    assert torch.equal(rows_a, rows_b)
    assert torch.equal(h_a.view(torch.int16), h_b.view(torch.int16)), "fused head-gradient rows differ"
    assert float(h_b.float().abs().sum()) > 0 and bool((h_b[:, c1 + 4 * C:] == 0).all())
    # the f32 per-sample gradients are optional in the fused form
    h_c = torch.full((B * S, ld), 7.0, dtype=BF, device=dev)
    ops.losses_head_grad(scores, deltas, tl, tb, idx, B, R, c1, S, 0.25, 1.0, out_b, None, None, h_c, ld, rows_b)
    torch.cuda.synchronize()
    assert torch.equal(h_c.view(torch.int16), h_a.view(torch.int16))


def test_losses_rpn_head_grad_fused_equals_two_launches(ops):
    """frcnn_losses_rpn_head_grad == frcnn_losses + frcnn_rpn_head_grad (float atomics: equal up to the order of additions)."""
    g = torch.Generator().manual_seed(22)
    dev = "cuda"
    B, S, locs, apl, ld, n = 3, 256, 60, 12, 128, 500
    A = locs * apl
    keep = torch.randperm(A, generator=g)[:n].sort().values.to(torch.int32).to(dev)
    scores = torch.softmax(torch.randn(B, n, 2, generator=g), -1).to(dev)
    deltas = (torch.randn(B, n, 1, 4, generator=g) * 1.5).to(dev)
    tl = F.one_hot(torch.randint(0, 2, (B, n), generator=g), 2).float()
    tb = torch.zeros(B, n, 1, 4)
    fg = tl[..., 1] == 1
    tb[fg] = torch.randn(int(fg.sum()), 1, 4, generator=g)
    idx = torch.randint(0, n, (B, S), generator=g, dtype=torch.int32)
    idx[:, 5] = idx[:, 4]                                                 # duplicated samples accumulate
    tl, tb, idx = tl.to(dev), tb.to(dev), idx.to(dev)
    out_a, out_b = torch.empty(2, device=dev), torch.empty(2, device=dev)
    dl, dd = torch.empty(B, S, 2, device=dev), torch.empty(B, S, 1, 4, device=dev)
    h_a, h_b = torch.zeros(B * locs, ld, device=dev), torch.zeros(B * locs, ld, device=dev)
    ops.losses(scores, deltas, tl, tb, idx, B, n, 2, S, 0.5, 1.0, out_a, dl, dd)
    ops.rpn_head_grad(dl, dd, idx, keep, B, S, A, apl, h_a, ld)
    ops.losses_rpn_head_grad(scores, deltas, tl, tb, idx, B, n, S, 0.5, 1.0, out_b, None, None, keep, A, apl, h_b, ld)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert float(h_a.abs().sum()) > 0
    _close(h_b, h_a, 1e-6, 1e-9, "fused RPN head-gradient scatter")


def test_rcnn_head_post(ops):
    g = torch.Generator().manual_seed(15)
    R, ld = 77, 64
    logits = torch.randn(R, ld, generator=g)
    bias = torch.randn(36, generator=g)
    s, d = torch.empty(R, 8, device="cuda"), torch.empty(R, 28, device="cuda")
    ops.rcnn_head_post(logits.cuda(), ld, bias.cuda(), R, 8, s, d)
    torch.cuda.synchronize()
    _close(s, torch.softmax(logits[:, :8] + bias[:8], -1), 1e-5, 1e-6, "rcnn scores")
    _close(d, logits[:, 8:36] + bias[8:], 0, 1e-6, "rcnn deltas")


# ------------------------------------------------------------------ call surface: box helpers, per-image targets, metrics
def test_box_helpers_under_reference_names(ops):
    """utils/boxes.py decode / encode / to_relative / to_absolute / clip_to_window against the oracle's literal restatement of
    reference utils/boxes.py:4-93, for the broadcasts the reference uses (post_processing.py:39-44, training.py:69)."""
    UB = importlib.import_module("2d_object_detection_amd.utils.boxes")
    from oracle import boxes as OB
    g = torch.Generator().manual_seed(21)
    B, R, C = 2, 37, 7
    mins = torch.rand(B, R, C, 2, generator=g) * 300
    boxes = torch.cat([mins, mins + torch.rand(B, R, C, 2, generator=g) * 200 + 1], -1)
    rmin = torch.rand(R, 2, generator=g) * 300
    ref = torch.cat([rmin, rmin + torch.rand(R, 2, generator=g) * 150 + 1], -1)
    refb = ref[None].repeat(B, 1, 1) + torch.rand(B, R, 1, generator=g)
    # encode: [B,R,C,4] x [R,4] (tiled regions), x [B,R,1,4]; [R,4] x [R,4] (training.py:69)
    for bx, rf, rf_o in ((boxes, ref, ref[None, :, None, :]), (boxes, refb[:, :, None, :], refb[:, :, None, :]), (boxes[0, :, 0], ref, ref)):
        enc = UB.encode(bx.cuda(), rf.cuda())
        exp = OB.encode(bx, rf_o)
        assert enc.shape == bx.shape
        _close(enc, exp, 1e-5, 1e-6, "encode")
        dec = UB.decode(enc, rf.cuda())
        _close(dec, OB.decode(exp, rf_o), 1e-5, 1e-4, "decode")
        _close(dec, bx, 1e-4, 1e-3, "decode(encode(x)) == x")
    shape = (375, 1242, 3)
    rel = UB.to_relative(boxes.cuda(), shape)
    assert torch.equal(rel.cpu(), OB.to_relative(boxes, shape)), "to_relative is a true division"
    assert torch.equal(UB.to_absolute(rel, shape).cpu(), OB.to_absolute(rel.cpu(), shape))
    with pytest.raises(ValueError):
        UB.encode(boxes.cuda(), ref[:5].cuda())


def test_generate_targets_per_image_form(ops):
    """reference utils/training.py:7 takes ONE image ([G,C+1], [G,4], [R,4]); the batched form must equal it row for row."""
    UT = importlib.import_module("2d_object_detection_amd.utils.training")
    g = torch.Generator().manual_seed(4)
    B, G, R, C1 = 3, 100, 500, 8
    gl, gb = torch.zeros(B, G, C1), torch.zeros(B, G, 4)
    for b in range(B):
        n = 5 + b
        mn = torch.rand(n, 2, generator=g) * 0.6
        gb[b, :n] = torch.cat([mn, mn + torch.rand(n, 2, generator=g) * 0.3 + 0.05], -1)
        gl[b, torch.arange(n), torch.randint(1, C1, (n,), generator=g)] = 1.0
    mn = torch.rand(R, 2, generator=g) * torch.tensor([900.0, 250.0])
    regions = torch.cat([mn, mn + torch.rand(R, 2, generator=g) * 300 + 8], -1)
    tl, tb = UT.generate_targets(gl.cuda(), gb.cuda(), regions.cuda(), (375, 1242, 3), [0.5, 1.0], [0.0, 0.5])
    for b in range(B):
        tl1, tb1 = UT.generate_targets(gl[b].cuda(), gb[b].cuda(), regions.cuda(), (375, 1242, 3), [0.5, 1.0], [0.0, 0.5])
        assert tl1.shape == (R, C1) and tb1.shape == (R, C1 - 1, 4)
        assert torch.equal(tl1, tl[b]) and torch.equal(tb1, tb[b])


def test_metrics_on_device_match_loop_oracle():
    """AP / mAP (utils/metrics.py:4-133) with CUDA tensors -- as the training driver feeds them, including the reuse of the
    train step's static prediction buffers -- against the loop oracle."""
    MET = importlib.import_module("2d_object_detection_amd.utils.metrics")
    from oracle import metrics as om
    from test_data_metrics import _random_case
    g = torch.Generator().manual_seed(9)
    B, G, P, C = 2, 6, 24, 3
    pb_buf, ps_buf, pc_buf = torch.zeros(B, P, 4, device="cuda"), torch.zeros(B, P, device="cuda"), torch.zeros(B, P, dtype=torch.int32, device="cuda")
    ap, apo = MET.AveragePrecision(0.5), om.AveragePrecisionOracle(0.5)
    mp, mpo = MET.MeanAveragePrecision(C, 0.5), om.MeanAveragePrecisionOracle(C, 0.5)
    for _ in range(3):
        gt, lab, pb, ps, pc = _random_case(g, B, G, P, C)
        pb_buf.copy_(pb)
        ps_buf.copy_(ps)
        pc_buf.copy_(pc)
        ap.update_state(gt.cuda(), pb_buf, ps_buf)
        mp.update_state(gt.cuda(), lab.cuda(), pb_buf, ps_buf, pc_buf)
        apo.update_state(gt, pb, ps)
        mpo.update_state(gt, lab, pb, ps, pc)
    assert abs(ap.result() - apo.result()) < 1e-6
    assert abs(mp.result() - mpo.result()) < 1e-6
    b1 = torch.rand(5, 4)
    b1[:, 2:] += b1[:, :2]
    assert torch.equal(MET.iou(b1.cuda(), b1.cuda(), pairwise=True).cpu(), om.iou(b1, b1, pairwise=True))


# ------------------------------------------------------------------ round 4: launches folded into their neighbours (each fused form == its parts)
def test_sgd_fused_equals_separate_launches(ops):
    """frcnn_sgd_momentum_fused == frcnn_sgd_momentum per decay range + frcnn_stem_pack_weights + frcnn_step_increment, bit for bit; the
    step counter moves exactly once however many workgroups the launch has, and the arrival counter is left at zero."""
    g = torch.Generator().manual_seed(31)
    dev = "cuda"
    n, decay_end, stem_begin, cout = 300_000, 70_016, 120_000, 64
    w, gr, v = torch.randn(n, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    bounds = torch.tensor([40000, 80000, 0], dtype=torch.int64, device=dev)
    values = torch.tensor([1e-3, 1e-4, 1e-5], device=dev)
    for step_val in (0, 40001):
        wa, va, wb_a = w.to(dev), v.to(dev), torch.zeros(n, dtype=BF, device=dev)
        wbb, vb, wb_b = w.to(dev), v.to(dev), torch.zeros(n, dtype=BF, device=dev)
        gd = gr.to(dev)
        step_a = torch.tensor([step_val], dtype=torch.int64, device=dev)
        step_b = step_a.clone()
        pk_a = torch.full((cout, 7, 8, 4), 9.0, dtype=BF, device=dev)
        pk_b = torch.zeros(cout, 7, 8, 4, dtype=BF, device=dev)        # (the fused form only writes the 7 x 7 x 3 real taps: padding stays as allocated, zero)
        ops.sgd_momentum(wa[:decay_end], gd[:decay_end], va[:decay_end], wb_a[:decay_end], decay_end, 0.9, 0.0005, 1.0, step_a, bounds, values, 2)
        ops.sgd_momentum(wa[decay_end:], gd[decay_end:], va[decay_end:], wb_a[decay_end:], n - decay_end, 0.9, 0.0, 1.0, step_a, bounds, values, 2)
        ops.stem_pack_weights(wa[stem_begin:stem_begin + cout * 147], pk_a, cout)
        ops.step_increment(step_a)
        arrive = torch.zeros(4, dtype=torch.int32, device=dev)
        fused = ops.sgd_fused_args(decay_end, 0.0005, arrive, stem_begin, cout, pk_b)
        ops.sgd_momentum_fused(wbb, gd, vb, wb_b, n, 0.9, 1.0, step_b, bounds, values, 2, fused)
        torch.cuda.synchronize()
        assert torch.equal(wa, wbb) and torch.equal(va, vb) and torch.equal(wb_a.view(torch.int16), wb_b.view(torch.int16))
        assert torch.equal(pk_a.view(torch.int16), pk_b.view(torch.int16)), "packed stem taps"
        assert int(step_a) == int(step_b) == step_val + 1 and int(arrive[0]) == 0
        ops.sgd_momentum_fused(wbb, gd, vb, wb_b, n, 0.9, 1.0, step_b, bounds, values, 2, fused)       # a second launch on the reset counter
        torch.cuda.synchronize()
        assert int(step_b) == step_val + 2 and int(arrive[0]) == 0


def test_sgd_bucket_launches_equal_the_one_launch_update(ops):
    """SGD.apply_bucket_plan + apply_plan(first=...) (round 5: a finished gradient bucket is updated under the rest of the backward pass):
    frcnn_sgd_momentum_fused over PARTS of the flat buffers -- arrive = NULL: no counter update -- followed by the launch over the last
    part (stem re-pack, step counter) == the one launch over everything, bit for bit; the partial launches leave the step counter alone."""
    g = torch.Generator().manual_seed(32)
    dev = "cuda"
    n, decay_end, stem_begin, cout = 300_032, 70_016, 280_000, 64
    cuts = [0, 69_952, 70_400, 200_000, 270_016, n]                # (the decay boundary lies INSIDE the second part; the stem in the last)
    w, gr, v = torch.randn(n, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    bounds = torch.tensor([40000, 80000, 0], dtype=torch.int64, device=dev)
    values = torch.tensor([1e-3, 1e-4, 1e-5], device=dev)
    gd = gr.to(dev)
    wa, va, wb_a = w.to(dev), v.to(dev), torch.zeros(n, dtype=BF, device=dev)
    wbb, vb, wb_b = w.to(dev), v.to(dev), torch.zeros(n, dtype=BF, device=dev)
    step_a = torch.tensor([40001], dtype=torch.int64, device=dev)
    step_b = step_a.clone()
    pk_a, pk_b = torch.zeros(cout, 7, 8, 4, dtype=BF, device=dev), torch.zeros(cout, 7, 8, 4, dtype=BF, device=dev)
    arrive_a, arrive_b = torch.zeros(4, dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
    ops.sgd_momentum_fused(wa, gd, va, wb_a, n, 0.9, 1.0, step_a, bounds, values, 2, ops.sgd_fused_args(decay_end, 0.0005, arrive_a, stem_begin, cout, pk_a))
    for b, e in zip(cuts[:-2], cuts[1:-1]):
        part = ops.sgd_fused_args(min(max(decay_end - b, 0), e - b), 0.0005, None)
        ops.sgd_momentum_fused(wbb[b:e], gd[b:e], vb[b:e], wb_b[b:e], e - b, 0.9, 1.0, step_b, bounds, values, 2, part)
        torch.cuda.synchronize()
        assert int(step_b) == 40001, "a partial launch moved the step counter"
    b = cuts[-2]
    last = ops.sgd_fused_args(0, 0.0005, arrive_b, stem_begin - b, cout, pk_b)
    ops.sgd_momentum_fused(wbb[b:], gd[b:], vb[b:], wb_b[b:], n - b, 0.9, 1.0, step_b, bounds, values, 2, last)
    torch.cuda.synchronize()
    assert torch.equal(wa, wbb) and torch.equal(va, vb) and torch.equal(wb_a.view(torch.int16), wb_b.view(torch.int16))
    assert torch.equal(pk_a.view(torch.int16), pk_b.view(torch.int16)) and float(pk_a.float().abs().sum()) > 0
    assert int(step_a) == int(step_b) == 40002 and int(arrive_b[0]) == 0


@pytest.mark.parametrize("m,c", [(7488, 128), (200, 64), (29952, 256)])
def test_cast_and_relu_colsum_fused_equal_two_launches(ops, m, c):
    """frcnn_cast_colsum == frcnn_cast_f32_bf16 + frcnn_colsum_bf16 and frcnn_relu_bwd_colsum == frcnn_relu_bwd + frcnn_colsum_bf16: the
    stored matrices bit for bit; the sums equal up to the arrival order of one float atomic per (column, 256-row chunk) -- exactly when
    there is a single chunk."""
    g = torch.Generator().manual_seed(m + c)
    dev = "cuda"
    src = (torch.randn(m, c, generator=g) * 0.1).to(dev)
    src[::3] = 0.0
    d_a, d_b = torch.empty(m, c, dtype=BF, device=dev), torch.empty(m, c, dtype=BF, device=dev)
    s_a, s_b = torch.full((c,), 0.5, device=dev), torch.full((c,), 0.5, device=dev)
    ops.cast_f32_bf16(src, d_a)
    ops.colsum_bf16(d_a, m, c, c, s_a)
    ops.cast_colsum(src, d_b, m, c, s_b)
    gg = torch.randn(m, c, generator=g).to(BF).to(dev)
    act = torch.randn(m, c, generator=g).clamp(min=0).to(BF).to(dev)
    r_a, r_b = torch.empty_like(gg), torch.empty_like(gg)
    t_a, t_b = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.relu_bwd(gg, act, r_a)
    ops.colsum_bf16(r_a, m, c, c, t_a)
    ops.relu_bwd_colsum(gg, act, r_b, m, c, t_b)
    torch.cuda.synchronize()
    assert torch.equal(d_a.view(torch.int16), d_b.view(torch.int16)) and torch.equal(r_a.view(torch.int16), r_b.view(torch.int16))
    if m <= 256:
        assert torch.equal(s_a, s_b) and torch.equal(t_a, t_b)
    _close(s_b, s_a.cpu(), 1e-5, 1e-4, "cast + column sums")                   # (one float atomic per column and 256-row chunk: arrival order)
    _close(t_b, t_a.cpu(), 1e-5, 1e-4, "ReLU backward + column sums")
    _close(t_b, r_a.double().sum(0).float().cpu(), 1e-5, 1e-3, "column sums against torch")


@pytest.mark.parametrize("B,N,q,C,mpc,mt", [(2, 8768, 1, 1, 300, 300), (2, 300, 7, 7, 100, 300)])
def test_nms_combined_abs_equals_nms_then_scale(ops, B, N, q, C, mpc, mt):
    """frcnn_nms_combined_abs == frcnn_nms_combined + frcnn_boxes_scale on its boxes (single-class and merged forms), bit for bit."""
    g = torch.Generator().manual_seed(N)
    dev = "cuda"
    boxes = _random_boxes(g, B, N, q).to(dev)
    scores = torch.stack([torch.randperm(N * C, generator=g) for _ in range(B)]).reshape(B, N, C).float().add(1).div(N * C + 1).to(dev)
    outs = []
    for fused in (False, True):
        ob, os_ = torch.full((B, mt, 4), -1.0, device=dev), torch.full((B, mt), -1.0, device=dev)
        oc, ov = torch.full((B, mt), -1, dtype=torch.int32, device=dev), torch.full((B,), -1, dtype=torch.int32, device=dev)
        oa = torch.full((B, mt, 4), -1.0, device=dev)
        ws = torch.zeros(ops.nms_workspace_bytes(B, N, C, mpc, mt), dtype=torch.uint8, device=dev)
        if fused:
            ops.nms_combined_abs(boxes, scores, B, N, q, C, C, 0, mpc, mt, 0.7, 0.0, ob, os_, oc, ov, ws, oa, 1242.0, 375.0)
        else:
            ops.nms_combined(boxes, scores, B, N, q, C, C, 0, mpc, mt, 0.7, 0.0, ob, os_, oc, ov, ws)
            ops.boxes_scale(ob, oa, 1242.0, 375.0)
        torch.cuda.synchronize()
        outs.append((ob, os_, oc, ov, oa))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[1][4].max()) > 300.0


def test_rcnn_head_post_decode_equals_two_launches(ops):
    """frcnn_rcnn_head_post_decode == frcnn_rcnn_head_post + frcnn_decode_boxes (per-image regions), bit for bit."""
    g = torch.Generator().manual_seed(16)
    dev = "cuda"
    B, P, c1, ld = 2, 300, 8, 64
    R = B * P
    logits = torch.randn(R, ld, generator=g).to(dev)
    bias = torch.randn(ld, generator=g).to(dev)
    ctr = torch.rand(B, P, 2, generator=g) * torch.tensor([1242.0, 375.0])
    sz = torch.rand(B, P, 2, generator=g) * 200 + 8
    regions = torch.cat([ctr - sz / 2, ctr + sz / 2], -1).to(dev)
    s_a, d_a = torch.empty(B, P, c1, device=dev), torch.empty(B, P, c1 - 1, 4, device=dev)
    s_b, d_b = torch.empty_like(s_a), torch.empty_like(d_a)
    dec_a, dec_b = torch.empty(B, P, c1 - 1, 4, device=dev), torch.empty(B, P, c1 - 1, 4, device=dev)
    ops.rcnn_head_post(logits, ld, bias, R, c1, s_a, d_a)
    ops.decode_boxes(regions, d_a, dec_a, B, P, c1 - 1, 1242, 375)
    ops.rcnn_head_post_decode(logits, ld, bias, R, c1, s_b, d_b, regions, dec_b, 1242, 375)
    torch.cuda.synchronize()
    assert torch.equal(s_a, s_b) and torch.equal(d_a, d_b)
    assert torch.equal(dec_a, dec_b) and bool(torch.isfinite(dec_b).all())


@pytest.mark.parametrize("B,S", [(4, 64), (8, 64), (2, 16)])
def test_losses_head_grad_bias_gradient_equals_colsum(ops, B, S):
    """frcnn_losses_head_grad(bias_grad=...) adds exactly what frcnn_colsum_bf16 adds from the head-gradient rows it writes (same
    chunking, same order of additions: bit for bit, also over two 256-row chunks at batch 8)."""
    g = torch.Generator().manual_seed(23 + B)
    dev = "cuda"
    R, c1, C, ld = 300, 8, 7, 64
    scores = torch.softmax(torch.randn(B, R, c1, generator=g), -1).to(dev)
    deltas = (torch.randn(B, R, C, 4, generator=g) * 1.5).to(dev)
    tl = F.one_hot(torch.randint(0, c1, (B, R), generator=g), c1).float()
    tb = torch.zeros(B, R, C, 4)
    fg = tl[..., 0] == 0
    tb[fg, tl[..., 1:].argmax(-1)[fg]] = torch.randn(int(fg.sum()), 4, generator=g)
    idx = torch.randint(0, R, (B, S), generator=g, dtype=torch.int32).to(dev)
    tl, tb = tl.to(dev), tb.to(dev)
    out = torch.empty(2, device=dev)
    rows = torch.empty(B * S, dtype=torch.int32, device=dev)
    h_a, h_b = torch.empty(B * S, ld, dtype=BF, device=dev), torch.empty(B * S, ld, dtype=BF, device=dev)
    bg_a, bg_b = torch.zeros(ld, device=dev), torch.zeros(ld, device=dev)
    ops.losses_head_grad(scores, deltas, tl, tb, idx, B, R, c1, S, 0.25, 1.0, out, None, None, h_a, ld, rows)
    ops.colsum_bf16(h_a, B * S, ld, ld, bg_a)
    ops.losses_head_grad(scores, deltas, tl, tb, idx, B, R, c1, S, 0.25, 1.0, out, None, None, h_b, ld, rows, bias_grad=bg_b)
    torch.cuda.synchronize()
    assert torch.equal(h_a.view(torch.int16), h_b.view(torch.int16))
    assert float(bg_a.abs().sum()) > 0 and torch.equal(bg_a, bg_b), (bg_a - bg_b).abs().max()


@pytest.mark.parametrize("case", conv_cases.BNIN, ids=[c["id"] for c in conv_cases.BNIN])
def test_conv2d_fprop_bnin_equals_bn_apply_then_conv(ops, case):
    """frcnn_conv2d_fprop_bnin == frcnn_bn_train_apply (ReLU, bit mask) followed by frcnn_conv2d_fprop on its output: the activation, its
    ReLU mask, mean / invstd and the moving statistics bit for bit, the convolution output bit for bit (both run the weights-resident
    kernel on identical patches), its statistics up to the order of the f64 slot sums.  (4, 94, 311) is conv2's shape at the benchmark's
    batch; the others end inside tiles in both directions, one with two channel parts."""
    n, h, w, cout, stats = case["n"], case["h"], case["w"], case["cout"], case["stats"]
    g = torch.Generator().manual_seed(7 + h)
    dev, cin = "cuda", case["cin"]
    m = n * h * w
    z = (torch.randn(m, cin, generator=g) * 1.3 + 0.2).to(BF).to(dev)
    zf = z.double()
    zstats = torch.zeros(16, 2, cin, dtype=torch.float64, device=dev)
    for s_ in range(16):                                   # the sums spread over the slots, as the producing convolution leaves them
        rows = slice(s_ * m // 16, (s_ + 1) * m // 16)
        zstats[s_, 0], zstats[s_, 1] = zf[rows].sum(0), (zf[rows] * zf[rows]).sum(0)
    gamma, beta = (torch.rand(cin, generator=g) + 0.5).to(dev), (torch.randn(cin, generator=g) * 0.2).to(dev)
    kk = case.get("k", 3)
    wt = (torch.randn(cout, kk, kk, cin, generator=g) / (8.0 * kk)).to(BF).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    flags = ops.CONV_BIAS | (ops.CONV_STATS if stats else 0)
    d = conv_cases.bnin_desc(ops, case)
    assert ops.conv2d_bnin_supported(d)
    assert not ops.conv2d_bnin_supported(ops.conv_desc(1, 24, 78, 1024, 1, 1, 1, 0, 0, 24, 78, 256))         # (a 1x1 layer with 16 slices: three-slot ring)
    assert not ops.conv2d_bnin_supported(ops.conv_desc(4, 47, 156, 256, 1, 1, 2, 0, 0, 24, 78, 512))         # (a strided 1x1 layer)
    assert not ops.conv2d_bnin_supported(ops.conv_desc(8, 94, 311, 256, 3, 3, 1, 1, 1, 94, 311, 256))         # (3x3 on the tile kernel: too many tiles for the patch forms)
    # the instantiation the plain form runs: the fused form must be ITS ,BNIN=1 twin (the three-workgroup tile form's twin holds two per CU)
    kernel = ops.conv2d_describe(d).split(" grid")[0]

    def buffers():
        return dict(act=torch.zeros(m, cin, dtype=BF, device=dev), mask=torch.zeros(m, cin // 8, dtype=torch.uint8, device=dev),
                    mean=torch.zeros(cin, device=dev), invstd=torch.zeros(cin, device=dev), mm=torch.full((cin,), 0.25, device=dev),
                    mv=torch.full((cin,), 1.5, device=dev), y=torch.zeros(m, cout, dtype=BF, device=dev),
                    ystats=torch.zeros(16, 2, cout, dtype=torch.float64, device=dev))

    a, b = buffers(), buffers()
    ops.bn_train_apply(z, zstats, 16, m, gamma, beta, a["mm"], a["mv"], 0.99, 1.001e-5, a["act"], a["mean"], a["invstd"], m, cin, relu=True,
                       relu_mask=a["mask"])
    ops.conv2d_fprop(d, a["act"], wt, a["y"], bias=bias, stats=a["ystats"] if stats else None)
    assert ops.last_conv_instantiation().split(" grid")[0] == kernel and kernel.startswith(("conv3x3_wres<", "conv3x3_patch<", "conv_tile<")), kernel
    bn = ops.bn_in_args(zstats, gamma, beta, b["mm"], b["mv"], 0.99, 1.001e-5, m, b["act"], b["mask"], b["mean"], b["invstd"])
    ops.conv2d_fprop_bnin(d, z, wt, b["y"], bn, bias=bias, stats=b["ystats"] if stats else None)
    assert ops.last_conv_instantiation().split(" grid")[0] == kernel[:-1].replace("OCC=3", "OCC=2") + ",BNIN=1>", ops.last_conv_instantiation()
    torch.cuda.synchronize()
    for k in ("mean", "invstd", "mm", "mv"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["act"].view(torch.int16), b["act"].view(torch.int16)), "activation"
    assert torch.equal(a["mask"], b["mask"]), "ReLU mask"
    assert torch.equal(a["y"].view(torch.int16), b["y"].view(torch.int16)), "convolution output"
    if stats:
        sa, sb = a["ystats"].sum(0), b["ystats"].sum(0)
        assert float(sa.abs().max()) > 0 and float(((sa - sb).abs() / (sa.abs() + 1.0)).max()) < 1e-9
