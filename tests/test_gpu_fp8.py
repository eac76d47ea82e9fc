"""GPU parity of the fp8 (OCP e4m3) convolution path (BASELINE.json configs[4]: "fp8 weights ... CDNA4 fp8 MFMA conv path";
reference contraction: the Conv2D layers of models/feature_extractor.py:8-10 and models/detectors/rpn_detector.py:26-34).

The fp8 kernel is compared with torch-CPU fp32 on the DEQUANTISED operands: products of two e4m3 values are exact in fp32 and
the MFMA accumulates in fp32, so only the accumulation order and the bf16 rounding of the output differ -- the tolerance is the
bf16 kernels' (2^-7 relative + the absolute term of cancelling sums), not an fp8 tolerance.  The quantisers are compared bit for
bit with torch's float8_e4m3fn conversion (round to nearest even) of the same fp32 product."""
import pytest
import torch
import torch.nn.functional as F

from conv_cases import DGRAD_FP8, FPROP_FP8, WGRAD_FP8, WGRAD_GROUPS_FP8, conv_desc, dgrad_desc, fprop_desc

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
E4M3 = torch.float8_e4m3fn
E5M2 = torch.float8_e5m2


def _close(a, b, rtol, atol, what):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bad = int((err > atol + rtol * b.abs()).sum())
    assert bad == 0, "%s: %d/%d mismatches, max err %g (ref max %g)" % (what, bad, a.numel(), float(err.max()), float(b.abs().max()))


def _rand_fp8(shape, g, spread=1.0):
    """random e4m3 bytes with a wide spread of exponents (and exact zeros), as uint8 + their fp32 values"""
    v = torch.randn(shape, generator=g) * spread * torch.exp2(torch.randint(-6, 4, shape, generator=g).float())
    v = torch.where(torch.rand(shape, generator=g) < 0.1, torch.zeros(()), v).clamp(-448, 448)
    q = v.to(E4M3)
    return q.view(torch.uint8), q.float()


def test_fp8_mfma_operand_layout_with_exact_integers(ops):
    """K-permutation check of the 128-deep MFMA step: integer-valued operands (exact in e4m3 and in fp32 sums) whose every K
    position carries a different weight -- a lane map that does not pair the SAME k of both operands gives wrong integers."""
    g = torch.Generator().manual_seed(1)
    n, h, w, cin, cout = 1, 8, 16, 256, 64
    x = torch.randint(-3, 4, (n, h, w, cin), generator=g).float()
    wt = torch.zeros(cout, cin)
    for co in range(cout):                                       # row co: +-1 / +-2 at positions that depend on co and k (asymmetric)
        wt[co] = ((torch.arange(cin) * 7 + co * 13) % 5 - 2).float()
    ref = (x.reshape(-1, cin) @ wt.t())
    d = ops.conv_desc(n, h, w, cin, 1, 1, 1, 0, 0, h, w, cout)
    x8, w8 = x.to(E4M3).view(torch.uint8).cuda(), wt.to(E4M3).view(torch.uint8).cuda()
    one = torch.ones(1, device="cuda")
    y = torch.empty(n * h * w, cout, dtype=BF, device="cuda")
    ops.conv2d_fprop_fp8(d, x8, w8, one, torch.ones(cout, device="cuda"), y)
    torch.cuda.synchronize()
    assert float(ref.abs().max()) < 256                          # integers below 2^8: exact in bf16
    assert torch.equal(y.float().cpu(), ref), "fp8 MFMA: wrong K pairing (max err %g)" % float((y.float().cpu() - ref).abs().max())


@pytest.mark.parametrize("case", FPROP_FP8, ids=[c["id"] for c in FPROP_FP8])
def test_conv_fprop_fp8_vs_fp32_on_dequantised_operands(ops, case):
    g = torch.Generator().manual_seed(0)
    n, h, w, cin, cout, k, s, p = (case[x] for x in ("n", "h", "w", "cin", "cout", "k", "s", "p"))
    x8, xf = _rand_fp8((n, h, w, cin), g)
    w8, wf = _rand_fp8((cout, k, k, cin), g, spread=0.5)        # OHWI
    x_scale = torch.tensor([0.0371])
    w_scale = torch.rand(cout, generator=g) * 0.02 + 0.001
    bias = torch.randn(cout, generator=g)
    xd = xf * x_scale
    wd = wf * w_scale.view(-1, 1, 1, 1)
    ref = F.conv2d(xd.permute(0, 3, 1, 2), wd.permute(0, 3, 1, 2), bias if case["bias"] else None, stride=s, padding=p)
    if case["relu"]:
        ref = F.relu(ref)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    ho, wo = ref.shape[1], ref.shape[2]
    d = fprop_desc(ops, case, "cuda")
    y = torch.full((n, ho, wo, cout), float("nan"), dtype=BF, device="cuda")
    stats = torch.zeros(ops.conv_stat_tiles(d), 2, cout, dtype=torch.float64, device="cuda")
    ops.conv2d_fprop_fp8(d, x8.cuda(), w8.cuda(), x_scale.cuda(), w_scale.cuda(), y, bias=bias.cuda(), stats=stats if case["stats"] else None)
    got = ops.last_conv_instantiation().split(" grid")[0]
    assert got == ops.conv2d_describe_fp8(d).split(" grid")[0] and "F8=1" in got, (got, ops.conv2d_describe_fp8(d))
    torch.cuda.synchronize()
    yc, rc = y.float().cpu().reshape(-1, cout), ref.reshape(-1, cout)
    # the sums cancel (random signs over K up to 9216): absolute term scaled to the typical magnitude of a term-sum
    atol = 2e-3 * float(rc.abs().max())
    _close(yc, rc, 2 ** -7, atol, "fp8 conv output")
    rel = float((yc - rc).norm() / rc.norm())
    assert rel < 3e-3, "fp8 conv: relative L2 error %g beyond bf16 output rounding" % rel
    if case["stats"]:
        _close(stats[:, 0].sum(0), yc.double().sum(0), 1e-4, 1e-2 * float(rc.abs().max()), "stats sum")
        _close(stats[:, 1].sum(0), (yc.double() * yc.double()).sum(0), 1e-4, 1e-2 * float(rc.abs().max()) ** 2, "stats sumsq")


def test_fp8_mfma_mixed_formats_with_exact_integers(ops):
    """Data-gradient form: the x operand in e5m2, the weights in e4m3 (format selectors of the scaled MFMA).  Integers up to 4 are
    exact in e5m2; a swapped format selector decodes the bytes of one operand in the other's format and misses by far."""
    g = torch.Generator().manual_seed(2)
    n, h, w, cin, cout = 1, 8, 16, 256, 64
    x = torch.randint(-4, 5, (n, h, w, cin), generator=g).float()
    wt = torch.zeros(cout, cin)
    for co in range(cout):
        wt[co] = ((torch.arange(cin) * 5 + co * 11) % 7 - 3).float()
    ref = x.reshape(-1, cin) @ wt.t()
    d = ops.conv_desc(n, h, w, cin, 1, 1, 1, 0, 0, h, w, cout)
    y = torch.empty(n * h * w, cout, dtype=BF, device="cuda")
    ops.conv2d_dgrad_fp8(d, x.to(E5M2).view(torch.uint8).cuda(), wt.to(E4M3).view(torch.uint8).cuda(), torch.ones(1, device="cuda"),
                         torch.ones(cout, device="cuda"), y)
    torch.cuda.synchronize()
    assert float(ref.abs().max()) < 256 * 8
    err = float((y.float().cpu() - ref).abs().max())
    assert err <= 2 ** -8 * float(ref.abs().max()), "mixed-format MFMA: max err %g" % err


@pytest.mark.parametrize("case", DGRAD_FP8, ids=[c["id"] for c in DGRAD_FP8])
def test_conv_dgrad_fp8_vs_fp32_on_dequantised_operands(ops, case):
    """As tests/test_gpu_conv.py::test_conv_dgrad, with the gradient in e5m2 and the transposed weights in e4m3: gx against an fp32
    convolution of the dequantised operands (+ residual, masks, scatter), the fused BatchNorm-backward sums against an fp64
    evaluation of their definition on the kernel's own bf16 gx."""
    g = torch.Generator().manual_seed(11)
    n, h, w, cin, cout, k, sc = (case[x] for x in ("n", "h", "w", "cin", "cout", "k", "scatter"))
    d = dgrad_desc(ops, case, "cuda")
    oh, ow = d.out_h, d.out_w
    m, mo = n * h * w, n * oh * ow
    v = torch.randn(n, h, w, cin, generator=g) * torch.exp2(torch.randint(-8, 6, (n, h, w, cin), generator=g).float())
    dz8q = v.clamp(-57344, 57344).to(E5M2)
    w8q = (torch.randn(cout, k, k, cin, generator=g) * 0.5 * torch.exp2(torch.randint(-5, 3, (cout, k, k, cin), generator=g).float())).clamp(-448, 448).to(E4M3)
    dz_scale = torch.tensor([2.3e-3])
    w_scale = torch.rand(cout, generator=g) * 0.02 + 0.001
    dzf = dz8q.float() * dz_scale
    wf = w8q.float() * w_scale.view(-1, 1, 1, 1)                       # the kernel's "w_t" operand, OHWI
    conv = F.conv2d(dzf.permute(0, 3, 1, 2), wf.permute(0, 3, 1, 2), None, stride=1, padding=k // 2).permute(0, 2, 3, 1).contiguous()
    amp = float(conv.abs().max())
    base = (torch.randn(n, oh, ow, cout, generator=g) * 0.3 * amp).to(BF).float() if (case["res"] or sc > 1) else None
    rmask = torch.randint(0, 256, (mo, cout // 8), generator=g, dtype=torch.uint8) if case["res_mask"] else None
    res_eff = base
    if rmask is not None:
        bits = ((rmask[:, :, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(n, oh, ow, cout).bool()
        res_eff = torch.where(bits, base, torch.zeros_like(base))
    if sc == 1:
        ref = conv + (res_eff if case["res"] else 0)
        touched = torch.ones(mo, dtype=torch.bool)
    else:
        ref = base.clone() if case["res"] else torch.zeros(n, oh, ow, cout)
        ref[:, ::sc, ::sc][:, :h, :w] = conv + (res_eff[:, ::sc, ::sc][:, :h, :w] if case["res"] else 0)
        t2 = torch.zeros(n, oh, ow, dtype=torch.bool)
        t2[:, ::sc, ::sc][:, :h, :w] = True
        touched = t2.reshape(-1)
    z = (torch.randn(mo, cout, generator=g) * 2 + 0.5).to(BF).float()
    zmask = torch.randint(0, 256, (mo, cout // 8), generator=g, dtype=torch.uint8) if case["mask"] else None
    mean = z.mean(0).contiguous()
    invstd = (1.0 / (z.var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
    if case["res"]:
        out = base.to(BF).cuda()
        res_d = out
    else:
        out = torch.zeros(n, oh, ow, cout, dtype=BF, device="cuda") if sc > 1 else torch.full((n, oh, ow, cout), float("nan"), dtype=BF, device="cuda")
        res_d = None
    part = torch.zeros(ops.STAT_SLOTS, 2, cout, device="cuda")
    z_d = z.to(BF).cuda()
    zmask_d, rmask_d = (zmask.cuda() if zmask is not None else None), (rmask.cuda() if rmask is not None else None)
    mean_d, invstd_d = mean.cuda(), invstd.cuda()
    red = ops.bn_reduce_args(z_d, zmask_d, mean_d, invstd_d, part) if case["red"] else None
    ops.conv2d_dgrad_fp8(d, dz8q.view(torch.uint8).cuda(), w8q.view(torch.uint8).cuda(), dz_scale.cuda(), w_scale.cuda(), out, red=red, res=res_d,
                         res_mask=rmask_d)
    got = ops.last_conv_instantiation().split(" grid")[0]
    assert got == ops.conv2d_describe_dgrad_fp8(d, case["red"]).split(" grid")[0] and "F8=2" in got, got
    torch.cuda.synchronize()
    oc, rc = out.float().cpu().reshape(mo, cout), ref.reshape(mo, cout)
    # with a residual the kernel rounds the convolution term to bf16 before the add (as the bf16 kernel does): up to 2^-8 of |conv| more
    _close(oc, rc, 2 ** -7, (6e-3 if case["res"] else 2e-3) * amp, "fp8 data gradient")
    assert float((oc - rc).norm() / rc.norm()) < 3e-3
    if case["red"]:
        gm = oc.double()[touched]
        if zmask is not None:
            zb = ((zmask[:, :, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(mo, cout).bool()[touched]
            gm = torch.where(zb, gm, torch.zeros_like(gm))
        zt = z.double()[touched]
        sg = gm.sum(0)
        sgz = invstd.double() * ((gm * zt).sum(0) - mean.double() * sg)
        got_p = part.sum(0).double().cpu()
        for name, a, b in (("sum g*m", got_p[0], sg), ("sum g*m*xhat", got_p[1], sgz)):
            scale = float(b.abs().max()) + 1e-6
            err = float((a - b).abs().max()) / scale
            assert err < 2e-4, "%s: fused reduce %s off by %.3g of its scale" % (case["id"], name, err)


def test_bn_bwd_apply_writes_the_e5m2_twin(ops):
    g = torch.Generator().manual_seed(6)
    m, c = 911, 128
    dev = "cuda"
    z = (torch.randn(m, c, generator=g) * 2 + 0.5).to(BF)
    gout = (torch.randn(m, c, generator=g) * 1e-3).to(BF)
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev)
    mean, invstd = z.float().mean(0).to(dev), (1.0 / (z.float().var(0, unbiased=False) + 1e-5).sqrt()).to(dev)
    mask = torch.randint(0, 256, (m, c // 8), generator=g, dtype=torch.uint8).to(dev)
    nb = ops.bn_bwd_blocks(m)
    partial = torch.zeros(nb, 2, c, device=dev)
    ops.bn_bwd_reduce(gout.to(dev), None, z.to(dev), mean, invstd, partial, m, c, relu_mask=mask)
    outs = []
    qs = torch.tensor([3.0e5], device=dev)
    for with_f8 in (False, True):
        dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
        dz = torch.empty(m, c, dtype=BF, device=dev)
        dz8 = torch.zeros(m, c, dtype=torch.uint8, device=dev)
        amax = torch.zeros(ops.FP8_AMAX_SLOTS, device=dev)
        ops.bn_bwd_apply_fused(gout.to(dev), None, z.to(dev), mean, invstd, gamma, partial, nb, dg, db, dz, None, m, c, relu_mask=mask,
                               f8=ops.fp8_out(dz8, qs, amax) if with_f8 else None)
        torch.cuda.synchronize()
        outs.append(dz.clone())
    assert torch.equal(outs[0], outs[1]), "the e5m2 twin must not change dz"
    exp = (outs[1].float().cpu() * 3.0e5).clamp(-57344, 57344).to(E5M2).view(torch.uint8)
    got = dz8.cpu()
    same = (got == exp) | ((got & 0x7F) == 0) & ((exp & 0x7F) == 0)
    assert bool(same.all()), "e5m2 twin: %d bytes differ" % int((~same).sum())
    assert float(amax.max()) == float(outs[1].float().abs().max())


def test_weight_quantiser_bf16_source(ops):
    g = torch.Generator().manual_seed(8)
    wt = (torch.randn(256, 3 * 3 * 128, generator=g) * 0.03).to(BF).cuda()
    w8 = torch.zeros(256, 3 * 3 * 128, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(256, device="cuda")
    table, total = ops.make_weight_quant_table([(wt, w8, sc)], "cuda")
    ops.quantize_weights_fp8_batched(table, total)
    torch.cuda.synchronize()
    wc = wt.float().cpu()
    exp_sc = wc.abs().amax(1) * torch.tensor(1.0 / 448.0, dtype=torch.float32)
    assert torch.equal(sc.cpu(), exp_sc)
    exp8 = (wc * (1.0 / exp_sc).view(-1, 1)).clamp(-448, 448).to(E4M3).view(torch.uint8)
    got = w8.cpu()
    assert bool(((got == exp8) | ((got & 0x7F) == 0) & ((exp8 & 0x7F) == 0)).all())


def _ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous()


def _wgrad_fp8_problem(case, g, integers=False):
    """x8 (e4m3), dz8 (e5m2) bytes, their scales and the fp32 weight gradient of the dequantised operands"""
    n, h, w, cin, cout, k, s, p = (case[x] for x in ("n", "h", "w", "cin", "cout", "k", "s", "p"))
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    if integers:                                           # small integers: exact in both formats, exact fp32 sums
        xq = torch.randint(-3, 4, (n, cin, h, w), generator=g).float().to(E4M3)
        zq = torch.randint(-2, 3, (n, cout, ho, wo), generator=g).float().to(E5M2)
        xs, zs = torch.tensor([1.0]), torch.tensor([1.0])
    else:
        xq = (torch.randn(n, cin, h, w, generator=g) * torch.exp2(torch.randint(-5, 4, (n, cin, h, w), generator=g).float())).clamp(-448, 448).to(E4M3)
        zq = (torch.randn(n, cout, ho, wo, generator=g) * torch.exp2(torch.randint(-8, 6, (n, cout, ho, wo), generator=g).float())).clamp(-57344, 57344).to(E5M2)
        xs, zs = torch.tensor([3.1e-2]), torch.tensor([2.3e-3])
    wt = torch.zeros(cout, cin, k, k, requires_grad=True)
    y = F.conv2d(xq.float() * xs, wt, stride=s, padding=p)
    y.backward(zq.float() * zs)
    x8 = xq.view(torch.uint8).permute(0, 2, 3, 1).contiguous().cuda()
    z8 = zq.view(torch.uint8).permute(0, 2, 3, 1).contiguous().cuda()
    return x8, z8, xs.cuda(), zs.cuda(), _ohwi(wt.grad), n * ho * wo


@pytest.mark.parametrize("case", WGRAD_FP8, ids=[c["id"] for c in WGRAD_FP8])
def test_conv_wgrad_fp8_exact_on_integers(ops, case):
    """Small-integer operands (exact in e4m3 / e5m2, sums exact in fp32): the weight gradient must equal the integer result bit for
    bit -- this pins the ds_read_b64_tr_b8 transposition, the 16-byte-chunk swizzle, the shared K permutation of the two MFMA
    operands, the im2col walk and the pixel tail of the 128-pixel slices all at once."""
    g = torch.Generator().manual_seed(21)
    x8, z8, xs, zs, ref, m = _wgrad_fp8_problem(case, g, integers=True)
    d = conv_desc(ops, case)
    dw = torch.zeros(case["cout"], case["k"], case["k"], case["cin"], device="cuda")
    ops.conv2d_wgrad_fp8(d, x8, z8, xs, zs, dw)
    assert ops.last_conv_instantiation().split(" grid")[0] == ops.conv2d_wgrad_describe_fp8(d).split(" grid")[0]
    assert ",F8>" in ops.last_conv_instantiation()
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu(), ref), "fp8 wgrad on integers: %d mismatches, max err %g" % (int((dw.cpu() != ref).sum()), float((dw.cpu() - ref).abs().max()))


@pytest.mark.parametrize("case", WGRAD_FP8, ids=[c["id"] for c in WGRAD_FP8])
def test_conv_wgrad_fp8_vs_fp32_on_dequantised_operands(ops, case):
    """Random fp8 operands with a wide exponent spread: every e4m3 x e5m2 product is exact in fp32 and the MFMA accumulates in fp32,
    so the tolerance is the bf16 weight-gradient test's accumulation-order term (scaled by the operands' magnitudes), not an fp8 one."""
    g = torch.Generator().manual_seed(22)
    x8, z8, xs, zs, ref, m = _wgrad_fp8_problem(case, g)
    d = conv_desc(ops, case)
    dw = torch.zeros(case["cout"], case["k"], case["k"], case["cin"], device="cuda")
    ops.conv2d_wgrad_fp8(d, x8, z8, xs, zs, dw)
    torch.cuda.synchronize()
    _close(dw, ref, 2e-4, 2e-4 * float(ref.abs().max()) + 1e-12, "fp8 wgrad")


def test_conv_wgrad_fp8_shared_dw_accumulates_over_launches(ops):
    """fp8 twin of test_conv_wgrad_shared_dw_accumulates_over_launches: a level of <= 128 pixels is one 128-pixel slice, hence one
    split; with FRCNN_CONV_WGRAD_ACCUMULATE it must add to what the larger levels left in dw (integers: exact)."""
    g = torch.Generator().manual_seed(56)
    cin, cout, k = 256, 256, 3
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    ref = torch.zeros(cout, k, k, cin)
    for lv in (dict(n=1, h=32, w=32), dict(n=1, h=16, w=16), dict(n=1, h=8, w=8), dict(n=2, h=8, w=8)):
        case = dict(lv, cin=cin, cout=cout, k=k, s=1, p=1)
        x8, z8, xs, zs, r, m = _wgrad_fp8_problem(case, g, integers=True)
        ops.conv2d_wgrad_fp8(conv_desc(ops, case, flags=ops.CONV_WGRAD_ACCUMULATE), x8, z8, xs, zs, dw)
        ref += r
    torch.cuda.synchronize()
    assert torch.equal(dw.cpu(), ref), "fp8 wgrad shared over launches: max err %g" % float((dw.cpu() - ref).abs().max())


@pytest.mark.parametrize("grp", WGRAD_GROUPS_FP8, ids=[c["id"] for c in WGRAD_GROUPS_FP8])
def test_conv_wgrad_grouped_with_fp8_layers(ops, grp):
    """bf16 and fp8 layers in one group table: one launch per (precision, addressing mode) that has layers."""
    g = torch.Generator().manual_seed(23)
    items, refs = [], []
    for case in grp["layers"]:
        dw = torch.zeros(case["cout"], case["k"], case["k"], case["cin"], device="cuda")
        if case["f8"]:
            x8, z8, xs, zs, ref, m = _wgrad_fp8_problem(case, g)
            items.append((conv_desc(ops, case), x8, z8, dw, xs, zs))
        else:
            x = torch.randn(case["n"], case["cin"], case["h"], case["w"], generator=g).to(BF).float()
            wt = torch.zeros(case["cout"], case["cin"], case["k"], case["k"], requires_grad=True)
            y = F.conv2d(x, wt, stride=case["s"], padding=case["p"])
            dz = torch.randn(y.shape, generator=g).to(BF).float()
            y.backward(dz)
            ref = _ohwi(wt.grad)
            items.append((conv_desc(ops, case), x.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), dz.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), dw))
        refs.append(ref)
    group = ops.WgradGroup(items, "cuda")
    ops.conv2d_wgrad_grouped(group)
    got = ops.last_conv_instantiation()
    want = ops.conv2d_wgrad_describe(group=group)
    assert [p.split(" grid")[0] for p in got.split("; ")] == [p.split(" grid")[0] for p in want.split("; ")], (got, want)
    assert ",F8>" in got
    torch.cuda.synchronize()
    for it, ref in zip(items, refs):
        _close(it[3], ref, 2e-4, 2e-4 * float(ref.abs().max()) + 1e-12, "grouped wgrad (%s)" % ("fp8" if len(it) > 4 else "bf16"))


def test_quantize_fp8_bit_exact_and_amax(ops):
    g = torch.Generator().manual_seed(3)
    n = 8 * 1000 + 8 * 37
    x = (torch.randn(n, generator=g) * torch.exp2(torch.randint(-10, 10, (n,), generator=g).float())).to(BF)
    x[:8] = torch.tensor([0.0, -0.0, 1e30, -1e30, 448.0, 464.0, 2.0 ** -9, 2.0 ** -11]).to(BF)    # zeros, saturation, subnormals
    qs = torch.tensor([3.17])
    out8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    amax = torch.zeros(ops.FP8_AMAX_SLOTS, device="cuda")
    ops.quantize_fp8(x.cuda(), qs.cuda(), out8, amax)
    torch.cuda.synchronize()
    exp = (x.float() * qs).clamp(-448, 448).to(E4M3).view(torch.uint8)
    got = out8.cpu()
    same = (got == exp) | ((got & 0x7F) == 0) & ((exp & 0x7F) == 0)       # (+0 / -0 both mean zero)
    assert bool(same.all()), "quantize_fp8: %d of %d bytes differ from torch's e4m3fn conversion" % (int((~same).sum()), n)
    assert float(amax.max()) == float(x.float().abs().max())
    ops.quantize_fp8(x.cuda(), qs.cuda(), out8, None, e5m2=True)
    torch.cuda.synchronize()
    exp = (x.float() * qs).clamp(-57344, 57344).to(E5M2).view(torch.uint8)
    got = out8.cpu()
    same = (got == exp) | ((got & 0x7F) == 0) & ((exp & 0x7F) == 0)
    assert bool(same.all()), "quantize_fp8 (e5m2): %d of %d bytes differ" % (int((~same).sum()), n)


def test_weight_quantiser_per_row_scales(ops):
    g = torch.Generator().manual_seed(4)
    shapes = [(64, 3, 3, 128), (256, 1, 1, 1024), (24, 1, 1, 256)]
    entries, masters = [], []
    for (co, kh, kw, ci) in shapes:
        wm = (torch.randn(co, kh, kw, ci, generator=g) * 0.05).cuda()
        wm[1] = 0                                                 # an all-zero row: scale 1, bytes 0
        w8 = torch.full((co, kh, kw, ci), 0xFF, dtype=torch.uint8, device="cuda")
        sc = torch.zeros(co, device="cuda")
        entries.append((wm.view(co, -1), w8, sc))
        masters.append(wm)
    table, total = ops.make_weight_quant_table(entries, "cuda")
    ops.quantize_weights_fp8_batched(table, total)
    torch.cuda.synchronize()
    for (wm2, w8, sc), wm in zip(entries, masters):
        wc = wm2.cpu()
        mx = wc.abs().amax(1)
        exp_sc = torch.where(mx > 0, mx * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones(()))
        assert torch.equal(sc.cpu(), exp_sc)
        exp8 = (wc * (1.0 / exp_sc).view(-1, 1)).clamp(-448, 448).to(E4M3).view(torch.uint8)
        got = w8.cpu().view(wc.shape)
        same = (got == exp8) | ((got & 0x7F) == 0) & ((exp8 & 0x7F) == 0)
        assert bool(same.all()), "weight bytes differ"
        # the largest entry of every row maps to +-448
        assert bool(((got.view(torch.float8_e4m3fn).float().abs().amax(1) == 448) | (mx == 0)).all())


def test_bn_train_apply_writes_the_fp8_twin(ops):
    g = torch.Generator().manual_seed(5)
    m, c = 777, 192
    z = (torch.randn(m, c, generator=g) * 2 + 0.5).to(BF)
    z2 = (torch.randn(m, c, generator=g)).to(BF)
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    dev = "cuda"
    parts = torch.zeros(16, 2, c, dtype=torch.float64, device=dev)
    parts[0, 0], parts[0, 1] = z.double().sum(0).to(dev), (z.double() ** 2).sum(0).to(dev)
    parts2 = torch.zeros(16, 2, c, dtype=torch.float64, device=dev)
    parts2[0, 0], parts2[0, 1] = z2.double().sum(0).to(dev), (z2.double() ** 2).sum(0).to(dev)
    new = lambda: (torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.empty(c, device=dev), torch.empty(c, device=dev))
    qs = torch.tensor([37.5], device=dev)
    for dual in (False, True):
        outs = []
        for with_f8 in (False, True):
            mm, mv, mean, invstd = new()
            out = torch.empty(m, c, dtype=BF, device=dev)
            out8 = torch.zeros(m, c, dtype=torch.uint8, device=dev)
            amax = torch.zeros(ops.FP8_AMAX_SLOTS, device=dev)
            f8 = ops.fp8_out(out8, qs, amax) if with_f8 else None
            if dual:
                mm2, mv2, mean2, invstd2 = new()
                ops.bn_train_apply_dual(z.to(dev), parts, gamma.to(dev), beta.to(dev), mm, mv, mean, invstd, z2.to(dev), parts2, beta.to(dev) + 1,
                                        gamma.to(dev) - 1, mm2, mv2, mean2, invstd2, 16, m, 0.99, 1.001e-5, out, m, c, relu=True, f8=f8)
            else:
                ops.bn_train_apply(z.to(dev), parts, 16, m, gamma.to(dev), beta.to(dev), mm, mv, 0.99, 1.001e-5, out, mean, invstd, m, c,
                                   res=z2.to(dev), relu=True, f8=f8)
            torch.cuda.synchronize()
            outs.append(out.clone())
        assert torch.equal(outs[0], outs[1]), "the fp8 twin must not change the bf16 output"
        exp = (outs[1].float().cpu() * 37.5).clamp(-448, 448).to(E4M3).view(torch.uint8)
        got = out8.cpu()
        same = (got == exp) | ((got & 0x7F) == 0) & ((exp & 0x7F) == 0)
        assert bool(same.all()), "fp8 twin (dual=%s): %d bytes differ" % (dual, int((~same).sum()))
        assert float(amax.max()) == float(outs[1].float().abs().max())


@pytest.mark.parametrize("n", [8 * 4096, 8 * 37])        # every lane of the writing waves active / a tail wave with exited lanes
def test_amax_records_a_nan_without_an_inf(ops, n):
    """fmaxf drops a NaN operand, so a tensor holding NaN but no Inf used to leave a FINITE amax behind and
    frcnn_fp8_update_scales never counted it.  atomic_amax now reduces the bit patterns of |x| (every NaN orders above +Inf): the slot
    row of such a tensor holds a NaN, the scale is kept and status[1] counts the event; a clean tensor beside it is updated as before."""
    x = torch.linspace(-3.0, 5.0, n).to(BF)
    clean = x.clone()
    x[n // 2 + 3] = float("nan")
    assert not bool(torch.isinf(x.float()).any())
    qs = torch.tensor([1.0]).cuda()
    out8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    amax = torch.zeros(2, ops.FP8_AMAX_SLOTS, device="cuda")
    ops.quantize_fp8(x.cuda(), qs, out8, amax[0])
    ops.quantize_fp8(clean.cuda(), qs, out8, amax[1])
    torch.cuda.synchronize()
    assert bool(torch.isnan(amax[0]).any()) and float(amax[1].max()) == 5.0
    scale, qscale = torch.tensor([7.0, 7.0], device="cuda"), torch.tensor([9.0, 9.0], device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.fp8_update_scales(amax, scale, qscale, 2, margin=1.0, status=status)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0, 1], status
    assert float(scale[0]) == 7.0 and float(qscale[0]) == 9.0                      # non-finite amax: the scale is kept
    assert abs(float(scale[1]) - 5.0 / 448.0) < 1e-7


def test_fp8_delayed_scaling_update(ops):
    amax = torch.zeros(3, ops.FP8_AMAX_SLOTS, device="cuda")
    amax[0, 5], amax[0, 63], amax[2, 0] = 4.48, 1.0, 896.0        # (the maximum over a tensor's slots counts)
    scale = torch.tensor([7.0, 7.0, 7.0], device="cuda")
    qscale = torch.tensor([9.0, 9.0, 9.0], device="cuda")
    ops.fp8_update_scales(amax, scale, qscale, 3, margin=1.0)
    torch.cuda.synchronize()
    assert torch.allclose(scale.cpu(), torch.tensor([0.01, 7.0, 2.0])) and torch.allclose(qscale.cpu(), torch.tensor([100.0, 9.0, 0.5]))


# ---------------------------------------------------------------------------------------------------------------------------
# End to end: one training step with precision="fp8" against the same step in bf16, from identical weights and inputs.  This
# is a DEVIATION measurement (quoted in DESIGN.md section 5), not a parity gate: e4m3 has 3 significand bits, so the fp8 forward
# pass is a different (quantised) function; the gates on the kernels themselves are the dequantised-operand tests above.
def _small_step(precision, steps=3):
    import importlib
    from oracle import faster_rcnn as O
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    OPT = importlib.import_module("2d_object_detection_amd.optimizers")
    cfg = O.default_config((192, 256, 3))
    cfg["rpn"]["anchors"]["base_anchor_shape"] = [32, 32]
    cfg["rpn"]["nms"].update(max_total_size=40, max_output_size_per_class=40)
    cfg["rpn"]["sampling"]["num_samples"] = 32
    cfg["rcnn"]["sampling"]["num_samples"] = 16
    params = O.init_params(cfg, seed=3, randomize_affine=True)
    for k in params:
        if k.endswith("/kernel"):
            params[k] = params[k].to(BF).float()
        if k.endswith("_3_bn/gamma"):
            params[k] = params[k] * 0.25            # (the regime of trained nets: rounding noise is not amplified 100x)
    images, gl, gb = O.synthetic_batch(2, cfg["image_shape"], seed=5)
    model = M.FasterRCNN(cfg, sampling_seed=11, precision=precision)
    model.use_graphs = False
    model.set_weights(params)
    opt = OPT.SGD(learning_rate=1e-4, momentum=0.9)
    hist = []
    for _ in range(steps):
        losses, _ = model.train_step(images.cuda(), gl.cuda(), gb.cuda(), opt)
        torch.cuda.synchronize()
        hist.append({k: float(v) for k, v in losses.items()})
    aux = model._train_plan["aux"]
    return {"feat": aux["feature_maps"].float().cpu(), "losses": hist, "g": model.store.g.cpu().clone(), "buckets": list(model.store.buckets),
            "launches": model._train_plan["plan"].num_launches, "model": model}


def _backbone_fwd_bwd(precision):
    """ResNet-50 backbone alone, forward + backward from a FIXED feature-map gradient (nothing discrete in between: the full step's
    proposals and samples differ as soon as a score moves, which makes its head gradients incomparable).  Two runs: the first
    calibrates the delayed scales."""
    import importlib
    FE = importlib.import_module("2d_object_detection_amd.models.feature_extractor")
    RT = importlib.import_module("2d_object_detection_amd.runtime")
    shape = (192, 256, 3)
    fe = FE.FeatureExtractor(shape, depth=50, device="cuda", precision=precision)
    g = torch.Generator().manual_seed(5)
    for u in fe.conv_units():
        fe.store.weight(u.name + "_bn/gamma").copy_((torch.rand(u.cout, generator=g) + 0.5) * (0.25 if u.name.endswith("_3") else 1.0))
        fe.store.weight(u.name + "_bn/beta").copy_(torch.randn(u.cout, generator=g) * 0.1)
    batch = 2
    fe.setup(batch, True)
    fe.images.copy_(torch.randint(0, 256, (batch,) + shape, generator=g, dtype=torch.uint8))
    fe.store.refresh_bf16()
    _, gh, gw, cf = fe.output_shape
    g_feat = (torch.randn(batch * gh * gw, cf, generator=g) * 1e-2).to(BF).cuda()
    plan = RT.Plan("backbone")
    plan.zero(fe.store.g)
    fe.refresh_weights(plan)
    fe.forward_plan(plan, True)
    fe.backward_plan(plan, g_feat, g_feat_reduced=False)
    if fe.f8 is not None:
        fe.f8.plan_update(plan)
    for _ in range(2):
        plan.run()
        torch.cuda.synchronize()
    first = fe.specs[0][0]
    return {"feat": fe.feature_maps.float().cpu(), "gin": fe.acts[first]["gin"].float().cpu(), "g": fe.store.g.cpu().clone(),
            "buckets": list(fe.store.buckets), "fe": fe}


def test_fp8_backbone_deviation_from_bf16():
    import importlib
    FE = importlib.import_module("2d_object_detection_amd.models.feature_extractor")
    ref, f8 = _backbone_fwd_bwd("bf16"), _backbone_fwd_bwd("fp8")
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-20))
    fe = f8["fe"]
    n_f8 = sum(1 for u in fe.conv_units() if u.fp8), sum(1 for u in fe.conv_units() if u.fp8_bwd and u.dz8 is not None)
    cosines = {n: round(cos(f8["g"][b:e], ref["g"][b:e]), 4) for n, b, e in ref["buckets"]}
    print("fp8 vs bf16 backbone (R50, 192x256, batch 2; %d fp8 forward convs, %d fp8 data gradients): feature maps rel L2 %.4f, "
          "block-input gradient rel L2 %.4f, parameter-gradient cosine per bucket %s" % (n_f8[0], n_f8[1], rel(f8["feat"], ref["feat"]),
                                                                                       rel(f8["gin"], ref["gin"]), cosines))
    assert n_f8[0] >= 25 and (n_f8[1] >= 25 or not FE.FP8_BWD)
    sc = fe.f8.buf[0, :fe.f8.n].cpu()
    assert bool((sc != 1.0).all()) and bool(torch.isfinite(sc).all()) and bool((sc > 0).all()), "delayed scaling did not calibrate every tensor"
    # e4m3 / e5m2 operands: 2^-4 / 2^-3 relative per element, averaged down by the contractions; ReLU masks flip under the forward
    # difference, which is what the gradient deviation mostly is
    # gates = 1.3x what MI355X measures (rounds 3 and 4: feature maps 0.105, block-input gradient rel L2 0.62, bucket cosines 0.80 .. 0.84;
    # gradients with the forward pass alone in fp8 (FRCNN_FP8_BWD=0) vs forward + backward: DESIGN.md 5)
    assert rel(f8["feat"], ref["feat"]) < 0.13
    assert rel(f8["gin"], ref["gin"]) < 0.8 and cos(f8["gin"], ref["gin"]) > 0.75, (rel(f8["gin"], ref["gin"]), cos(f8["gin"], ref["gin"]))
    assert min(cosines.values()) > 0.75, cosines
    assert fe.f8.status.cpu().tolist() == [0, 0], "a steady two-step run must not clamp or overflow: %s" % fe.f8.status.cpu().tolist()
    # the fp8 weight gradients' own share: the same fp8 run with the weight gradients alone back in bf16 (same forward pass, same
    # data gradients, same twins: the two runs differ only in the operands of the pixel contraction)
    n_wg = sum(1 for u in fe.conv_units() if u.fp8_wgrad and u.dz8 is not None)
    assert n_wg >= 25 or not FE.FP8_WGRAD
    if FE.FP8_WGRAD:
        FE.FP8_WGRAD = False
        try:
            f8_bw = _backbone_fwd_bwd("fp8")
        finally:
            FE.FP8_WGRAD = True
        own = {n: (round(cos(f8["g"][b:e], f8_bw["g"][b:e]), 4), round(rel(f8["g"][b:e], f8_bw["g"][b:e]), 4)) for n, b, e in ref["buckets"]}
        print("fp8 weight gradients (%d layers) vs bf16 weight gradients on the same fp8 forward / data-gradient pass: (cosine, rel L2) per bucket %s"
              % (n_wg, own))
        assert torch.equal(f8["feat"], f8_bw["feat"])
        assert min(c for c, _ in own.values()) > 0.99, own


def test_fp8_delayed_scaling_amax_jump_is_visible_and_bounded(ops):
    """Delayed scaling has ONE step of history: a tensor that outgrows margin x last step's amax is clamped by the quantiser.  That
    must not be silent.  Kernel level: step t calibrates on x, step t+1 feeds 8 x -- the twin saturates at limit x scale (a BOUNDED
    error: nothing wraps, nothing turns Inf / NaN), frcnn_fp8_update_scales counts the tensor in status[0], and step t+2 (same 8 x)
    is clean again.  A non-finite amax keeps the previous scale and is counted in status[1]."""
    g = torch.Generator().manual_seed(77)
    n = 1 << 16
    x = torch.randn(n, generator=g).to(BF).cuda()
    amax = torch.zeros(2, ops.FP8_AMAX_SLOTS, device="cuda")
    buf = torch.ones(2, 2, device="cuda")                        # [scale | 1 / scale] x 2 tensors (0: e4m3, 1: e5m2)
    limit = torch.tensor([448.0, 57344.0], device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    out = torch.zeros(2, n, dtype=torch.uint8, device="cuda")
    margin = 2.0

    def step(src):
        amax.zero_()
        for i, e5 in ((0, False), (1, True)):
            ops.quantize_fp8(src, buf[1, i:i + 1], out[i], amax[i], e5m2=e5)
        deq = [out[0].view(E4M3).float() * buf[0, 0], out[1].view(E5M2).float() * buf[0, 1]]
        ops.fp8_update_scales(amax, buf[0], buf[1], 2, margin, limit, status)
        torch.cuda.synchronize()
        return deq

    step(x)                                                      # t - 1: scale 1 -> calibrated
    a0 = float(x.float().abs().max())
    assert status.tolist() == [0, 0] and abs(float(buf[0, 0]) - margin * a0 / 448) < 1e-6
    deq = step(x)                                                # t: calibrated, nothing clamps
    assert status.tolist() == [0, 0]
    assert float((deq[0] - x.float()).abs().max()) <= a0 * 2 ** -3            # e4m3: 3 mantissa bits at the top binade
    x8 = (x.float() * 8).to(BF)
    sc_t = [float(buf[0, 0]), float(buf[0, 1])]
    deq = step(x8)                                               # t + 1: amax jumps 8 x > margin
    # e4m3 clamps at 448 x scale = margin x a0; e5m2 has 57344 / 448 = 128 x head-room: it does not
    assert status.tolist() == [1, 0], status.tolist()
    assert bool(torch.isfinite(deq[0]).all()) and abs(float(deq[0].abs().max()) - 448 * sc_t[0]) < 1e-3 * 448 * sc_t[0]
    inside = x8.float().abs() <= 448 * sc_t[0]
    assert float((deq[0] - x8.float())[inside].abs().max()) <= 448 * sc_t[0] * 2 ** -3       # unclamped values keep their precision
    assert float((deq[1] - x8.float()).abs().max()) <= 8 * a0 * 2 ** -2                        # e5m2 (2 mantissa bits): not clamped
    deq = step(x8)                                               # t + 2: the scale has followed
    assert status.tolist() == [1, 0]
    assert float((deq[0] - x8.float()).abs().max()) <= 8 * a0 * 2 ** -3
    # a tensor that overflowed to Inf: scale kept, event counted, later steps still finite
    keep = buf.clone()
    xi = x8.clone()
    xi[5] = float("inf")
    step(xi)
    assert status.tolist()[1] == 2 and torch.equal(buf, keep), (status.tolist(), buf, keep)
    deq = step(x8)
    assert bool(torch.isfinite(deq[0]).all()) and bool(torch.isfinite(buf).all())


def test_fp8_backbone_amax_jump_sets_the_status_word():
    """The same event in the model: every BatchNorm gamma x 8 between two steps makes the activations of step t+1 eight times what
    the delayed scales were measured on -- FasterRCNN.fp8_status()'s source (Fp8Scales.status[0]) must count the clamped twins, the
    outputs must stay finite, and one step later the run must be back at the steady-state deviation from bf16."""
    import importlib
    FE = importlib.import_module("2d_object_detection_amd.models.feature_extractor")
    RT = importlib.import_module("2d_object_detection_amd.runtime")
    shape, batch = (192, 256, 3), 2

    def build(precision):
        fe = FE.FeatureExtractor(shape, depth=50, device="cuda", precision=precision)
        g = torch.Generator().manual_seed(5)
        for u in fe.conv_units():
            fe.store.weight(u.name + "_bn/gamma").copy_((torch.rand(u.cout, generator=g) + 0.5) * (0.25 if u.name.endswith("_3") else 1.0))
            fe.store.weight(u.name + "_bn/beta").copy_(torch.randn(u.cout, generator=g) * 0.1)
        fe.setup(batch, True)
        fe.images.copy_(torch.randint(0, 256, (batch,) + shape, generator=g, dtype=torch.uint8))
        _, gh, gw, cf = fe.output_shape
        g_feat = (torch.randn(batch * gh * gw, cf, generator=g) * 1e-2).to(BF).cuda()
        plan = RT.Plan("backbone")
        plan.zero(fe.store.g)
        plan.add(fe.store.refresh_bf16)
        fe.refresh_weights(plan)
        fe.forward_plan(plan, True)
        fe.backward_plan(plan, g_feat, g_feat_reduced=False)
        if fe.f8 is not None:
            fe.f8.plan_update(plan)
        return fe, plan

    def scale_gammas(fe, k):
        for u in fe.conv_units():
            if not u.name.endswith("_3"):                     # (the inner activations a1 / a2 of every block: twins written by BatchNorm kernels)
                fe.store.weight(u.name + "_bn/gamma").mul_(k)

    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    fe, plan = build("fp8")
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize()
    assert fe.f8.status.cpu().tolist() == [0, 0]
    scale_gammas(fe, 8.0)
    plan.run()                                                   # step t + 1 on step t's scales
    torch.cuda.synchronize()
    clamped = int(fe.f8.status[0])
    assert clamped >= 10, "an 8x activation jump under margin 2 must clamp the inner-activation twins (status[0] = %d)" % clamped
    assert int(fe.f8.status[1]) == 0 and bool(torch.isfinite(fe.feature_maps.float()).all())
    plan.run()                                                   # step t + 2: the scales have followed
    plan.run()
    torch.cuda.synchronize()
    after = int(fe.f8.status[0])
    fe_ref, plan_ref = build("bf16")
    scale_gammas(fe_ref, 8.0)
    plan_ref.run()
    torch.cuda.synchronize()
    e = rel(fe.feature_maps.float().cpu(), fe_ref.feature_maps.float().cpu())
    print("fp8 amax jump (gamma x 8): %d twins clamped in the jump step, %d more in the two steps after; feature maps vs bf16 afterwards: rel L2 %.4f" % (
        clamped, after - clamped, e))
    assert e < 0.13, e


def test_fp8_twin_only_stores_change_nothing():
    """fp8 mode skips the bf16 store of tensors whose every reader takes the fp8 twin (BatchNorm backward outputs, inner activations).
    The twin is the image of the bf16-ROUNDED value either way, so forward pass, data gradients and parameter gradients must not move
    (up to the order of float-atomic sums in the backward partial sums and split weight gradients)."""
    import importlib
    FE = importlib.import_module("2d_object_detection_amd.models.feature_extractor")
    assert FE.FP8_DZ_TWIN_ONLY
    a = _backbone_fwd_bwd("fp8")
    n_skip_dz = sum(1 for u in a["fe"].conv_units() if getattr(u, "dz_twin_only", False))
    n_skip_act = sum(1 for n in a["fe"].acts for k in ("a1_twin_only", "a2_twin_only") if a["fe"].acts[n].get(k))
    assert n_skip_dz >= 20 and n_skip_act >= 10, (n_skip_dz, n_skip_act)
    FE.FP8_DZ_TWIN_ONLY = False
    try:
        b = _backbone_fwd_bwd("fp8")
    finally:
        FE.FP8_DZ_TWIN_ONLY = True
    assert not any(getattr(u, "dz_twin_only", False) for u in b["fe"].conv_units())
    rel = lambda x, y: float((x - y).norm() / (y.norm() + 1e-20))
    assert torch.equal(a["feat"], b["feat"]), "forward pass moved"
    # Two runs of ONE configuration already differ by the order of their float-atomic sums, and the train-mode backward pass of a
    # random-init ResNet amplifies that on its way down (measured, same configuration twice: conv4 2e-9, conv3 4e-7, conv2 + stem
    # 4e-3, block-input gradient 3e-3 .. 1.3e-2 -- in bf16 as in fp8).  The two modes must agree to that noise: tight where it is small.
    tol = {"conv4": 1e-6, "conv3": 1e-5}
    for n, lo, hi in a["buckets"]:
        assert rel(a["g"][lo:hi], b["g"][lo:hi]) < tol.get(n, 3e-2), (n, rel(a["g"][lo:hi], b["g"][lo:hi]))
    assert rel(a["gin"], b["gin"]) < 3e-2, rel(a["gin"], b["gin"])


def test_fp8_train_step_runs_and_stays_close():
    ref = _small_step("bf16")
    f8 = _small_step("fp8")
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-20))
    e_feat = rel(f8["feat"], ref["feat"])
    line = "fp8 vs bf16 after 3 steps: feature maps rel L2 %.3f; losses %s vs %s; gradient cosine per bucket %s" % (
        e_feat, {k: round(v, 4) for k, v in f8["losses"][-1].items()}, {k: round(v, 4) for k, v in ref["losses"][-1].items()},
        {n: round(cos(f8["g"][b:e], ref["g"][b:e]), 3) for n, b, e in ref["buckets"]})
    print(line)
    assert all(torch.isfinite(torch.tensor(list(h.values()))).all() for h in f8["losses"])
    # the fp8 step really ran fp8 kernels, and its scales moved off their initial 1.0 (delayed scaling is alive)
    fe = f8["model"]._train.fe
    assert fe.f8 is not None and fe.f8.n >= 20
    sc = fe.f8.buf[0, :fe.f8.n].cpu()
    assert bool((sc != 1.0).all()) and bool(torch.isfinite(sc).all()) and bool((sc > 0).all())
    assert e_feat < 0.25, e_feat                                      # (e4m3: ~2^-4 relative per element, partly averaged by the contraction)
    # rpn_cls averages 256 samples per image of ~9 k anchors and is stable; rcnn_cls is the mean over 64 sampled RoIs of a 3-step-old
    # head on proposals that differ between the two runs (observed over this round's boxes: fp8 0.385-0.472 against bf16 0.407-0.410)
    for k, tol in (("rpn_cls", 0.1), ("rcnn_cls", 0.25)):
        assert abs(f8["losses"][-1][k] - ref["losses"][-1][k]) < tol * abs(ref["losses"][-1][k]) + 0.02, (k, f8["losses"][-1], ref["losses"][-1])
    # (no gradient comparison here: proposals and samples are discrete functions of the scores -- once one differs, the head gradients
    # of the two runs belong to different samples; test_fp8_backbone_deviation_from_bf16 measures the backward path on a fixed gradient)
