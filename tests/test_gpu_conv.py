"""GPU parity of the MFMA conv family (through the C ABI) against torch-CPU fp32 on the same
bf16-rounded inputs.  Tolerances: outputs are bf16 (8 significand bits -> 2^-9 relative
rounding) of fp32-accumulated sums; fp32 outputs only differ by accumulation order."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rt(t):
    return t.to(BF).to(torch.float32)


def _ohwi(w_oihw):
    return w_oihw.permute(0, 2, 3, 1).contiguous()


def _close(a, b, rtol, atol, what):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    bad = int((err > bound).sum())
    assert bad == 0, "%s: %d/%d mismatches, max err %g (ref max %g)" % (what, bad, a.numel(), float(err.max()), float(b.abs().max()))


from conv_cases import DGRAD, F32, FPROP, WGRAD, WGRAD_GROUPS, conv_desc, dgrad_desc, f32_desc, fprop_desc


def _inst(ops):
    return ops.last_conv_instantiation().split(" grid")[0]


def _assert_dispatch(ops, expected_full, what):
    """the kernel that really ran == the one the dispatcher's dry run names (tests/test_conv_dispatch.py ties those names to
    the train plans' launches)"""
    got = _inst(ops)
    assert got == expected_full.split(" grid")[0], "%s: launched %s, dry run said %s" % (what, got, expected_full)


def _edge_masks(n, ho, wo):
    """boolean [n*ho*wo] masks of the pixels where a 3x3 window leaves the image row / the image"""
    ox = torch.arange(wo).repeat(n * ho)
    oy = torch.arange(ho).repeat_interleave(wo).repeat(n)
    return {"first pixel of an image row": ox == 0, "last pixel of an image row": ox == wo - 1, "first image row": oy == 0,
            "last image row": oy == ho - 1}


@pytest.mark.parametrize("case", FPROP, ids=[c["id"] for c in FPROP])
def test_conv_fprop(ops, case):
    g = torch.Generator().manual_seed(0)
    n, h, w, cin, cout, k, s, p = (case[x] for x in ("n", "h", "w", "cin", "cout", "k", "s", "p"))
    x = _rt(torch.randn(n, h, w, cin, generator=g))
    wt = _rt(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2), wt, bias if case["bias"] else None, stride=s, padding=p)
    if case["relu"]:
        ref = F.relu(ref)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    ho, wo = ref.shape[1], ref.shape[2]
    d = fprop_desc(ops, case, "cuda")
    assert (d.ho, d.wo) == (ho, wo)
    xd, wd, bd = x.to(BF).cuda(), _ohwi(wt).to(BF).cuda(), bias.cuda()
    y = torch.full((n, ho, wo, cout), float("nan"), dtype=BF, device="cuda")
    tiles = ops.conv_stat_tiles(d)
    stats = torch.zeros(tiles, 2, cout, dtype=torch.float64, device="cuda")
    ops.conv2d_fprop(d, xd, wd, y, bias=bd, stats=stats if case["stats"] else None)
    _assert_dispatch(ops, ops.conv2d_describe(d), case["id"])
    torch.cuda.synchronize()
    yc, rc = y.float().cpu().reshape(-1, cout), ref.reshape(-1, cout)
    if k == 3:
        for name, mk in _edge_masks(n, ho, wo).items():
            _close(yc[mk], rc[mk], 2 ** -7, 2e-2, "conv output, " + name)
    _close(yc, rc, 2 ** -7, 2e-2, "conv output")
    if case["stats"]:
        _close(stats[:, 0].sum(0), yc.double().sum(0), 1e-4, 1e-2, "stats sum")
        _close(stats[:, 1].sum(0), (yc.double() * yc.double()).sum(0), 1e-4, 1e-2, "stats sumsq")


@pytest.mark.parametrize("case", DGRAD, ids=[c["id"] for c in DGRAD])
def test_conv_dgrad(ops, case):
    """Data-gradient launches (the forward kernel on transposed, tap-flipped weights): gx = conv(dz, w_t) [+ res (* mask bits)],
    plain or scattered with stride 2, with or without the fused BatchNorm-backward reduce of the consuming layer.  gx against
    torch-CPU fp32; the partial sums against an fp64 evaluation of their definition on the kernel's own (bf16) gx."""
    g = torch.Generator().manual_seed(11)
    n, h, w, cin, cout, k, sc = (case[x] for x in ("n", "h", "w", "cin", "cout", "k", "scatter"))
    d = dgrad_desc(ops, case, "cuda")
    oh, ow = d.out_h, d.out_w
    m, mo = n * h * w, n * oh * ow
    dz = _rt(torch.randn(n, h, w, cin, generator=g))
    wt = _rt(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)           # the kernel's "w_t" operand (OIHW here)
    conv = F.conv2d(dz.permute(0, 3, 1, 2), wt, None, stride=1, padding=k // 2).permute(0, 2, 3, 1).contiguous()
    base = _rt(torch.randn(n, oh, ow, cout, generator=g)) if (case["res"] or sc > 1) else None
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
    z = _rt(torch.randn(mo, cout, generator=g) * 2 + 0.5)
    zmask = torch.randint(0, 256, (mo, cout // 8), generator=g, dtype=torch.uint8) if case["mask"] else None
    mean = z.mean(0).contiguous()
    invstd = (1.0 / (z.var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
    if case["res"]:
        out = base.to(BF).cuda()                        # the residual aliases the output, as in the training plan
        res_d = out
    else:
        out = torch.zeros(n, oh, ow, cout, dtype=BF, device="cuda") if sc > 1 else torch.full((n, oh, ow, cout), float("nan"), dtype=BF, device="cuda")
        res_d = None
    part = torch.zeros(ops.STAT_SLOTS, 2, cout, device="cuda")
    dz_d, w_d = dz.to(BF).cuda(), _ohwi(wt).to(BF).cuda()
    z_d = z.to(BF).cuda()
    # (frcnn_bn_reduce holds raw pointers: every operand stays referenced until the kernel has run)
    zmask_d, rmask_d = (zmask.cuda() if zmask is not None else None), (rmask.cuda() if rmask is not None else None)
    mean_d, invstd_d = mean.cuda(), invstd.cuda()
    if case["red"]:
        red = ops.bn_reduce_args(z_d, zmask_d, mean_d, invstd_d, part)
        ops.conv2d_dgrad_bnreduce(d, dz_d, w_d, out, red, res=res_d, res_mask=rmask_d)
    else:
        assert rmask is None
        ops.conv2d_fprop(d, dz_d, w_d, out, res=res_d)
    _assert_dispatch(ops, ops.conv2d_describe(d, case["red"]), case["id"])
    torch.cuda.synchronize()
    oc, rc = out.float().cpu().reshape(mo, cout), ref.reshape(mo, cout)
    if k == 3:
        for name, mk in _edge_masks(n, h, w).items():
            _close(oc[mk], rc[mk], 2 ** -7, 3e-2, "data gradient, " + name)
    _close(oc, rc, 2 ** -7, 3e-2, "data gradient")
    if case["red"]:
        gm = oc.double()[touched]
        if zmask is not None:
            zb = ((zmask[:, :, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(mo, cout).bool()[touched]
            gm = torch.where(zb, gm, torch.zeros_like(gm))
        zt = z.double()[touched]
        sg = gm.sum(0)
        sgz = invstd.double() * ((gm * zt).sum(0) - mean.double() * sg)
        got = part.sum(0).double().cpu()
        for name, a, b in (("sum g*m", got[0], sg), ("sum g*m*xhat", got[1], sgz)):
            scale = float(b.abs().max()) + 1e-6
            err = float((a - b).abs().max()) / scale
            assert err < 2e-4, "%s: fused reduce %s off by %.3g of its scale" % (case["id"], name, err)


def test_conv_stem(ops):
    """uint8 image -> preprocess kernel (BGR - mean, pad 3, 4 channels) -> 7x7/2 conv as a K=7x32 row-gather GEMM."""
    g = torch.Generator().manual_seed(1)
    n, h, w = 2, 37, 45
    img = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    wt = _rt(torch.randn(64, 3, 7, 7, generator=g) * 0.05)
    bias = torch.randn(64, generator=g)
    mean = torch.tensor([103.939, 116.779, 123.68])
    xp = _rt(img.float()[..., [2, 1, 0]] - mean)
    ref = F.conv2d(xp.permute(0, 3, 1, 2), wt, bias, stride=2, padding=3).permute(0, 2, 3, 1)
    ho, wo = ref.shape[1], ref.shape[2]
    hp, wp = h + 6, max(w + 6, 2 * (wo - 1) + 8)
    xpad = torch.full((n * hp * wp * 4 + 64,), float("nan"), dtype=BF, device="cuda")
    xpad[n * hp * wp * 4:] = 0
    ops.preprocess(img.cuda(), xpad[: n * hp * wp * 4].view(n, hp, wp, 4), pad=3)
    pre = xpad[: n * hp * wp * 4].view(n, hp, wp, 4).float().cpu()
    assert torch.equal(pre[:, 3:3 + h, 3:3 + w, :3], xp), "preprocess interior"
    assert float(pre[:, :3].abs().sum()) == 0 and float(pre[..., 3].abs().sum()) == 0, "preprocess padding"
    w_master = _ohwi(wt).cuda()                         # [64,7,7,3] fp32
    w_packed = torch.empty(64, 7, 8, 4, dtype=BF, device="cuda")
    ops.stem_pack_weights(w_master, w_packed)
    d = ops.conv_desc(n, hp, wp, 32, 7, 1, 2, 0, 0, ho, wo, 64, in_pix_stride=4, flags=ops.CONV_BIAS)
    y = torch.empty(n, ho, wo, 64, dtype=BF, device="cuda")
    ops.conv2d_fprop(d, xpad, w_packed, y, bias=bias.cuda())
    assert ops.last_conv_instantiation().startswith("conv_stem<STATS=0>"), ops.last_conv_instantiation()
    torch.cuda.synchronize()
    _close(y, ref, 2 ** -7, 0.5, "stem conv")          # |x| up to 150 -> outputs O(100)
    # training form: the same outputs + the BatchNorm statistics of the ROUNDED outputs (column sums / sums of squares, f64 slots)
    ds = ops.conv_desc(n, hp, wp, 32, 7, 1, 2, 0, 0, ho, wo, 64, in_pix_stride=4, flags=ops.CONV_BIAS | ops.CONV_STATS)
    y2 = torch.full((n, ho, wo, 64), 7.0, dtype=BF, device="cuda")
    stats = torch.zeros(16, 2, 64, dtype=torch.float64, device="cuda")
    ops.conv2d_fprop(ds, xpad, w_packed, y2, bias=bias.cuda(), stats=stats)
    assert ops.last_conv_instantiation().startswith("conv_stem<STATS=1>"), ops.last_conv_instantiation()
    torch.cuda.synchronize()
    assert torch.equal(y2.view(torch.int16), y.view(torch.int16)), "the statistics form must not change the outputs"
    yd = y2.double().view(-1, 64).cpu()
    got = stats.sum(0).cpu()
    assert float((got[0] - yd.sum(0)).abs().max()) <= 1e-4 * float(yd.abs().sum(0).max()), "column sums"
    assert float((got[1] - (yd * yd).sum(0)).abs().max()) <= 1e-4 * float((yd * yd).sum(0).max()), "column sums of squares"


def test_conv_dgrad_3x3_with_residual(ops):
    g = torch.Generator().manual_seed(2)
    n, h, w, cin, cout = 2, 11, 9, 64, 128
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = _rt(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    dz = _rt(torch.randn(n, cout, h, w, generator=g))
    res = _rt(torch.randn(n, h, w, cin, generator=g))
    F.conv2d(x, wt, padding=1).backward(dz)
    ref = x.grad.permute(0, 2, 3, 1) + res
    w_t = torch.empty(cin, 3, 3, cout, dtype=BF, device="cuda")
    ops.weights_transpose_flip(_ohwi(wt).cuda(), w_t, cout, 3, 3, cin)
    d = ops.conv_desc(n, h, w, cout, 3, 3, 1, 1, 1, h, w, cin, flags=ops.CONV_ADD_RES)
    out = res.to(BF).cuda()                                # in-place accumulate: res aliases y
    ops.conv2d_fprop(d, dz.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), w_t, out, res=out)
    torch.cuda.synchronize()
    _close(out, ref, 2 ** -7, 3e-2, "dgrad 3x3 + residual")


def test_conv_dgrad_stride2_scatter(ops):
    g = torch.Generator().manual_seed(3)
    n, h, w, cin, cout = 2, 15, 21, 64, 128
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = _rt(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5)
    y = F.conv2d(x, wt, stride=2)
    ho, wo = y.shape[2], y.shape[3]
    dz = _rt(torch.randn(n, cout, ho, wo, generator=g))
    y.backward(dz)
    ref = x.grad.permute(0, 2, 3, 1)
    w_t = torch.empty(cin, 1, 1, cout, dtype=BF, device="cuda")
    ops.weights_transpose_flip(_ohwi(wt).cuda(), w_t, cout, 1, 1, cin)
    d = ops.conv_desc(n, ho, wo, cout, 1, 1, 1, 0, 0, ho, wo, cin, out_h=h, out_w=w, out_scatter=2)
    out = torch.zeros(n, h, w, cin, dtype=BF, device="cuda")
    ops.conv2d_fprop(d, dz.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), w_t, out)
    torch.cuda.synchronize()
    _close(out, ref, 2 ** -7, 3e-2, "dgrad stride-2 scatter")


@pytest.mark.parametrize("case", F32, ids=[c["id"] for c in F32])
def test_conv_f32_out_and_splitk(ops, case):
    g = torch.Generator().manual_seed(4)
    m, k, cout = case["m"], case["k"], case["cout"]
    x = _rt(torch.randn(m, k, generator=g))
    wt = _rt(torch.randn(cout, k, generator=g) / k ** 0.5)
    wt[cout * 9 // 16:] = 0                                       # zero-padded filter rows, as the merged head banks have
    bias = torch.randn(cout, generator=g)
    ref = x @ wt.t() + (bias if case["bias"] else 0)
    d = f32_desc(ops, case)
    y = torch.zeros(m, cout, device="cuda")
    ops.conv2d_fprop(d, x.to(BF).cuda(), wt.to(BF).cuda(), y, bias=bias.cuda() if case["bias"] else None)
    _assert_dispatch(ops, ops.conv2d_describe(d), case["id"])
    torch.cuda.synchronize()
    _close(y, ref, 1e-4, 1e-3 * max(1.0, (k / 6400) ** 0.5), "fp32 / split-K output")


def _wgrad_problem(case, g):
    n, h, w, cin, cout, k, s, p = (case[x] for x in ("n", "h", "w", "cin", "cout", "k", "s", "p"))
    x = _rt(torch.randn(n, cin, h, w, generator=g))
    wt = torch.zeros(cout, cin, k, k, requires_grad=True)
    y = F.conv2d(x, wt, stride=s, padding=p)
    ho, wo = y.shape[2], y.shape[3]
    dz = _rt(torch.randn(n, cout, ho, wo, generator=g))
    y.backward(dz)
    return x.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), dz.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), _ohwi(wt.grad), n * ho * wo


@pytest.mark.parametrize("case", WGRAD, ids=[c["id"] for c in WGRAD])
def test_conv_wgrad(ops, case):
    g = torch.Generator().manual_seed(5)
    x, dz, ref, m = _wgrad_problem(case, g)
    d = conv_desc(ops, case)
    dw = torch.zeros(case["cout"], case["k"], case["k"], case["cin"], device="cuda")
    ops.conv2d_wgrad(d, x, dz, dw)
    _assert_dispatch(ops, ops.conv2d_wgrad_describe(d), case["id"])
    torch.cuda.synchronize()
    _close(dw, ref, 2e-4, 2e-3 * m ** 0.5, "wgrad")


@pytest.mark.parametrize("k", [1, 3])
def test_conv_wgrad_shared_dw_accumulates_over_launches(ops, k):
    """The RPN weights of the feature-pyramid plan collect their gradient over four launches (P2..P5) into ONE dw
    (models/fpn.py backward_params_plan).  A level with M <= 64 pixels (P5 of a 256 x 256 image at batch 1: 8 x 8) has a single
    pixel split; without FRCNN_CONV_WGRAD_ACCUMULATE such a launch stores plainly and wipes the other levels' sums."""
    g = torch.Generator().manual_seed(55)
    cin, cout = (256, 128) if k == 1 else (256, 256)
    levels = [dict(n=1, h=32, w=32), dict(n=1, h=16, w=16), dict(n=1, h=8, w=8)]      # the last one: M = 64, one split
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    ref = torch.zeros(cout, k, k, cin)
    m_tot = 0
    for lv in levels:
        case = dict(lv, cin=cin, cout=cout, k=k, s=1, p=k // 2)
        x, dz, r, m = _wgrad_problem(case, g)
        d = conv_desc(ops, case, flags=ops.CONV_WGRAD_ACCUMULATE)
        ops.conv2d_wgrad(d, x, dz, dw)
        ref += r
        m_tot += m
    assert ops.last_conv_instantiation().endswith("split=1"), ops.last_conv_instantiation()      # the 8 x 8 level: the path under test
    torch.cuda.synchronize()
    _close(dw, ref, 2e-4, 2e-3 * m_tot ** 0.5, "wgrad shared over launches")
    # and the contract of the unflagged form: a one-split launch REPLACES (single writer of a zeroed dw)
    case = dict(levels[-1], cin=cin, cout=cout, k=k, s=1, p=k // 2)
    x, dz, r, m = _wgrad_problem(case, g)
    ops.conv2d_wgrad(conv_desc(ops, case), x, dz, dw)
    torch.cuda.synchronize()
    _close(dw, r, 2e-4, 2e-3 * m ** 0.5, "wgrad plain store")


def test_conv_wgrad_grouped_honours_the_accumulate_flag(ops):
    """frcnn_conv2d_wgrad_group_plan set plain_store for every layer with one effective pixel split and ignored
    FRCNN_CONV_WGRAD_ACCUMULATE, which include/frcnn_hip.h documents for the weight-gradient entry points generally: two flagged layers
    of ONE grouped launch that share a dw (8 x 8 pixels each: one split) must leave the SUM of their gradients there, on top of what
    dw held before; unflagged, a one-split layer replaces."""
    g = torch.Generator().manual_seed(56)
    cin, cout = 128, 128
    case = dict(n=1, h=8, w=8, cin=cin, cout=cout, k=1, s=1, p=0)
    other = dict(n=1, h=8, w=8, cin=cin, cout=64, k=1, s=1, p=0)               # an unflagged neighbour in the same group
    (x0, dz0, r0, m), (x1, dz1, r1, _), (x2, dz2, r2, _) = _wgrad_problem(case, g), _wgrad_problem(case, g), _wgrad_problem(other, g)
    before = torch.randn(cout, 1, 1, cin, generator=g)
    dw = before.clone().cuda()
    dw2 = torch.full((64, 1, 1, cin), 7.0, device="cuda")
    fl = ops.CONV_WGRAD_ACCUMULATE
    group = ops.WgradGroup([(conv_desc(ops, case, flags=fl), x0, dz0, dw), (conv_desc(ops, case, flags=fl), x1, dz1, dw),
                            (conv_desc(ops, other), x2, dz2, dw2)], "cuda")
    ops.conv2d_wgrad_grouped(group)
    torch.cuda.synchronize()
    _close(dw, before + r0 + r1, 2e-4, 2e-3 * (2 * m) ** 0.5, "grouped wgrad, shared dw")
    _close(dw2, r2, 2e-4, 2e-3 * m ** 0.5, "grouped wgrad, plain store of the unflagged layer")


@pytest.mark.parametrize("grp", WGRAD_GROUPS, ids=[c["id"] for c in WGRAD_GROUPS])
def test_conv_wgrad_grouped(ops, grp):
    """Several layers (3x3, strided 1x1, plain 1x1 -- both addressing modes) in one grouped launch with a common pixel split."""
    g = torch.Generator().manual_seed(8)
    items, refs = [], []
    for case in grp["layers"]:
        x, dz, ref, m = _wgrad_problem(case, g)
        refs.append((ref, m))
        dw = torch.zeros(case["cout"], case["k"], case["k"], case["cin"], device="cuda")        # (pre-zeroed: the group may use a pixel split)
        items.append((conv_desc(ops, case), x, dz, dw))
    group = ops.WgradGroup(items, "cuda")
    ops.conv2d_wgrad_grouped(group)
    got = ops.last_conv_instantiation()
    want = ops.conv2d_wgrad_describe(group=group)
    assert [p.split(" grid")[0] for p in got.split("; ")] == [p.split(" grid")[0] for p in want.split("; ")], (got, want)
    torch.cuda.synchronize()
    for (d, x, dz, dw), (ref, m) in zip(items, refs):
        _close(dw, ref, 2e-4, 2e-3 * m ** 0.5, "grouped wgrad")


def test_conv_wgrad_group_rejects_narrow_layers(ops):
    with pytest.raises(RuntimeError):
        d = ops.conv_desc(1, 8, 8, 32, 1, 1, 1, 0, 0, 8, 8, 64)                                   # cin = 32: not groupable
        ops.WgradGroup([(d, torch.zeros(64, 32, dtype=BF, device="cuda"), torch.zeros(64, 64, dtype=BF, device="cuda"),
                         torch.zeros(64, 32, device="cuda"))], "cuda")


def test_conv_wgrad_stem_and_rowindex(ops):
    g = torch.Generator().manual_seed(6)
    # stem: packed [64][7][8][4] gradient, then unpack to [64][7][7][3]
    n, h, w = 2, 37, 45
    xp = _rt(torch.randn(n, 3, h, w, generator=g))
    wt = torch.zeros(64, 3, 7, 7, requires_grad=True)
    y = F.conv2d(xp, wt, stride=2, padding=3)
    ho, wo = y.shape[2], y.shape[3]
    dz = _rt(torch.randn(n, 64, ho, wo, generator=g))
    y.backward(dz)
    ref = _ohwi(wt.grad)
    hp, wp = h + 6, max(w + 6, 2 * (wo - 1) + 8)
    xpad = torch.zeros(n, hp, wp, 4)
    xpad[:, 3:3 + h, 3:3 + w, :3] = xp.permute(0, 2, 3, 1)
    xflat = torch.zeros(n * hp * wp * 4 + 64, dtype=BF, device="cuda")
    xflat[: n * hp * wp * 4] = xpad.reshape(-1).to(BF).cuda()
    d = ops.conv_desc(n, hp, wp, 32, 7, 1, 2, 0, 0, ho, wo, 64, in_pix_stride=4)
    dwp = torch.zeros(64, 7, 8, 4, device="cuda")
    ops.conv2d_wgrad(d, xflat, dz.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), dwp)
    dw = torch.empty(64, 7, 7, 3, device="cuda")
    ops.stem_unpack_grad(dwp, dw)
    # ... and the form the train plan uses: the 7 x 3 real values of every tap row stored straight into the Keras-layout gradient
    du = ops.conv_desc(n, hp, wp, 32, 7, 1, 2, 0, 0, ho, wo, 64, in_pix_stride=4, flags=ops.CONV_WGRAD_STEM_UNPACK)
    dw_direct = torch.zeros(64, 7, 7, 3, device="cuda")
    ops.conv2d_wgrad(du, xflat, dz.permute(0, 2, 3, 1).contiguous().to(BF).cuda(), dw_direct)
    torch.cuda.synchronize()
    _close(dw, ref, 2e-4, 2e-3 * (n * ho * wo) ** 0.5, "stem wgrad")
    _close(dw_direct, ref, 2e-4, 2e-3 * (n * ho * wo) ** 0.5, "stem wgrad, unpacked in the store")
    _close(dw_direct, dw.cpu(), 1e-5, 1e-4, "stem wgrad: direct store against packed + unpack")      # (float-atomic order)
    # Dense-head form: rows gathered through row_index, dz padded to 64 columns
    R, K, S = 40, 1024, 24
    pooled = _rt(torch.randn(R, K, generator=g))
    rows = torch.randint(0, R, (S,), generator=g, dtype=torch.int32)
    dl = torch.zeros(S, 64)
    dl[:, :36] = _rt(torch.randn(S, 36, generator=g))
    ref2 = dl.t() @ pooled[rows.long()]
    d2 = ops.conv_desc(1, 1, S, K, 1, 1, 1, 0, 0, 1, S, 64)
    dw2 = torch.zeros(64, K, device="cuda")
    ops.conv2d_wgrad(d2, pooled.to(BF).cuda(), dl.to(BF).cuda(), dw2, dz_stride=64, row_index=rows.cuda())
    torch.cuda.synchronize()
    _close(dw2, ref2, 2e-4, 1e-2, "head wgrad (row_index)")


@pytest.mark.parametrize("case", [
    dict(n=2, h=12, w=39, cin=256, cout=64, k=1, res=True, mask=True),       # short K: tile run
    dict(n=4, h=24, w=78, cin=64, cout=256, k=1, res=True, mask=True),
    dict(n=2, h=13, w=17, cin=128, cout=128, k=3, res=False, mask=True),
    dict(n=1, h=24, w=78, cin=1024, cout=256, k=1, res=False, mask=False),
])
def test_conv_dgrad_with_fused_bn_reduce(ops, case):
    """conv2d_dgrad_bnreduce == conv2d_fprop followed by bn_bwd_reduce on its output (same gx bits; partial sums equal up to
    fp32 summation order and the sum(g*z) - mean*sum(g) form of sum(g*xhat))."""
    g = torch.Generator(device="cuda").manual_seed(7)
    n, h, w, cin, cout, k = (case[x] for x in ("n", "h", "w", "cin", "cout", "k"))
    m = n * h * w
    dz = torch.randn(m, cin, device="cuda", generator=g).to(BF)
    wt = (torch.randn(cout, k, k, cin, device="cuda", generator=g) / (cin * k * k) ** 0.5).to(BF)
    res = torch.randn(m, cout, device="cuda", generator=g).to(BF) if case["res"] else None
    z = (torch.randn(m, cout, device="cuda", generator=g) * 2 + 0.5).to(BF)
    mask = torch.randint(0, 256, (m, cout // 8), device="cuda", generator=g, dtype=torch.uint8) if case["mask"] else None
    mean = z.float().mean(0).contiguous()
    invstd = (1.0 / (z.float().var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
    d = ops.conv_desc(n, h, w, cin, k, k, 1, k // 2, k // 2, h, w, cout, flags=ops.CONV_ADD_RES if res is not None else 0)
    slots = ops.STAT_SLOTS
    gx_a, gx_b = torch.empty(m, cout, dtype=BF, device="cuda"), torch.empty(m, cout, dtype=BF, device="cuda")
    pa, pb = torch.zeros(slots, 2, cout, device="cuda"), torch.zeros(slots, 2, cout, device="cuda")
    ops.conv2d_fprop(d, dz, wt, gx_a, res=res)
    ops.bn_bwd_reduce(gx_a, None, z, mean, invstd, pa, m, cout, relu_mask=mask)
    red = ops.bn_reduce_args(z, mask, mean, invstd, pb)
    ops.conv2d_dgrad_bnreduce(d, dz, wt, gx_b, red, res=res)
    torch.cuda.synchronize()
    assert torch.equal(gx_a, gx_b), "fused kernel changes the data gradient"
    sa, sb = pa.sum(0).double(), pb.sum(0).double()
    scale = sa.abs().max(dim=1, keepdim=True).values + 1e-6
    assert float(((sa - sb).abs() / scale).max()) < 2e-4, float(((sa - sb).abs() / scale).max())


def test_conv_dgrad_scatter_with_fused_bn_reduce(ops):
    g = torch.Generator(device="cuda").manual_seed(8)
    n, ho, wo, cin, cout = 2, 12, 39, 512, 256
    hi, wi = 2 * ho, 2 * wo
    dz = torch.randn(n * ho * wo, cin, device="cuda", generator=g).to(BF)
    wt = (torch.randn(cout, 1, 1, cin, device="cuda", generator=g) / cin ** 0.5).to(BF)
    mo = n * hi * wi
    z = torch.randn(mo, cout, device="cuda", generator=g).to(BF)
    mask = torch.randint(0, 256, (mo, cout // 8), device="cuda", generator=g, dtype=torch.uint8)
    mean, invstd = z.float().mean(0).contiguous(), (1.0 / (z.float().var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
    base = torch.randn(mo, cout, device="cuda", generator=g).to(BF)
    d = ops.conv_desc(n, ho, wo, cin, 1, 1, 1, 0, 0, ho, wo, cout, out_h=hi, out_w=wi, out_scatter=2, flags=ops.CONV_ADD_RES)
    slots = ops.STAT_SLOTS
    ya, yb = base.clone(), base.clone()
    pa, pb = torch.zeros(slots, 2, cout, device="cuda"), torch.zeros(slots, 2, cout, device="cuda")
    ops.conv2d_fprop(d, dz, wt, ya, res=ya)
    # reference partial sums over the rows the scatter touches only (the fused kernel sees exactly those)
    rows = (torch.arange(n)[:, None, None] * hi * wi + torch.arange(ho)[None, :, None] * 2 * wi + torch.arange(wo)[None, None, :] * 2).reshape(-1).cuda()
    ops.bn_bwd_reduce(ya[rows].contiguous(), None, z[rows].contiguous(), mean, invstd, pa, rows.numel(), cout, relu_mask=mask[rows].contiguous())
    ops.conv2d_dgrad_bnreduce(d, dz, wt, yb, ops.bn_reduce_args(z, mask, mean, invstd, pb), res=yb)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    sa, sb = pa.sum(0).double(), pb.sum(0).double()
    scale = sa.abs().max(dim=1, keepdim=True).values + 1e-6
    assert float(((sa - sb).abs() / scale).max()) < 2e-4


def test_conv_dgrad_masked_residual(ops):
    """res_mask: the residual enters the add through a bit mask == adding the pre-masked tensor."""
    g = torch.Generator(device="cuda").manual_seed(9)
    n, h, w, cin, cout = 2, 24, 39, 256, 1024
    m = n * h * w
    dz = torch.randn(m, cin, device="cuda", generator=g).to(BF)
    wt = (torch.randn(cout, 1, 1, cin, device="cuda", generator=g) / cin ** 0.5).to(BF)
    res = torch.randn(m, cout, device="cuda", generator=g).to(BF)
    mask = torch.randint(0, 256, (m, cout // 8), device="cuda", generator=g, dtype=torch.uint8)
    bits = ((mask[:, :, None] >> torch.arange(8, dtype=torch.uint8, device="cuda")) & 1).reshape(m, cout).bool()
    z = torch.randn(m, cout, device="cuda", generator=g).to(BF)
    mean, invstd = z.float().mean(0).contiguous(), torch.ones(cout, device="cuda")
    d = ops.conv_desc(n, h, w, cin, 1, 1, 1, 0, 0, h, w, cout, flags=ops.CONV_ADD_RES)
    ya, yb = torch.empty(m, cout, dtype=BF, device="cuda"), torch.empty(m, cout, dtype=BF, device="cuda")
    ops.conv2d_fprop(d, dz, wt, ya, res=torch.where(bits, res, torch.zeros_like(res)))
    part = torch.zeros(ops.STAT_SLOTS, 2, cout, device="cuda")
    ops.conv2d_dgrad_bnreduce(d, dz, wt, yb, ops.bn_reduce_args(z, None, mean, invstd, part), res=res, res_mask=mask)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
