"""GPU parity of the feature-pyramid pieces (BASELINE.json configs[4]; the reference has no FPN -- models/faster_rcnn.py:25-34 -- so the
oracle is oracle/fpn.py, a restatement of Lin et al., CVPR 2017, on the reference's own pieces; parity unpinned as stated there).
Index / discrete outputs bit-exact; bf16 sums of a few terms to a bf16 ulp."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from oracle import faster_rcnn as O
from oracle import fpn as OF
from oracle import roi as oroi

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rt(t):
    return t.to(BF).float()


def _close(a, b, rtol, atol, what):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    bad = int((err > atol + rtol * b.abs()).sum())
    assert bad == 0, "%s: %d/%d mismatches, max err %g" % (what, bad, a.numel(), float(err.max()))


@pytest.mark.parametrize("hw", [((24, 78), (47, 156)), ((47, 156), (94, 311)), ((6, 8), (12, 16)), ((3, 5), (5, 9))])
def test_upsample_add_forward_and_backward(ops, hw):
    (ht, wt), (h, w) = hw
    g = torch.Generator().manual_seed(0)
    B, C = 2, 64
    top = _rt(torch.randn(B, ht, wt, C, generator=g)).requires_grad_(True)
    lat = _rt(torch.randn(B, h, w, C, generator=g))
    up = OF.upsample_nearest(top.permute(0, 3, 1, 2), (h, w)).permute(0, 2, 3, 1)
    ref = (lat + up)
    out = torch.empty(B, h, w, C, dtype=BF, device="cuda")
    ops.upsample_add(top.detach().to(BF).cuda(), ht, wt, lat.to(BF).cuda(), out, B, h, w, C)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref.detach().to(BF)), "upsample_add"
    gout = _rt(torch.randn(B, h, w, C, generator=g))
    ref.backward(gout)
    prior = _rt(torch.randn(B, ht, wt, C, generator=g))
    for acc in (False, True):
        gtop = prior.to(BF).cuda().clone()
        ops.upsample_add_bwd(gout.to(BF).cuda(), h, w, gtop, B, ht, wt, C, accumulate=acc)
        torch.cuda.synchronize()
        exp = top.grad + (prior if acc else 0)
        _close(gtop, exp, 2 ** -8, 1e-6, "upsample_add_bwd (accumulate=%s)" % acc)
    # every fine pixel reads exactly one coarse pixel: the gradient sums match
    assert abs(float(top.grad.sum()) - float(gout.sum())) < 1e-2 * float(gout.abs().sum()) ** 0.5 + 1e-3


def test_subsample2_forward_and_backward(ops):
    g = torch.Generator().manual_seed(1)
    for (h, w) in ((24, 78), (7, 5)):
        B, C = 2, 256
        x = _rt(torch.randn(B, h, w, C, generator=g))
        ho, wo = (h + 1) // 2, (w + 1) // 2
        y = torch.empty(B, ho, wo, C, dtype=BF, device="cuda")
        ops.subsample2(x.to(BF).cuda(), y, B, h, w, C)
        torch.cuda.synchronize()
        assert torch.equal(y.cpu().float(), x[:, ::2, ::2])
        gy = _rt(torch.randn(B, ho, wo, C, generator=g))
        gx = x.to(BF).cuda().clone()
        ops.subsample2_bwd_add(gy.to(BF).cuda(), gx, B, h, w, C)
        torch.cuda.synchronize()
        exp = x.clone()
        exp[:, ::2, ::2] += gy
        assert torch.equal(gx.cpu(), exp.to(BF))


def test_roi_level_assignment_bit_exact(ops):
    g = torch.Generator().manual_seed(2)
    n = 4096
    shape = (375, 1242, 3)
    x0, y0 = torch.rand(n, generator=g) * 0.9, torch.rand(n, generator=g) * 0.9
    rois = torch.stack([x0, y0, x0 + torch.rand(n, generator=g) * 0.6, y0 + torch.rand(n, generator=g) * 0.9], 1)
    rois[0] = 0.0                                                       # NMS padding: level 2
    rois[1] = torch.tensor([0.0, 0.0, 112.0 / 1242, 112.0 / 375])       # on the first threshold
    rois[2] = torch.tensor([0.0, 0.0, 224.0 / 1242, 224.0 / 375])       # on the second
    rois[3] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    lv = torch.zeros(n, dtype=torch.int32, device="cuda")
    ops.roi_assign_levels(rois.cuda(), 1242, 375, lv)
    torch.cuda.synchronize()
    exp = OF.roi_levels(rois, shape)
    assert torch.equal(lv.cpu(), exp)
    assert set(exp.tolist()) == {2, 3, 4} and int(exp[0]) == 2 and int(exp[3]) == 4


def test_roi_pooling_per_level(ops):
    """Every RoI pooled from its own level's map, forward and backward, against the oracle's per-level pooling."""
    g = torch.Generator().manual_seed(3)
    B, P, C = 2, 24, 64
    grids = {2: (30, 40), 3: (15, 20), 4: (8, 10)}
    feats = {l: _rt(torch.randn(B, gh, gw, C, generator=g)).requires_grad_(True) for l, (gh, gw) in grids.items()}
    x0, y0 = torch.rand(B, P, generator=g) * 0.6, torch.rand(B, P, generator=g) * 0.6
    rois = torch.stack([x0, y0, x0 + torch.rand(B, P, generator=g) * 0.4 + 0.01, y0 + torch.rand(B, P, generator=g) * 0.4 + 0.01], -1)
    levels = torch.randint(2, 5, (B, P), generator=g, dtype=torch.int32)
    exp = sum(oroi.roi_pooling(feats[l], rois, 7, 2) * (levels == l).unsqueeze(-1) for l in grids)
    dev = "cuda"
    pooled = torch.full((B * P, 49 * C), 7.0, dtype=BF, device=dev)
    am = torch.zeros(B * P, 49 * C, dtype=torch.uint8, device=dev)
    lv = levels.reshape(-1).to(dev)
    for l, (gh, gw) in grids.items():
        ops.roi_crop_pool_fwd_level(feats[l].detach().to(BF).to(dev), rois.to(dev), B, P, gh, gw, C, 7, 2, pooled, am, lv, l)
    torch.cuda.synchronize()
    _close(pooled.view(B, P, -1), exp.detach(), 2 ** -7, 2e-2, "per-level roi pooling")
    rows = torch.arange(0, B * P, 2, dtype=torch.int32)
    gp = _rt(torch.randn(len(rows), 49 * C, generator=g))
    gfull = torch.zeros(B * P, 49 * C)
    gfull[rows.long()] = gp
    exp.backward(gfull.view(B, P, -1))
    for l, (gh, gw) in grids.items():
        gf = torch.full((B, gh, gw, C), 9.0, dtype=BF, device=dev)
        ops.roi_crop_pool_bwd_bf16_level(gp.to(BF).to(dev), am, rois.to(dev), rows.to(dev), len(rows), B, P, gh, gw, C, 7, 2, gf, lv, l)
        torch.cuda.synchronize()
        _close(gf, feats[l].grad, 2 ** -6, 5e-2, "per-level roi gradient, level %d" % l)


def test_rpn_head_windows(ops):
    """head post / head gradient of one level write / read their window of the concatenated per-image anchor list exactly as the
    single-map kernels do on a list of their own."""
    g = torch.Generator().manual_seed(4)
    B, apl, ld = 2, 3, 128
    dev = "cuda"
    locs = {2: 40, 3: 12}
    n_tot, off = 0, {}
    keep, heads, regions = {}, {}, {}
    for l, nl in locs.items():
        k = torch.sort(torch.randperm(nl * apl, generator=g)[: nl * apl * 2 // 3]).values.to(torch.int32)
        keep[l], heads[l] = k, torch.randn(B * nl, ld, generator=g)
        regions[l] = torch.rand(len(k), 4, generator=g) * 100
        regions[l][:, 2:] += regions[l][:, :2] + 5
        off[l] = n_tot
        n_tot += len(k)
    scores, deltas = torch.zeros(B, n_tot, 2, device=dev), torch.zeros(B, n_tot, 1, 4, device=dev)
    decoded = torch.zeros(B, n_tot, 1, 4, device=dev)
    reg_all = torch.cat([regions[l] for l in locs]).to(dev)
    for l, nl in locs.items():
        n = len(keep[l])
        ops.rpn_head_post_level(heads[l].to(dev), ld, B, nl * apl, apl, keep[l].to(dev), n, scores, deltas, reg_all[off[l]:off[l] + n], decoded,
                                1242.0, 375.0, n_tot, off[l])
        s1, d1, dc1 = torch.zeros(B, n, 2, device=dev), torch.zeros(B, n, 1, 4, device=dev), torch.zeros(B, n, 1, 4, device=dev)
        ops.rpn_head_post_decode(heads[l].to(dev), ld, B, nl * apl, apl, keep[l].to(dev), n, s1, d1, regions[l].to(dev), dc1, 1242.0, 375.0)
        torch.cuda.synchronize()
        assert torch.equal(scores[:, off[l]:off[l] + n], s1) and torch.equal(deltas[:, off[l]:off[l] + n], d1)
        assert torch.equal(decoded[:, off[l]:off[l] + n], dc1)
    S = 16
    idx = torch.randint(0, n_tot, (B, S), generator=g, dtype=torch.int32)
    dl, dd = torch.randn(B, S, 2, generator=g), torch.randn(B, S, 1, 4, generator=g)
    for l, nl in locs.items():
        n = len(keep[l])
        dh = torch.zeros(B * nl, ld, device=dev)
        ops.rpn_head_grad_level(dl.to(dev), dd.to(dev), idx.to(dev), keep[l].to(dev), B, S, nl * apl, apl, dh, ld, off[l], n)
        torch.cuda.synchronize()
        exp = torch.zeros(B * nl, ld)
        for b in range(B):
            for s in range(S):
                j = int(idx[b, s]) - off[l]
                if 0 <= j < n:
                    a = int(keep[l][j])
                    loc, k = a // apl, a % apl
                    exp[b * nl + loc, 2 * k:2 * k + 2] += dl[b, s]
                    exp[b * nl + loc, 2 * apl + 4 * k:2 * apl + 4 * k + 4] += dd[b, s, 0]
        _close(dh, exp, 1e-6, 1e-6, "rpn head gradient window, level %d" % l)
