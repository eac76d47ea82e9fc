"""CPU: every MFMA conv kernel instantiation the BASELINE train plans launch is covered by an oracle-compared GPU case.

The dispatcher (csrc/conv_tile.hip: conv_tile_dispatch, csrc/conv_wgrad.hip) is host logic; its dry-run entry points
frcnn_conv2d_describe / frcnn_conv2d_wgrad_describe name the kernel a descriptor would launch without touching a device.
Here the full train plans of BASELINE.json configs[1] (ResNet-50, batch 4, 375x1242) and configs[3] (ResNet-101, batch 2,
1000 proposals) are built on CPU tensors (nothing is launched), the instantiation of every conv launch is collected, and
each must appear among the instantiations that tests/conv_cases.py's cases dispatch to -- the cases tests/test_gpu_conv.py
runs against torch-CPU on the GPU box, asserting there again which kernel really ran.  When the tile heuristics move, this
test names the instantiation that lost its parity case."""
import copy
import importlib

import pytest
import torch

import conv_cases


def _plan_instantiations(monkeypatch, depth, batch, proposals=None, precision="bf16", topology="c4", image_shape=None):
    ops = importlib.import_module("2d_object_detection_amd.ops")
    # the two init-time kernels of the RPN detector (anchor table, clip) need a device; their values do not matter here
    monkeypatch.setattr(ops, "anchors_generate", lambda out, *a, **k: out.zero_())
    monkeypatch.setattr(ops, "clip_to_window", lambda boxes, out, w: out.copy_(boxes))
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    C = importlib.import_module("2d_object_detection_amd.config")
    OPT = importlib.import_module("2d_object_detection_amd.optimizers")
    cfg = copy.deepcopy(C.default_config(image_shape) if image_shape else C.default_config())
    if proposals:
        cfg["rpn"]["nms"]["max_total_size"] = cfg["rpn"]["nms"]["max_output_size_per_class"] = proposals
    model = M.FasterRCNN(cfg, depth=depth, device="cpu", precision=precision, topology=topology)
    opt = OPT.SGD(learning_rate=1e-3, momentum=0.9)
    opt.bind(model.store)
    plan = model._build(model._train, batch, True, opt)["plan"]
    names = {}
    BN_FUSED_FORMS = {getattr(ops, n): n for n in conv_cases.BN_FUSED_ENTRY_POINTS if hasattr(ops, n)}
    assert BN_FUSED_FORMS, "ops lost its BatchNorm-carrying convolution entry points"
    for seg in plan.segments:
        for fn, args, kwargs, _br in seg:
            found = []
            if fn is ops.conv2d_fprop or fn is ops.conv2d_dgrad_bnreduce:
                found = [ops.conv2d_describe(args[0], fn is ops.conv2d_dgrad_bnreduce)]
            elif fn in BN_FUSED_FORMS:
                # the convolutions that carry a neighbouring BatchNorm (forward apply on the input side, ...): the kernel frcnn_conv2d_describe
                # names for the plain form, tagged with the form -- a parity case must have run THAT form on THAT instantiation
                found = [ops.conv2d_describe(args[0], False).replace(" grid", "+" + BN_FUSED_FORMS[fn] + " grid", 1)]
            elif fn is ops.conv2d_fprop_fp8:
                found = [ops.conv2d_describe_fp8(args[0])]
            elif fn is ops.conv2d_dgrad_fp8:
                found = [ops.conv2d_describe_dgrad_fp8(args[0], kwargs.get("red") is not None)]
            elif fn is ops.conv2d_wgrad_fp8:
                found = [ops.conv2d_wgrad_describe_fp8(args[0])]
            elif fn is ops.conv2d_wgrad:
                found = [ops.conv2d_wgrad_describe(args[0], with_row_index=kwargs.get("row_index") is not None)]
            elif fn is ops.conv2d_wgrad_grouped:
                found = [p for p in ops.conv2d_wgrad_describe(group=args[0]).split("; ") if p.strip()]
            for f in found:
                key = f.split(" grid")[0]
                names[key] = names.get(key, 0) + 1
    return names, plan.num_launches


@pytest.mark.parametrize("depth,batch,proposals", [(50, 4, None), (101, 2, 1000)])
def test_every_plan_instantiation_has_a_parity_case(monkeypatch, ops, depth, batch, proposals):
    used, launches = _plan_instantiations(monkeypatch, depth, batch, proposals)
    covered = conv_cases.covered_instantiations(ops)
    assert len(used) >= 20 and launches >= 150          # (200 before round 5 folded BatchNorm launches into their consumers)
    missing = sorted(k for k in used if k not in covered)
    assert not missing, "conv kernels of the R%d batch-%d train plan without an oracle-compared GPU case:\n  %s" % (depth, batch, "\n  ".join(missing))


def test_reference_default_plan_instantiations_have_parity_cases(monkeypatch, ops):
    """The reference's OWN configuration -- /root/reference config.json:3 image_shape [600, 1987, 3], train_faster_rcnn.py:52-54 batch 2 --
    which INTEGRATION.md tells a maintainer to load unchanged: 38 x 125 feature grid, odd extents at every stage, other workgroup-count
    branches of the 3x3 dispatch than 375 x 1242 takes.  (tests/test_gpu_reference_default.py runs the step itself.)"""
    used, launches = _plan_instantiations(monkeypatch, 50, 2, image_shape=(600, 1987, 3))
    covered = conv_cases.covered_instantiations(ops)
    assert len(used) >= 20 and launches >= 150
    missing = sorted(k for k in used if k not in covered)
    assert not missing, "conv kernels of the 600x1987 batch-2 train plan without an oracle-compared GPU case:\n  %s" % "\n  ".join(missing)


def test_every_fp8_plan_instantiation_has_a_parity_case(monkeypatch, ops):
    """BASELINE.json configs[4]'s precision (fp8 forward convolutions) at its batch of 8, and at the bench's batch 4."""
    covered = conv_cases.covered_instantiations(ops)
    for depth, batch, proposals in ((50, 8, None), (50, 4, None), (101, 2, 1000)):       # (ResNet-101: ~250 fp8 tensors in the scale table)
        used, launches = _plan_instantiations(monkeypatch, depth, batch, proposals, precision="fp8")
        f8 = sorted(k for k in used if "F8=1" in k)
        assert len(f8) >= 5, f8
        assert any(",F8>" in k for k in used), "no fp8 weight gradient in the fp8 plan"
        missing = sorted(k for k in used if k not in covered)
        assert not missing, "conv kernels of the fp8 R%d batch-%d train plan without an oracle-compared GPU case:\n  %s" % (depth, batch, "\n  ".join(missing))


def test_every_fpn_plan_instantiation_has_a_parity_case(monkeypatch, ops):
    """The feature-pyramid topology (BASELINE.json configs[4]): batch 8 in fp8 (its own configuration), batch 4 and 2 in bf16."""
    covered = conv_cases.covered_instantiations(ops)
    for batch, precision in ((8, "fp8"), (4, "bf16"), (2, "bf16")):
        used, launches = _plan_instantiations(monkeypatch, 50, batch, precision=precision, topology="fpn")
        assert launches >= 220           # (round 4: 12 + 4 x 2 launches folded into their neighbours; round 5: BatchNorm launches into their consumers)
        missing = sorted(k for k in used if k not in covered)
        assert not missing, "conv kernels of the FPN R50 batch-%d %s train plan without an oracle-compared GPU case:\n  %s" % (
            batch, precision, "\n  ".join(missing))


def test_describe_reports_errors(ops):
    d = ops.conv_desc(1, 8, 8, 64, 7, 7, 1, 3, 3, 8, 8, 64)            # 49 taps: unsupported
    with pytest.raises(RuntimeError, match="32 taps"):
        ops.conv2d_describe(d)
    assert "KWS=1" in ops.conv2d_describe(ops.conv_desc(1, 94, 311, 64, 3, 3, 1, 1, 1, 94, 311, 64))
    assert "KWS=0" in ops.conv2d_describe(ops.conv_desc(1, 24, 78, 64, 3, 3, 1, 1, 1, 24, 78, 64))


def test_workspace_query_matches_the_dispatch_rule(ops):
    """frcnn_conv2d_workspace_bytes: > 0 exactly where the dispatcher takes the split-K fix-up form (fewer 128 x 64 tiles than CUs,
    >= 64 K slices); a descriptor without workspace keeps the one-workgroup-per-tile kernel."""
    rpn = ops.conv_desc(4, 24, 15, 1024, 3, 3, 1, 1, 1, 24, 15, 256, flags=ops.CONV_BIAS | ops.CONV_RELU)        # 48 tiles, 144 slices
    tiles = ((4 * 24 * 15 + 127) // 128) * (256 // 64)
    assert ops.conv_workspace_bytes(rpn) == tiles * 2 * 128 * 64 * 4 + tiles * 4
    assert "FIX" not in ops.conv2d_describe(rpn)
    assert ops.conv_attach_workspace(rpn, "cpu") is not None and "FIX=1" in ops.conv2d_describe(rpn)
    # the RPN's own 3x3 (rows of 78 pixels) runs on the patch-resident kernel in bf16 since round 4 and ignores the workspace; the query
    # still answers for the tile kernel, which the fp8 entry points use with the same descriptor
    real = ops.conv_desc(4, 24, 78, 1024, 3, 3, 1, 1, 1, 24, 78, 256, flags=ops.CONV_BIAS | ops.CONV_RELU)
    assert ops.conv_workspace_bytes(real) > 0 and ops.conv_attach_workspace(real, "cpu") is not None
    assert ops.conv2d_describe(real).startswith("conv3x3_patch<SB=4,SMODE=0,LW=4> grid=240x1")
    assert "FIX=1" in ops.conv2d_describe_fp8(real)
    for d in (ops.conv_desc(4, 24, 78, 1024, 1, 1, 1, 0, 0, 24, 78, 256, flags=ops.CONV_BIAS | ops.CONV_STATS),       # 16 slices
              ops.conv_desc(4, 24, 78, 256, 3, 3, 1, 1, 1, 24, 78, 256, flags=ops.CONV_BIAS | ops.CONV_STATS),        # patch-resident kernel
              ops.conv_desc(4, 94, 311, 64, 3, 3, 1, 1, 1, 94, 311, 64),                                              # kw-shared, 914 tiles
              ops.conv_desc(4, 24, 78, 256, 3, 3, 1, 1, 1, 24, 78, 1024)):                                            # 128 x 128 tiles
        assert ops.conv_workspace_bytes(d) == 0
        assert ops.conv_attach_workspace(d, "cpu") is None
    small = ops.conv_desc(4, 24, 15, 1024, 3, 3, 1, 1, 1, 24, 15, 256)
    ops.conv_attach_workspace(small, "cpu")
    small.workspace_bytes = 1000
    with pytest.raises(Exception):
        ops.conv2d_describe(small)
