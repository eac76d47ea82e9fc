"""Faster-RCNN with the call surface of reference models/faster_rcnn.py, MI355X-native.

    model = FasterRCNN(config)
    losses, preds = model.train_step(images, gt_labels, gt_boxes, optimizer)
    losses, preds = model.test_step(images, gt_labels, gt_boxes)
    rpn_output, rcnn_output = model(images, training)

The whole train step (forward, RPN NMS, RoI pooling, target assignment, sampling, losses,
backward, SGD, prediction NMS) is one static launch plan over pre-allocated HBM buffers, replayed
as hipGraphs: no host synchronisation inside a step, ~zero launch overhead.  The RPN proposals are
computed once per step (the reference computes them twice with identical inputs,
faster_rcnn.py:53,106).
"""
import os

import torch

from .. import ops
from ..optimizers import SGD
from ..runtime import ParamStore, Plan
from ..utils.post_processing import NmsBuffers, postprocess_plan
from .detectors.fast_rcnn_detector import FastRCNNDetector
from .detectors.rpn_detector import RPNDetector
from .feature_extractor import FeatureExtractor, get_feature_extractor_model
from .fpn import LEVELS as FPN_LEVELS
from .fpn import FastRCNNDetectorFPN, FPNNeck, RPNDetectorFPN

BF16 = torch.bfloat16
LOSS_NAMES = ("rpn_cls", "rpn_reg", "rcnn_cls", "rcnn_reg")


# The RPN's 3x3 data gradient (a 50 us launch that needs nothing from the RoI branch) runs on the RPN's side stream, beside proposal NMS /
# RoI pooling / heads, and WRITES g_feat; the RoI backward pass then ADDS its gradient and runs the last backbone unit's BatchNorm-backward
# reduce (ops.roi_crop_pool_bwd_bf16_add).  Round 3 had it on the main chain after the RoI backward pass.  Same-box A/B (tools/ab_lib.sh,
# FRCNN_RPN_DGRAD_SIDE=0 / 1, twice): 4.107 -> 4.071, 4.104 -> 4.089 ms.  (C4 plan; the pyramid plan keeps the old order.)
RPN_DGRAD_ON_SIDE_STREAM = os.environ.get("FRCNN_RPN_DGRAD_SIDE", "1") != "0"
# RPN target assignment + sampling (anchors and ground truth only: no prediction) run on the weight re-layout side stream, under the
# forward pass, instead of at the head of the RPN's side stream behind the RPN convolutions.  Same-box A/B (FRCNN_RPN_TARGETS_EARLY=0 / 1,
# twice each): pyramid + fp8 + batch 8 (82 k regions per image: 53 + 83 us of one-workgroup-per-image kernels) 8.697 -> 8.605 ms; C4 batch 4
# 4.087 -> 4.089 (neutral: that chain is not the critical one there).
RPN_TARGETS_UNDER_FORWARD = os.environ.get("FRCNN_RPN_TARGETS_EARLY", "1") != "0"
# Data-parallel steps as ONE hipGraph with the gradient-bucket all-reduces (and the synchronised-BatchNorm sync points) captured inside it
# (Plan.capture_with_hook) instead of segment graphs with the collectives launched from the host between them.  Opt-in: RCCL's stream
# capture has run on this pool's hardware at world 1 only (tests/test_gpu_model.py, bench.py's captured_collectives_world1 leg) -- no
# multi-GPU node was available to any round -- and a capture that hangs at N > 1 would take a whole job down, a refused one falls back.
CAPTURE_COLLECTIVES = os.environ.get("FRCNN_CAPTURE_COLLECTIVES", "0") not in ("", "0")
# Opt-in (FRCNN_SGD_EARLY=1, single GPU): a gradient bucket (heads + RPN, conv4, conv3) is updated as soon as it is final, on a side stream
# under the rest of the backward pass (SGD.apply_bucket_plan: the fused kernel over that part of the flat buffers); the plan's last launch
# updates conv2 + stem, re-packs the stem and moves the step counter.  Same arithmetic element for element (tests), and MEASURED SLOWER:
# same-box A/B 3.95 -> 4.09 / 4.11 ms (gpurun_out/r5_ab4.txt).  The trace says why: every fork of the replayed hipGraph moves the main
# chain to another hardware queue and leaves the chip idle for 10 - 17 us (three forks + the join), and the update kernel -- a streaming
# kernel on all CUs -- doubles the BatchNorm kernel it runs beside (24 -> 45 us); 64 us of update at the end of the step cost less.
SGD_EARLY = os.environ.get("FRCNN_SGD_EARLY", "0") != "0"
# The accumulation targets only the backward pass touches (flat gradient 57 MB, the stride-2 scatter targets 90 MB) are zeroed on the
# side stream that runs under the forward pass (Plan.late_zero_fill) instead of by the fill in front of the stem; FRCNN_LATE_ZERO=0: one
# fill in front, as up to round 5's first bench.
LATE_ZERO_FILL = os.environ.get("FRCNN_LATE_ZERO", "1") != "0"
# At the head chain's two forks (Fast-RCNN targets after the proposal NMS, detection NMS after the head's post-processing) the MAIN
# stream's next launches are enqueued / captured before the side block (Plan.mark keeps the fork point): same graph edges, another order
# of node creation.
MAIN_FIRST_AT_FORKS = os.environ.get("FRCNN_MAIN_FIRST", "0") != "0"


class _Modules:
    """One set of module instances (= one set of activation buffers) attached to the shared store."""

    def __init__(self, config, depth, store, device, first, sync_bn_world=1, precision="bf16", topology="c4"):
        image_shape = config["image_shape"]
        # registration order = gradient-bucket order: regularised kernels, head biases, then backbone
        fe_shape = _feature_shape(image_shape, depth)
        self.neck = None
        if topology == "fpn":
            # feature pyramid over C2..C4 (models/fpn.py; BASELINE.json configs[4]): heads and RPN read 256-channel pyramid levels
            grids = _stage_grids(image_shape)
            self.rcnn = FastRCNNDetectorFPN(image_shape, config["num_classes"], config["rcnn"], grids, store=store, device=device)
            self.rpn = RPNDetectorFPN(image_shape, grids, config["rpn"], store=store, device=device)
            self.neck = FPNNeck(store, {2: 256, 3: 512, 4: 1024}, float(config["rpn"]["weight_decay"]), device)
        else:
            self.rcnn = FastRCNNDetector(image_shape, config["num_classes"], config["rcnn"], feature_channels=fe_shape[3], store=store,
                                         device=device)
            self.rpn = RPNDetector(image_shape, fe_shape, config["rpn"], store=store, device=device, precision=precision)
        self.rcnn.register_biases()
        self.rpn.register_biases()
        if self.neck is not None:
            self.neck.register_biases()
        store.end_bucket("heads")
        self.fe = get_feature_extractor_model(image_shape, depth=depth, store=store, device=device, sync_bn_world=sync_bn_world,
                                              precision=precision)
        assert tuple(self.fe.output_shape) == tuple(fe_shape)


def _feature_shape(image_shape, depth):
    h, w = image_shape[0], image_shape[1]
    f1 = lambda n: (n + 6 - 7) // 2 + 1
    f2 = lambda n: (n + 2 - 3) // 2 + 1
    f3 = lambda n: (n - 1) // 2 + 1
    return (None, f3(f3(f2(f1(h)))), f3(f3(f2(f1(w)))), 1024)


def _stage_grids(image_shape):
    """(h, w) of the backbone stage outputs C2, C3, C4 and of the subsampled pyramid level 5."""
    h, w = image_shape[0], image_shape[1]
    f1 = lambda n: (n + 6 - 7) // 2 + 1
    f2 = lambda n: (n + 2 - 3) // 2 + 1
    f3 = lambda n: (n - 1) // 2 + 1
    g2 = (f2(f1(h)), f2(f1(w)))
    g3 = (f3(g2[0]), f3(g2[1]))
    g4 = (f3(g3[0]), f3(g3[1]))
    return {2: g2, 3: g3, 4: g4, 5: ((g4[0] + 1) // 2, (g4[1] + 1) // 2)}


class FasterRCNN:
    def __init__(self, config, name="faster_rcnn", depth=50, device="cuda", seed=0, sampling_seed=0, world_size=1, sync_bn=False,
                 sampling_image_base=0, precision="bf16", topology="c4"):
        """reference faster_rcnn.py:11-37.  `config`: dict with the reference's config.json schema.
        world_size: data-parallel ranks (classification losses are means over the GLOBAL batch).  sync_bn: BatchNorm batch
        statistics and their backward sums are all-reduced over the ranks, so world_size x b images behave like the reference's
        one device with world_size*b images (faster_rcnn.py:50); off (default) every replica normalises with its own batch."""
        self.name = name
        self.config = config
        self._image_shape = tuple(config["image_shape"])
        self._rpn_config = config["rpn"]
        self._rcnn_config = config["rcnn"]
        self.depth = depth
        self.device = torch.device(device)
        self.sampling_seed = int(sampling_seed)
        # index of this replica's first image in the global batch: with one sampling_seed for all ranks and image_base = rank * b,
        # N ranks x b images draw exactly the fg / bg samples one device draws for N*b images (the sampler's Philox counter
        # carries the global image index)
        self.sampling_image_base = int(sampling_image_base)
        self.world_size = int(world_size)
        self.sync_bn = bool(sync_bn)
        # precision "fp8" (BASELINE.json configs[4]): the training step's forward convolutions with cin % 128 == 0 (backbone from
        # conv2_block2 on, RPN 3x3) multiply e4m3 operands on the fp8 MFMA path; everything else as in "bf16"
        self.precision = precision
        # topology "fpn" (BASELINE.json configs[4]): feature pyramid over C2..C4, RPN on P2..P5 with shared weights, RoI heads on
        # the level of each proposal (models/fpn.py, oracle/fpn.py); "c4": the reference's single conv4 map
        assert topology in ("c4", "fpn")
        self.topology = topology
        self.store = ParamStore(self.device)
        self._train = _Modules(config, depth, self.store, self.device, True, self.world_size if self.sync_bn else 1, precision, topology)
        self.store.finalize()
        self._eval = None
        self.feature_extractor, self.rpn_detector, self.rcnn_detector = self._train.fe, self._train.rpn, self._train.rcnn
        self.init_weights(seed)
        self._train_plan = None
        self._eval_plan = None
        self._fwd_train, self._fwd_plan = None, None
        self._eval_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.status = torch.zeros(4, dtype=torch.int32, device=self.device)     # [0] |= 1: empty background set while sampling
        self.use_graphs = True
        self._inject_proposals = False       # train_step(..., proposals_override=...): the plan being built takes its proposals from the caller
        self.capture_collectives = CAPTURE_COLLECTIVES      # data-parallel steps as one graph with their collectives inside (see above)

    # ------------------------------------------------------------------ parameters
    def init_weights(self, seed=0):
        self.feature_extractor.init_weights(seed)
        self.rpn_detector.init_weights(seed + 1)
        self.rcnn_detector.init_weights(seed + 2)
        if self._train.neck is not None:
            self._train.neck.init_weights(seed + 3)
        self._weights_dirty = True
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1

    def set_weights(self, weights):
        """dict Keras-variable-name -> array in Keras layout (see oracle/faster_rcnn.py for the names)."""
        self.feature_extractor.set_weights(weights)
        self.rpn_detector.set_weights(weights)
        self.rcnn_detector.set_weights(weights)
        if self._train.neck is not None:
            self._train.neck.set_weights(weights)
        self._weights_dirty = True
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1

    def get_weights(self):
        out = {}
        for m in (self.feature_extractor, self.rpn_detector, self.rcnn_detector):
            out.update(m.get_weights())
        if self._train.neck is not None:
            out.update(self._train.neck.get_weights())
        return out

    def save_weights(self, path):
        """reference train_faster_rcnn.py:242-243 (torch.save of the Keras-named weight dict)."""
        torch.save(self.get_weights(), path)

    def load_weights(self, path):
        self.set_weights(torch.load(path))

    @property
    def trainable_variables(self):
        return self.store.order

    def _sync_derived_weights(self, mods):
        """fp32 masters -> bf16 working copies, transposed data-gradient weights, packed stem."""
        p = Plan("refresh")
        p.add(self.store.refresh_bf16)
        mods.fe.refresh_weights(p)
        mods.rpn.refresh_weights(p)
        mods.rcnn.refresh_weights(p)
        if mods.neck is not None:
            mods.neck.refresh_weights(p)
        p.run()

    # ------------------------------------------------------------------ plan builders
    def _build(self, mods, batch, training, optimizer):
        if self.topology == "fpn":
            return self._build_fpn(mods, batch, training, optimizer)
        cfg = self.config
        dev = self.device
        H, W = self._image_shape[0], self._image_shape[1]
        nc1 = cfg["num_classes"] + 1
        plan = Plan("train_step" if training else "test_step")
        io = {}
        io["images"] = mods.fe.setup(batch, training)
        _, gh, gw, cf = mods.fe.output_shape
        G = 100
        io["gt_labels"] = torch.zeros(batch, G, nc1, device=dev)
        io["gt_boxes"] = torch.zeros(batch, G, 4, device=dev)
        mods.rpn.setup(batch, training, f8_scales=mods.fe.f8)
        P = int(self._rpn_config["nms"]["max_total_size"])
        rs, cs = self._rpn_config["sampling"], self._rcnn_config["sampling"]
        S_rpn, S_rcnn = int(rs["num_samples"]), int(cs["num_samples"])
        mods.rcnn.setup(batch, P, gh, gw, training, S_rcnn)
        step = optimizer.iterations if training else self._eval_step

        rpn_targets_early = False
        if training:
            plan.zero(self.store.g, late=True)
            # derived weights for the backward pass: all tap-flipped transposes (backbone, RPN, heads) in ONE launch, on a
            # side stream under the forward pass (first needed by the head backward passes)
            table, total = ops.make_transpose_flip_table(mods.fe.flip_entries() + mods.rpn.flip_entries() + mods.rcnn.flip_entries(), dev)
            plan.hold(table)
            with plan.branch("weight_flips"):
                if LATE_ZERO_FILL:                   # the backward pass's accumulation targets: zeroed here, not in front of the stem
                    plan.late_zero_fill()
                plan.add(ops.weights_transpose_flip_batched, table, total)
                if mods.fe.f8 is not None:           # fp8 mode: e4m3 twins of the transposes, for the fp8 data gradients
                    mods.fe.quantize_bwd_weights_plan(plan, extra=mods.rpn.quant_entries_bwd())
                rpn_targets_early = RPN_TARGETS_UNDER_FORWARD
        # ---- targets, sampling, losses (+ per-sample gradients)
        n = mods.rpn.n
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        t = {}
        t["rpn_tl"], t["rpn_tb"] = torch.empty(batch, n, 2, **f32), torch.empty(batch, n, 1, 4, **f32)
        t["rpn_idx"], t["rpn_ws"] = torch.empty(batch, S_rpn, **i32), torch.empty(batch, 2 * n, **i32)
        t["rcnn_tl"], t["rcnn_tb"] = torch.empty(batch, P, nc1, **f32), torch.empty(batch, P, nc1 - 1, 4, **f32)
        t["rcnn_idx"], t["rcnn_ws"] = torch.empty(batch, S_rcnn, **i32), torch.empty(batch, 2 * P, **i32)
        losses = torch.zeros(4, **f32)
        if training:
            t["rpn_dl"], t["rpn_dd"] = torch.empty(batch, S_rpn, 2, **f32), torch.empty(batch, S_rpn, 1, 4, **f32)
            t["rcnn_dl"], t["rcnn_dd"] = torch.empty(batch, S_rcnn, nc1, **f32), torch.empty(batch, S_rcnn, nc1 - 1, 4, **f32)
        plan.hold(t)
        # data-parallel: classification losses are means over the GLOBAL batch (scale 1/world before the SUM
        # all-reduce); regression losses are sums over rows (utils/losses.py:40) -> scale 1
        cls_scale = 1.0 / self.world_size
        # The RPN target / loss chain only needs the RPN outputs: it runs on a side stream next to proposal NMS, RoI pooling
        # and the Fast-RCNN heads, and continues with the part of the RPN backward pass that does not need the RoI-branch
        # gradient.  The detection NMS of the step's predictions is a second branch under the head backward passes.
        def rpn_targets(regions):
            plan.add(ops.assign_targets, regions, io["gt_labels"], io["gt_boxes"], batch, n, G, nc1, True, W, H,
                     rs["foreground_iou_interval"], rs["background_iou_interval"], t["rpn_tl"], t["rpn_tb"])
            plan.add(ops.sample_indices, t["rpn_tl"], batch, n, 2, S_rpn, rs["foreground_proportion"], self.sampling_seed, step, 0,
                     t["rpn_idx"], t["rpn_ws"], self.status, image_base=self.sampling_image_base)

        def rpn_losses():
            if training:
                # loss, per-sample gradients and their scatter into the dense head gradient in one launch
                plan.add(ops.losses_rpn_head_grad, rpn_out["pred_scores"], rpn_out["pred_boxes"], t["rpn_tl"], t["rpn_tb"], t["rpn_idx"],
                         batch, n, S_rpn, cls_scale, 1.0, losses[0:2], t["rpn_dl"], t["rpn_dd"], *mods.rpn.head_grad_target(plan))
            else:
                plan.add(ops.losses, rpn_out["pred_scores"], rpn_out["pred_boxes"], t["rpn_tl"], t["rpn_tb"], t["rpn_idx"], batch, n, 2,
                         S_rpn, cls_scale, 1.0, losses[0:2], None, None)

        def rcnn_targets(regions):
            plan.add(ops.assign_targets, regions, io["gt_labels"], io["gt_boxes"], batch, P, G, nc1, False, W, H,
                     cs["foreground_iou_interval"], cs["background_iou_interval"], t["rcnn_tl"], t["rcnn_tb"])
            plan.add(ops.sample_indices, t["rcnn_tl"], batch, P, nc1, S_rcnn, cs["foreground_proportion"], self.sampling_seed, step, 2,
                     t["rcnn_idx"], t["rcnn_ws"], self.status, image_base=self.sampling_image_base)

        def rcnn_losses():
            if training:
                # loss, per-sample gradients and the head's bf16 gradient rows in one launch (frcnn_losses_head_grad)
                plan.add(ops.losses_head_grad, rcnn_out["pred_scores"], rcnn_out["pred_boxes"], t["rcnn_tl"], t["rcnn_tb"], t["rcnn_idx"],
                         batch, P, nc1, S_rcnn, cls_scale, 1.0, losses[2:4], t["rcnn_dl"], t["rcnn_dd"], *mods.rcnn.head_grad_rows(),
                         bias_grad=self.store.grad("fast_rcnn_heads/bias"))
            else:
                plan.add(ops.losses, rcnn_out["pred_scores"], rcnn_out["pred_boxes"], t["rcnn_tl"], t["rcnn_tb"], t["rcnn_idx"], batch, P,
                         nc1, S_rcnn, cls_scale, 1.0, losses[2:4], None, None)

        if rpn_targets_early:
            # The RPN's targets and sample indices depend on the anchors and the ground truth, not on the predictions: they join the
            # side stream that already runs under the forward pass (no new fork), so that the RPN's own side stream -- since round 4
            # the longer of the two chains between the proposals and the backbone's backward pass -- starts with the loss launch
            with plan.branch("weight_flips"):
                rpn_targets(mods.rpn.regions(True))
        feat = mods.fe.forward_plan(plan, training)
        feat2d = feat.view(batch * gh * gw, cf)
        nms_cfg = self._rpn_config["nms"]
        rpn_nms = NmsBuffers(batch, mods.rpn.n, 1, nms_cfg["max_output_size_per_class"], nms_cfg["max_total_size"], dev)
        rpn_out = mods.rpn.forward_plan(plan, feat2d, training, decoded=rpn_nms.decoded, feature_maps8=mods.fe.feature_maps8 if training else None)
        assert n == rpn_out["pred_scores"].shape[1]
        if training:
            g_feat = torch.empty(batch * gh * gw, cf, dtype=BF16, device=dev)
            plan.hold(g_feat)
            t["g_feat"] = g_feat
            plan.join("weight_flips")
            with plan.branch("rpn_side"):
                if not rpn_targets_early:
                    rpn_targets(rpn_out["regions"])
                rpn_losses()
                mods.rpn.backward_params_plan(plan, None, None, t["rpn_idx"], S_rpn, feat2d, head_grad_done=True)
                if RPN_DGRAD_ON_SIDE_STREAM:
                    # the RPN's data gradient needs nothing from the RoI branch either: it WRITES g_feat here, beside the proposal NMS /
                    # RoI pooling / heads, and the RoI backward pass adds to it
                    mods.rpn.backward_data_plan(plan, g_feat, plain=True)
        else:
            rpn_targets(rpn_out["regions"])
            rpn_losses()
        # (the proposal NMS also writes the proposals in absolute coordinates: the Fast-RCNN stage's `regions`, fast_rcnn_detector.py:67)
        nms_rpn = postprocess_plan(plan, self._image_shape, **rpn_out, **self._rpn_config["nms"], buffers=rpn_nms, decoded_done=True,
                                   abs_boxes=mods.rcnn.regions_abs)
        rois = nms_rpn["pred_boxes"]
        self._inject_proposals_plan(plan, io, training, rois, mods.rcnn.regions_abs, batch, P, W, H)
        det_cfg = self._rcnn_config["nms"]
        det_nms = NmsBuffers(batch, P, nc1 - 1, det_cfg["max_output_size_per_class"], det_cfg["max_total_size"], dev)
        # Target assignment and sampling of the Fast-RCNN stage need the proposals and the ground truth, not the head's
        # predictions: a side stream under RoI pooling and the head GEMM (the chain from the proposals to the head's loss
        # gradient is a string of one-workgroup kernels with the chip idle).  Branches are not free -- a forked hipGraph is
        # enqueued node by node with cross-queue waits, a linear one in one piece: of the seven branches tried in round 2
        # (tools/ab_plan.sh, one box) this one, the RPN side chain and the detection NMS pay; targets of the RPN at the head
        # of the step (+0.25 ms), a late zero fill (+0.12 ms), the step counter and the head's parameter gradients do not.
        if MAIN_FIRST_AT_FORKS:
            plan.mark("rcnn_targets")
        else:
            with plan.branch("rcnn_targets"):
                rcnn_targets(mods.rcnn.regions_abs)
        # (the head-post launch also decodes the detections' boxes: the first step of their NMS)
        rcnn_out = mods.rcnn.forward_plan(plan, feat, rois, regions_done=True, decoded=det_nms.decoded)
        if MAIN_FIRST_AT_FORKS:
            with plan.branch("rcnn_targets"):
                rcnn_targets(mods.rcnn.regions_abs)
        plan.join("rcnn_targets")
        if training and MAIN_FIRST_AT_FORKS:
            plan.mark("detections")
            rcnn_losses()
            with plan.branch("detections"):
                nms_rcnn = postprocess_plan(plan, self._image_shape, **rcnn_out, **det_cfg, buffers=det_nms, decoded_done=mods.rcnn.decoded_done)
        else:
            if training:
                with plan.branch("detections"):
                    nms_rcnn = postprocess_plan(plan, self._image_shape, **rcnn_out, **det_cfg, buffers=det_nms, decoded_done=mods.rcnn.decoded_done)
            rcnn_losses()

        if training:
            if RPN_DGRAD_ON_SIDE_STREAM:
                mods.rcnn.backward_plan(plan, None, None, t["rcnn_idx"], S_rcnn, rois, g_feat, head_grad_done=True, bias_grad_done=True,
                                        add_to_g_feat_after="rpn_side", consumer=mods.fe.last_unit())
            else:
                mods.rcnn.backward_plan(plan, None, None, t["rcnn_idx"], S_rcnn, rois, g_feat, head_grad_done=True, bias_grad_done=True)
                plan.join("rpn_side")
                mods.rpn.backward_data_plan(plan, g_feat, consumer=mods.fe.last_unit())
            plan.join("detections")
            early = self._early_update_hook(plan, optimizer)
            if early:
                early("heads")
            plan.cut("bwd_conv4")
            mods.fe.backward_plan(plan, g_feat, g_feat_reduced=True, on_stage_done=(lambda stage: early("conv%d" % stage)) if early else None)
            plan.cut("update")
            self._final_update(plan, optimizer, mods.fe.stem, early)     # SGD (of what is left) + the stem's packed taps + the step counter: one launch
            if mods.fe.f8 is not None:
                # fp8 mode: next step's e4m3 weights from the updated masters (one launch), next step's activation scales from this
                # step's amax row
                mods.fe.quantize_weights_plan(plan, extra=mods.rpn.quant_entries())
                mods.fe.f8.plan_update(plan)
        else:
            nms_rcnn = postprocess_plan(plan, self._image_shape, **rcnn_out, **det_cfg, buffers=det_nms, decoded_done=mods.rcnn.decoded_done)
        preds = {"rpn_boxes": nms_rpn["pred_boxes"], "rpn_scores": nms_rpn["pred_scores"], "rcnn_boxes": nms_rcnn["pred_boxes"],
                 "rcnn_scores": nms_rcnn["pred_scores"], "rcnn_classes": nms_rcnn["pred_classes"]}
        aux = {"rpn_out": rpn_out, "rcnn_out": rcnn_out, "nms_rpn": nms_rpn, "nms_rcnn": nms_rcnn, "targets": t, "feature_maps": feat}
        return {"plan": plan, "io": io, "losses": losses, "preds": preds, "aux": aux, "batch": batch}

    def _build_fpn(self, mods, batch, training, optimizer):
        """The step of _build on the feature pyramid (models/fpn.py): backbone -> neck -> RPN on P2..P5 -> ONE proposal NMS ->
        per-level RoI pooling -> heads; backward through heads, RPN and neck into the backbone at C4, C3 and C2; the same four
        side-stream branches as the C4 plan (weight re-layouts, RPN targets -> loss, Fast-RCNN targets, detection NMS)."""
        cfg, dev = self.config, self.device
        H, W = self._image_shape[0], self._image_shape[1]
        nc1 = cfg["num_classes"] + 1
        plan = Plan("train_step_fpn" if training else "test_step_fpn")
        io = {"images": mods.fe.setup(batch, training)}
        G = 100
        io["gt_labels"] = torch.zeros(batch, G, nc1, device=dev)
        io["gt_boxes"] = torch.zeros(batch, G, 4, device=dev)
        fe, neck, rpn, rcnn = mods.fe, mods.neck, mods.rpn, mods.rcnn
        last = {2: "conv2_block3", 3: "conv3_block4", 4: fe.specs[-1][0]}
        grids = {l: (fe.units[last[l]][1].ho, fe.units[last[l]][1].wo) for l in FPN_LEVELS}
        neck.setup(batch, grids, training, f8_scales=fe.f8)
        rpn.setup(batch, training, f8_scales=fe.f8)
        P = int(self._rpn_config["nms"]["max_total_size"])
        rs, cs = self._rpn_config["sampling"], self._rcnn_config["sampling"]
        S_rpn, S_rcnn = int(rs["num_samples"]), int(cs["num_samples"])
        rcnn.setup(batch, P, training, S_rcnn)
        step = optimizer.iterations if training else self._eval_step
        n = rpn.n
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        t = {"rpn_tl": torch.empty(batch, n, 2, **f32), "rpn_tb": torch.empty(batch, n, 1, 4, **f32),
             "rpn_idx": torch.empty(batch, S_rpn, **i32), "rpn_ws": torch.empty(batch, 2 * n, **i32),
             "rcnn_tl": torch.empty(batch, P, nc1, **f32), "rcnn_tb": torch.empty(batch, P, nc1 - 1, 4, **f32),
             "rcnn_idx": torch.empty(batch, S_rcnn, **i32), "rcnn_ws": torch.empty(batch, 2 * P, **i32)}
        losses = torch.zeros(4, **f32)
        if training:
            t["rpn_dl"], t["rpn_dd"] = torch.empty(batch, S_rpn, 2, **f32), torch.empty(batch, S_rpn, 1, 4, **f32)
            t["rcnn_dl"], t["rcnn_dd"] = torch.empty(batch, S_rcnn, nc1, **f32), torch.empty(batch, S_rcnn, nc1 - 1, 4, **f32)
        plan.hold(t)
        cls_scale = 1.0 / self.world_size
        if training:
            plan.zero(self.store.g, late=True)
            table, total = ops.make_transpose_flip_table(fe.flip_entries() + neck.flip_entries() + rpn.flip_entries() + rcnn.flip_entries(), dev)
            plan.hold(table)
            with plan.branch("weight_flips"):        # (first needed by the backward pass: a side stream under the forward pass)
                if LATE_ZERO_FILL:
                    plan.late_zero_fill()
                plan.add(ops.weights_transpose_flip_batched, table, total)
                if fe.f8 is not None:                # precision "fp8": the backbone + the pyramid's 3x3 convolutions
                    fe.quantize_bwd_weights_plan(plan, extra=neck.quant_entries()[1] + rpn.quant_entries()[1])
        rpn_targets_early = training and RPN_TARGETS_UNDER_FORWARD

        def rpn_targets():
            plan.add(ops.assign_targets, rpn.regions_all, io["gt_labels"], io["gt_boxes"], batch, n, G, nc1, True, W, H,
                     rs["foreground_iou_interval"], rs["background_iou_interval"], t["rpn_tl"], t["rpn_tb"])
            plan.add(ops.sample_indices, t["rpn_tl"], batch, n, 2, S_rpn, rs["foreground_proportion"], self.sampling_seed, step, 0,
                     t["rpn_idx"], t["rpn_ws"], self.status, image_base=self.sampling_image_base)
        if rpn_targets_early:                        # (as in the C4 plan: on the side stream that runs under the forward pass)
            with plan.branch("weight_flips"):
                rpn_targets()
        fe.forward_plan(plan, training)
        stage_maps = {l: fe.acts[last[l]]["out"] for l in FPN_LEVELS}
        pyramid = neck.forward_plan(plan, stage_maps, {l: fe.acts[last[l]].get("out_8") for l in FPN_LEVELS} if training else None)
        nms_cfg = self._rpn_config["nms"]
        rpn_nms = NmsBuffers(batch, n, 1, nms_cfg["max_output_size_per_class"], nms_cfg["max_total_size"], dev)
        rpn_out = rpn.forward_plan(plan, pyramid, training, decoded=rpn_nms.decoded)
        # one-workgroup-per-image chains over ~82 k regions (targets -> sampling -> loss): side streams next to the proposal NMS, the
        # RoI pooling and the head GEMM, as in the C4 plan
        if training:
            plan.join("weight_flips")                # (the RPN's parameter gradients below read the transposed head weights)
        with plan.branch("rpn_side"):
            if not rpn_targets_early:
                rpn_targets()
            plan.add(ops.losses, rpn_out["pred_scores"], rpn_out["pred_boxes"], t["rpn_tl"], t["rpn_tb"], t["rpn_idx"], batch, n, 2, S_rpn,
                     cls_scale, 1.0, losses[0:2], t.get("rpn_dl"), t.get("rpn_dd"))
            if training:
                # the part of the RPN backward pass that does not need the RoI branch's gradient fills the chip while the proposal
                # NMS (one workgroup per image over ~82 k candidates: ~0.2 ms) holds the main chain
                rpn.backward_params_plan(plan, t["rpn_dl"], t["rpn_dd"], t["rpn_idx"], S_rpn, pyramid)
        nms_rpn = postprocess_plan(plan, self._image_shape, **rpn_out, **self._rpn_config["nms"], buffers=rpn_nms, decoded_done=True,
                                   abs_boxes=rcnn.regions_abs)
        rois = nms_rpn["pred_boxes"]
        self._inject_proposals_plan(plan, io, training, rois, rcnn.regions_abs, batch, P, W, H)
        det_cfg = self._rcnn_config["nms"]
        det_nms = NmsBuffers(batch, P, nc1 - 1, det_cfg["max_output_size_per_class"], det_cfg["max_total_size"], dev)
        with plan.branch("rcnn_targets"):
            regions_abs = rcnn.regions_abs
            plan.add(ops.assign_targets, regions_abs, io["gt_labels"], io["gt_boxes"], batch, P, G, nc1, False, W, H,
                     cs["foreground_iou_interval"], cs["background_iou_interval"], t["rcnn_tl"], t["rcnn_tb"])
            plan.add(ops.sample_indices, t["rcnn_tl"], batch, P, nc1, S_rcnn, cs["foreground_proportion"], self.sampling_seed, step, 2,
                     t["rcnn_idx"], t["rcnn_ws"], self.status, image_base=self.sampling_image_base)
        rcnn_out = rcnn.forward_plan(plan, pyramid, rois, regions_done=True, decoded=det_nms.decoded)
        plan.join("rcnn_targets")
        if training:
            plan.add(ops.losses_head_grad, rcnn_out["pred_scores"], rcnn_out["pred_boxes"], t["rcnn_tl"], t["rcnn_tb"], t["rcnn_idx"],
                     batch, P, nc1, S_rcnn, cls_scale, 1.0, losses[2:4], t["rcnn_dl"], t["rcnn_dd"], *rcnn.head_grad_rows(),
                     bias_grad=self.store.grad("fast_rcnn_heads/bias"))
        else:
            plan.add(ops.losses, rcnn_out["pred_scores"], rcnn_out["pred_boxes"], t["rcnn_tl"], t["rcnn_tb"], t["rcnn_idx"], batch, P, nc1,
                     S_rcnn, cls_scale, 1.0, losses[2:4], None, None)
        with plan.branch("detections"):              # the step's predictions: nothing on the main chain needs them
            nms_rcnn = postprocess_plan(plan, self._image_shape, **rcnn_out, **det_cfg, buffers=det_nms, decoded_done=rcnn.decoded_done)
        if training:
            rcnn.backward_plan(plan, rois, neck.gp, bias_grad_done=True)                                  # gp[2..4]: complete RoI-branch gradients
            plan.join("rpn_side")
            rpn.backward_data_plan(plan, neck.gp, {2: True, 3: True, 4: True, 5: False})
            _, gh, gw, cf = fe.output_shape
            g_feat = torch.empty(batch * gh * gw, cf, dtype=BF16, device=dev)
            plan.hold(g_feat)
            t["g_feat"] = g_feat
            first_of = {3: "conv3_block1", 4: "conv4_block1"}
            targets = {4: g_feat, 3: fe.acts[first_of[4]]["gin"], 2: fe.acts[first_of[3]]["gin"]}
            red4 = fe.last_unit().reduce_args(relu=True)
            plan.hold(red4)
            neck.backward_plan(plan, stage_maps, targets, red4=red4)
            plan.join("detections")
            early = self._early_update_hook(plan, optimizer)
            if early:
                early("heads")
            plan.cut("bwd_conv4")
            fe.backward_plan(plan, g_feat, g_feat_reduced=True, injected=(first_of[4], first_of[3]),
                             on_stage_done=(lambda stage: early("conv%d" % stage)) if early else None)
            plan.cut("update")
            self._final_update(plan, optimizer, fe.stem, early)          # SGD (of what is left) + the stem's packed taps + the step counter: one launch
            if fe.f8 is not None:
                fe.quantize_weights_plan(plan, extra=neck.quant_entries()[0] + rpn.quant_entries()[0])
                fe.f8.plan_update(plan)
        preds = {"rpn_boxes": nms_rpn["pred_boxes"], "rpn_scores": nms_rpn["pred_scores"], "rcnn_boxes": nms_rcnn["pred_boxes"],
                 "rcnn_scores": nms_rcnn["pred_scores"], "rcnn_classes": nms_rcnn["pred_classes"]}
        aux = {"rpn_out": rpn_out, "rcnn_out": rcnn_out, "nms_rpn": nms_rpn, "nms_rcnn": nms_rcnn, "targets": t, "feature_maps": fe.feature_maps,
               "pyramid": pyramid, "stage_maps": stage_maps, "roi_levels": rcnn.levels}
        return {"plan": plan, "io": io, "losses": losses, "preds": preds, "aux": aux, "batch": batch}

    def _early_update_hook(self, plan, optimizer):
        """(SGD_EARLY) callable(bucket name) that appends the early update of that gradient bucket on the trailing side stream, or None."""
        if not (SGD_EARLY and self.world_size == 1 and optimizer.early_ok()):
            return None
        buckets = {name: (b, e) for name, b, e in self.store.buckets}
        last = self.store.buckets[-1][0]
        done = []

        def early(name):
            if name in buckets and name != last and name not in done:
                b, e = buckets[name]
                assert b == (done and buckets[done[-1]][1] or 0), "gradient buckets become final in registration order"
                with plan.branch("sgd_early", follow=True):
                    optimizer.apply_bucket_plan(plan, b, e)
                done.append(name)
        early.done = done
        early.buckets = buckets
        return early

    def _final_update(self, plan, optimizer, stem, early):
        if early and early.done:
            plan.join("sgd_early")               # (the early launches have read the step counter: the last launch may move it)
            optimizer.apply_plan(plan, stem=stem, first=early.buckets[early.done[-1]][1])
        else:
            optimizer.apply_plan(plan, stem=stem)

    def _inject_proposals_plan(self, plan, io, training, rois, regions_abs, batch, P, W, H):
        """Test hook (train_step(..., proposals_override=...)): two launches behind the proposal NMS overwrite its kept boxes -- `rois`
        [B,P,4] relative, and their absolute form with the NMS launch's own arithmetic (x * W, y * H: frcnn_nms_combined_abs) -- with
        the caller's, so that everything downstream (RoI levels, pooling, Fast-RCNN targets, sample indices, losses, gradients) of two
        runs whose RPN scores differ in the last bits is computed on the SAME regions (the reference's stop_gradient'ed rois,
        models/faster_rcnn.py:53-55).  Not part of a plan built without an override: the benchmark's step is unchanged."""
        if not (training and self._inject_proposals):
            return
        io["proposals"] = torch.zeros(batch, P, 4, device=self.device)
        plan.add(ops.copy_bytes, io["proposals"], rois)
        plan.add(ops.boxes_scale, rois, regions_abs, float(W), float(H))

    def _build_forward(self, mods, batch):
        """Training-mode forward only (reference faster_rcnn.py:39-57 with training=True): BatchNorm on batch statistics (its
        moving averages are updated, as Keras does), RPN on the in-image anchors, proposal NMS, Fast-RCNN heads."""
        if self.topology == "fpn":
            return self._build_forward_fpn(mods, batch)
        plan = Plan("call_training")
        io = {"images": mods.fe.setup(batch, True)}
        _, gh, gw, cf = mods.fe.output_shape
        mods.rpn.setup(batch, True, f8_scales=None)
        P = int(self._rpn_config["nms"]["max_total_size"])
        mods.rcnn.setup(batch, P, gh, gw, False)
        feat = mods.fe.forward_plan(plan, True)
        rpn_out = mods.rpn.forward_plan(plan, feat.view(batch * gh * gw, cf), True, feature_maps8=mods.fe.feature_maps8)
        if mods.fe.f8 is not None:
            mods.fe.f8.plan_update(plan)
        nms_rpn = postprocess_plan(plan, self._image_shape, **rpn_out, **self._rpn_config["nms"])
        rcnn_out = mods.rcnn.forward_plan(plan, feat, nms_rpn["pred_boxes"])
        return {"plan": plan, "io": io, "batch": batch, "aux": {"rpn_out": rpn_out, "rcnn_out": rcnn_out, "nms_rpn": nms_rpn, "feature_maps": feat}}

    def _build_forward_fpn(self, mods, batch):
        """_build_forward on the feature pyramid: backbone (BatchNorm on batch statistics) -> neck -> RPN over P2..P5 on the in-image
        anchors -> ONE proposal NMS -> per-level RoI pooling -> heads (the forward half of _build_fpn; neck and RPN in bf16)."""
        plan = Plan("call_training_fpn")
        fe, neck, rpn, rcnn = mods.fe, mods.neck, mods.rpn, mods.rcnn
        io = {"images": fe.setup(batch, True)}
        last = {2: "conv2_block3", 3: "conv3_block4", 4: fe.specs[-1][0]}
        grids = {l: (fe.units[last[l]][1].ho, fe.units[last[l]][1].wo) for l in FPN_LEVELS}
        neck.setup(batch, grids, False)
        rpn.setup(batch, True, f8_scales=None)
        P = int(self._rpn_config["nms"]["max_total_size"])
        rcnn.setup(batch, P, False)
        fe.forward_plan(plan, True)
        if fe.f8 is not None:
            fe.f8.plan_update(plan)
        stage_maps = {l: fe.acts[last[l]]["out"] for l in FPN_LEVELS}
        pyramid = neck.forward_plan(plan, stage_maps)
        rpn_out = rpn.forward_plan(plan, pyramid, True)
        nms_rpn = postprocess_plan(plan, self._image_shape, **rpn_out, **self._rpn_config["nms"])
        rcnn_out = rcnn.forward_plan(plan, pyramid, nms_rpn["pred_boxes"])
        return {"plan": plan, "io": io, "batch": batch, "aux": {"rpn_out": rpn_out, "rcnn_out": rcnn_out, "nms_rpn": nms_rpn,
                                                                "feature_maps": fe.feature_maps, "pyramid": pyramid, "roi_levels": rcnn.levels}}

    def _feed(self, built, images, gt_labels, gt_boxes, proposals=None):
        io = built["io"]
        pairs = ((images, io["images"]), (gt_labels, io["gt_labels"]), (gt_boxes, io["gt_boxes"]))
        if proposals is not None:
            pairs += ((proposals, io["proposals"]),)
        for s, d in pairs:
            if tuple(s.shape) != tuple(d.shape):      # (copy_ would broadcast a smaller input silently)
                raise ValueError("input of shape %s where the step expects %s" % (tuple(s.shape), tuple(d.shape)))
        if all(s.is_cuda and s.dtype == d.dtype and s.shape == d.shape and s.is_contiguous() and s.data_ptr() % 16 == 0 for s, d in pairs):
            # device-resident inputs of the right types: ONE copy launch into the plan's static buffers (three launches of the
            # runtime's blit kernel were 22 us at the head of every step; its 5.6 MB image copy alone took 24 us in round 1)
            ops.copy_bytes_multi(pairs)
        else:
            for s, d in pairs:
                d.copy_(s, non_blocking=True)

    def _losses_dict(self, built):
        l = built["losses"]
        return {k: l[i] for i, k in enumerate(LOSS_NAMES)}

    # ------------------------------------------------------------------ reference call surface
    def train_step(self, images, gt_labels, gt_boxes, optimizer, sync_fn=None, proposals_override=None):
        """reference faster_rcnn.py:59-117.  images uint8 [B,H,W,3]; gt_labels fp32 [B,100,C+1];
        gt_boxes fp32 [B,100,4] relative (CUDA tensors).  Returns (losses, preds): device tensors
        living in static buffers (valid until the next step; clone to keep).
        sync_fn(segment_index) is the data-parallel hook called after each backward segment.
        proposals_override (tests): fp32 CUDA [B,P,4] relative boxes that replace the proposal NMS's output inside the step
        (_inject_proposals_plan); a model steps either always with or always without one (the plan is rebuilt on a change)."""
        b = int(images.shape[0])
        self._weights_epoch += 1                  # (test_step re-derives its modules' weight copies when this has moved)
        inject = proposals_override is not None
        if inject and not (proposals_override.is_cuda and proposals_override.dtype == torch.float32):
            raise TypeError("proposals_override: float32 CUDA tensor [B, P, 4] expected")
        if (self._train_plan is None or self._train_plan["batch"] != b or self._train_plan["optimizer"] is not optimizer
                or self._train_plan["inject"] != inject):
            optimizer.bind(self.store)
            self._inject_proposals = inject
            built = self._build(self._train, b, True, optimizer)
            built["optimizer"] = optimizer
            built["inject"] = inject
            self._train_plan = built
            self._sync_derived_weights(self._train)
            self._weights_dirty = False
            self._feed(built, images, gt_labels, gt_boxes, proposals_override)
            if self.use_graphs or self.get_fp8_state() is not None:
                # warm-up eagerly on a scratch copy of the mutable state, then capture.  In fp8 mode the same run is the calibration
                # pass of the delayed scales (kept: the first real step quantises with measured scales, not with 1) -- also without
                # graphs, so that eager and replayed runs are the same computation
                state = self._snapshot(optimizer)
                self._run_with_collectives(built["plan"])
                torch.cuda.synchronize()
                self._restore(state, optimizer, fp8_scales=False)
                if self.use_graphs:
                    built["plan"].capture()
                    self._restore(state, optimizer, fp8_scales=False)
            if getattr(self, "_pending_fp8_state", None) is not None:          # a checkpoint's scales win over the warm-up run's
                self.set_fp8_state(self._pending_fp8_state)
                self._pending_fp8_state = None
        built = self._train_plan
        if self._weights_dirty:
            self._sync_derived_weights(self._train)
            self._weights_dirty = False
        self._feed(built, images, gt_labels, gt_boxes, proposals_override)
        plan = built["plan"]
        nseg = len(plan.segments)
        collectives = self.world_size > 1 and bool(plan.pre_sync)
        if sync_fn is None and plan.captured and not collectives:
            plan.replay()                        # nothing to interleave between segments: the whole step is one graph
            return self._losses_dict(built), built["preds"]
        if sync_fn is not None and plan.captured and self.capture_collectives:
            # the data-parallel step as ONE graph with its collectives inside (Plan.capture_with_hook); falls back to the segment
            # graphs below if the capture is refused
            dp = built.get("dp")
            owner = getattr(sync_fn, "__self__", sync_fn)
            if dp is None or dp["owner"] is not owner:
                dp = built["dp"] = self._capture_data_parallel(built, sync_fn, owner, (images, gt_labels, gt_boxes, proposals_override))
            if dp["graph"] is not None:
                dp["graph"].replay()
                if hasattr(owner, "note_replay"):
                    owner.note_replay(dp["calls"])
                return self._losses_dict(built), built["preds"]
        done = 0                                 # gradient buckets handed to the data-parallel hook so far
        for i in range(nseg):
            if collectives:
                plan.sync_before(i)              # synchronised BatchNorm: partial sums of every rank -> sums of the global batch
            if plan.captured:
                plan.replay_segment(i)
            else:
                plan.run_segment(i)
            if sync_fn is not None and done < len(plan.bucket_ends) and i == plan.bucket_ends[done]:
                sync_fn(done, len(plan.bucket_ends) + 1)     # (hook protocol: bucket index, buckets + the update segment)
                done += 1
        return self._losses_dict(built), built["preds"]

    def _capture_data_parallel(self, built, sync_fn, owner, feed):
        """One eager data-parallel step on a scratch copy of the state (the communicator's first collective must not happen inside a
        capture), then the capture.  Returns {"graph": CUDAGraph or None, ...}; a refused capture is remembered and reported once."""
        plan, optimizer = built["plan"], built["optimizer"]
        calls0 = getattr(owner, "calls", 0)
        state = self._snapshot(optimizer)
        try:
            done = 0
            for i in range(len(plan.segments)):
                plan.sync_before(i)
                plan.run_segment(i)
                if done < len(plan.bucket_ends) and i == plan.bucket_ends[done]:
                    sync_fn(done, len(plan.bucket_ends) + 1)
                    done += 1
            torch.cuda.synchronize()
            calls = getattr(owner, "calls", 0) - calls0
            self._restore(state, optimizer)
            graph = plan.capture_with_hook(sync_fn)
            torch.cuda.synchronize()
            if hasattr(owner, "calls"):
                owner.calls = calls0              # (the warm-up step and the capture pass are not training steps)
            err = None
        except Exception as e:                    # capture refused (backend without stream-capture support, ...): segment graphs
            graph, calls, err = None, 0, "%s: %s" % (type(e).__name__, e)
            import warnings
            warnings.warn("data-parallel step could not be captured as one graph (%s): running its segment graphs" % err)
        self._restore(state, optimizer)
        self._feed(built, *feed)
        return {"graph": graph, "owner": owner, "calls": calls, "error": err}

    def _run_with_collectives(self, plan):
        """Eager run of a plan, with the all-reduces of its sync points (synchronised BatchNorm) when there are several ranks."""
        plan.run_synced()

    def _snapshot(self, optimizer):
        st = self.store
        return {"w": st.w.clone(), "wb": st.wb.clone(), "v": optimizer.velocity.clone(), "it": optimizer.iterations.clone(),
                "stats": {k: v.clone() for k, v in st.stats.items()}, "fp8": self.get_fp8_state()}

    def _restore(self, s, optimizer, fp8_scales=True):
        """fp8_scales=False keeps the delayed-scaling state the steps since the snapshot left behind (the plan builder's warm-up run
        doubles as the calibration pass: the first real step then quantises with measured scales instead of 1)."""
        st = self.store
        st.w.copy_(s["w"])
        st.wb.copy_(s["wb"])
        optimizer.velocity.copy_(s["v"])
        optimizer.iterations.copy_(s["it"])
        for k, v in s["stats"].items():
            st.stats[k].copy_(v)
        if fp8_scales and s.get("fp8") is not None:
            self.set_fp8_state(s["fp8"])
        self._weights_epoch += 1
        self._sync_derived_weights(self._train)

    def get_fp8_state(self):
        """Delayed-scaling state of the fp8 twins ([scale | 1 / scale] per tensor) or None (bf16, or no train plan yet).  Part of a
        checkpoint: without it a restored model's first step runs on scale 1 (e4m3 activations clamp at 448, e5m2 gradients below
        ~1.5e-5 flush to zero)."""
        f8 = self._train.fe.f8 if self._train is not None and getattr(self._train.fe, "f8", None) is not None else None
        return None if f8 is None else f8.state()

    def set_fp8_state(self, state):
        f8 = self._train.fe.f8 if self._train is not None and getattr(self._train.fe, "f8", None) is not None else None
        if f8 is None:
            self._pending_fp8_state = state          # (applied when the train plan, and with it the scale table, exists)
        elif state is not None:
            f8.load_state(state)

    def fp8_status(self):
        """{"clamped": (tensor, step) pairs whose fp8 twin held clamped values -- the tensor outgrew margin x the previous step's
        amax --, "nonfinite": non-finite amax events (scale kept)} since the model was built; zeros in bf16."""
        f8 = self._train.fe.f8 if self._train is not None and getattr(self._train.fe, "f8", None) is not None else None
        if f8 is None:
            return {"clamped": 0, "nonfinite": 0}
        c = f8.status.cpu()
        return {"clamped": int(c[0]), "nonfinite": int(c[1])}

    def test_step(self, images, gt_labels, gt_boxes):
        """reference faster_rcnn.py:119-169 (BN in inference mode, all anchors clipped to the image)."""
        b = int(images.shape[0])
        if self._eval is None:
            self._eval = _Modules(self.config, self.depth, self.store, self.device, False, topology=self.topology)
        if self._eval_plan is None or self._eval_plan["batch"] != b:
            self._eval_plan = self._build(self._eval, b, False, None)
        built = self._eval_plan
        if built.get("weights_version") != self._weights_version():
            # once per change of the weights (a validation pass after an epoch of training), not per step: the evaluation modules' own
            # re-laid-out copies (packed stem, transposed heads) from the masters the train step keeps current
            self._sync_derived_weights(self._eval)
            built["weights_version"] = self._weights_version()
        self._feed(built, images, gt_labels, gt_boxes)
        # (eager on purpose: replaying a captured evaluation plan was tried at the end of round 5 -- tools/driver_rate.py's validation leg
        # then crashed in the NEXT model built in the same process, after the first model and its evaluation graph had been released;
        # not understood, not kept)
        built["plan"].run()
        return self._losses_dict(built), built["preds"]

    def _weights_version(self):
        """Changes whenever the masters may have: every train step, set_weights / init_weights, a restored snapshot."""
        return self._weights_epoch

    def __call__(self, images, training=False):
        """reference faster_rcnn.py:39-57: returns (rpn_output, rcnn_output) dicts."""
        if training:
            # its own module instances (activation buffers): the captured train plan's buffers stay untouched
            b = int(images.shape[0])
            if self._fwd_train is None:
                self._fwd_train = _Modules(self.config, self.depth, self.store, self.device, False, self.world_size if self.sync_bn else 1,
                                           self.precision, topology=self.topology)
            if self._fwd_plan is None or self._fwd_plan["batch"] != b:
                self._fwd_plan = self._build_forward(self._fwd_train, b)
            built = self._fwd_plan
            self._sync_derived_weights(self._fwd_train)
            dst = built["io"]["images"]
            if tuple(images.shape) != tuple(dst.shape):
                raise ValueError("images of shape %s do not match the model's input [%d, %d, %d, 3] (config image_shape)" % (
                    tuple(images.shape), b, dst.shape[1], dst.shape[2]))
            if images.is_cuda and images.dtype == dst.dtype and images.is_contiguous() and images.data_ptr() % 16 == 0:
                ops.copy_bytes(images, dst)
            else:
                dst.copy_(images, non_blocking=True)
            self._run_with_collectives(built["plan"])
            return built["aux"]["rpn_out"], built["aux"]["rcnn_out"]
        z = torch.zeros
        b = int(images.shape[0])
        self.test_step(images, z(b, 100, self.config["num_classes"] + 1, device=self.device), z(b, 100, 4, device=self.device))
        return self._eval_plan["aux"]["rpn_out"], self._eval_plan["aux"]["rcnn_out"]
