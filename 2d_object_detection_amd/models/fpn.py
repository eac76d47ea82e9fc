"""Feature-pyramid variant of the detector (BASELINE.json configs[4]: "ResNet-50 FPN Faster-RCNN").

The reference has no FPN: models/faster_rcnn.py:25-34 wires the single conv4_block6 map of models/feature_extractor.py:8-9 into
RPNDetector and FastRCNNDetector.  This module restates Lin et al., "Feature Pyramid Networks for Object Detection" (CVPR 2017) on
top of the reference's pieces, exactly as oracle/fpn.py writes it down (the test oracle; its header lists every choice):

  neck      lateral 1x1 -> 256 on C2, C3, C4 (conv2_block3_out, conv3_block4_out, conv4_block6_out); top-down nearest-neighbour
            upsampling + add; 3x3 output convolution per level; P5 = P4[::2, ::2]                                   (sec. 3, 4.1)
  RPN       the reference's head (rpn_detector.py:26-58) on P2..P5 with SHARED weights, one anchor scale per level (the
            config's four scales), three ratios, strides 4 / 8 / 16 / 32; regions of all levels concatenated per image,
            ONE combined NMS (utils/post_processing.py)                                                                (sec. 4.1)
  heads     RoI level k = 2 + [wh >= 112^2] + [wh >= 224^2] (eq. 1, k0 = 4, clamped to 2..4); the reference's crop_and_resize +
            max-pool RoI pooling (fast_rcnn_detector.py:133-177) on that level; the two Dense layers on 7x7x256      (sec. 4.2)

Every convolution is a launch of the implicit-GEMM kernel (bf16); the merge, the subsampling, the level assignment and the
per-level windows of the RoI / RPN-head kernels are csrc/fpn.hip, roi.hip, boxes_nms.hip, targets_losses.hip.
"""
import math

import numpy as np
import torch

from .. import ops
from .detectors.fast_rcnn_detector import HEAD_LD as RCNN_LD
from .detectors.fast_rcnn_detector import FastRCNNDetector
# padded row length of the RPN head's [cls | reg] output: 3 anchors per location x 6 = 18 real columns here (the single-map detector
# has 12 anchors: 72 of 128) -- 64 keeps the head's fp32 output, its gradient and their casts / column sums at half the bytes of 128
RPN_LD = 64
from .feature_extractor import FP8_WGRAD

BF16 = torch.bfloat16
FPN_DIM = 256
LEVELS = (2, 3, 4)
RPN_LEVELS = (2, 3, 4, 5)
STRIDE = {2: 4, 3: 8, 4: 16, 5: 32}


class _Conv:
    """conv + bias (no BatchNorm): forward, weight / bias gradient, data gradient."""

    def __init__(self, store, name, cin, cout, k, decay):
        self.store, self.name, self.cin, self.cout, self.k = store, name, cin, cout, k
        store.register(name + "/kernel", (cout, k, k, cin), decay=decay)        # OHWI (Keras: HWIO)

    def setup(self, n, h, w, device, relu=False, fp8=False):
        """fp8: allocate e4m3 twins of the forward weights (per output channel) and of the tap-flipped transposes (per input
        channel) for frcnn_conv2d_fprop_fp8 / frcnn_conv2d_dgrad_fp8 (precision "fp8": the 3x3 convolutions of the pyramid)."""
        k, p = self.k, self.k // 2
        self.n, self.h, self.w, self.m = n, h, w, n * h * w
        self.w8 = None
        if fp8 and self.cin % 128 == 0 and self.cout % 128 == 0:
            self.w8 = torch.zeros(self.cout, k, k, self.cin, dtype=ops.FP8, device=device)
            self.w8_scale = torch.ones(self.cout, dtype=torch.float32, device=device)
            self.w_t8 = torch.zeros(self.cin, k, k, self.cout, dtype=ops.FP8, device=device)
            self.w_t8_scale = torch.ones(self.cin, dtype=torch.float32, device=device)
        self.desc = ops.conv_desc(n, h, w, self.cin, k, k, 1, p, p, h, w, self.cout, flags=ops.CONV_BIAS | (ops.CONV_RELU if relu else 0))
        self.d_bwd = ops.conv_desc(n, h, w, self.cout, k, k, 1, p, p, h, w, self.cin)
        self.d_bwd_res = ops.conv_desc(n, h, w, self.cout, k, k, 1, p, p, h, w, self.cin, flags=ops.CONV_ADD_RES)
        self.w_t = torch.zeros(self.cin, k, k, self.cout, dtype=BF16, device=device)
        self._ws = [ops.conv_attach_workspace(d, device) for d in (self.desc, self.d_bwd, self.d_bwd_res)]

    def flip_entry(self):
        return (self.store.weight(self.name + "/kernel"), self.w_t, self.cout, self.k, self.k, self.cin)

    def quant_entries(self):
        if self.w8 is None:
            return [], []
        return ([(self.store.weight(self.name + "/kernel").view(self.cout, -1), self.w8, self.w8_scale)],
                [(self.w_t.view(self.cin, -1), self.w_t8, self.w_t8_scale)])

    def forward(self, plan, x, y, x8=None):
        ops.conv_zero_counters(plan, self.desc)
        if x8 is not None and self.w8 is not None:
            plan.add(ops.conv2d_fprop_fp8, self.desc, x8.data, self.w8, x8.scale, self.w8_scale, y, bias=self.store.weight(self.name + "/bias"))
            return
        plan.add(ops.conv2d_fprop, self.desc, x, self.store.weight_bf16(self.name + "/kernel"), y, bias=self.store.weight(self.name + "/bias"))

    def backward_params(self, plan, x, dz, x8=None, dz8=None):
        """x8 / dz8: Fp8Twin of x (e4m3) and of dz (e5m2, already quantised in this plan): the fp8 weight gradient."""
        from .feature_extractor import FP8_WGRAD
        st = self.store
        plan.add(ops.colsum_bf16, dz, self.m, self.cout, self.cout, st.grad(self.name + "/bias"))
        if FP8_WGRAD and x8 is not None and dz8 is not None and self.cin % 64 == 0 and self.cout % 64 == 0:
            plan.add(ops.conv2d_wgrad_fp8, self.desc, x8.data, dz8.data, x8.scale, dz8.scale, st.grad(self.name + "/kernel"))
        else:
            plan.add(ops.conv2d_wgrad, self.desc, x, dz, st.grad(self.name + "/kernel"))

    def backward_data(self, plan, dz, gx, add_to_gx=False, red=None, dz8=None):
        """gx = conv^T(dz) [+ gx]; red: fused BatchNorm-backward reduce of the layer that consumes gx; dz8: e5m2 twin of dz."""
        d = self.d_bwd_res if add_to_gx else self.d_bwd
        ops.conv_zero_counters(plan, d)
        if dz8 is not None and self.w8 is not None:
            plan.add(ops.conv2d_dgrad_fp8, d, dz8.data, self.w_t8, dz8.scale, self.w_t8_scale, gx, red=red, res=gx if add_to_gx else None)
        elif red is not None:
            plan.add(ops.conv2d_dgrad_bnreduce, d, dz, self.w_t, gx, red, res=gx if add_to_gx else None)
        else:
            plan.add(ops.conv2d_fprop, d, dz, self.w_t, gx, res=gx if add_to_gx else None)


def _twin(scales, like, device, e5m2=False):
    from .feature_extractor import Fp8Twin
    return Fp8Twin(scales, tuple(like.shape), device, e5m2=e5m2)


def _quantize(plan, twin, src, e5m2=False):
    """one pass: src (bf16) -> twin bytes with this step's scale, amax for the next step's"""
    sc = twin.scales
    plan.add(ops.quantize_fp8, src, sc.qscale(twin.idx), twin.data, sc.amax(twin.idx), e5m2=e5m2)


def _quantize_weights(mod, plan):
    """(outside the train plan: initial / restored weights) e4m3 twins of a module's forward weights and of the transposes just rebuilt"""
    fwd, bwd = mod.quant_entries()
    if fwd or bwd:
        key = tuple(e[1].data_ptr() for e in fwd + bwd)
        if getattr(mod, "_qt_key", None) != key:
            mod._qt, mod._qt_key = ops.make_weight_quant_table(fwd + bwd, mod.device), key
        plan.add(ops.quantize_weights_fp8_batched, *mod._qt)


class FPNNeck:
    def __init__(self, store, stage_channels, decay, device):
        self.store, self.device = store, torch.device(device)
        self.lateral = {l: _Conv(store, "fpn_lateral%d" % l, stage_channels[l], FPN_DIM, 1, decay) for l in LEVELS}
        self.output = {l: _Conv(store, "fpn_output%d" % l, FPN_DIM, FPN_DIM, 3, decay) for l in LEVELS}
        self._deferred = [("fpn_lateral%d/bias" % l, (FPN_DIM,)) for l in LEVELS] + [("fpn_output%d/bias" % l, (FPN_DIM,)) for l in LEVELS]

    def register_biases(self):
        for n, s in self._deferred:
            self.store.register(n, s)

    def convs(self):
        return [self.lateral[l] for l in LEVELS] + [self.output[l] for l in LEVELS]

    def init_weights(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        for c in self.convs():
            fan_in, fan_out = c.k * c.k * c.cin, c.k * c.k * c.cout
            lim = math.sqrt(6.0 / (fan_in + fan_out))                           # Glorot-uniform, as the reference's Dense heads
            self.store.weight(c.name + "/kernel").copy_((torch.rand(c.cout, c.k, c.k, c.cin, generator=g) * 2 - 1) * lim)
            self.store.weight(c.name + "/bias").zero_()

    def set_weights(self, w):
        for c in self.convs():
            self.store.weight(c.name + "/kernel").copy_(torch.as_tensor(w[c.name + "/kernel"]).permute(3, 0, 1, 2))
            self.store.weight(c.name + "/bias").copy_(torch.as_tensor(w[c.name + "/bias"]))

    def get_weights(self):
        out = {}
        for c in self.convs():
            out[c.name + "/kernel"] = self.store.weight(c.name + "/kernel").permute(1, 2, 3, 0).contiguous().cpu()
            out[c.name + "/bias"] = self.store.weight(c.name + "/bias").clone().cpu()
        return out

    def flip_entries(self):
        return [c.flip_entry() for c in self.convs()]

    def refresh_weights(self, plan):
        for c in self.convs():
            plan.add(ops.weights_transpose_flip, *c.flip_entry())
        _quantize_weights(self, plan)

    def setup(self, batch, grids, training, f8_scales=None):
        """grids: {level: (h, w)} of the backbone stage outputs.  f8_scales (precision "fp8", training): the backbone's scale table;
        the 3x3 output convolutions then run on e4m3 / e5m2 twins of the merged maps / of the pyramid gradients, written by one
        quantise pass each (these tensors have no BatchNorm kernel that could write the twin on the way)."""
        dev = self.device
        self.batch, self.grids = batch, dict(grids)
        self.merged, self.p, self.gp, self.gm = {}, {}, {}, {}
        self.f8 = f8_scales if training else None
        self.merged8, self.gp8 = {}, {}
        for l in LEVELS:
            h, w = grids[l]
            self.lateral[l].setup(batch, h, w, dev, fp8=self.f8 is not None)     # (forward on the backbone's twin of the stage output)
            self.output[l].setup(batch, h, w, dev, fp8=self.f8 is not None)
            self.merged[l] = torch.empty(batch * h * w, FPN_DIM, dtype=BF16, device=dev)     # lateral output, merged in place
            self.p[l] = torch.empty(batch, h, w, FPN_DIM, dtype=BF16, device=dev)
            if training:
                self.gp[l] = torch.empty(batch * h * w, FPN_DIM, dtype=BF16, device=dev)       # gradient w.r.t. P_l (RoI + RPN branches)
                self.gm[l] = torch.empty(batch * h * w, FPN_DIM, dtype=BF16, device=dev)       # gradient w.r.t. the merged map
                if self.f8 is not None:
                    self.merged8[l] = _twin(self.f8, self.merged[l], dev)
                    self.gp8[l] = _twin(self.f8, self.gp[l], dev, e5m2=True)
        h4, w4 = grids[4]
        self.grids[5] = ((h4 + 1) // 2, (w4 + 1) // 2)
        self.p[5] = torch.empty(batch, self.grids[5][0], self.grids[5][1], FPN_DIM, dtype=BF16, device=dev)
        if training:
            self.gp[5] = torch.empty(batch * self.grids[5][0] * self.grids[5][1], FPN_DIM, dtype=BF16, device=dev)

    def forward_plan(self, plan, stage_maps, stage_maps8=None):
        """stage_maps {level: bf16 [B*h*w, C_level]} -> pyramid {2..5: bf16 [B, h, w, 256]}; stage_maps8: their Fp8Twins (fp8 training)"""
        b = self.batch
        for l in LEVELS:
            self.lateral[l].forward(plan, stage_maps[l], self.merged[l], (stage_maps8 or {}).get(l) if self.f8 is not None else None)
        for l in (3, 2):
            (ht, wt), (h, w) = self.grids[l + 1], self.grids[l]
            plan.add(ops.upsample_add, self.merged[l + 1], ht, wt, self.merged[l], self.merged[l], b, h, w, FPN_DIM)
        for l in LEVELS:
            if self.merged8:
                _quantize(plan, self.merged8[l], self.merged[l])
            self.output[l].forward(plan, self.merged[l], self.p[l], self.merged8.get(l))
        h4, w4 = self.grids[4]
        plan.add(ops.subsample2, self.p[4], self.p[5], b, h4, w4, FPN_DIM)
        return self.p

    def quant_entries(self):
        fwd, bwd = [], []
        for c in self.convs():
            f, b_ = c.quant_entries()
            fwd += f
            bwd += b_
        return fwd, bwd

    def backward_plan(self, plan, stage_maps, targets, red4=None):
        """self.gp[l] hold the gradients w.r.t. P2..P5 (complete).  targets {level: bf16 [B*h*w, C_level]}: receive the gradient
        w.r.t. the stage outputs (plain writes); red4: BatchNorm-backward reduce of the backbone's last unit, fused into C4's."""
        b = self.batch
        h4, w4 = self.grids[4]
        plan.add(ops.subsample2_bwd_add, self.gp[5], self.gp[4], b, h4, w4, FPN_DIM)
        for l in LEVELS:                                               # fine -> coarse: a merged map's gradient feeds the coarser one
            if self.gp8:
                _quantize(plan, self.gp8[l], self.gp[l], e5m2=True)
            self.output[l].backward_data(plan, self.gp[l], self.gm[l], dz8=self.gp8.get(l))
            if l > 2:
                (h, w), (ht, wt) = self.grids[l - 1], self.grids[l]
                plan.add(ops.upsample_add_bwd, self.gm[l - 1], h, w, self.gm[l], b, ht, wt, FPN_DIM, True)
            self.output[l].backward_params(plan, self.merged[l], self.gp[l], self.merged8.get(l), self.gp8.get(l))
            self.lateral[l].backward_params(plan, stage_maps[l], self.gm[l])
        for l in LEVELS:
            self.lateral[l].backward_data(plan, self.gm[l], targets[l], red=red4 if l == 4 else None)


class RPNDetectorFPN:
    """The reference's RPN head on every pyramid level with shared weights (interface of RPNDetector where the plan builder uses it)."""

    def __init__(self, image_shape, grids, config, store, device):
        self._image_shape = tuple(image_shape)
        self.config, self.store, self.device = config, store, torch.device(device)
        a = config["anchors"]
        assert len(a["scales"]) == len(RPN_LEVELS), "one anchor scale per pyramid level (oracle/fpn.py)"
        self.apl = len(a["aspect_ratios"])
        self.ws = int(config["window_size"])
        self.grids = dict(grids)
        wd = float(config["weight_decay"])
        store.register("rpn_intermediate_layer/kernel", (256, self.ws, self.ws, FPN_DIM), decay=wd)
        store.register("rpn_heads/kernel", (RPN_LD, 1, 1, 256), decay=wd)
        self._deferred = [("rpn_intermediate_layer/bias", (256,)), ("rpn_heads/bias", (RPN_LD,))]
        h, w = image_shape[0], image_shape[1]
        self.anchors, self.keep, self.inside, self.clipped, self.num_anchors = {}, {}, {}, {}, {}
        for i, l in enumerate(RPN_LEVELS):
            gh, gw = grids[l]
            na = gh * gw * self.apl
            an = torch.empty(na, 4, device=self.device)
            ops.anchors_generate(an, gh, gw, [a["scales"][i]], a["aspect_ratios"], float(a["base_anchor_shape"][0]), float(a["base_anchor_shape"][1]),
                                 float(STRIDE[l]), float(STRIDE[l]))
            c = an.cpu().numpy()
            ins = (c[:, 0] >= 0) & (c[:, 1] >= 0) & (c[:, 2] <= w) & (c[:, 3] <= h)
            self.anchors[l], self.num_anchors[l] = an, na
            self.keep[l] = torch.from_numpy(np.nonzero(ins)[0].astype(np.int32)).to(self.device)
            self.inside[l] = an[self.keep[l].long()].contiguous()
            cl = torch.empty_like(an)
            ops.clip_to_window(an, cl, [0, 0, w, h])
            self.clipped[l] = cl

    def register_biases(self):
        for n, s in self._deferred:
            self.store.register(n, s)

    def init_weights(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        st, A = self.store, self.apl
        t = torch.empty(256, self.ws, self.ws, FPN_DIM)
        torch.nn.init.trunc_normal_(t, 0.0, 0.01, -0.02, 0.02, generator=g)
        st.weight("rpn_intermediate_layer/kernel").copy_(t)
        hd = torch.zeros(RPN_LD, 1, 1, 256)
        t = torch.empty(6 * A, 1, 1, 256)
        torch.nn.init.trunc_normal_(t, 0.0, 0.01, -0.02, 0.02, generator=g)
        hd[:6 * A] = t
        st.weight("rpn_heads/kernel").copy_(hd)
        st.weight("rpn_intermediate_layer/bias").zero_()
        st.weight("rpn_heads/bias").zero_()

    def set_weights(self, w):
        st, A = self.store, self.apl
        st.weight("rpn_intermediate_layer/kernel").copy_(torch.as_tensor(w["rpn_intermediate_layer/kernel"]).permute(3, 0, 1, 2))
        st.weight("rpn_intermediate_layer/bias").copy_(torch.as_tensor(w["rpn_intermediate_layer/bias"]))
        hd = torch.zeros(RPN_LD, 1, 1, 256)
        hd[:2 * A] = torch.as_tensor(w["rpn_classification_head/kernel"]).permute(3, 0, 1, 2)
        hd[2 * A:6 * A] = torch.as_tensor(w["rpn_regression_head/kernel"]).permute(3, 0, 1, 2)
        st.weight("rpn_heads/kernel").copy_(hd)
        b = torch.zeros(RPN_LD)
        b[:2 * A] = torch.as_tensor(w["rpn_classification_head/bias"])
        b[2 * A:6 * A] = torch.as_tensor(w["rpn_regression_head/bias"])
        st.weight("rpn_heads/bias").copy_(b)

    def get_weights(self):
        st, A = self.store, self.apl
        hd, b = st.weight("rpn_heads/kernel").cpu(), st.weight("rpn_heads/bias").cpu()
        return {"rpn_intermediate_layer/kernel": st.weight("rpn_intermediate_layer/kernel").permute(1, 2, 3, 0).contiguous().cpu(),
                "rpn_intermediate_layer/bias": st.weight("rpn_intermediate_layer/bias").clone().cpu(),
                "rpn_classification_head/kernel": hd[:2 * A].permute(1, 2, 3, 0).contiguous(), "rpn_classification_head/bias": b[:2 * A].clone(),
                "rpn_regression_head/kernel": hd[2 * A:6 * A].permute(1, 2, 3, 0).contiguous(), "rpn_regression_head/bias": b[2 * A:6 * A].clone()}

    def setup(self, batch, training, f8_scales=None):
        """f8_scales (precision "fp8", training): the shared 3x3 convolution runs on e4m3 twins of the pyramid levels and e5m2 twins of
        its incoming gradients (one quantise pass each), with one e4m3 copy of its weights per direction."""
        dev = self.device
        self.batch = batch
        self.f8 = f8_scales if training else None
        self.w_inter8 = None
        if self.f8 is not None:
            self.w_inter8 = torch.zeros(256, self.ws, self.ws, FPN_DIM, dtype=ops.FP8, device=dev)
            self.w_inter8_scale = torch.ones(256, dtype=torch.float32, device=dev)
            self.w_inter_t8 = torch.zeros(FPN_DIM, self.ws, self.ws, 256, dtype=ops.FP8, device=dev)
            self.w_inter_t8_scale = torch.ones(FPN_DIM, dtype=torch.float32, device=dev)
        self.n_level = {l: int(self.keep[l].numel()) if training else self.num_anchors[l] for l in RPN_LEVELS}
        self.offset, n = {}, 0
        for l in RPN_LEVELS:
            self.offset[l] = n
            n += self.n_level[l]
        self.n = n
        self.regions_all = torch.cat([(self.inside if training else self.clipped)[l] for l in RPN_LEVELS], 0).contiguous()
        self.scores = torch.empty(batch, n, 2, device=dev)
        self.deltas = torch.empty(batch, n, 1, 4, device=dev)
        self.w_inter_t = torch.zeros(FPN_DIM, self.ws, self.ws, 256, dtype=BF16, device=dev)
        self.w_heads_t = torch.zeros(256, 1, 1, RPN_LD, dtype=BF16, device=dev)
        p = self.ws // 2
        self.lv = {}
        for l in RPN_LEVELS:
            gh, gw = self.grids[l]
            m = batch * gh * gw
            e = {"m": m, "gh": gh, "gw": gw}
            e["d_inter"] = ops.conv_desc(batch, gh, gw, FPN_DIM, self.ws, self.ws, 1, p, p, gh, gw, 256, flags=ops.CONV_BIAS | ops.CONV_RELU | ops.CONV_WGRAD_ACCUMULATE)
            e["d_heads"] = ops.conv_desc(batch, gh, gw, 256, 1, 1, 1, 0, 0, gh, gw, RPN_LD, flags=ops.CONV_BIAS | ops.CONV_OUT_F32 | ops.CONV_WGRAD_ACCUMULATE)
            e["f"] = torch.empty(m, 256, dtype=BF16, device=dev)
            e["head"] = torch.empty(m, RPN_LD, device=dev)
            if self.f8 is not None:
                e["p8"] = _twin(self.f8, torch.empty(m, FPN_DIM, device="meta"), dev)
                e["dz_f8"] = _twin(self.f8, torch.empty(m, 256, device="meta"), dev, e5m2=True)
            e["ws"] = [ops.conv_attach_workspace(e["d_inter"], dev)]
            if training:
                e["dhead32"] = torch.zeros(m, RPN_LD, device=dev)
                e["dhead"] = torch.empty(m, RPN_LD, dtype=BF16, device=dev)
                e["g_f"] = torch.empty(m, 256, dtype=BF16, device=dev)
                e["dz_f"] = torch.empty(m, 256, dtype=BF16, device=dev)
                e["d_heads_bwd"] = ops.conv_desc(batch, gh, gw, RPN_LD, 1, 1, 1, 0, 0, gh, gw, 256)
                e["d_inter_bwd"] = ops.conv_desc(batch, gh, gw, 256, self.ws, self.ws, 1, p, p, gh, gw, FPN_DIM)
                e["d_inter_bwd_res"] = ops.conv_desc(batch, gh, gw, 256, self.ws, self.ws, 1, p, p, gh, gw, FPN_DIM, flags=ops.CONV_ADD_RES)
                e["ws"] += [ops.conv_attach_workspace(e["d_inter_bwd"], dev), ops.conv_attach_workspace(e["d_inter_bwd_res"], dev)]
            self.lv[l] = e

    def flip_entries(self):
        st = self.store
        return [(st.weight("rpn_intermediate_layer/kernel"), self.w_inter_t, 256, self.ws, self.ws, FPN_DIM),
                (st.weight("rpn_heads/kernel"), self.w_heads_t, RPN_LD, 1, 1, 256)]

    def quant_entries(self):
        if self.w_inter8 is None:
            return [], []
        return ([(self.store.weight("rpn_intermediate_layer/kernel").view(256, -1), self.w_inter8, self.w_inter8_scale)],
                [(self.w_inter_t.view(FPN_DIM, -1), self.w_inter_t8, self.w_inter_t8_scale)])

    def refresh_weights(self, plan):
        for e in self.flip_entries():
            plan.add(ops.weights_transpose_flip, *e)
        _quantize_weights(self, plan)

    def forward_plan(self, plan, pyramid, training, decoded=None):
        st = self.store
        W, H = float(self._image_shape[1]), float(self._image_shape[0])
        for l in RPN_LEVELS:
            e = self.lv[l]
            ops.conv_zero_counters(plan, e["d_inter"])
            if self.f8 is not None:
                _quantize(plan, e["p8"], pyramid[l])
                plan.add(ops.conv2d_fprop_fp8, e["d_inter"], e["p8"].data, self.w_inter8, e["p8"].scale, self.w_inter8_scale, e["f"],
                         bias=st.weight("rpn_intermediate_layer/bias"))
            else:
                plan.add(ops.conv2d_fprop, e["d_inter"], pyramid[l], st.weight_bf16("rpn_intermediate_layer/kernel"), e["f"],
                         bias=st.weight("rpn_intermediate_layer/bias"))
            plan.add(ops.conv2d_fprop, e["d_heads"], e["f"], st.weight_bf16("rpn_heads/kernel"), e["head"], bias=st.weight("rpn_heads/bias"))
            off, n = self.offset[l], self.n_level[l]
            plan.add(ops.rpn_head_post_level, e["head"], RPN_LD, self.batch, self.num_anchors[l], self.apl, self.keep[l] if training else None, n,
                     self.scores, self.deltas, self.regions_all[off:off + n], decoded, W, H, self.n, off)
        return {"regions": self.regions_all, "pred_scores": self.scores, "pred_boxes": self.deltas}

    def backward_plan(self, plan, dlogits_s, ddeltas_s, indices, num_samples, pyramid, gp, gp_written):
        """Per-sample loss gradients -> shared head parameter gradients (accumulated over the levels) and the gradient w.r.t. every
        level's map: ADDED into gp[l] where gp_written[l] (the RoI branch wrote it), plain otherwise."""
        self.backward_params_plan(plan, dlogits_s, ddeltas_s, indices, num_samples, pyramid)
        self.backward_data_plan(plan, gp, gp_written)

    def backward_params_plan(self, plan, dlogits_s, ddeltas_s, indices, num_samples, pyramid):
        """Everything of the RPN backward pass that does not need the RoI branch's gradient: the training plan runs it on the RPN's
        side stream, under proposal NMS / RoI pooling / the Fast-RCNN heads (as RPNDetector.backward_params_plan in the C4 plan)."""
        st = self.store
        for l in RPN_LEVELS:
            e = self.lv[l]
            plan.zero(e["dhead32"])
            plan.add(ops.rpn_head_grad_level, dlogits_s, ddeltas_s, indices, self.keep[l], self.batch, num_samples, self.num_anchors[l], self.apl,
                     e["dhead32"], RPN_LD, self.offset[l], self.n_level[l])
            plan.add(ops.cast_colsum, e["dhead32"], e["dhead"], e["m"], RPN_LD, st.grad("rpn_heads/bias"))
            plan.add(ops.conv2d_wgrad, e["d_heads"], e["f"], e["dhead"], st.grad("rpn_heads/kernel"))
            plan.add(ops.conv2d_fprop, e["d_heads_bwd"], e["dhead"], self.w_heads_t, e["g_f"])
            plan.add(ops.relu_bwd_colsum, e["g_f"], e["f"], e["dz_f"], e["m"], 256, st.grad("rpn_intermediate_layer/bias"))
            if self.f8 is not None:
                _quantize(plan, e["dz_f8"], e["dz_f"], e5m2=True)
            if self.f8 is not None and FP8_WGRAD:
                plan.add(ops.conv2d_wgrad_fp8, e["d_inter"], e["p8"].data, e["dz_f8"].data, e["p8"].scale, e["dz_f8"].scale,
                         st.grad("rpn_intermediate_layer/kernel"))
            else:
                plan.add(ops.conv2d_wgrad, e["d_inter"], pyramid[l], e["dz_f"], st.grad("rpn_intermediate_layer/kernel"))

    def backward_data_plan(self, plan, gp, gp_written):
        """gp[l] (+)= the 3x3 convolution's data gradient of every level (after backward_params_plan)."""
        for l in RPN_LEVELS:
            e = self.lv[l]
            d = e["d_inter_bwd_res"] if gp_written.get(l) else e["d_inter_bwd"]
            res = gp[l] if gp_written.get(l) else None
            ops.conv_zero_counters(plan, d)
            if self.f8 is not None:
                plan.add(ops.conv2d_dgrad_fp8, d, e["dz_f8"].data, self.w_inter_t8, e["dz_f8"].scale, self.w_inter_t8_scale, gp[l], res=res)
            else:
                plan.add(ops.conv2d_fprop, d, e["dz_f"], self.w_inter_t, gp[l], res=res)


class FastRCNNDetectorFPN(FastRCNNDetector):
    """The reference's Fast-RCNN heads with every RoI pooled from its pyramid level (Lin et al. eq. 1)."""

    def __init__(self, image_shape, num_classes, config, grids, store, device):
        super().__init__(image_shape, num_classes, config, feature_channels=FPN_DIM, store=store, device=device)
        self.grids = {l: grids[l] for l in LEVELS}

    def setup(self, batch, num_rois, training, num_samples=0):
        super().setup(batch, num_rois, self.grids[4][0], self.grids[4][1], training, num_samples)
        self.levels = torch.zeros(batch * num_rois, dtype=torch.int32, device=self.device)

    def forward_plan(self, plan, pyramid, rois, regions_done=False, decoded=None):
        st = self.store
        plan.add(ops.roi_assign_levels, rois, float(self._image_shape[1]), float(self._image_shape[0]), self.levels)
        for l in LEVELS:
            gh, gw = self.grids[l]
            plan.add(ops.roi_crop_pool_fwd_level, pyramid[l], rois, self.batch, self.p, gh, gw, self.cf, self.ps, self.ks, self.pooled, self.argmax,
                     self.levels, l)
        plan.zero(self.logits)
        plan.add(ops.conv2d_fprop, self.d_fwd, self.pooled, st.weight_bf16("fast_rcnn_heads/kernel"), self.logits)
        self.decoded_done = self.head_post_plan(plan, regions_done, decoded)
        if not regions_done:
            self.regions_plan(plan, rois)
        return {"regions": self.regions_abs, "pred_scores": self.scores, "pred_boxes": self.deltas}

    def backward_plan(self, plan, rois, gp, bias_grad_done=False):
        """(the loss launch has written self.dhead_s / self.rows) -> head parameter gradients; gp[l] (l = 2..4) receive the
        COMPLETE RoI-branch gradient of their level's map."""
        st = self.store
        plan.add(ops.conv2d_fprop, self.d_dgrad, self.dhead_s, self.w_t, self.dpooled_s)
        for l in LEVELS:
            gh, gw = self.grids[l]
            plan.add(ops.roi_crop_pool_bwd_bf16_level, self.dpooled_s, self.argmax, rois, self.rows, self.rs, self.batch, self.p, gh, gw, self.cf,
                     self.ps, self.ks, gp[l], self.levels, l)
        if not bias_grad_done:
            plan.add(ops.colsum_bf16, self.dhead_s, self.rs, RCNN_LD, RCNN_LD, st.grad("fast_rcnn_heads/bias"))
        plan.add(ops.conv2d_wgrad, self.d_wgrad, self.pooled, self.dhead_s, st.grad("fast_rcnn_heads/kernel"), RCNN_LD, self.rows)
