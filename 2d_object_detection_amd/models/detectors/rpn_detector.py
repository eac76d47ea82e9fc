"""Region Proposal Network detector with the call surface of reference
models/detectors/rpn_detector.py, on gfx950 kernels.

The two 1x1 heads (24 objectness logits + 48 box deltas per location) are ONE implicit GEMM with
the filters concatenated and zero-padded to 128 output channels (fp32 output), followed by a
small kernel that does the pair-softmax and the in-image anchor gather (training) in one pass.
"""
import math

import numpy as np
import torch

from ... import ops
from ...runtime import ParamStore, Plan

BF16 = torch.bfloat16
HEAD_LD = 128


class RPNDetector:
    def __init__(self, image_shape, feature_maps_shape, config, name="region_proposal_network_detector", store=None, device="cuda",
                 precision="bf16"):
        """reference rpn_detector.py:8-58 (same positional arguments).  precision "fp8": the 3x3 convolution's training forward
        pass on e4m3 operands (frcnn_conv2d_fprop_fp8)."""
        self.name = name
        self.precision = precision
        self._image_shape = tuple(image_shape)
        _, gh, gw, cf = feature_maps_shape
        self.gh, self.gw, self.cf = gh, gw, cf
        self.config = config
        self.device = torch.device(device)
        a = config["anchors"]
        self.apl = len(a["scales"]) * len(a["aspect_ratios"])
        assert 6 * self.apl <= HEAD_LD
        self.ws = int(config["window_size"])
        self.num_anchors = gh * gw * self.apl
        # anchors (rpn_detector.py:22,162-199) -- generated once on the device
        self._anchors = torch.empty(self.num_anchors, 4, device=self.device)
        ops.anchors_generate(self._anchors, gh, gw, a["scales"], a["aspect_ratios"], float(a["base_anchor_shape"][0]),
                             float(a["base_anchor_shape"][1]))
        # in-image anchor indices (rpn_detector.py:216-225): init-time, host side
        an = self._anchors.cpu().numpy()
        h, w = image_shape[0], image_shape[1]
        inside = (an[:, 0] >= 0) & (an[:, 1] >= 0) & (an[:, 2] <= w) & (an[:, 3] <= h)
        self._keep = torch.from_numpy(np.nonzero(inside)[0].astype(np.int32)).to(self.device)
        self._anchors_inside = self._anchors[self._keep.long()].contiguous()
        self._anchors_clipped = torch.empty_like(self._anchors)
        ops.clip_to_window(self._anchors, self._anchors_clipped, [0, 0, w, h])          # rpn_detector.py:93-94

        self.own_store = store is None
        self.store = store if store is not None else ParamStore(self.device)
        wd = float(config["weight_decay"])
        st = self.store
        st.register("rpn_intermediate_layer/kernel", (256, self.ws, self.ws, cf), decay=wd)
        st.register("rpn_heads/kernel", (HEAD_LD, 1, 1, 256), decay=wd)     # rows [0,2A) cls, [2A,6A) reg, rest 0
        self._deferred = [("rpn_intermediate_layer/bias", (256,)), ("rpn_heads/bias", (HEAD_LD,))]
        if self.own_store:
            self.register_biases()
            st.finalize()
            self.init_weights(0)
        self._plans = {}

    def register_biases(self):
        for n, s in self._deferred:
            self.store.register(n, s)

    # ------------------------------------------------------------------ parameters
    def init_weights(self, seed=0):
        """TruncatedNormal(0, 0.01) kernels, zero biases (rpn_detector.py:24)."""
        g = torch.Generator().manual_seed(seed)
        st, A = self.store, self.apl
        t = torch.empty(256, self.ws, self.ws, self.cf)
        torch.nn.init.trunc_normal_(t, 0.0, 0.01, -0.02, 0.02, generator=g)
        st.weight("rpn_intermediate_layer/kernel").copy_(t)
        h = torch.zeros(HEAD_LD, 1, 1, 256)
        t = torch.empty(6 * A, 1, 1, 256)
        torch.nn.init.trunc_normal_(t, 0.0, 0.01, -0.02, 0.02, generator=g)
        h[:6 * A] = t
        st.weight("rpn_heads/kernel").copy_(h)
        st.weight("rpn_intermediate_layer/bias").zero_()
        st.weight("rpn_heads/bias").zero_()

    def set_weights(self, w):
        st, A = self.store, self.apl
        st.weight("rpn_intermediate_layer/kernel").copy_(torch.as_tensor(w["rpn_intermediate_layer/kernel"]).permute(3, 0, 1, 2))
        st.weight("rpn_intermediate_layer/bias").copy_(torch.as_tensor(w["rpn_intermediate_layer/bias"]))
        h = torch.zeros(HEAD_LD, 1, 1, 256)
        h[:2 * A] = torch.as_tensor(w["rpn_classification_head/kernel"]).permute(3, 0, 1, 2)
        h[2 * A:6 * A] = torch.as_tensor(w["rpn_regression_head/kernel"]).permute(3, 0, 1, 2)
        st.weight("rpn_heads/kernel").copy_(h)
        b = torch.zeros(HEAD_LD)
        b[:2 * A] = torch.as_tensor(w["rpn_classification_head/bias"])
        b[2 * A:6 * A] = torch.as_tensor(w["rpn_regression_head/bias"])
        st.weight("rpn_heads/bias").copy_(b)

    def get_weights(self):
        st, A = self.store, self.apl
        h = st.weight("rpn_heads/kernel").cpu()
        b = st.weight("rpn_heads/bias").cpu()
        return {
            "rpn_intermediate_layer/kernel": st.weight("rpn_intermediate_layer/kernel").permute(1, 2, 3, 0).contiguous().cpu(),
            "rpn_intermediate_layer/bias": st.weight("rpn_intermediate_layer/bias").clone().cpu(),
            "rpn_classification_head/kernel": h[:2 * A].permute(1, 2, 3, 0).contiguous(),
            "rpn_classification_head/bias": b[:2 * A].clone(),
            "rpn_regression_head/kernel": h[2 * A:6 * A].permute(1, 2, 3, 0).contiguous(),
            "rpn_regression_head/bias": b[2 * A:6 * A].clone(),
        }

    # ------------------------------------------------------------------ plans
    def setup(self, batch, training, f8_scales=None):
        """f8_scales: the backbone's Fp8Scales table (fp8 training): the e5m2 twin of the 3x3 convolution's incoming gradient gets
        its column there, so one update launch per step serves every fp8 tensor."""
        dev, gh, gw, cf = self.device, self.gh, self.gw, self.cf
        self.batch = batch
        m = batch * gh * gw
        self.m = m
        p = self.ws // 2
        self.d_inter = ops.conv_desc(batch, gh, gw, cf, self.ws, self.ws, 1, p, p, gh, gw, 256, flags=ops.CONV_BIAS | ops.CONV_RELU)
        self.d_heads = ops.conv_desc(batch, gh, gw, 256, 1, 1, 1, 0, 0, gh, gw, HEAD_LD, flags=ops.CONV_BIAS | ops.CONV_OUT_F32)
        self._conv_ws = [ops.conv_attach_workspace(self.d_inter, dev)]      # (K = 9 * 1024 on fewer tiles than CUs: split-K fix-up form)
        self.f = torch.empty(m, 256, dtype=BF16, device=dev)
        self.head = torch.empty(m, HEAD_LD, device=dev)
        self.n = int(self._keep.numel()) if training else self.num_anchors
        self.scores = torch.empty(batch, self.n, 2, device=dev)
        self.deltas = torch.empty(batch, self.n, 1, 4, device=dev)
        self.w_inter_t = torch.zeros(cf, self.ws, self.ws, 256, dtype=BF16, device=dev)
        self.w_heads_t = torch.zeros(256, 1, 1, HEAD_LD, dtype=BF16, device=dev)
        self.w_inter8 = None
        self._quant_table = None
        if training and getattr(self, "precision", "bf16") == "fp8" and cf % 128 == 0:
            # fp8 forward of the 3x3 intermediate convolution (K = 9 * 1024: the largest single layer of the step)
            self.w_inter8 = torch.zeros(256, self.ws, self.ws, cf, dtype=ops.FP8, device=dev)
            self.w_inter8_scale = torch.ones(256, dtype=torch.float32, device=dev)
        self.dz_f8 = None
        if training:
            self.dhead32 = torch.zeros(m, HEAD_LD, device=dev)
            self.dhead = torch.empty(m, HEAD_LD, dtype=BF16, device=dev)
            self.g_f = torch.empty(m, 256, dtype=BF16, device=dev)
            self.dz_f = torch.empty(m, 256, dtype=BF16, device=dev)
            self.d_heads_bwd = ops.conv_desc(batch, gh, gw, HEAD_LD, 1, 1, 1, 0, 0, gh, gw, 256)
            self.d_inter_bwd = ops.conv_desc(batch, gh, gw, 256, self.ws, self.ws, 1, p, p, gh, gw, cf, flags=ops.CONV_ADD_RES)
            self._conv_ws.append(ops.conv_attach_workspace(self.d_inter_bwd, dev))
            # the same data gradient WRITTEN (not added): the form that runs on the RPN's side stream before the RoI backward pass
            self.d_inter_bwd_plain = ops.conv_desc(batch, gh, gw, 256, self.ws, self.ws, 1, p, p, gh, gw, cf)
            self._conv_ws.append(ops.conv_attach_workspace(self.d_inter_bwd_plain, dev))
            from ..feature_extractor import FP8_BWD
            if self.w_inter8 is not None and f8_scales is not None and FP8_BWD:
                # fp8 data gradient of the 3x3 convolution: e5m2 twin of dz_f (written by a quantise pass behind the ReLU backward),
                # e4m3 twin of the tap-flipped transposed weights, one scale per row (= per feature-map channel)
                from ..feature_extractor import Fp8Twin
                self.dz_f8 = Fp8Twin(f8_scales, (m, 256), dev, e5m2=True)
                self.w_inter_t8 = torch.zeros(cf, self.ws, self.ws, 256, dtype=ops.FP8, device=dev)
                self.w_inter_t8_scale = torch.ones(cf, dtype=torch.float32, device=dev)

    def refresh_weights(self, plan):
        st = self.store
        plan.add(ops.weights_transpose_flip, st.weight("rpn_intermediate_layer/kernel"), self.w_inter_t, 256, self.ws, self.ws, self.cf)
        plan.add(ops.weights_transpose_flip, st.weight("rpn_heads/kernel"), self.w_heads_t, HEAD_LD, 1, 1, 256)
        if self.quant_entries():
            if getattr(self, "_quant_table", None) is None:
                self._quant_table = ops.make_weight_quant_table(self.quant_entries() + self.quant_entries_bwd(), self.device)
            plan.add(ops.quantize_weights_fp8_batched, *self._quant_table)

    def quant_entries_bwd(self):
        if getattr(self, "dz_f8", None) is None:
            return []
        return [(self.w_inter_t.view(self.cf, -1), self.w_inter_t8, self.w_inter_t8_scale)]

    def quant_entries(self):
        """fp8 mode: (fp32 master rows, e4m3 destination, per-row scale) of the 3x3 intermediate convolution's weights."""
        if getattr(self, "w_inter8", None) is None:
            return []
        return [(self.store.weight("rpn_intermediate_layer/kernel").view(256, -1), self.w_inter8, self.w_inter8_scale)]

    def flip_entries(self):
        st = self.store
        return [(st.weight("rpn_intermediate_layer/kernel"), self.w_inter_t, 256, self.ws, self.ws, self.cf),
                (st.weight("rpn_heads/kernel"), self.w_heads_t, HEAD_LD, 1, 1, 256)]

    def regions(self, training):
        """The regions the RPN predicts on: the anchors inside the image when training (reference rpn_detector.py:112-117),
        all anchors clipped to the image otherwise."""
        return self._anchors_inside if training else self._anchors_clipped

    def forward_plan(self, plan, feature_maps, training, decoded=None, feature_maps8=None):
        """decoded [B,n,1,4]: also receives the decoded proposals (first launch of post-processing) from the head-post kernel.
        feature_maps8: Fp8Twin of the feature maps (fp8 training): the 3x3 convolution reads e4m3 operands."""
        st = self.store
        ops.conv_zero_counters(plan, self.d_inter)
        self._feature_maps8 = feature_maps8 if (feature_maps8 is not None and self.w_inter8 is not None) else None
        if feature_maps8 is not None and self.w_inter8 is not None:
            plan.add(ops.conv2d_fprop_fp8, self.d_inter, feature_maps8.data, self.w_inter8, feature_maps8.scale, self.w_inter8_scale, self.f,
                     bias=st.weight("rpn_intermediate_layer/bias"))
        else:
            plan.add(ops.conv2d_fprop, self.d_inter, feature_maps, st.weight_bf16("rpn_intermediate_layer/kernel"), self.f,
                     bias=st.weight("rpn_intermediate_layer/bias"))
        plan.add(ops.conv2d_fprop, self.d_heads, self.f, st.weight_bf16("rpn_heads/kernel"), self.head, bias=st.weight("rpn_heads/bias"))
        keep = self._keep if training else None
        regions = self.regions(training)
        if decoded is None:
            plan.add(ops.rpn_head_post, self.head, HEAD_LD, self.batch, self.num_anchors, self.apl, keep, self.n, self.scores, self.deltas)
        else:
            plan.add(ops.rpn_head_post_decode, self.head, HEAD_LD, self.batch, self.num_anchors, self.apl, keep, self.n, self.scores,
                     self.deltas, regions, decoded, float(self._image_shape[1]), float(self._image_shape[0]))
        return {"regions": regions, "pred_scores": self.scores, "pred_boxes": self.deltas}

    def backward_plan(self, plan, dlogits_s, ddeltas_s, indices, num_samples, feature_maps, g_feat):
        """Per-sample loss gradients -> parameter gradients; ADDS the feature-map gradient into
        g_feat (bf16 [M, C], already holding the RoI-branch gradient)."""
        self.backward_params_plan(plan, dlogits_s, ddeltas_s, indices, num_samples, feature_maps)
        self.backward_data_plan(plan, g_feat)

    def head_grad_target(self, plan):
        """(keep, num_anchors, anchors per location, dhead32, ld): where the fused loss launch (ops.losses_rpn_head_grad)
        scatter-adds the per-sample gradients; registers dhead32 with the plan's zero fill."""
        plan.zero(self.dhead32)
        return self._keep, self.num_anchors, self.apl, self.dhead32, HEAD_LD

    def backward_params_plan(self, plan, dlogits_s, ddeltas_s, indices, num_samples, feature_maps, head_grad_done=False):
        """Everything of the RPN backward pass that does not need the RoI-branch gradient (a side-stream branch of the
        training step runs it next to the Fast-RCNN backward pass).  head_grad_done: dhead32 already holds the scattered
        per-sample gradients (head_grad_target)."""
        st = self.store
        if not head_grad_done:
            plan.zero(self.dhead32)
            plan.add(ops.rpn_head_grad, dlogits_s, ddeltas_s, indices, self._keep, self.batch, num_samples, self.num_anchors, self.apl,
                     self.dhead32, HEAD_LD)
        # (each bias gradient comes out of the pass that produces its matrix: cast + column sums, ReLU backward + column sums)
        plan.add(ops.cast_colsum, self.dhead32, self.dhead, self.m, HEAD_LD, st.grad("rpn_heads/bias"))
        plan.add(ops.conv2d_wgrad, self.d_heads, self.f, self.dhead, st.grad("rpn_heads/kernel"))
        plan.add(ops.conv2d_fprop, self.d_heads_bwd, self.dhead, self.w_heads_t, self.g_f)
        plan.add(ops.relu_bwd_colsum, self.g_f, self.f, self.dz_f, self.m, 256, st.grad("rpn_intermediate_layer/bias"))
        if self.dz_f8 is not None:
            sc = self.dz_f8.scales
            plan.add(ops.quantize_fp8, self.dz_f, sc.qscale(self.dz_f8.idx), self.dz_f8.data, sc.amax(self.dz_f8.idx), e5m2=True)
        from ..feature_extractor import FP8_WGRAD
        f8 = getattr(self, "_feature_maps8", None)
        if FP8_WGRAD and f8 is not None and self.dz_f8 is not None:
            # fp8 weight gradient of the 3x3 convolution: the feature maps' e4m3 twin x the e5m2 twin of dz_f quantised above
            plan.add(ops.conv2d_wgrad_fp8, self.d_inter, f8.data, self.dz_f8.data, f8.scale, self.dz_f8.scale, st.grad("rpn_intermediate_layer/kernel"))
        else:
            plan.add(ops.conv2d_wgrad, self.d_inter, feature_maps, self.dz_f, st.grad("rpn_intermediate_layer/kernel"))

    def backward_data_plan(self, plan, g_feat, consumer=None, plain=False):
        """consumer: the backbone's last conv unit -- g_feat is complete after this kernel, so it also runs that unit's
        BatchNorm-backward reduce.  plain: g_feat = the RPN's data gradient alone (written, not added: the RoI backward pass adds its
        own afterwards, ops.roi_crop_pool_bwd_bf16_add) -- the kernel then needs nothing from the RoI branch."""
        if plain:
            assert consumer is None
            ops.conv_zero_counters(plan, self.d_inter_bwd_plain)
            if self.dz_f8 is not None:
                plan.add(ops.conv2d_dgrad_fp8, self.d_inter_bwd_plain, self.dz_f8.data, self.w_inter_t8, self.dz_f8.scale, self.w_inter_t8_scale, g_feat)
            else:
                plan.add(ops.conv2d_fprop, self.d_inter_bwd_plain, self.dz_f, self.w_inter_t, g_feat)
            return
        ops.conv_zero_counters(plan, self.d_inter_bwd)
        red = None
        if consumer is not None:
            red = consumer.reduce_args(relu=True)
            plan.hold(red)
        if self.dz_f8 is not None:
            plan.add(ops.conv2d_dgrad_fp8, self.d_inter_bwd, self.dz_f8.data, self.w_inter_t8, self.dz_f8.scale, self.w_inter_t8_scale, g_feat,
                     red=red, res=g_feat)
        elif consumer is not None:
            plan.add(ops.conv2d_dgrad_bnreduce, self.d_inter_bwd, self.dz_f, self.w_inter_t, g_feat, red, res=g_feat)
        else:
            plan.add(ops.conv2d_fprop, self.d_inter_bwd, self.dz_f, self.w_inter_t, g_feat, res=g_feat)

    # ------------------------------------------------------------------ reference call surface
    def __call__(self, feature_maps, training=False):
        """reference rpn_detector.py:60-96.  feature_maps: bf16 [B,gh,gw,C] CUDA.  Returns dict
        regions [A',4], pred_scores [B,A',2], pred_boxes [B,A',1,4] (fp32)."""
        key = (int(feature_maps.shape[0]), bool(training))
        if key not in self._plans:
            self.setup(key[0], training)
            self._x = torch.empty(feature_maps.shape, dtype=BF16, device=self.device)
            plan = Plan("rpn")
            plan.add(self.store.refresh_bf16)
            self._out = self.forward_plan(plan, self._x, training)
            self._plans = {key: plan}
        self._x.copy_(feature_maps)
        self._plans[key].run()
        return self._out

    def get_training_samples(self, gt_labels, gt_boxes, regions, pred_scores, pred_boxes, foreground_iou_interval,
                             background_iou_interval, num_samples, foreground_proportion, seed=0, step=None):
        """reference rpn_detector.py:98-160 (forward values)."""
        from ...utils.training import generate_targets, get_sample_indices
        tl, tb = generate_targets(gt_labels, gt_boxes, regions, self._image_shape, foreground_iou_interval, background_iou_interval,
                                  objectness=True)
        idx = get_sample_indices(tl, num_samples, foreground_proportion, seed=seed, step=step, stream_base=0).long()
        ar = torch.arange(tl.shape[0], device=tl.device)[:, None]
        return {"target_labels": tl[ar, idx], "pred_scores": pred_scores[ar, idx], "target_boxes": tb[ar, idx],
                "pred_boxes": pred_boxes[ar, idx], "sample_indices": idx}
