"""Fast-RCNN detector with the call surface of reference
models/detectors/fast_rcnn_detector.py, on gfx950 kernels.

RoI pooling is the fused crop_and_resize(14x14)+MaxPool(2x2) kernel; the two Dense heads are ONE
split-K implicit GEMM over the pooled rows with the (8 + 28) output filters concatenated and
zero-padded to 64.  Only the sampled RoI rows take part in the backward pass.
"""
import math
import os

import torch

from ... import ops
from ...runtime import ParamStore, Plan

BF16 = torch.bfloat16
HEAD_LD = 64


class ROIPooling:
    """reference fast_rcnn_detector.py:133-177 (flatten=True, keep_batch_dim=True form)."""

    def __init__(self, pooled_size, kernel_size, name="regions_of_interest_pooling"):
        self._pooled_size, self._kernel_size, self.name = int(pooled_size), int(kernel_size), name

    def __call__(self, feature_maps, rois, flatten=True, keep_batch_dim=True):
        b, hf, wf, c = feature_maps.shape
        p = rois.shape[1]
        ps, ks = self._pooled_size, self._kernel_size
        pooled = torch.empty(b * p, ps * ps * c, dtype=BF16, device=feature_maps.device)
        argmax = torch.empty(b * p, ps * ps * c, dtype=torch.uint8, device=feature_maps.device)
        ops.roi_crop_pool_fwd(feature_maps.contiguous(), rois.contiguous(), b, p, hf, wf, c, ps, ks, pooled, argmax)
        out = pooled if flatten else pooled.view(b * p, ps, ps, c)
        return out.view(b, p, *out.shape[1:]) if keep_batch_dim else out


class FastRCNNDetector:
    def __init__(self, image_shape, num_classes, config, name="fast_rcnn_detector", feature_channels=1024, store=None,
                 device="cuda"):
        """reference fast_rcnn_detector.py:8-41 (same positional arguments)."""
        self.name = name
        self._image_shape = tuple(image_shape)
        self.num_classes = int(num_classes)
        self.c1 = self.num_classes + 1
        assert self.c1 + 4 * self.num_classes <= HEAD_LD
        self.config = config
        self.device = torch.device(device)
        self.ps = int(config["roi_pooling"]["pooled_size"])
        self.ks = int(config["roi_pooling"]["kernel_size"])
        self.cf = feature_channels
        self.flat = self.ps * self.ps * self.cf
        self._roi_pooling = ROIPooling(self.ps, self.ks)
        self.own_store = store is None
        self.store = store if store is not None else ParamStore(self.device)
        self.store.register("fast_rcnn_heads/kernel", (HEAD_LD, 1, 1, self.flat), decay=float(config["weight_decay"]))
        self._deferred = [("fast_rcnn_heads/bias", (HEAD_LD,))]
        if self.own_store:
            self.register_biases()
            self.store.finalize()
            self.init_weights(0)
        self._plans = {}

    def register_biases(self):
        for n, s in self._deferred:
            self.store.register(n, s)

    # ------------------------------------------------------------------ parameters
    def init_weights(self, seed=0):
        """Glorot-uniform kernels (VarianceScaling(1, fan_avg, uniform), fast_rcnn_detector.py:20), zero biases."""
        g = torch.Generator().manual_seed(seed)
        c1, nr = self.c1, 4 * self.num_classes
        w = torch.zeros(HEAD_LD, self.flat)
        w[:c1] = (torch.rand(c1, self.flat, generator=g) * 2 - 1) * math.sqrt(6.0 / (self.flat + c1))
        w[c1:c1 + nr] = (torch.rand(nr, self.flat, generator=g) * 2 - 1) * math.sqrt(6.0 / (self.flat + nr))
        self.store.weight("fast_rcnn_heads/kernel").copy_(w.view(HEAD_LD, 1, 1, self.flat))
        self.store.weight("fast_rcnn_heads/bias").zero_()

    def set_weights(self, w):
        c1, nr = self.c1, 4 * self.num_classes
        k = torch.zeros(HEAD_LD, self.flat)
        k[:c1] = torch.as_tensor(w["fast_rcnn_classification_head/kernel"]).t()
        k[c1:c1 + nr] = torch.as_tensor(w["fast_rcnn_regression_head/kernel"]).t()
        self.store.weight("fast_rcnn_heads/kernel").copy_(k.view(HEAD_LD, 1, 1, self.flat))
        b = torch.zeros(HEAD_LD)
        b[:c1] = torch.as_tensor(w["fast_rcnn_classification_head/bias"])
        b[c1:c1 + nr] = torch.as_tensor(w["fast_rcnn_regression_head/bias"])
        self.store.weight("fast_rcnn_heads/bias").copy_(b)

    def get_weights(self):
        c1, nr = self.c1, 4 * self.num_classes
        k = self.store.weight("fast_rcnn_heads/kernel").view(HEAD_LD, self.flat).cpu()
        b = self.store.weight("fast_rcnn_heads/bias").cpu()
        return {"fast_rcnn_classification_head/kernel": k[:c1].t().contiguous(), "fast_rcnn_classification_head/bias": b[:c1].clone(),
                "fast_rcnn_regression_head/kernel": k[c1:c1 + nr].t().contiguous(), "fast_rcnn_regression_head/bias": b[c1:c1 + nr].clone()}

    # ------------------------------------------------------------------ plans
    def setup(self, batch, num_rois, hf, wf, training, num_samples=0):
        dev = self.device
        self.batch, self.p, self.hf, self.wf = batch, num_rois, hf, wf
        r = batch * num_rois
        self.r = r
        self.pooled = torch.empty(r, self.flat, dtype=BF16, device=dev)
        self.argmax = torch.empty(r, self.flat, dtype=torch.uint8, device=dev)
        self.logits = torch.zeros(r, HEAD_LD, device=dev)
        self.scores = torch.empty(batch, num_rois, self.c1, device=dev)
        self.deltas = torch.empty(batch, num_rois, self.num_classes, 4, device=dev)
        self.regions_abs = torch.empty(batch, num_rois, 4, device=dev)
        # K split of the Dense-head GEMM ([B*P] x 50176 x 64): every split adds one fp32 tile of float atomics (memory side,
        # 1.3 TB/s); measured over 4..98 splits back to back (tools/head_gemm_bench.py): 12-24 are fastest (29-30 us against 36 at 64).
        # IN THE STEP the GEMM shares HBM with the write-back of what earlier kernels left in the caches and with the RPN's side stream
        # (DESIGN 0.1b), and twice the workgroups hold their own better: same-box A/B, 16 -> 32, three alternations: 3.755 -> 3.746,
        # 3.758 -> 3.742, 3.774 -> 3.760 ms (49: 3.759 / 3.780 / 3.755)
        split = max(1, min(int(os.environ.get("FRCNN_HEAD_SPLIT", "32")), self.flat // 64 // 8))
        self.d_fwd = ops.conv_desc(1, 1, r, self.flat, 1, 1, 1, 0, 0, 1, r, HEAD_LD, flags=ops.CONV_SPLITK_ATOMIC, split_k=split)
        self.w_t = torch.zeros(self.flat, 1, 1, HEAD_LD, dtype=BF16, device=dev)
        if training:
            rs = batch * num_samples
            self.rs = rs
            self.dhead_s = torch.empty(rs, HEAD_LD, dtype=BF16, device=dev)
            self.rows = torch.empty(rs, dtype=torch.int32, device=dev)
            self.dpooled_s = torch.empty(rs, self.flat, dtype=BF16, device=dev)
            self.d_wgrad = ops.conv_desc(1, 1, rs, self.flat, 1, 1, 1, 0, 0, 1, rs, HEAD_LD)
            self.d_dgrad = ops.conv_desc(1, 1, rs, HEAD_LD, 1, 1, 1, 0, 0, 1, rs, self.flat)

    def refresh_weights(self, plan):
        plan.add(ops.weights_transpose_flip, self.store.weight("fast_rcnn_heads/kernel"), self.w_t, HEAD_LD, 1, 1, self.flat)

    def flip_entries(self):
        return [(self.store.weight("fast_rcnn_heads/kernel"), self.w_t, HEAD_LD, 1, 1, self.flat)]

    def regions_plan(self, plan, rois):
        """The proposals in absolute image coordinates (reference fast_rcnn_detector.py:67): needed by target assignment and
        box decoding, independent of the head -- the train plan runs it on its target-assignment side stream."""
        plan.add(ops.boxes_scale, rois, self.regions_abs, float(self._image_shape[1]), float(self._image_shape[0]))   # :67
        return self.regions_abs

    def head_grad_rows(self):
        """(dhead_s, ld, rows): destination of the fused loss + head-gradient launch (ops.losses_head_grad)."""
        return self.dhead_s, HEAD_LD, self.rows

    def head_post_plan(self, plan, regions_done, decoded):
        """bias + softmax / split of the head GEMM's logits; decoded [B,P,C,4] (with regions_done): also the decode step of detection
        NMS in the same launch (self.regions_abs must be complete: the proposal NMS writes it, ops.nms_combined_abs)."""
        st = self.store
        if decoded is not None and regions_done:
            plan.add(ops.rcnn_head_post_decode, self.logits, HEAD_LD, st.weight("fast_rcnn_heads/bias"), self.r, self.c1, self.scores, self.deltas,
                     self.regions_abs, decoded, float(self._image_shape[1]), float(self._image_shape[0]))
            return True
        plan.add(ops.rcnn_head_post, self.logits, HEAD_LD, st.weight("fast_rcnn_heads/bias"), self.r, self.c1, self.scores, self.deltas)
        return False

    def forward_plan(self, plan, feature_maps, rois, regions_done=False, decoded=None):
        """Returns the output dict; self.decoded_done tells whether `decoded` was filled (head_post_plan)."""
        st = self.store
        plan.add(ops.roi_crop_pool_fwd, feature_maps, rois, self.batch, self.p, self.hf, self.wf, self.cf, self.ps, self.ks, self.pooled,
                 self.argmax)
        plan.zero(self.logits)                      # (split-K float atomics)
        plan.add(ops.conv2d_fprop, self.d_fwd, self.pooled, st.weight_bf16("fast_rcnn_heads/kernel"), self.logits)
        self.decoded_done = self.head_post_plan(plan, regions_done, decoded)
        if not regions_done:
            self.regions_plan(plan, rois)
        return {"regions": self.regions_abs, "pred_scores": self.scores, "pred_boxes": self.deltas}

    def backward_plan(self, plan, dlogits_s, ddeltas_s, indices, num_samples, rois, g_feat_bf16, head_grad_done=False, bias_grad_done=False,
                      add_to_g_feat_after=None, consumer=None):
        """Per-sample loss gradients -> head parameter gradients and the RoI-branch feature-map
        gradient, written (bf16) to g_feat_bf16 [B*hf*wf, C].  head_grad_done: self.dhead_s / self.rows were already written
        by the loss launch (ops.losses_head_grad)."""
        st = self.store
        if not head_grad_done:
            plan.add(ops.rcnn_head_grad, dlogits_s, ddeltas_s, indices, self.batch, self.p, self.c1, num_samples, self.dhead_s,
                     HEAD_LD, self.rows)
        # data gradient first (the chain to the RoI backward pass and the backbone); the head's parameter gradients, which nobody
        # needs before the update, after it.  (As a side branch they measured 0.015 ms SLOWER than in line: see faster_rcnn.py.)
        plan.add(ops.conv2d_fprop, self.d_dgrad, self.dhead_s, self.w_t, self.dpooled_s)
        # gather form: every element of g_feat is written once, in bf16, without global atomics (no memset / cast passes)
        if add_to_g_feat_after is not None:
            # g_feat already receives another branch's gradient (the RPN's, on the side stream named here): wait for it, then add
            # (consumer: the backbone's last conv unit -- g_feat is complete after this kernel, so it also runs that unit's
            # BatchNorm-backward reduce)
            plan.join(add_to_g_feat_after)
            red = None
            if consumer is not None:
                red = consumer.reduce_args(relu=True)
                plan.hold(red)
            plan.add(ops.roi_crop_pool_bwd_bf16_add, self.dpooled_s, self.argmax, rois, self.rows, self.rs, self.batch, self.p, self.hf, self.wf,
                     self.cf, self.ps, self.ks, g_feat_bf16, red=red)
        else:
            plan.add(ops.roi_crop_pool_bwd_bf16, self.dpooled_s, self.argmax, rois, self.rows, self.rs, self.batch, self.p, self.hf, self.wf,
                     self.cf, self.ps, self.ks, g_feat_bf16)
        if not bias_grad_done:                      # (the fused loss launch adds the bias gradient itself: ops.losses_head_grad(bias_grad=...))
            plan.add(ops.colsum_bf16, self.dhead_s, self.rs, HEAD_LD, HEAD_LD, st.grad("fast_rcnn_heads/bias"))
        plan.add(ops.conv2d_wgrad, self.d_wgrad, self.pooled, self.dhead_s, st.grad("fast_rcnn_heads/kernel"), HEAD_LD, self.rows)

    # ------------------------------------------------------------------ reference call surface
    def __call__(self, feature_maps, rois):
        """reference fast_rcnn_detector.py:43-69.  feature_maps bf16 [B,hf,wf,C]; rois fp32 [B,P,4]
        relative.  Returns dict regions [B,P,4] abs, pred_scores [B,P,C+1], pred_boxes [B,P,C,4]."""
        b, hf, wf, _ = feature_maps.shape
        key = (int(b), int(rois.shape[1]), int(hf), int(wf))
        if key not in self._plans:
            self.setup(key[0], key[1], hf, wf, False)
            self._x = torch.empty(feature_maps.shape, dtype=BF16, device=self.device)
            self._rois = torch.empty(rois.shape, device=self.device)
            plan = Plan("fast_rcnn")
            plan.add(self.store.refresh_bf16)
            self._out = self.forward_plan(plan, self._x, self._rois)
            self._plans = {key: plan}
        self._x.copy_(feature_maps)
        self._rois.copy_(rois)
        self._plans[key].run()
        return self._out

    def get_training_samples(self, gt_labels, gt_boxes, regions, pred_scores, pred_boxes, foreground_iou_interval,
                             background_iou_interval, num_samples, foreground_proportion, seed=0, step=None):
        """reference fast_rcnn_detector.py:71-130 (forward values)."""
        from ...utils.training import generate_targets, get_sample_indices
        tl, tb = generate_targets(gt_labels, gt_boxes, regions, self._image_shape, foreground_iou_interval, background_iou_interval)
        idx = get_sample_indices(tl, num_samples, foreground_proportion, seed=seed, step=step, stream_base=2).long()
        ar = torch.arange(tl.shape[0], device=tl.device)[:, None]
        return {"target_labels": tl[ar, idx], "pred_scores": pred_scores[ar, idx], "target_boxes": tb[ar, idx],
                "pred_boxes": pred_boxes[ar, idx], "sample_indices": idx}
