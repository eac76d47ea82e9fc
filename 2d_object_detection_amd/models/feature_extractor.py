"""Backbone: Keras-applications ResNet50 (v1) truncated at conv4_block6_out, on hand-written
gfx950 kernels.  Mirrors reference models/feature_extractor.py:4-11:

    m = get_feature_extractor_model(image_shape)
    feature_maps = m(images_uint8[B,H,W,3], training=bool)     # bf16 [B, gh, gw, 1024]
    m.output_shape == (None, gh, gw, 1024)

Layout: NHWC bf16 activations.  conv -> (bias) -> BN statistics are fused into the implicit-GEMM
epilogue; BN-apply(+residual)(+ReLU) is one elementwise pass.  Backward is an explicit reverse
plan (bn_bwd reduce/apply, data gradient as an implicit GEMM with transposed tap-flipped weights,
weight gradient with transposing LDS reads).
"""
import math

import os

import torch

from .. import ops
from ..runtime import ParamStore, Plan

BF16 = torch.bfloat16
BN_EPS = 1.001e-5          # Keras ResNet50 BatchNormalization epsilon [TF-ext]
BN_MOMENTUM = 0.99
GROUPED_WGRAD = os.environ.get("FRCNN_GROUPED_WGRAD", "1") != "0"      # (0: one weight-gradient launch per layer)
# measuring aid: the grouped weight-gradient launches of conv4 / conv3 on a side stream under the next stage's backward chain (single GPU: no bucket cuts)
WGRAD_TRAIL = os.environ.get("FRCNN_WGRAD_TRAIL", "0") != "0"
# the backward reduce of a first block's shortcut BatchNorm inside the block-final BatchNorm's backward-apply launch (0: its own launch)
BN_RED2 = os.environ.get("FRCNN_BN_RED2", "1") != "0"
STACKS = {50: ((64, 3, 1), (128, 4, 2), (256, 6, 2)), 101: ((64, 3, 1), (128, 4, 2), (256, 23, 2))}


def _out_hw(h, w):
    f1 = lambda n: (n + 6 - 7) // 2 + 1
    f2 = lambda n: (n + 2 - 3) // 2 + 1
    return f1(h), f1(w), f2(f1(h)), f2(f1(w))


FP8_MIN_STAGE = int(os.environ.get("FRCNN_FP8_MIN_STAGE", "2"))      # first ResNet stage whose tensors get fp8 twins (measuring aid: 3 = conv3 on)
# The 3x3 convolution of a bottleneck block applies the block's first BatchNorm + ReLU itself where the C ABI offers it
# (frcnn_conv2d_fprop_bnin: conv2's layers at the benchmark's sizes); 0: the separate bn_train_apply launch (measuring aid)
# Which convolutions apply the BatchNorm + ReLU of their INPUT layer themselves (frcnn_conv2d_fprop_bnin): "0" none, "wres" round 4's 64-channel
# 3x3 layers (weights-resident kernel), "3x3" (default) those + the 3x3 layers on the patch-resident kernel (conv3 / conv4), "1" also the 1x1
# layer behind a block's second BatchNorm (tile kernel).  Measured on MI355X (round 5; tools/bnin_bench.py, isolated graph replays, us,
# bn_train_apply + convolution back to back -> fused): 3x3 conv2 31.5 -> 25.5, conv3 24.8 -> 21.5, conv4 20.4 -> 20.6; 1x1 conv2 33.7 -> 32.2,
# conv3 25.7 -> 27.6, conv4 20.0 -> 20.3 (byte mask stores; 36.0 / 28.9 / 21.5 with the mask bytes pooled by cross-lane shuffles).  In the step
# (same-box A/B, ms): "wres" 3.885 -> "3x3" 3.858; "3x3" 3.903 / 3.909 -> "1" 3.932 / 3.903.  The in-place transform of a landed tile is VALU
# work that EVERY channel-part workgroup of a pixel tile repeats (4 - 8 x for the 1x1 layers: 2.5 - 5 us per launch) and the ReLU bit mask
# costs as much again (1.3 - 4 us): on the 1x1 layers that is what the 5 - 11 us BatchNorm launch cost.  DESIGN.md section 0.2.
BN_IN_FUSED = os.environ.get("FRCNN_BN_IN", "3x3")
FP8_BWD = os.environ.get("FRCNN_FP8_BWD", "1") != "0"                # measuring aid: 0 keeps the data gradients in bf16 (fp8 forward only)
FP8_WGRAD = os.environ.get("FRCNN_FP8_WGRAD", "1") != "0"            # measuring aid: 0 keeps the weight gradients in bf16
FP8_DZ_TWIN_ONLY = os.environ.get("FRCNN_FP8_DZ_TWIN_ONLY", "1") != "0"   # measuring aid: 0 always stores the bf16 dz beside its twin
FP8_FWD_MIN_STAGE = int(os.environ.get("FRCNN_FP8_FWD_MIN_STAGE", "3"))  # first stage whose FORWARD convolutions run in fp8 (measuring aid)
FP8_OUT8_ALWAYS = os.environ.get("FRCNN_FP8_OUT8_ALWAYS", "1") != "0"  # measuring aid: 0 = a block output gets a twin only if its OWN stage's layers read one
FP8_MIN_K = int(os.environ.get("FRCNN_FP8_MIN_K", "256"))            # measuring aid: shortest contraction (k * k * channels) that runs in fp8
FP8_MARGIN = 2.0           # delayed scaling: next step's scale = margin * this step's amax / 448 (e4m3 is a float format: head-room costs no precision)


class Fp8Scales:
    """Per-tensor scales of the fp8 (e4m3) activation twins, device resident so that captured graphs follow them: rows
    [dequantisation scale | 1 / scale] x one column per tensor, plus the amax slots of this step [tensor][8192].  The amax slots are an
    accumulation target of the step (atomic max, zeroed by the plan's fill); `update` -- once per step, after the last producer -- turns it into the NEXT
    step's scales (delayed scaling with one step of history; the first step runs on scale 1)."""

    def __init__(self, device, capacity=384):      # (ResNet-101 under the pyramid takes ~250 columns)
        self.buf = torch.zeros(2, capacity, dtype=torch.float32, device=device)             # [scale | 1 / scale]
        self.buf.fill_(1.0)
        self.limit = torch.full((capacity,), 448.0, dtype=torch.float32, device=device)     # clamp of each twin's format (e4m3 448, e5m2 57344)
        # [0]: (tensor, step) pairs whose values were clamped (amax beyond limit x the scale they were quantised with: the delayed rule
        # has one step of history), [1]: non-finite amax (scale kept).  Counters, never reset by a step: FasterRCNN.fp8_status() reads them
        self.status = torch.zeros(2, dtype=torch.int32, device=device)
        self.amax_buf = torch.zeros(capacity, ops.FP8_AMAX_SLOTS, dtype=torch.float32, device=device)   # slots per tensor (one word would serialise the atomics)
        self.n = 0

    def new(self, e5m2=False):
        assert self.n < self.buf.shape[1], "Fp8Scales: more than %d fp8 tensors" % self.buf.shape[1]
        self.n += 1
        if e5m2:
            self.limit[self.n - 1] = 57344.0
        return self.n - 1

    def state(self):
        """the delayed-scaling state (for snapshots / checkpoints: a restored model otherwise runs its first step on scale 1)"""
        return self.buf.clone()

    def load_state(self, buf):
        self.buf.copy_(buf.to(self.buf.device))

    def amax(self, i):
        return self.amax_buf[i]

    def scale(self, i):
        return self.buf[0, i:i + 1]

    def qscale(self, i):
        return self.buf[1, i:i + 1]

    def plan_zero(self, plan):
        plan.zero(self.amax_buf[:max(self.n, 1)])          # (called after every twin has taken its column)

    def plan_update(self, plan):
        if self.n:
            plan.add(ops.fp8_update_scales, self.amax_buf, self.buf[0], self.buf[1], self.n, FP8_MARGIN, self.limit, self.status)


class Fp8Twin:
    """fp8 copy of an activation tensor: bytes, its column in the scale table, and the frcnn_fp8_out the producing kernel takes."""

    def __init__(self, scales, shape, device, e5m2=False):
        self.scales, self.idx = scales, scales.new(e5m2)
        self.data = torch.zeros(shape, dtype=ops.FP8, device=device)
        self.out = ops.fp8_out(self.data, scales.qscale(self.idx), scales.amax(self.idx))

    @property
    def scale(self):
        return self.scales.scale(self.idx)


class _ConvBN:
    """One conv + BatchNorm unit (Keras names <name>_conv / <name>_bn)."""

    def __init__(self, store, name, cin, cout, k, stride, pad, sync_world=1, fp8=False, fp8_bwd=None, fp8_wgrad=None):
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        # fp8 forward convolution (e4m3 operands, 128-deep MFMA steps): a layer whose cin is a multiple of 128
        # ... and whose contraction is at least two 128-deep steps long (a single step has no K loop to shorten: measured at
        # 375x1242, batch 4, the 1x1 128 -> 512 layers of conv3 run 16.4 -> 15.3 us forward and 30.9 -> 35.0 us backward in fp8)
        self.fp8 = bool(fp8) and k != 7 and cin % 128 == 0 and k * k * cin >= FP8_MIN_K
        # fp8 data gradient (e5m2 gradient x e4m3 transposed weights): its contraction runs over this layer's OUTPUT channels
        self.fp8_bwd = FP8_BWD and bool(fp8 if fp8_bwd is None else fp8_bwd) and k != 7 and cout % 128 == 0 and k * k * cout >= FP8_MIN_K
        # fp8 weight gradient (e4m3 input twin x e5m2 gradient twin, contraction over the pixels): bound by the bytes its workgroups
        # stream, so it pays whatever the channel counts (tools/wgrad_sweep.py fp8: 1.3-1.8x) -- wherever the input already has a twin
        self.fp8_wgrad = FP8_WGRAD and bool(fp8 if fp8_wgrad is None else fp8_wgrad) and k != 7 and cin % 64 == 0 and cout % 64 == 0
        self.store = store
        self.sync_world = int(sync_world)            # > 1: BatchNorm statistics are summed over this many data-parallel ranks
        store.register(name + "_conv/kernel", (cout, k, k, cin))          # OHWI (Keras: HWIO)
        store.register(name + "_conv/bias", (cout,))
        store.register(name + "_bn/gamma", (cout,))
        store.register(name + "_bn/beta", (cout,))
        self.mm = store.register_stat(name + "_bn/moving_mean", (cout,), 0.0)
        self.mv = store.register_stat(name + "_bn/moving_variance", (cout,), 1.0)
        self.is_stem = (k == 7)

    # -- buffers for a fixed input geometry
    def setup(self, n, hi, wi, device, training, acc=None):
        s, p, k = self.stride, self.pad, self.k
        self.n, self.hi, self.wi = n, hi, wi
        if self.is_stem:
            self.ho, self.wo = (hi + 6 - 7) // 2 + 1, (wi + 6 - 7) // 2 + 1
            self.hp, self.wp = hi + 6, max(wi + 6, 2 * (self.wo - 1) + 8)
            self.desc = ops.conv_desc(n, self.hp, self.wp, 32, 7, 1, 2, 0, 0, self.ho, self.wo, self.cout, in_pix_stride=4,
                                      flags=ops.CONV_BIAS | (ops.CONV_STATS if training else 0))
            self.desc_wgrad = ops.conv_desc(n, self.hp, self.wp, 32, 7, 1, 2, 0, 0, self.ho, self.wo, self.cout, in_pix_stride=4,
                                            flags=ops.CONV_WGRAD_STEM_UNPACK)
            self.w_packed = torch.zeros(self.cout, 7, 8, 4, dtype=BF16, device=device)
        else:
            self.ho, self.wo = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
            self.desc = ops.conv_desc(n, hi, wi, self.cin, k, k, s, p, p, self.ho, self.wo, self.cout,
                                      flags=ops.CONV_BIAS | (ops.CONV_STATS if training else 0))
            self.w_t = torch.zeros(self.cin, k, k, self.cout, dtype=BF16, device=device)   # data-gradient weights
            if self.fp8 and training:
                self.w8 = torch.zeros(self.cout, k, k, self.cin, dtype=ops.FP8, device=device)
                self.w8_scale = torch.ones(self.cout, dtype=torch.float32, device=device)
            if self.fp8_bwd and training:
                self.w_t8 = torch.zeros(self.cin, k, k, self.cout, dtype=ops.FP8, device=device)
                self.w_t8_scale = torch.ones(self.cin, dtype=torch.float32, device=device)
            # fewer output tiles than CUs and a long K (conv4 at 375x1242): scratch for the split-K fix-up form of the conv kernel
            self.conv_ws = ops.conv_attach_workspace(self.desc, device)
        self._device = device
        self.m = n * self.ho * self.wo
        c = self.cout
        f32 = dict(dtype=torch.float32, device=device)
        self.z = torch.empty(self.m, c, dtype=BF16, device=device)
        self.scale, self.shift = torch.empty(c, **f32), torch.empty(c, **f32)
        if training:
            self.tiles = ops.conv_stat_tiles(self.desc)
            # atomically accumulated buffers: carved from ONE flat tensor that is zeroed by a single fill per step
            # f64 partial sums: the arrival order of the conv kernels' atomics does not show in the statistics
            self.stats = (acc(self.tiles * 2 * c * 2).view(torch.float64).view(self.tiles, 2, c) if acc
                          else torch.zeros(self.tiles, 2, c, dtype=torch.float64, device=device))
            self.mean, self.invstd = torch.empty(c, **f32), torch.empty(c, **f32)
            self.bwd_blocks = ops.bn_bwd_blocks(self.m)
            self.bwd_partial = (acc(self.bwd_blocks * 2 * c).view(self.bwd_blocks, 2, c) if acc
                                else torch.zeros(self.bwd_blocks, 2, c, **f32))
            self.c1, self.c2 = torch.empty(c, **f32), torch.empty(c, **f32)
            self.dz = torch.empty(self.m, c, dtype=BF16, device=device)
            self.dz8 = None                          # Fp8Twin (e5m2) of dz, attached by FeatureExtractor.setup in fp8 mode
            # ReLU bit mask written by the forward BN kernel: the backward kernels read 1 bit instead of 16 per element
            self.relu_mask = torch.empty(self.m, c // 8, dtype=torch.uint8, device=device)

    def refresh_weights(self, plan):
        """(re)build the derived bf16 weight forms from the fp32 masters."""
        st = self.store
        if self.is_stem:
            plan.add(ops.stem_pack_weights, st.weight(self.name + "_conv/kernel"), self.w_packed, self.cout)
        else:
            plan.add(ops.weights_transpose_flip, st.weight(self.name + "_conv/kernel"), self.w_t, self.cout, self.k, self.k, self.cin)

    def w_fwd(self):
        return self.w_packed if self.is_stem else self.store.weight_bf16(self.name + "_conv/kernel")

    # -- forward: x -> z (raw conv output) -> scale/shift
    def quant_entry(self):
        """(fp32 master rows, fp8 destination, per-row scale) of this layer's forward weights, or None (bf16 layer)."""
        if not (self.fp8 and hasattr(self, "w8")):
            return None
        return (self.store.weight(self.name + "_conv/kernel").view(self.cout, -1), self.w8, self.w8_scale)

    def quant_entry_bwd(self):
        """(bf16 tap-flipped transposed weights as rows per input channel, e4m3 destination, per-row scale) or None."""
        if not (self.fp8_bwd and hasattr(self, "w_t8")):
            return None
        return (self.w_t.view(self.cin, -1), self.w_t8, self.w_t8_scale)

    def forward(self, plan, x, training, x8=None):
        """x8: Fp8Twin of x (training, fp8 layers): the convolution reads the e4m3 bytes instead of the bf16 tensor."""
        st = self.store
        ops.conv_zero_counters(plan, self.desc)
        if x8 is not None and self.fp8 and training:
            plan.add(ops.conv2d_fprop_fp8, self.desc, x8.data, self.w8, x8.scale, self.w8_scale, self.z, bias=st.weight(self.name + "_conv/bias"),
                     stats=self.stats)
        else:
            plan.add(ops.conv2d_fprop, self.desc, x, self.w_fwd(), self.z, bias=st.weight(self.name + "_conv/bias"),
                     stats=self.stats if training else None)
        self._training = training
        if training and self.sync_world > 1:
            plan.sync_point(self.name + "_bn_stats", [self.stats])       # f64 slot sums of every rank -> sums of the global batch
        if not training:
            g, b = st.weight(self.name + "_bn/gamma"), st.weight(self.name + "_bn/beta")
            plan.add(ops.bn_finalize_eval, self.cout, g, b, self.mm, self.mv, BN_EPS, self.scale, self.shift)

    def forward_bnin(self, plan, prev, act_out):
        """This layer's forward convolution on the RAW output of `prev`, applying prev's training-mode BatchNorm + ReLU itself
        (ops.conv2d_fprop_bnin == prev.apply(act_out) followed by self.forward(act_out), same bits): prev.apply is not launched; the
        activation act_out, prev's ReLU mask, mean / invstd and moving statistics are written by this launch."""
        st = self.store
        assert prev.tiles == ops.conv_stat_tiles(self.desc) and prev.cout == self.cin
        self._bnin = ops.bn_in_args(prev.stats, st.weight(prev.name + "_bn/gamma"), st.weight(prev.name + "_bn/beta"), prev.mm, prev.mv, BN_MOMENTUM,
                                    BN_EPS, prev.m * prev.sync_world, act_out, prev.relu_mask, prev.mean, prev.invstd)
        plan.add(ops.conv2d_fprop_bnin, self.desc, prev.z, self.w_fwd(), self.z, self._bnin, bias=st.weight(self.name + "_conv/bias"), stats=self.stats)
        self._training = True
        if self.sync_world > 1:
            plan.sync_point(self.name + "_bn_stats", [self.stats])

    def apply(self, plan, out, res=None, relu=True, dual=None, out8=None, twin_only=False):
        """dual: a second conv unit of the same output shape whose BatchNorm (no ReLU) is added before the ReLU -- the shortcut
        branch of a stage's first block: out = ReLU(BN(z) + BN_dual(z_dual)) in one kernel, the shortcut's output never stored."""
        if self._training and dual is not None:
            st = self.store
            assert res is None and dual.m == self.m and dual.cout == self.cout and dual.tiles == self.tiles
            plan.add(ops.bn_train_apply_dual, self.z, self.stats, st.weight(self.name + "_bn/gamma"), st.weight(self.name + "_bn/beta"),
                     self.mm, self.mv, self.mean, self.invstd, dual.z, dual.stats, st.weight(dual.name + "_bn/gamma"),
                     st.weight(dual.name + "_bn/beta"), dual.mm, dual.mv, dual.mean, dual.invstd, self.tiles, self.m * self.sync_world,
                     BN_MOMENTUM, BN_EPS, out, self.m, self.cout, relu=relu, relu_mask=self.relu_mask if relu else None,
                     f8=out8.out if out8 is not None else None)
            return
        if self._training:
            # batch statistics -> scale/shift inside the apply kernel (every workgroup reduces its own 64 channels)
            st = self.store
            if twin_only:
                assert out8 is not None and res is None
                out = None                               # every consumer of this activation reads its e4m3 twin: no bf16 store
            plan.add(ops.bn_train_apply, self.z, self.stats, self.tiles, self.m * self.sync_world, st.weight(self.name + "_bn/gamma"),
                     st.weight(self.name + "_bn/beta"), self.mm, self.mv, BN_MOMENTUM, BN_EPS, out, self.mean, self.invstd, self.m,
                     self.cout, res=res, relu=relu, relu_mask=self.relu_mask if relu else None, f8=out8.out if out8 is not None else None)
        else:
            plan.add(ops.bn_apply, self.z, self.scale, self.shift, out, self.m, self.cout, res=res, relu=relu)

    # -- backward: gout (grad of the BN[+res][+relu] output), act = that output (None when no ReLU)
    def reduce_args(self, relu=True):
        """frcnn_bn_reduce of this layer's BatchNorm for the data-gradient kernel that produces its incoming gradient."""
        if getattr(self, "_red", None) is None or self._red[0] != relu:
            self._red = (relu, ops.bn_reduce_args(self.z, self.relu_mask if relu else None, self.mean, self.invstd, self.bwd_partial))
        return self._red[1]

    def backward_bn(self, plan, gout, act, gpre=None, reduced=False, mask=None, also_reduce=None):
        """reduced: the kernel that produced gout already accumulated this layer's backward sums (conv2d_dgrad_bnreduce).
        mask: ReLU bit mask to apply to gout instead of this layer's own (the shortcut BN of a block receives the block-output
        gradient masked by the block's final ReLU).
        also_reduce: a second conv unit whose BatchNorm receives the SAME masked gradient (the shortcut unit of a stage's first block):
        this launch accumulates its backward sums too (ops.bn_bwd_apply_fused_red2); its own backward_bn then runs with reduced=True."""
        st = self.store
        if mask is None:
            mask = self.relu_mask if act is not None else None   # (act only says whether the layer ends in a ReLU)
        if not reduced:
            plan.add(ops.bn_bwd_reduce, gout, None, self.z, self.mean, self.invstd, self.bwd_partial, self.m, self.cout, relu_mask=mask)
        if self.sync_world > 1:
            plan.sync_point(self.name + "_bn_bwd", [self.bwd_partial])
        # fp8 mode: where both consumers of dz (data gradient, weight gradient) read its e5m2 twin, the bf16 tensor is not stored
        dz = None if (getattr(self, "dz_twin_only", False) and self.dz8 is not None) else self.dz
        if also_reduce is not None:
            assert mask is not None and gpre is None and self.cout % 64 == 0 and also_reduce.cout == self.cout and also_reduce.m == self.m
            red2 = also_reduce.reduce_args(relu=False)
            plan.hold(red2)
            plan.add(ops.bn_bwd_apply_fused_red2, gout, self.z, self.mean, self.invstd, st.weight(self.name + "_bn/gamma"), self.bwd_partial,
                     self.bwd_blocks, st.grad(self.name + "_bn/gamma"), st.grad(self.name + "_bn/beta"), dz, self.m, self.cout, mask, red2,
                     count=self.m * self.sync_world, param_grad_scale=1.0 / self.sync_world, f8=self.dz8.out if self.dz8 is not None else None)
            return
        plan.add(ops.bn_bwd_apply_fused, gout, None, self.z, self.mean, self.invstd, st.weight(self.name + "_bn/gamma"),
                 self.bwd_partial, self.bwd_blocks, st.grad(self.name + "_bn/gamma"), st.grad(self.name + "_bn/beta"), dz, gpre,
                 self.m, self.cout, relu_mask=mask, count=self.m * self.sync_world, param_grad_scale=1.0 / self.sync_world,
                 f8=self.dz8.out if self.dz8 is not None else None)
        # the conv bias feeds a training-mode BN: its gradient is identically zero (flat grad buffer is pre-zeroed)

    def backward_weights(self, plan, x, defer=None, x8=None):
        """defer: list collecting (desc, x, dz, dw) of layers whose weight gradients are launched together at the end of their
        stage (ops.WgradGroup) instead of one launch each.  x8: Fp8Twin of x -- with the e5m2 twin of dz the fp8 form."""
        st = self.store
        f8 = self.fp8_wgrad and x8 is not None and self.dz8 is not None
        assert f8 or not getattr(self, "dz_twin_only", False), self.name + ": the bf16 dz is not stored, but its weight gradient would read it"
        if defer is not None and not self.is_stem and self.cin % 64 == 0 and self.cout % 64 == 0:
            if f8:
                defer.append((self.desc, x8.data, self.dz8.data, st.grad(self.name + "_conv/kernel"), x8.scale, self.dz8.scale))
            else:
                defer.append((self.desc, x, self.dz, st.grad(self.name + "_conv/kernel")))
            return
        if f8:
            plan.add(ops.conv2d_wgrad_fp8, self.desc, x8.data, self.dz8.data, x8.scale, self.dz8.scale, st.grad(self.name + "_conv/kernel"))
            return
        if self.is_stem:
            # the 7 x 3 real values of every packed tap row go straight into the (zeroed) Keras-layout gradient
            plan.add(ops.conv2d_wgrad, self.desc_wgrad, x, self.dz, st.grad(self.name + "_conv/kernel"))
        else:
            plan.add(ops.conv2d_wgrad, self.desc, x, self.dz, st.grad(self.name + "_conv/kernel"))

    def backward_data(self, plan, gx, res=None, consumer=None, res_mask=None):
        """gx[n,hi,wi,cin] = conv_transpose(dz) (+ res); stride-2 1x1 scatters into a pre-zeroed gx.  consumer: the conv unit
        whose BatchNorm(+ReLU) output gradient gx is -- its backward reduce is fused into this kernel."""
        k, s = self.k, self.stride
        if s == 1:
            d = ops.conv_desc(self.n, self.ho, self.wo, self.cout, k, k, 1, self.pad, self.pad, self.hi, self.wi, self.cin,
                              flags=ops.CONV_ADD_RES if res is not None else 0)
        else:
            assert k == 1
            d = ops.conv_desc(self.n, self.ho, self.wo, self.cout, 1, 1, 1, 0, 0, self.ho, self.wo, self.cin, out_h=self.hi,
                              out_w=self.wi, out_scatter=s, flags=ops.CONV_ADD_RES if res is not None else 0)
        plan.hold(d)
        plan.hold(ops.conv_attach_workspace(d, self._device))
        ops.conv_zero_counters(plan, d)
        red = None
        if consumer is not None:
            red = consumer.reduce_args(relu=True)
            plan.hold(red)
        else:
            assert res_mask is None
        assert (self.fp8_bwd and self.dz8 is not None) or not getattr(self, "dz_twin_only", False)
        if self.fp8_bwd and self.dz8 is not None:
            plan.add(ops.conv2d_dgrad_fp8, d, self.dz8.data, self.w_t8, self.dz8.scale, self.w_t8_scale, gx, red=red, res=res, res_mask=res_mask)
        elif consumer is not None:
            plan.add(ops.conv2d_dgrad_bnreduce, d, self.dz, self.w_t, gx, red, res=res, res_mask=res_mask)
        else:
            plan.add(ops.conv2d_fprop, d, self.dz, self.w_t, gx, res=res)


class FeatureExtractor:
    """ResNet-50/101 C4 backbone (callable like the Keras model returned by the reference)."""

    def __init__(self, image_shape, depth=50, store=None, device="cuda", sync_bn_world=1, precision="bf16"):
        """sync_bn_world > 1: synchronised BatchNorm over that many data-parallel ranks -- the plan gets a sync point (an
        all-reduce of the layer's partial sums, runtime.Plan.sync_point) between every statistics-producing kernel and the
        kernel that consumes them, so that N ranks x b images reproduce the reference's single device with N*b images
        (models/faster_rcnn.py:50).  One small collective per BatchNorm layer and direction: the statistics of layer k+1
        depend on layer k's normalised output, so they cannot be batched across layers."""
        self.image_shape = tuple(image_shape)
        self.sync_bn_world = int(sync_bn_world)
        assert precision in ("bf16", "fp8")
        # "fp8": training-mode forward convolutions with cin % 128 == 0 run on e4m3 operands (BASELINE.json configs[4]'s precision;
        # weights quantised per output channel from the fp32 masters, activations per tensor by the BatchNorm kernel that writes
        # them); backward pass, statistics and every stored tensor stay as in "bf16"
        self.precision = precision
        self.depth = depth
        self.device = torch.device(device)
        self.own_store = store is None
        self.store = store if store is not None else ParamStore(self.device)
        h, w = image_shape[0], image_shape[1]
        h1, w1, h2, w2 = _out_hw(h, w)
        # registration in reverse execution order (gradient-bucket order): conv4 .. conv2, stem
        self.blocks = []            # (name, [conv0?], conv1, conv2, conv3, stride)
        cin = 64
        specs = []
        for si, (f, nb, s1) in enumerate(STACKS[depth]):
            for b in range(1, nb + 1):
                specs.append(("conv%d_block%d" % (si + 2, b), cin, f, s1 if b == 1 else 1, b == 1))
                cin = 4 * f
        units = {}
        fp8 = precision == "fp8"
        stage_of = lambda n: int(n[4])
        last_stage = None
        for (n, ci, f, s, first) in reversed(specs):
            if last_stage is not None and stage_of(n) != last_stage:
                self.store.end_bucket("conv%d" % last_stage)
            last_stage = stage_of(n)
            u = {}
            # fp8 from conv2 on (FP8_MIN_STAGE).  At conv2's resolution (94 x 311) the forward convolutions are bound by HBM and
            # their epilogues, not by operand fill, and in the first half of round 3 a twin cost its BatchNorm kernel more than it
            # saved the one convolution reading it -- conv2 stayed bf16.  With the weight gradients in fp8 as well (a twin has two or
            # three readers) and the bf16 tensors that lose their last reader no longer stored (dz_twin_only), conv2's twins pay:
            # same-box A/B of stage 3 -> 2: batch 8 6.41 -> 6.35 ms, pyramid 8.73 -> 8.67, batch 4 3.94 -> 3.92.
            stage = stage_of(n)
            f_in = fp8 and (stage - (1 if first else 0)) >= FP8_MIN_STAGE          # units reading the block input
            f_blk = fp8 and stage >= FP8_MIN_STAGE                                 # units reading this block's own activations
            # FORWARD convolutions in fp8 only from conv3 on (FP8_FWD_MIN_STAGE): quantisation noise injected in conv2 is carried --
            # and, in a train-mode random-init network, amplified -- through every later layer (backbone deviation test: feature maps
            # 10.5 % -> 14.5 % relative L2, gradient cosines 0.84 / 0.80 -> 0.79 / 0.70, for 1 % of step time); the backward-only uses
            # of conv2's twins (data and weight gradients) do not touch the forward pass
            ff_in = f_in and (stage - (1 if first else 0)) >= FP8_FWD_MIN_STAGE
            ff_blk = f_blk and stage >= FP8_FWD_MIN_STAGE
            if first:
                u[0] = _ConvBN(self.store, n + "_0", ci, 4 * f, 1, s, 0, self.sync_bn_world, ff_in, f_blk, f_in)
            u[1] = _ConvBN(self.store, n + "_1", ci, f, 1, s, 0, self.sync_bn_world, ff_in, f_blk, f_in)
            u[2] = _ConvBN(self.store, n + "_2", f, f, 3, 1, 1, self.sync_bn_world, ff_blk, f_blk, f_blk)
            u[3] = _ConvBN(self.store, n + "_3", f, 4 * f, 1, 1, 0, self.sync_bn_world, ff_blk, f_blk, f_blk)
            units[n] = u
        self.stem = _ConvBN(self.store, "conv1", 3, 64, 7, 2, 3, self.sync_bn_world)
        self.store.end_bucket("conv2+stem")
        self.specs = specs
        self.units = units
        self.out_channels = cin
        gh, gw = h2, w2
        for (_, _, _, s, _) in specs:
            if s == 2:
                gh, gw = (gh - 1) // 2 + 1, (gw - 1) // 2 + 1
        self.output_shape = (None, gh, gw, cin)
        self._plans = {}
        if self.own_store:
            self.store.finalize()
            self.init_weights(0)

    # ------------------------------------------------------------------ parameters
    def conv_units(self):
        yield self.stem
        for (n, _, _, _, _) in self.specs:
            for k in sorted(self.units[n]):
                yield self.units[n][k]

    def init_weights(self, seed=0):
        """He-normal conv kernels, zero bias, gamma 1 / beta 0 (no ImageNet download possible here)."""
        g = torch.Generator().manual_seed(seed)
        st = self.store
        for u in self.conv_units():
            fan_in = u.k * u.k * u.cin
            w = torch.randn(u.cout, u.k, u.k, u.cin, generator=g) * math.sqrt(2.0 / fan_in)
            st.weight(u.name + "_conv/kernel").copy_(w)
            st.weight(u.name + "_conv/bias").zero_()
            st.weight(u.name + "_bn/gamma").fill_(1.0)
            st.weight(u.name + "_bn/beta").zero_()
            u.mm.zero_()
            u.mv.fill_(1.0)

    def set_weights(self, weights):
        """weights: dict Keras-name -> array in Keras layout (conv kernel HWIO)."""
        st = self.store
        for u in self.conv_units():
            st.weight(u.name + "_conv/kernel").copy_(torch.as_tensor(weights[u.name + "_conv/kernel"]).permute(3, 0, 1, 2))
            for s in ("_conv/bias", "_bn/gamma", "_bn/beta"):
                st.weight(u.name + s).copy_(torch.as_tensor(weights[u.name + s]))
            u.mm.copy_(torch.as_tensor(weights[u.name + "_bn/moving_mean"]))
            u.mv.copy_(torch.as_tensor(weights[u.name + "_bn/moving_variance"]))

    def get_weights(self):
        st, out = self.store, {}
        for u in self.conv_units():
            out[u.name + "_conv/kernel"] = st.weight(u.name + "_conv/kernel").permute(1, 2, 3, 0).contiguous().cpu()
            for s in ("_conv/bias", "_bn/gamma", "_bn/beta"):
                out[u.name + s] = st.weight(u.name + s).clone().cpu()
            out[u.name + "_bn/moving_mean"] = u.mm.clone().cpu()
            out[u.name + "_bn/moving_variance"] = u.mv.clone().cpu()
        return out

    def refresh_weights(self, plan):
        for u in self.conv_units():
            u.refresh_weights(plan)
        self.quantize_weights_plan(plan)
        self.quantize_bwd_weights_plan(plan)

    def quant_entries(self):
        return [e for e in (u.quant_entry() for u in self.conv_units()) if e is not None]

    def quant_entries_bwd(self):
        return [e for e in (u.quant_entry_bwd() for u in self.conv_units()) if e is not None]

    def quantize_bwd_weights_plan(self, plan, extra=()):
        """e4m3 twins of the tap-flipped transposed weights (after the transposes that produce them): ONE launch."""
        entries = self.quant_entries_bwd() + list(extra)
        if entries:
            key = tuple(e[1].data_ptr() for e in entries)
            cache = self.__dict__.setdefault("_quant_tables", {})
            if key not in cache:
                cache[key] = ops.make_weight_quant_table(entries, self.device)
            plan.add(ops.quantize_weights_fp8_batched, *cache[key])

    def quantize_weights_plan(self, plan, extra=()):
        """fp8 forward weights of every fp8 layer (+ extra entries, e.g. the RPN's) from the fp32 masters: ONE launch."""
        entries = self.quant_entries() + list(extra)
        if entries:
            key = tuple(e[1].data_ptr() for e in entries)
            cache = self.__dict__.setdefault("_quant_tables", {})
            if key not in cache:
                cache[key] = ops.make_weight_quant_table(entries, self.device)
            table, total = cache[key]
            plan.add(ops.quantize_weights_fp8_batched, table, total)

    def flip_entries(self):
        """(fp32 master, data-gradient weight buffer, cout, kh, kw, cin) of every conv except the stem (batched refresh)."""
        st = self.store
        return [(st.weight(u.name + "_conv/kernel"), u.w_t, u.cout, u.k, u.k, u.cin) for u in self.conv_units() if not u.is_stem]

    # ------------------------------------------------------------------ plans
    def setup(self, batch, training):
        """Allocate activations for a batch size; returns the static input buffer [B,H,W,3] uint8."""
        dev = self.device
        h, w = self.image_shape[0], self.image_shape[1]
        self.batch = batch
        self.images = torch.zeros(batch, h, w, 3, dtype=torch.uint8, device=dev)
        st = self.stem
        acc = None
        if training:
            slots = ops.STAT_SLOTS                              # FRCNN_STAT_SLOTS
            # per BN layer: forward statistics [slots][2][c] in f64 (two floats each) + backward partial sums [slots][2][c] fp32
            total = sum(3 * slots * 2 * u.cout for u in self.conv_units())
            self.acc_flat = torch.zeros(total, dtype=torch.float32, device=dev)
            cursor = [0]

            def acc(nfloats):
                v = self.acc_flat[cursor[0]:cursor[0] + nfloats]
                cursor[0] += nfloats
                return v
        st.setup(batch, h, w, dev, training, acc)
        npad = batch * st.hp * st.wp * 4
        self.xpad_flat = torch.zeros(npad + 256, dtype=BF16, device=dev)        # slack for the 8-wide tap reads
        self.xpad = self.xpad_flat[:npad].view(batch, st.hp, st.wp, 4)
        self.a_stem = None if training else torch.empty(st.m, 64, dtype=BF16, device=dev)   # (training: fused into the pool kernel)
        self.hp1, self.wp1 = (st.ho + 2 - 3) // 2 + 1, (st.wo + 2 - 3) // 2 + 1
        self.pool = torch.empty(batch * self.hp1 * self.wp1, 64, dtype=BF16, device=dev)
        self.pool_arg = torch.empty(batch * self.hp1 * self.wp1, 64, dtype=torch.uint8, device=dev)
        hi, wi = self.hp1, self.wp1
        self.acts = {}
        self.f8 = Fp8Scales(dev) if (training and self.precision == "fp8") else None
        prev_out8 = None
        for (n, ci, f, s, first) in self.specs:
            u = self.units[n]
            if first:
                u[0].setup(batch, hi, wi, dev, training, acc)
            u[1].setup(batch, hi, wi, dev, training, acc)
            ho, wo = u[1].ho, u[1].wo
            u[2].setup(batch, ho, wo, dev, training, acc)
            u[3].setup(batch, ho, wo, dev, training, acc)
            m = batch * ho * wo
            a = {"a1": torch.empty(m, f, dtype=BF16, device=dev), "a2": torch.empty(m, f, dtype=BF16, device=dev),
                 "out": torch.empty(m, 4 * f, dtype=BF16, device=dev)}
            if first and not training:
                a["sc"] = torch.empty(m, 4 * f, dtype=BF16, device=dev)
            elif first:
                a["sc"] = None                    # (training: the shortcut BatchNorm is fused into the block-final one)
            if self.f8 is not None:
                for k_ in sorted(u):                     # e5m2 twins of the BatchNorm-backward outputs that feed fp8 data / weight gradients
                    if u[k_].fp8_bwd or u[k_].fp8_wgrad:
                        u[k_].dz8 = Fp8Twin(self.f8, (u[k_].m, u[k_].cout), dev, e5m2=True)
                # fp8 twins of the activations that feed fp8 convolutions (forward and weight gradient): a1 -> 3x3, a2 -> 1x1
                # expansion, out -> the next block / RPN
                a["a1_8"] = Fp8Twin(self.f8, (m, f), dev) if u[2].fp8 or u[2].fp8_wgrad else None
                a["a2_8"] = Fp8Twin(self.f8, (m, f), dev) if u[3].fp8 or u[3].fp8_wgrad else None
                # (a block output gets its twin whenever the stage is in fp8 mode: the NEXT block's 1x1 convolutions -- forward pass
                # and weight gradient -- read it even where this block's own 64-channel layers (conv2) cannot run in fp8; same-box A/B:
                # batch 8 6.32 -> 6.25 ms, pyramid 8.67 -> 8.54, batch 4 3.93 -> 3.875)
                a["out_8"] = Fp8Twin(self.f8, (m, 4 * f), dev) if (u[3].fp8 or u[2].fp8 or (FP8_OUT8_ALWAYS and u[3].fp8_wgrad)) else None
                # a unit whose data gradient AND weight gradient both run on the e5m2 twin of dz never reads the bf16 tensor
                x8_of = {0: prev_out8, 1: prev_out8, 2: a["a1_8"], 3: a["a2_8"]}
                for k_ in sorted(u):
                    u[k_].dz_twin_only = bool(FP8_DZ_TWIN_ONLY and u[k_].dz8 is not None and u[k_].fp8_bwd and u[k_].fp8_wgrad
                                              and x8_of[k_] is not None)
                # likewise the bf16 activations a1 / a2 when the convolution that reads them runs its forward pass AND its weight
                # gradient on the e4m3 twin (the BatchNorm backward pass reads the ReLU bit mask, not the activation)
                a["a1_twin_only"] = bool(FP8_DZ_TWIN_ONLY and a["a1_8"] is not None and u[2].fp8 and u[2].fp8_wgrad and u[2].dz8 is not None)
                a["a2_twin_only"] = bool(FP8_DZ_TWIN_ONLY and a["a2_8"] is not None and u[3].fp8 and u[3].fp8_wgrad and u[3].dz8 is not None)
            prev_out8 = a.get("out_8")
            if training:
                a["g1"] = torch.empty(m, f, dtype=BF16, device=dev)      # grad wrt a1
                a["g2"] = torch.empty(m, f, dtype=BF16, device=dev)      # grad wrt a2
                a["gin"] = torch.empty(batch * hi * wi, ci, dtype=BF16, device=dev)   # grad wrt the block input
            self.acts[n] = a
            hi, wi = ho, wo
        if training:
            self.g_stem = torch.empty(st.m, 64, dtype=BF16, device=dev)
        self.feature_maps = self.acts[self.specs[-1][0]]["out"].view(batch, hi, wi, self.out_channels)
        self.feature_maps8 = self.acts[self.specs[-1][0]].get("out_8")           # Fp8Twin of the feature maps (fp8 training) or None
        return self.images

    def forward_plan(self, plan, training):
        st = self.stem
        if training:
            # BN statistics / backward partial sums are accumulated with atomics: one multi-tensor zero per step
            plan.zero(self.acc_flat)
        plan.add(ops.preprocess, self.images, self.xpad, 3)
        st.forward(plan, self.xpad_flat, training)
        if training:
            # conv1_bn + conv1_relu + pool1 in one pass over z: the 60 MB activation between them (375x1242, batch 4) is neither
            # written nor read back; the kernel also leaves the ReLU bit mask the backward pass reads
            plan.add(ops.bn_train_apply_maxpool, st.z, st.stats, st.tiles, st.m * st.sync_world, self.store.weight(st.name + "_bn/gamma"),
                     self.store.weight(st.name + "_bn/beta"), st.mm, st.mv, BN_MOMENTUM, BN_EPS, self.pool, self.pool_arg, st.relu_mask,
                     st.mean, st.invstd, self.batch, st.ho, st.wo, 64, self.hp1, self.wp1)
        else:
            st.apply(plan, self.a_stem, relu=True)
            plan.add(ops.maxpool_fwd, self.a_stem, self.pool, self.pool_arg, self.batch, st.ho, st.wo, 64, self.hp1, self.wp1)
        x, x8 = self.pool, None
        f8 = self.f8 if training else None
        if f8 is not None:
            f8.plan_zero(plan)                        # this step's amax row
        for (n, ci, f, s, first) in self.specs:
            u, a = self.units[n], self.acts[n]
            fused_shortcut = first and training      # shortcut BatchNorm applied inside the block-final BatchNorm kernel (no a["sc"])
            if first:
                u[0].forward(plan, x, training, x8)
                if not fused_shortcut:
                    u[0].apply(plan, a["sc"], relu=False)
                res = a["sc"]
            else:
                res = x
            u[1].forward(plan, x, training, x8)
            bnin = training and f8 is None and BN_IN_FUSED != "0"
            if bnin and ops.conv2d_bnin_supported(u[2].desc) and (BN_IN_FUSED != "wres" or u[2].cin == 64):
                # the 3x3 convolution applies the first BatchNorm + ReLU of the block itself (round 4: conv2's blocks on the weights-resident
                # kernel; round 5: conv3 / conv4 on the patch-resident kernel's loader-wave forms)
                u[2].forward_bnin(plan, u[1], a["a1"])
            else:
                u[1].apply(plan, a["a1"], out8=a.get("a1_8") if f8 is not None else None, twin_only=f8 is not None and a.get("a1_twin_only", False))
                u[2].forward(plan, a["a1"], training, a.get("a1_8") if f8 is not None else None)
            if bnin and (BN_IN_FUSED == "1" or (BN_IN_FUSED == "3x3c2" and u[3].cin == 64)) and ops.conv2d_bnin_supported(u[3].desc):
                # ... and the block's third convolution (1x1) its second BatchNorm + ReLU (round 5: the tile kernel transforms every landed A slice)
                u[3].forward_bnin(plan, u[2], a["a2"])
            else:
                u[2].apply(plan, a["a2"], out8=a.get("a2_8") if f8 is not None else None, twin_only=f8 is not None and a.get("a2_twin_only", False))
                u[3].forward(plan, a["a2"], training, a.get("a2_8") if f8 is not None else None)
            o8 = a.get("out_8") if f8 is not None else None
            if fused_shortcut:
                u[3].apply(plan, a["out"], relu=True, dual=u[0], out8=o8)
            else:
                u[3].apply(plan, a["out"], res=res, relu=True, out8=o8)
            x, x8 = a["out"], o8
        return self.feature_maps

    def last_unit(self):
        """The conv unit whose BatchNorm+ReLU produces feature_maps (consumer of the feature-map gradient)."""
        return self.units[self.specs[-1][0]][3]

    def backward_plan(self, plan, g_feat, g_feat_reduced=False, injected=(), on_stage_done=None):
        """g_feat: bf16 gradient w.r.t. feature_maps [B*gh*gw, C].  Cuts the plan after each stage.
        g_feat_reduced: the kernel that wrote g_feat already ran the BN-backward reduce of last_unit().
        injected: names of stride-2 first blocks whose input-gradient buffer acts[name]["gin"] ALREADY holds a gradient w.r.t. the
        previous stage's output (a second consumer of that output: the feature pyramid's lateral convolution, models/fpn.py);
        the block's own data gradients are then added to it instead of being scattered into a zeroed buffer."""
        gout = g_feat
        gout_reduced = g_feat_reduced
        prev_of = {}
        for i, spec in enumerate(self.specs):
            prev_of[spec[0]] = self.units[self.specs[i - 1][0]][3] if i > 0 else None
        prev_stage = None
        xs, xs8 = {}, {}
        x, x8 = self.pool, None
        for (n, ci, f, s, first) in self.specs:
            xs[n], xs8[n] = x, x8
            x, x8 = self.acts[n]["out"], self.acts[n].get("out_8")
        # the weight gradients of a stage are launched together (ops.WgradGroup): one pixel split sized for the whole group
        # instead of one per layer -- none for conv4 (19 layers, 1728 tiles), ~10 instead of 64-128 for conv2 -- i.e. a
        # fraction of the float atomics, and one ramp-up / tail per stage
        grouped_stage = (2, 3, 4) if GROUPED_WGRAD else ()
        deferred = []

        def flush_deferred():
            if deferred:
                group = ops.WgradGroup(deferred, self.device)
                plan.hold(group)
                if WGRAD_TRAIL:
                    # the stage's weight gradients on a side stream that trails the main chain (nothing before the update reads them)
                    with plan.branch("wgrad_trail", follow=True):
                        plan.add(ops.conv2d_wgrad_grouped, group)
                else:
                    plan.add(ops.conv2d_wgrad_grouped, group)
                del deferred[:]

        for (n, ci, f, s, first) in reversed(self.specs):
            stage = int(n[4])
            if prev_stage is not None and stage != prev_stage:
                flush_deferred()
                if on_stage_done is not None:     # every gradient of stage `prev_stage` is final (its bucket of the flat buffer)
                    on_stage_done(prev_stage)
                if not WGRAD_TRAIL:               # (a segment's end joins its side streams: the trailing form keeps the backbone's backward pass in one segment)
                    plan.cut("bwd_conv%d" % stage)
            prev_stage = stage
            defer = deferred if stage in grouped_stage else None
            u, a, xin = self.units[n], self.acts[n], xs[n]
            # every data-gradient kernel also runs the BN-backward reduce of the layer that consumes its output
            # the block-output gradient after the final ReLU (g * mask) is never materialised: its two consumers -- the
            # shortcut branch and the residual add of the block-input gradient -- read gout and the block's ReLU bit mask
            gblock, mblock = gout, u[3].relu_mask
            # (first block of a stage: the shortcut BatchNorm's backward sums -- same gradient, same mask -- in the same launch)
            red2 = first and BN_RED2 and u[3].cout % 64 == 0
            u[3].backward_bn(plan, gout, a["out"], reduced=gout_reduced, also_reduce=u[0] if red2 else None)
            u[3].backward_weights(plan, a["a2"], defer, a.get("a2_8"))
            u[3].backward_data(plan, a["g2"], consumer=u[2])
            u[2].backward_bn(plan, a["g2"], a["a2"], reduced=True)
            u[2].backward_weights(plan, a["a1"], defer, a.get("a1_8"))
            u[2].backward_data(plan, a["g1"], consumer=u[1])
            u[1].backward_bn(plan, a["g1"], a["a1"], reduced=True)
            u[1].backward_weights(plan, xin, defer, xs8[n])
            prev = prev_of[n]                     # block whose output this block's input gradient is (None: max-pool output)
            if first:
                u[0].backward_bn(plan, gblock, None, mask=mblock, reduced=red2)
                u[0].backward_weights(plan, xin, defer, xs8[n])
                if s != 1 and n in injected:
                    u[1].backward_data(plan, a["gin"], res=a["gin"])      # add to the gradient the other consumer left there
                else:
                    if s != 1:
                        plan.zero(a["gin"], late=True)         # (scatter target of the stride-2 data gradients)
                    u[1].backward_data(plan, a["gin"])
                if s != 1 and n in injected:
                    # the pixels the stride-2 scatter does not touch hold the other consumer's gradient, not zeros: the fused
                    # BatchNorm-backward reduce (which sums over the rows the kernel stores) would miss them -> stand-alone reduce
                    u[0].backward_data(plan, a["gin"], res=a["gin"])
                    fused = False
                else:
                    u[0].backward_data(plan, a["gin"], res=a["gin"], consumer=prev)   # gin is complete here (untouched pixels are zero)
                    fused = prev is not None
            else:
                u[1].backward_data(plan, a["gin"], res=gblock, consumer=prev, res_mask=mblock)
                fused = prev is not None
            gout = a["gin"]
            gout_reduced = fused
            if len(deferred) >= 24:               # (the parameter table holds 32 layers per addressing mode)
                flush_deferred()
        flush_deferred()
        st = self.stem
        # the pool's backward pass also accumulates the stem BatchNorm's backward sums (g_stem, z and the mask are not re-read)
        red = st.reduce_args(relu=True)
        plan.hold(red)
        plan.add(ops.maxpool_bwd_bnreduce, gout, self.pool_arg, self.g_stem, self.batch, st.ho, st.wo, 64, self.hp1, self.wp1, red)
        st.backward_bn(plan, self.g_stem, st.z, reduced=True)            # (second argument: only "this layer ends in a ReLU")
        st.backward_weights(plan, self.xpad_flat)

    # ------------------------------------------------------------------ Keras-model-like call
    def __call__(self, images, training=False):
        key = (int(images.shape[0]), bool(training))
        if key not in self._plans:
            self.setup(key[0], training)
            self.store.refresh_bf16()
            plan = Plan("feature_extractor")
            self.refresh_weights(plan)
            self.forward_plan(plan, training)
            self._plans = {key: plan}           # buffers are re-created per geometry: keep only the live plan
        self.images.copy_(images)
        self._plans[key].run_synced()
        return self.feature_maps


def get_feature_extractor_model(image_shape, depth=50, store=None, device="cuda", sync_bn_world=1, precision="bf16"):
    """reference models/feature_extractor.py:4 (weights: seeded synthetic init; load real ones with set_weights)."""
    return FeatureExtractor(image_shape, depth=depth, store=store, device=device, sync_bn_world=sync_bn_world, precision=precision)
