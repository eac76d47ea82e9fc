"""Torch-tensor front end of the C ABI (include/frcnn_hip.h).

PyTorch is plumbing only (device memory, streams): every function here hands raw device pointers
to lib2dod_hip.so on the current torch stream and launches hand-written gfx950 kernels.  Nothing
here computes with torch ops, and nothing falls back to them.
"""
import ctypes
from ctypes import byref, c_float, c_void_p

import torch

from . import _lib
from ._lib import (CONV_ADD_RES, CONV_BIAS, CONV_OUT_F32, CONV_RELU, CONV_SPLITK_ATOMIC, CONV_STATS, CONV_WGRAD_ACCUMULATE, CONV_WGRAD_STEM_UNPACK, BnReduce, ConvDesc, Fp8Out, call)

BF16 = torch.bfloat16


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError("%s must be a CUDA (HIP) tensor: the HIP path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s: expected %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def conv_desc(n, hi, wi, cin, kh, kw, stride, pad_h, pad_w, ho, wo, cout, in_pix_stride=None, out_h=None, out_w=None,
              out_scatter=1, flags=0, split_k=1):
    return ConvDesc(n, hi, wi, cin if in_pix_stride is None else in_pix_stride, cin, kh, kw, stride, pad_h, pad_w, ho, wo, cout,
                    ho if out_h is None else out_h, wo if out_w is None else out_w, out_scatter, flags, split_k)


def conv_workspace_bytes(d):
    """Scratch bytes with which conv2d_fprop / conv2d_dgrad_bnreduce run descriptor d in the split-K fix-up form (0: not used)."""
    return int(_lib.load().frcnn_conv2d_workspace_bytes(byref(d)))


def conv_attach_workspace(d, device):
    """Give descriptor d its own zeroed workspace when the dispatcher can use one (fewer tiles than CUs, long K).  Returns the
    tensor (the caller keeps it alive) or None."""
    n = conv_workspace_bytes(d)
    if n == 0:
        return None
    ws = torch.zeros(n, dtype=torch.uint8, device=device)      # (a CPU tensor serves the dispatcher's dry runs: tests/test_conv_dispatch.py)
    d.workspace = c_void_p(ws.data_ptr())
    d.workspace_bytes = n
    d._ws_tensor = ws                                          # the descriptor keeps its scratch alive
    cb = int(_lib.load().frcnn_conv2d_workspace_counter_bytes(byref(d)))
    d._ws_counters = ws[n - cb:] if cb else None               # arrival counters: plans re-zero them every step (conv_zero_counters)
    return ws


def conv_zero_counters(plan, d):
    """Register the arrival counters of descriptor d's split-K workspace (if it has one) with the plan's per-step zero fill: a
    counter left non-zero by an aborted launch must not outlive the step."""
    c = getattr(d, "_ws_counters", None)
    if c is not None:
        plan.zero(c)


STAT_SLOTS = 16          # FRCNN_STAT_SLOTS of include/frcnn_hip.h (checked against the library in tests/test_abi.py)


def last_conv_instantiation():
    """Which MFMA conv kernel (template arguments, grid) this thread launched last -- used by the parity tests."""
    return _lib.load().frcnn_last_conv_instantiation().decode()


def conv2d_describe(d, with_bn_reduce=False):
    """The kernel instantiation conv2d_fprop / conv2d_dgrad_bnreduce would launch for descriptor d (host logic, no device)."""
    r = _lib.load().frcnn_conv2d_describe(byref(d), 1 if with_bn_reduce else 0)
    if r is None:
        raise RuntimeError("frcnn_conv2d_describe: " + _lib.load().frcnn_last_error().decode())
    return r.decode()


def conv2d_wgrad_describe(d=None, with_row_index=False, group=None):
    """The kernel(s) conv2d_wgrad (or conv2d_wgrad_grouped for a WgradGroup) would launch (host logic, no device)."""
    r = _lib.load().frcnn_conv2d_wgrad_describe(byref(d) if d is not None else None, 1 if with_row_index else 0,
                                                c_void_p(group.host.data_ptr()) if group is not None else None)
    if r is None:
        raise RuntimeError("frcnn_conv2d_wgrad_describe: " + _lib.load().frcnn_last_error().decode())
    return r.decode()


def conv_stat_tiles(d):
    return _lib.load().frcnn_conv2d_stat_tiles(byref(d))


def conv2d_fprop(d, x, w, y, bias=None, res=None, stats=None):
    call("frcnn_conv2d_fprop", byref(d), _p(x), _p(w), _p(bias), _p(res), _p(y), _p(stats), _stream())


# ---------------------------------------------------------------- fp8 (e4m3) convolution path
FP8_AMAX_SLOTS = 8192        # FRCNN_FP8_AMAX_SLOTS
FP8 = torch.uint8            # storage type of OCP e4m3 bytes (torch.float8_e4m3fn views of these tensors are used by the tests only)


def conv2d_fprop_fp8(d, x8, w8, x_scale, w_scale, y, bias=None, stats=None):
    """y = bf16(x_scale * w_scale[co] * conv(x8, w8) + bias): both operands e4m3 bytes, x_scale a device scalar, w_scale [cout]."""
    call("frcnn_conv2d_fprop_fp8", byref(d), _p(x8), _p(w8), _p(x_scale), _p(w_scale), _p(bias), _p(y), _p(stats), _stream())


def conv2d_describe_fp8(d):
    r = _lib.load().frcnn_conv2d_describe_fp8(byref(d))
    if r is None:
        raise RuntimeError("frcnn_conv2d_describe_fp8: " + _lib.load().frcnn_last_error().decode())
    return r.decode()


def conv2d_dgrad_fp8(d, dz8, w_t8, dz_scale, w_scale, gx, red=None, res=None, res_mask=None):
    """gx = bf16(dz_scale * w_scale[ci] * conv(dz8 (e5m2), w_t8 (e4m3))) [+ res (* res_mask bits)] [+ fused BatchNorm-backward reduce]."""
    call("frcnn_conv2d_dgrad_fp8", byref(d), _p(dz8), _p(w_t8), _p(dz_scale), _p(w_scale), _p(res), _p(res_mask), _p(gx),
         byref(red) if red is not None else None, _stream())


def conv2d_describe_dgrad_fp8(d, with_bn_reduce=False):
    r = _lib.load().frcnn_conv2d_describe_dgrad_fp8(byref(d), 1 if with_bn_reduce else 0)
    if r is None:
        raise RuntimeError("frcnn_conv2d_describe_dgrad_fp8: " + _lib.load().frcnn_last_error().decode())
    return r.decode()


def quantize_fp8(x, qscale, out8, amax=None, e5m2=False):
    call("frcnn_quantize_fp8", _p(x), x.numel(), _p(qscale), _p(out8), _p(amax), 1 if e5m2 else 0, _stream())


def make_weight_quant_table(entries, device):
    """entries: list of (source [rows, K] view -- fp32 masters or bf16 derived weights --, fp8 destination, float scale[rows])
    -> (int64 table on device, total rows)."""
    rows, begin = [], 0
    for (w, w8, scale) in entries:
        r = int(w.shape[0])
        k = w.numel() // r
        assert k % 8 == 0 and w.is_contiguous() and w8.numel() == w.numel() and scale.numel() == r and w.dtype in (torch.float32, BF16)
        rows.append([w.data_ptr(), w8.data_ptr(), scale.data_ptr(), r, k, begin, 1 if w.dtype == BF16 else 0, 0])
        begin += r
    return torch.tensor(rows, dtype=torch.int64, device=device), begin


def quantize_weights_fp8_batched(table, total_rows):
    call("frcnn_quantize_weights_fp8_batched", _p(table), table.shape[0], total_rows, _stream())


def fp8_update_scales(amax, scale, qscale, n, margin=1.0, limit=None, status=None):
    """status (int32 [2], optional): [0] += tensors clamped this step (amax > limit * scale), [1] += non-finite amax (scale kept)"""
    call("frcnn_fp8_update_scales", _p(amax), _p(scale), _p(qscale), n, float(margin), _p(limit), _p(status), _stream())


def fp8_out(out8, qscale, amax=None):
    """frcnn_fp8_out for bn_train_apply / bn_train_apply_dual (keep the struct alive as long as a plan refers to it)."""
    return Fp8Out(out8.data_ptr(), qscale.data_ptr(), amax.data_ptr() if amax is not None else None)


def bn_reduce_args(z, relu_mask, mean, invstd, partial):
    """frcnn_bn_reduce for conv2d_dgrad_bnreduce (keep the returned struct alive as long as a plan refers to it)."""
    return BnReduce(z.data_ptr(), relu_mask.data_ptr() if relu_mask is not None else None, mean.data_ptr(), invstd.data_ptr(),
                    partial.data_ptr())


def bn_in_args(stats, gamma, beta, mm, mv, momentum, eps, count, act, relu_mask, mean, invstd):
    """frcnn_bn_in for conv2d_fprop_bnin (keep the returned struct alive as long as a plan refers to it)."""
    b = _lib.BnIn()
    b.stats_partial, b.gamma, b.beta, b.moving_mean, b.moving_var = _p(stats), _p(gamma), _p(beta), _p(mm), _p(mv)
    b.momentum, b.eps, b.count = float(momentum), float(eps), int(count)
    b.act, b.relu_mask, b.mean, b.invstd = _p(act), _p(relu_mask), _p(mean), _p(invstd)
    b._keep = (stats, gamma, beta, mm, mv, act, relu_mask, mean, invstd)
    return b


def conv2d_bnin_supported(d):
    """Can conv2d_fprop_bnin run descriptor d (the BatchNorm + ReLU of the input layer applied by the convolution itself)?"""
    return bool(_lib.load().frcnn_conv2d_bnin_supported(byref(d)))


def conv2d_fprop_bnin(d, z_in, w, y, bn, bias=None, stats=None):
    """== bn_train_apply(z_in -> bn.act, bn.relu_mask, ...) followed by conv2d_fprop(d, bn.act, w, y, ...), in one launch."""
    call("frcnn_conv2d_fprop_bnin", byref(d), _p(z_in), _p(w), _p(bias), _p(y), _p(stats), byref(bn), _stream())


def conv2d_dgrad_bnreduce(d, dz, w_t, gx, red, res=None, res_mask=None):
    """gx = conv(dz, w_t) [+ res (* res_mask bits)], fused with the BatchNorm-backward reduce of the layer that consumes gx."""
    call("frcnn_conv2d_dgrad_bnreduce", byref(d), _p(dz), _p(w_t), _p(res), _p(res_mask), _p(gx), byref(red), _stream())


def conv2d_wgrad(d, x, dz, dw, dz_stride=None, row_index=None):
    call("frcnn_conv2d_wgrad", byref(d), _p(x), _p(dz), d.cout if dz_stride is None else dz_stride, _p(row_index), _p(dw), _stream())


def conv2d_wgrad_fp8(d, x8, dz8, x_scale, dz_scale, dw, dz_stride=None):
    """dw += x_scale * dz_scale * sum_p dz8[p] (x) x8[im2col(p)]: the weight gradient from the fp8 twins (x8: e4m3, dz8: e5m2)."""
    call("frcnn_conv2d_wgrad_fp8", byref(d), _p(x8), _p(dz8), d.cout if dz_stride is None else dz_stride, _p(x_scale), _p(dz_scale), _p(dw),
         _stream())


def conv2d_wgrad_describe_fp8(d):
    r = _lib.load().frcnn_conv2d_wgrad_describe_fp8(byref(d))
    if r is None:
        raise RuntimeError("frcnn_conv2d_wgrad_describe_fp8: " + _lib.load().frcnn_last_error().decode())
    return r.decode()


class WgradGroup:
    """Several weight gradients launched together without a pixel split (frcnn_conv2d_wgrad_grouped).  items: list of
    (conv_desc, x, dz, dw) -- bf16 operands -- or (conv_desc, x8, dz8, dw, x_scale, dz_scale) -- fp8 twins and their
    dequantisation scales -- with static shapes and buffers; the parameter table is built once and kept on the device."""

    def __init__(self, items, device):
        self.items = list(items)                 # keeps descriptors and tensors alive
        n = len(self.items)
        arr = (_lib.WgradItem * n)()
        for i, it in enumerate(self.items):
            d, x, dz, dw = it[:4]
            arr[i].desc = ctypes.pointer(d)
            arr[i].x, arr[i].dz, arr[i].dw = _p(x), _p(dz), _p(dw)
            arr[i].dz_stride = d.cout
            if len(it) > 4:
                arr[i].x_scale, arr[i].dz_scale = _p(it[4]), _p(it[5])
        nbytes = int(_lib.load().frcnn_wgrad_group_bytes())
        self.host = torch.zeros(nbytes, dtype=torch.uint8)
        call("frcnn_conv2d_wgrad_group_plan", ctypes.cast(arr, c_void_p), n, self.host.data_ptr(), nbytes)
        self.dev = self.host.to(device)


def conv2d_wgrad_grouped(group):
    call("frcnn_conv2d_wgrad_grouped", group.host.data_ptr(), _p(group.dev), _stream())


def weights_transpose_flip(w, w_t, cout, kh, kw, cin):
    call("frcnn_weights_transpose_flip", _p(w), _p(w_t), cout, kh, kw, cin, _stream())


def make_transpose_flip_table(entries, device):
    """entries: list of (w fp32 master view, w_t bf16 buffer, cout, kh, kw, cin) -> (int64 table on device, total tiles).
    The kernel runs one workgroup per 32 x 32 (cout, cin) tile of one filter tap."""
    rows, begin = [], 0
    for (w, w_t, cout, kh, kw, cin) in entries:
        rows.append([w.data_ptr(), w_t.data_ptr(), cout, kh, kw, cin, begin, 0])
        begin += kh * kw * ((cout + 31) // 32) * ((cin + 31) // 32)
    return torch.tensor(rows, dtype=torch.int64, device=device), begin


def weights_transpose_flip_batched(table, total):
    call("frcnn_weights_transpose_flip_batched", _p(table), table.shape[0], total, _stream())


def copy_bytes(src, dst):
    """dst <- src (same dtype / shape, contiguous, 16-byte aligned): a full-width copy kernel instead of the runtime's blit."""
    nbytes = src.numel() * src.element_size()
    if nbytes != dst.numel() * dst.element_size() or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("copy_bytes: source of %d bytes, destination of %d bytes (contiguous tensors of equal size expected)" % (
            nbytes, dst.numel() * dst.element_size()))
    call("frcnn_copy_bytes", _p(src), _p(dst), nbytes, _stream())


def copy_bytes_multi(pairs):
    """[(src, dst), ...] (1..4 pairs of contiguous device tensors of equal byte size, 16-byte aligned): one launch."""
    n = len(pairs)
    srcs = (c_void_p * n)(*[s.data_ptr() for s, _ in pairs])
    dsts = (c_void_p * n)(*[d.data_ptr() for _, d in pairs])
    sizes = (ctypes.c_int64 * n)(*[s.numel() * s.element_size() for s, _ in pairs])
    for s_, d_ in pairs:
        if s_.numel() * s_.element_size() != d_.numel() * d_.element_size() or not (s_.is_contiguous() and d_.is_contiguous()):
            raise ValueError("copy_bytes_multi: every pair must be contiguous and of equal byte size")
    call("frcnn_copy_bytes_multi", srcs, dsts, sizes, n, _stream())


def make_zero_table(tensors, device):
    """Table for fill_zero_multi over `tensors` (contiguous, 16-byte aligned, byte sizes multiples of 16)."""
    rows, chunk = [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if not t.is_contiguous() or t.data_ptr() % 16 or nbytes % 16:
            raise ValueError("fill_zero_multi: buffers must be contiguous, 16-byte aligned, a multiple of 16 bytes long")
        rows.append([t.data_ptr(), chunk])
        chunk += nbytes // 16
    rows.append([0, chunk])
    return torch.tensor(rows, dtype=torch.int64, device=device), len(tensors), chunk


def fill_zero_multi(table, n, total_chunks):
    call("frcnn_fill_zero_multi", _p(table), n, total_chunks, _stream())


def cast_f32_bf16(src, dst, n=None):
    call("frcnn_cast_f32_bf16", _p(src), _p(dst), src.numel() if n is None else n, _stream())


def stem_pack_weights(w, w_packed, cout=64):
    call("frcnn_stem_pack_weights", _p(w), _p(w_packed), cout, _stream())


def stem_unpack_grad(dw_packed, dw, cout=64):
    call("frcnn_stem_unpack_grad", _p(dw_packed), _p(dw), cout, _stream())


def preprocess(images_u8, out, pad=3):
    _chk(images_u8, torch.uint8, "images")
    b, h, w, _ = images_u8.shape
    _, hp, wp, _ = out.shape
    call("frcnn_preprocess_u8_bgr_mean", _p(images_u8), _p(out), b, h, w, hp, wp, pad, _stream())


def bn_finalize_train(stats, tiles, c, count, gamma, beta, mm, mv, momentum, eps, scale, shift, mean, invstd):
    call("frcnn_bn_finalize_train", _p(stats), tiles, c, count, _p(gamma), _p(beta), _p(mm), _p(mv), momentum, eps,
         _p(scale), _p(shift), _p(mean), _p(invstd), _stream())


def bn_finalize_eval(c, gamma, beta, mm, mv, eps, scale, shift):
    call("frcnn_bn_finalize_eval", c, _p(gamma), _p(beta), _p(mm), _p(mv), eps, _p(scale), _p(shift), _stream())


def bn_apply(z, scale, shift, out, m, c, res=None, relu=True):
    call("frcnn_bn_apply", _p(z), _p(scale), _p(shift), _p(res), 1 if relu else 0, _p(out), m, c, _stream())


def bn_train_apply(z, stats, slots, count, gamma, beta, mm, mv, momentum, eps, out, mean, invstd, m, c, res=None, relu=True,
                   relu_mask=None, f8=None):
    call("frcnn_bn_train_apply", _p(z), _p(stats), slots, count, _p(gamma), _p(beta), _p(mm), _p(mv), momentum, eps, _p(res),
         1 if relu else 0, _p(out), _p(relu_mask), _p(mean), _p(invstd), m, c, byref(f8) if f8 is not None else None, _stream())


def bn_train_apply_dual(z, stats, gamma, beta, mm, mv, mean, invstd, z2, stats2, gamma2, beta2, mm2, mv2, mean2, invstd2, slots, count,
                        momentum, eps, out, m, c, relu=True, relu_mask=None, f8=None):
    """out = [ReLU](BN(z) + BN2(z2)), both with batch statistics, in one pass (block-final + shortcut BatchNorm of a first block)."""
    call("frcnn_bn_train_apply_dual", _p(z), _p(stats), _p(gamma), _p(beta), _p(mm), _p(mv), _p(mean), _p(invstd), _p(z2), _p(stats2),
         _p(gamma2), _p(beta2), _p(mm2), _p(mv2), _p(mean2), _p(invstd2), slots, count, momentum, eps, 1 if relu else 0, _p(out),
         _p(relu_mask), m, c, byref(f8) if f8 is not None else None, _stream())


def bn_train_apply_maxpool(z, stats, slots, count, gamma, beta, mm, mv, momentum, eps, pooled, argmax, relu_mask, mean, invstd, n, h, w, c,
                           ho, wo):
    """bn_train_apply (ReLU, bit mask) + maxpool_fwd in one pass: the activation between them is never written."""
    call("frcnn_bn_train_apply_maxpool", _p(z), _p(stats), slots, count, _p(gamma), _p(beta), _p(mm), _p(mv), momentum, eps, _p(pooled),
         _p(argmax), _p(relu_mask), _p(mean), _p(invstd), n, h, w, c, ho, wo, _stream())


def bn_bwd_apply_fused(gout, act, z, mean, invstd, gamma, partial, slots, dgamma, dbeta, dz, gpre, m, c, relu_mask=None, count=0,
                       param_grad_scale=1.0, f8=None):
    call("frcnn_bn_bwd_apply_fused", _p(gout), _p(act), _p(relu_mask), _p(z), _p(mean), _p(invstd), _p(gamma), _p(partial), slots,
         _p(dgamma), _p(dbeta), _p(dz), _p(gpre), m, c, count, float(param_grad_scale), byref(f8) if f8 is not None else None, _stream())


def bn_bwd_apply_fused_red2(gout, z, mean, invstd, gamma, partial, slots, dgamma, dbeta, dz, m, c, relu_mask, red2, count=0, param_grad_scale=1.0, f8=None):
    """bn_bwd_apply_fused (ReLU bit mask form) + the backward reduce of a second BatchNorm fed by the same masked gradient (red2 = bn_reduce_args
    of that layer: the shortcut BatchNorm of a stage's first block)."""
    call("frcnn_bn_bwd_apply_fused_red2", _p(gout), _p(relu_mask), _p(z), _p(mean), _p(invstd), _p(gamma), _p(partial), slots, _p(dgamma), _p(dbeta),
         _p(dz), m, c, count, float(param_grad_scale), byref(f8) if f8 is not None else None, byref(red2), _stream())


def bn_bwd_blocks(m):
    return _lib.load().frcnn_bn_bwd_blocks(m)


def bn_bwd_reduce(gout, act, z, mean, invstd, partial, m, c, relu_mask=None):
    call("frcnn_bn_bwd_reduce", _p(gout), _p(act), _p(relu_mask), _p(z), _p(mean), _p(invstd), _p(partial), m, c, _stream())


def bn_bwd_finalize(partial, blocks, c, m, dgamma, dbeta, c1, c2):
    call("frcnn_bn_bwd_finalize", _p(partial), blocks, c, m, _p(dgamma), _p(dbeta), _p(c1), _p(c2), _stream())


def bn_bwd_apply(gout, act, z, mean, invstd, gamma, c1, c2, dz, gpre, m, c):
    call("frcnn_bn_bwd_apply", _p(gout), _p(act), _p(z), _p(mean), _p(invstd), _p(gamma), _p(c1), _p(c2), _p(dz), _p(gpre), m, c,
         _stream())


def relu_bwd(g, act, out, n=None):
    call("frcnn_relu_bwd", _p(g), _p(act), _p(out), g.numel() if n is None else n, _stream())


def colsum_bf16(x, m, c, ld, out):
    call("frcnn_colsum_bf16", _p(x), m, c, ld, _p(out), _stream())


def maxpool_fwd(x, y, argmax, n, h, w, c, ho, wo):
    call("frcnn_maxpool3x3s2_fwd", _p(x), _p(y), _p(argmax), n, h, w, c, ho, wo, _stream())


def maxpool_bwd(gy, argmax, gx, n, h, w, c, ho, wo):
    call("frcnn_maxpool3x3s2_bwd", _p(gy), _p(argmax), _p(gx), n, h, w, c, ho, wo, _stream())


def maxpool_bwd_bnreduce(gy, argmax, gx, n, h, w, c, ho, wo, red):
    """maxpool_bwd + the BatchNorm-backward reduce of the layer that produced the pooled activation (red: bn_reduce_args), one launch"""
    call("frcnn_maxpool3x3s2_bwd_bnreduce", _p(gy), _p(argmax), _p(gx), n, h, w, c, ho, wo, byref(red), _stream())


def sgd_momentum(w, g, v, w_bf16, n, momentum, l2, grad_scale, step, boundaries, values, nb):
    call("frcnn_sgd_momentum", _p(w), _p(g), _p(v), _p(w_bf16), n, momentum, l2, grad_scale, _p(step), _p(boundaries), _p(values), nb,
         _stream())


def sgd_momentum_fused(w, g, v, w_bf16, n, momentum, grad_scale, step, boundaries, values, nb, fused):
    """the whole optimizer step in one launch (both decay ranges, the stem's packed taps, the step counter): fused = sgd_fused_args(...)"""
    call("frcnn_sgd_momentum_fused", _p(w), _p(g), _p(v), _p(w_bf16), n, momentum, grad_scale, _p(step), _p(boundaries), _p(values), nb,
         byref(fused), _stream())


def sgd_fused_args(decay_end, l2, arrive, stem_begin=-1, stem_cout=0, stem_packed=None):
    f = _lib.SgdFused(int(decay_end), float(l2), int(stem_begin), int(stem_cout), _p(stem_packed), _p(arrive))
    f._keep = (arrive, stem_packed)              # the struct is passed by value at launch; the tensors must outlive the plan
    return f


def cast_colsum(src, dst, m, c, colsum):
    """dst = bf16(src) and colsum[col] += column sums of dst, one launch (cast_f32_bf16 + colsum_bf16)"""
    call("frcnn_cast_colsum", _p(src), _p(dst), m, c, _p(colsum), _stream())


def relu_bwd_colsum(g, act, out, m, c, colsum):
    """out = act > 0 ? g : 0 and colsum[col] += column sums of out, one launch (relu_bwd + colsum_bf16)"""
    call("frcnn_relu_bwd_colsum", _p(g), _p(act), _p(out), m, c, _p(colsum), _stream())


def step_increment(step):
    call("frcnn_step_increment", _p(step), _stream())


def anchors_generate(out, gh, gw, scales, ratios, base_h, base_w, stride_h=16.0, stride_w=16.0):
    sc = (c_float * len(scales))(*scales)
    ra = (c_float * len(ratios))(*ratios)
    call("frcnn_anchors_generate", _p(out), gh, gw, sc, len(scales), ra, len(ratios), base_h, base_w, stride_h, stride_w, _stream())


def rpn_head_post(head, ld, b, num_anchors_total, a_per_loc, keep, n, scores, deltas):
    call("frcnn_rpn_head_post", _p(head), ld, b, num_anchors_total, a_per_loc, _p(keep), n, _p(scores), _p(deltas), _stream())


def rpn_head_post_decode(head, ld, b, num_anchors_total, a_per_loc, keep, n, scores, deltas, regions, decoded, img_w, img_h):
    """rpn_head_post + the decode launch of proposal NMS (shared regions [n,4]) in one kernel."""
    call("frcnn_rpn_head_post_decode", _p(head), ld, b, num_anchors_total, a_per_loc, _p(keep), n, _p(scores), _p(deltas), _p(regions),
         _p(decoded), float(img_w), float(img_h), _stream())


def clip_to_window(boxes, out, window):
    x0, y0, x1, y1 = [float(v) for v in window]
    call("frcnn_clip_to_window", _p(boxes), _p(out), boxes.numel() // 4, x0, y0, x1, y1, _stream())


def boxes_scale(inp, out, sx, sy):
    call("frcnn_boxes_scale", _p(inp), _p(out), inp.numel() // 4, sx, sy, _stream())


def decode_boxes(regions, deltas, out, b, r, c, img_w, img_h):
    call("frcnn_decode_boxes", _p(regions), 1 if regions.dim() == 3 else 0, _p(deltas), _p(out), b, r, c, float(img_w), float(img_h),
         _stream())


def encode_boxes(boxes, regions, out, b, r, c):
    call("frcnn_encode_boxes", _p(boxes), _p(regions), 1 if regions.dim() == 3 else 0, _p(out), b, r, c, _stream())


def boxes_divide(inp, out, w, h):
    call("frcnn_boxes_divide", _p(inp), _p(out), inp.numel() // 4, float(w), float(h), _stream())


def nms_workspace_bytes(b, n, c, max_per_class, max_total):
    return int(_lib.load().frcnn_nms_workspace_bytes(b, n, c, max_per_class, max_total))


def nms_combined(boxes, scores, b, n, q, c, score_stride, score_offset, max_per_class, max_total, iou_thr, score_thr, out_boxes,
                 out_scores, out_classes, out_valid, workspace):
    call("frcnn_nms_combined", _p(boxes), _p(scores), b, n, q, c, score_stride, score_offset, max_per_class, max_total, iou_thr,
         score_thr, _p(out_boxes), _p(out_scores), _p(out_classes), _p(out_valid), _p(workspace), workspace.numel() * workspace.element_size(),
         _stream())


def nms_combined_abs(boxes, scores, b, n, q, c, score_stride, score_offset, max_per_class, max_total, iou_thr, score_thr, out_boxes,
                     out_scores, out_classes, out_valid, workspace, out_boxes_abs, scale_x, scale_y):
    """nms_combined + out_boxes_abs = out_boxes * [scale_x, scale_y, scale_x, scale_y] in the same launch"""
    call("frcnn_nms_combined_abs", _p(boxes), _p(scores), b, n, q, c, score_stride, score_offset, max_per_class, max_total, iou_thr,
         score_thr, _p(out_boxes), _p(out_scores), _p(out_classes), _p(out_valid), _p(workspace), workspace.numel() * workspace.element_size(),
         _p(out_boxes_abs), float(scale_x), float(scale_y), _stream())


def roi_crop_pool_fwd(feat, rois, b, p, hf, wf, c, ps, ks, pooled, argmax):
    call("frcnn_roi_crop_pool_fwd", _p(feat), _p(rois), b, p, hf, wf, c, ps, ks, _p(pooled), _p(argmax), _stream())


def roi_crop_pool_bwd(gpooled, argmax, rois, rows, nrows, p, hf, wf, c, ps, ks, gfeat):
    call("frcnn_roi_crop_pool_bwd", _p(gpooled), _p(argmax), _p(rois), _p(rows), nrows, p, hf, wf, c, ps, ks, _p(gfeat), _stream())


def roi_crop_pool_bwd_bf16(gpooled, argmax, rois, rows, nrows, b, p, hf, wf, c, ps, ks, gfeat):
    call("frcnn_roi_crop_pool_bwd_bf16", _p(gpooled), _p(argmax), _p(rois), _p(rows), nrows, b, p, hf, wf, c, ps, ks, _p(gfeat), _stream())


def roi_crop_pool_bwd_bf16_add(gpooled, argmax, rois, rows, nrows, b, p, hf, wf, c, ps, ks, gfeat, red=None):
    """gfeat += the RoI-branch gradient (gfeat holds another consumer's gradient of the same map); red (bn_reduce_args): also the
    BatchNorm-backward sums of the layer gfeat arrives at"""
    call("frcnn_roi_crop_pool_bwd_bf16_add", _p(gpooled), _p(argmax), _p(rois), _p(rows), nrows, b, p, hf, wf, c, ps, ks, _p(gfeat),
         byref(red) if red is not None else None, _stream())


# ---------------------------------------------------------------- feature pyramid (BASELINE.json configs[4])
def upsample_add(top, ht, wt, lat, out, b, h, w, c):
    call("frcnn_upsample_add", _p(top), ht, wt, _p(lat), _p(out), b, h, w, c, _stream())


def upsample_add_bwd(g, h, w, gtop, b, ht, wt, c, accumulate=True):
    call("frcnn_upsample_add_bwd", _p(g), h, w, _p(gtop), b, ht, wt, c, 1 if accumulate else 0, _stream())


def subsample2(x, y, b, h, w, c):
    call("frcnn_subsample2", _p(x), _p(y), b, h, w, c, _stream())


def subsample2_bwd_add(gy, gx, b, h, w, c):
    call("frcnn_subsample2_bwd_add", _p(gy), _p(gx), b, h, w, c, _stream())


def roi_assign_levels(rois_rel, img_w, img_h, levels):
    call("frcnn_roi_assign_levels", _p(rois_rel), rois_rel.numel() // 4, float(img_w), float(img_h), _p(levels), _stream())


def roi_crop_pool_fwd_level(feat, rois, b, p, hf, wf, c, ps, ks, pooled, argmax, levels, level):
    call("frcnn_roi_crop_pool_fwd_level", _p(feat), _p(rois), b, p, hf, wf, c, ps, ks, _p(pooled), _p(argmax), _p(levels), level, _stream())


def roi_crop_pool_bwd_bf16_level(gpooled, argmax, rois, rows, nrows, b, p, hf, wf, c, ps, ks, gfeat, levels, level):
    call("frcnn_roi_crop_pool_bwd_bf16_level", _p(gpooled), _p(argmax), _p(rois), _p(rows), nrows, b, p, hf, wf, c, ps, ks, _p(gfeat), _p(levels),
         level, _stream())


def rpn_head_post_level(head, ld, b, num_anchors_level, a_per_loc, keep, n, scores, deltas, regions, decoded, img_w, img_h, n_total, offset):
    call("frcnn_rpn_head_post_level", _p(head), ld, b, num_anchors_level, a_per_loc, _p(keep), n, _p(scores), _p(deltas), _p(regions), _p(decoded),
         float(img_w), float(img_h), n_total, offset, _stream())


def rpn_head_grad_level(dlogits_s, ddeltas_s, indices, keep, b, s, num_anchors_level, a_per_loc, dhead, ld, offset, n):
    call("frcnn_rpn_head_grad_level", _p(dlogits_s), _p(ddeltas_s), _p(indices), _p(keep), b, s, num_anchors_level, a_per_loc, _p(dhead), ld, offset,
         n, _stream())


def rcnn_head_post(logits, ld, bias, r, nc1, scores, deltas):
    call("frcnn_rcnn_head_post", _p(logits), ld, _p(bias), r, nc1, _p(scores), _p(deltas), _stream())


def rcnn_head_post_decode(logits, ld, bias, r, nc1, scores, deltas, regions, decoded, img_w, img_h):
    """rcnn_head_post + the decode step of detection NMS (decode_boxes on the fresh deltas) in one launch"""
    call("frcnn_rcnn_head_post_decode", _p(logits), ld, _p(bias), r, nc1, _p(scores), _p(deltas), _p(regions), _p(decoded), float(img_w),
         float(img_h), _stream())


def assign_targets(regions, gt_labels, gt_boxes, b, r, g, c1g, objectness, img_w, img_h, fg_interval, bg_interval, target_labels,
                   target_boxes):
    call("frcnn_assign_targets", _p(regions), 1 if regions.dim() == 3 else 0, _p(gt_labels), _p(gt_boxes), b, r, g, c1g,
         1 if objectness else 0, float(img_w), float(img_h), float(fg_interval[0]), float(fg_interval[1]), float(bg_interval[0]),
         float(bg_interval[1]), _p(target_labels), _p(target_boxes), _stream())


def sample_indices(target_labels, b, r, c1, num_samples, fg_proportion, seed, step, stream_base, indices, workspace, status, image_base=0):
    call("frcnn_sample_indices", _p(target_labels), b, r, c1, num_samples, float(fg_proportion), ctypes.c_uint64(seed), _p(step),
         stream_base, _p(indices), _p(workspace), _p(status), int(image_base), _stream())


def losses(scores, deltas, target_labels, target_boxes, indices, b, r, c1, s, cls_scale, reg_scale, out_losses, dlogits_s=None,
           ddeltas_s=None):
    call("frcnn_losses", _p(scores), _p(deltas), _p(target_labels), _p(target_boxes), _p(indices), b, r, c1, s, float(cls_scale),
         float(reg_scale), _p(out_losses), _p(dlogits_s), _p(ddeltas_s), _stream())


def losses_head_grad(scores, deltas, target_labels, target_boxes, indices, b, r, c1, s, cls_scale, reg_scale, out_losses, dlogits_s,
                     ddeltas_s, dhead_s, ld, rows_out, bias_grad=None):
    """losses() + rcnn_head_grad() (+ the column sums of the gradient rows into bias_grad: colsum_bf16) in one launch."""
    call("frcnn_losses_head_grad", _p(scores), _p(deltas), _p(target_labels), _p(target_boxes), _p(indices), b, r, c1, s,
         float(cls_scale), float(reg_scale), _p(out_losses), _p(dlogits_s), _p(ddeltas_s), _p(dhead_s), ld, _p(rows_out), _p(bias_grad),
         _stream())


def losses_rpn_head_grad(scores, deltas, target_labels, target_boxes, indices, b, r, s, cls_scale, reg_scale, out_losses, dlogits_s,
                         ddeltas_s, keep, num_anchors_total, a_per_loc, dhead, ld):
    """losses() (objectness: C1 = 2) + rpn_head_grad() in one launch."""
    call("frcnn_losses_rpn_head_grad", _p(scores), _p(deltas), _p(target_labels), _p(target_boxes), _p(indices), b, r, s,
         float(cls_scale), float(reg_scale), _p(out_losses), _p(dlogits_s), _p(ddeltas_s), _p(keep), num_anchors_total, a_per_loc,
         _p(dhead), ld, _stream())


def rpn_head_grad(dlogits_s, ddeltas_s, indices, keep, b, s, num_anchors_total, a_per_loc, dhead, ld):
    call("frcnn_rpn_head_grad", _p(dlogits_s), _p(ddeltas_s), _p(indices), _p(keep), b, s, num_anchors_total, a_per_loc, _p(dhead), ld,
         _stream())


def rcnn_head_grad(dlogits_s, ddeltas_s, indices, b, r, c1, s, dhead_s, ld, rows_out):
    call("frcnn_rcnn_head_grad", _p(dlogits_s), _p(ddeltas_s), _p(indices), b, r, c1, s, _p(dhead_s), ld, _p(rows_out), _stream())
