"""ctypes binding of lib2dod_hip.so (C ABI: include/frcnn_hip.h).

The product path has NO fallback: if the HIP library is missing or a symbol does not resolve
this module raises, and every op raises on a non-zero status with frcnn_last_error().
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
# FRCNN_LIB: a kernel-development variant of the library (csrc/build.py: lib2dod_hip_sweep.so, lib2dod_hip_stamps.so)
LIB_PATH = os.path.join(_PKG, os.environ.get("FRCNN_LIB", "lib2dod_hip.so"))


class ConvDesc(Structure):
    """frcnn_conv_desc"""
    _fields_ = [(n, c_int) for n in (
        "n", "hi", "wi", "in_pix_stride", "cin", "kh", "kw", "stride", "pad_h", "pad_w",
        "ho", "wo", "cout", "out_h", "out_w", "out_scatter", "flags", "split_k")] + [("workspace", c_void_p), ("workspace_bytes", ctypes.c_size_t)]


class WgradItem(Structure):
    """struct frcnn_wgrad_item (include/frcnn_hip.h)."""
    _fields_ = [("desc", POINTER(ConvDesc)), ("x", c_void_p), ("dz", c_void_p), ("dw", c_void_p), ("dz_stride", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("x_scale", c_void_p), ("dz_scale", c_void_p)]


class Fp8Out(Structure):
    """frcnn_fp8_out: optional fp8 twin of a BatchNorm kernel's output"""
    _fields_ = [("out8", c_void_p), ("qscale", c_void_p), ("amax", c_void_p)]


class BnReduce(Structure):
    """frcnn_bn_reduce"""
    _fields_ = [("z", c_void_p), ("relu_mask", c_void_p), ("mean", c_void_p), ("invstd", c_void_p), ("partial", c_void_p)]


class BnIn(Structure):
    """frcnn_bn_in"""
    _fields_ = [("stats_partial", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("moving_mean", c_void_p), ("moving_var", c_void_p),
                ("momentum", c_float), ("eps", c_float), ("count", c_int64), ("act", c_void_p), ("relu_mask", c_void_p), ("mean", c_void_p),
                ("invstd", c_void_p)]


class SgdFused(Structure):
    """frcnn_sgd_fused"""
    _fields_ = [("decay_end", c_int64), ("l2", c_float), ("stem_begin", c_int64), ("stem_cout", c_int), ("stem_packed", c_void_p),
                ("arrive", c_void_p)]


ABI_VERSION = 7          # FRCNN_ABI_VERSION of include/frcnn_hip.h this table was written against (load() refuses any other library)

CONV_BIAS, CONV_RELU, CONV_OUT_F32, CONV_ADD_RES, CONV_STATS, CONV_SPLITK_ATOMIC, CONV_WGRAD_ACCUMULATE = 1, 2, 4, 8, 16, 32, 64
CONV_WGRAD_STEM_UNPACK = 128

P = c_void_p
_SIGNATURES = {
    # name: (restype, [argtypes])
    "frcnn_abi_version": (c_int, []),
    "frcnn_last_error": (c_char_p, []),
    "frcnn_source_hash": (c_char_p, []),
    "frcnn_last_conv_instantiation": (c_char_p, []),
    "frcnn_conv2d_workspace_bytes": (ctypes.c_size_t, [P]),
    "frcnn_conv2d_workspace_counter_bytes": (ctypes.c_size_t, [P]),
    "frcnn_conv2d_describe": (c_char_p, [POINTER(ConvDesc), c_int]),
    "frcnn_conv2d_wgrad_describe": (c_char_p, [POINTER(ConvDesc), c_int, P]),
    "frcnn_conv2d_stat_tiles": (c_int, [POINTER(ConvDesc)]),
    "frcnn_conv2d_fprop": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P]),
    "frcnn_conv2d_fprop_fp8": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P]),
    "frcnn_conv2d_describe_fp8": (c_char_p, [POINTER(ConvDesc)]),
    "frcnn_quantize_fp8": (c_int, [P, c_int64, P, P, P, c_int, P]),
    "frcnn_conv2d_dgrad_fp8": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, POINTER(BnReduce), P]),
    "frcnn_conv2d_describe_dgrad_fp8": (c_char_p, [POINTER(ConvDesc), c_int]),
    "frcnn_quantize_weights_fp8_batched": (c_int, [P, c_int, c_int64, P]),
    "frcnn_fp8_update_scales": (c_int, [P, P, P, c_int, c_float, P, P, P]),
    "frcnn_conv2d_dgrad_bnreduce": (c_int, [POINTER(ConvDesc), P, P, P, P, P, POINTER(BnReduce), P]),
    "frcnn_conv2d_fprop_bnin": (c_int, [POINTER(ConvDesc), P, P, P, P, P, POINTER(BnIn), P]),
    "frcnn_conv2d_bnin_supported": (c_int, [POINTER(ConvDesc)]),
    "frcnn_conv2d_wgrad": (c_int, [POINTER(ConvDesc), P, P, c_int, P, P, P]),
    "frcnn_conv2d_wgrad_fp8": (c_int, [POINTER(ConvDesc), P, P, c_int, P, P, P, P]),
    "frcnn_conv2d_wgrad_describe_fp8": (c_char_p, [POINTER(ConvDesc)]),
    "frcnn_wgrad_group_bytes": (c_size_t, []),
    "frcnn_conv2d_wgrad_group_plan": (c_int, [P, c_int, P, c_size_t]),
    "frcnn_conv2d_wgrad_grouped": (c_int, [P, P, P]),
    "frcnn_weights_transpose_flip": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "frcnn_weights_transpose_flip_batched": (c_int, [P, c_int, c_int64, P]),
    "frcnn_cast_f32_bf16": (c_int, [P, P, c_int64, P]),
    "frcnn_copy_bytes": (c_int, [P, P, c_int64, P]),
    "frcnn_copy_bytes_multi": (c_int, [P, P, P, c_int, P]),
    "frcnn_fill_zero_multi": (c_int, [P, c_int, c_int64, P]),
    "frcnn_stem_pack_weights": (c_int, [P, P, c_int, P]),
    "frcnn_stem_unpack_grad": (c_int, [P, P, c_int, P]),
    "frcnn_preprocess_u8_bgr_mean": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "frcnn_bn_finalize_train": (c_int, [P, c_int, c_int, c_int64, P, P, P, P, c_float, c_float, P, P, P, P, P]),
    "frcnn_bn_finalize_eval": (c_int, [c_int, P, P, P, P, c_float, P, P, P]),
    "frcnn_bn_apply": (c_int, [P, P, P, P, c_int, P, c_int64, c_int, P]),
    "frcnn_bn_train_apply": (c_int, [P, P, c_int, c_int64, P, P, P, P, c_float, c_float, P, c_int, P, P, P, P, c_int64, c_int, POINTER(Fp8Out), P]),
    "frcnn_bn_bwd_apply_fused": (c_int, [P, P, P, P, P, P, P, P, c_int, P, P, P, P, c_int64, c_int, c_int64, c_float, POINTER(Fp8Out), P]),
    "frcnn_bn_bwd_apply_fused_red2": (c_int, [P, P, P, P, P, P, P, c_int, P, P, P, c_int64, c_int, c_int64, c_float, POINTER(Fp8Out), POINTER(BnReduce), P]),
    "frcnn_bn_bwd_blocks": (c_int, [c_int64]),
    "frcnn_bn_bwd_reduce": (c_int, [P, P, P, P, P, P, P, c_int64, c_int, P]),
    "frcnn_bn_bwd_finalize": (c_int, [P, c_int, c_int, c_int64, P, P, P, P, P]),
    "frcnn_bn_bwd_apply": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int64, c_int, P]),
    "frcnn_relu_bwd": (c_int, [P, P, P, c_int64, P]),
    "frcnn_colsum_bf16": (c_int, [P, c_int64, c_int, c_int, P, P]),
    "frcnn_bn_train_apply_dual": (c_int, [P] * 16 + [c_int, c_int64, c_float, c_float, c_int, P, P, c_int64, c_int, POINTER(Fp8Out), P]),
    "frcnn_bn_train_apply_maxpool": (c_int, [P, P, c_int, c_int64, P, P, P, P, c_float, c_float, P, P, P, P, P, c_int, c_int, c_int, c_int,
                                             c_int, c_int, P]),
    "frcnn_maxpool3x3s2_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "frcnn_maxpool3x3s2_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "frcnn_sgd_momentum": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, P, P, P, c_int, P]),
    "frcnn_step_increment": (c_int, [P, P]),
    "frcnn_maxpool3x3s2_bwd_bnreduce": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(BnReduce), P]),
    "frcnn_cast_colsum": (c_int, [P, P, c_int64, c_int, P, P]),
    "frcnn_relu_bwd_colsum": (c_int, [P, P, P, c_int64, c_int, P, P]),
    "frcnn_sgd_momentum_fused": (c_int, [P, P, P, P, c_int64, c_float, c_float, P, P, P, c_int, POINTER(SgdFused), P]),
    "frcnn_anchors_generate": (c_int, [P, c_int, c_int, POINTER(c_float), c_int, POINTER(c_float), c_int,
                                       c_float, c_float, c_float, c_float, P]),
    "frcnn_rpn_head_post": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, P, P]),
    "frcnn_rpn_head_post_decode": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, c_float, P]),
    "frcnn_clip_to_window": (c_int, [P, P, c_int64, c_float, c_float, c_float, c_float, P]),
    "frcnn_decode_boxes": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_float, c_float, P]),
    "frcnn_encode_boxes": (c_int, [P, P, c_int, P, c_int, c_int, c_int, P]),
    "frcnn_boxes_divide": (c_int, [P, P, c_int64, c_float, c_float, P]),
    "frcnn_nms_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "frcnn_nms_combined": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                   P, P, P, P, P, c_size_t, P]),
    "frcnn_nms_combined_abs": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                   P, P, P, P, P, c_size_t, P, c_float, c_float, P]),
    "frcnn_roi_crop_pool_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "frcnn_roi_crop_pool_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "frcnn_roi_crop_pool_bwd_bf16": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "frcnn_roi_crop_pool_bwd_bf16_add": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, POINTER(BnReduce), P]),
    "frcnn_rcnn_head_post": (c_int, [P, c_int, P, c_int, c_int, P, P, P]),
    "frcnn_boxes_scale": (c_int, [P, P, c_int64, c_float, c_float, P]),
    "frcnn_assign_targets": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float,
                                     c_float, c_float, c_float, P, P, P]),
    "frcnn_sample_indices": (c_int, [P, c_int, c_int, c_int, c_int, c_float, c_uint64, P, c_int, P, P, P, c_int, P]),
    "frcnn_losses": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P, P, P, P]),
    "frcnn_losses_rpn_head_grad": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P, P, P, c_int, c_int, P, c_int, P]),
    "frcnn_losses_head_grad": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P, P, P, P, c_int, P, P, P]),
    "frcnn_rcnn_head_post_decode": (c_int, [P, c_int, P, c_int, c_int, P, P, P, P, c_float, c_float, P]),
    "frcnn_rpn_head_grad": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_int, P]),
    "frcnn_rcnn_head_grad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, P]),
    "frcnn_upsample_add": (c_int, [P, c_int, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "frcnn_upsample_add_bwd": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "frcnn_subsample2": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "frcnn_subsample2_bwd_add": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "frcnn_roi_assign_levels": (c_int, [P, c_int64, c_float, c_float, P, P]),
    "frcnn_roi_crop_pool_fwd_level": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P]),
    "frcnn_roi_crop_pool_bwd_bf16_level": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P]),
    "frcnn_rpn_head_post_level": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, c_float, c_int, c_int, P]),
    "frcnn_rpn_head_grad_level": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, P]),
    "frcnn_crc32c": (ctypes.c_uint32, [ctypes.c_uint32, P, c_size_t]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the library (after torch, so that both share one libamdhip64) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "%s not found: build it with `python 2d_object_detection_amd/csrc/build.py` "
            "(there is no CPU fallback)" % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads torch's bundled libamdhip64.so.7 first; same SONAME is then reused)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    try:
        lib.frcnn_abi_version.restype = c_int
        found = int(lib.frcnn_abi_version())
    except AttributeError as e:
        raise HipLibraryError("symbol frcnn_abi_version missing from %s" % LIB_PATH) from e
    if found != ABI_VERSION:
        raise HipLibraryError("%s implements ABI version %d, this binding needs %d (struct layouts / signatures differ): rebuild it with "
                              "`python 2d_object_detection_amd/csrc/build.py`" % (LIB_PATH, found, ABI_VERSION))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError("symbol %s missing from %s" % (name, LIB_PATH)) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().frcnn_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what or "frcnn call", rc, msg.decode() if msg else "?"))


def call(name, *args):
    """Call an int-returning entry point and raise on error."""
    rc = getattr(load(), name)(*args)
    check(rc, name)
