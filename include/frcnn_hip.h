/* frcnn_hip.h -- C ABI of the MI355X-native (gfx950) Faster-RCNN hot path.
 *
 * The reference (antoineBarbez/2D_object_detection) is 100% Python on TensorFlow and has no
 * FFI of its own; the boundary it offers is the Python call surface of
 * models/faster_rcnn.py, models/feature_extractor.py, models/detectors/{rpn,fast_rcnn}_detector.py and
 * utils/post_processing.py.  Each entry point below names the reference call site (file:line)
 * whose TensorFlow op(s) it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer unless marked "host";
 *   - `stream` is a hipStream_t passed as void*; nothing here synchronises the host,
 *     allocates, or keeps global mutable state (graph-capture safe, re-entrant);
 *   - activations are NHWC, bf16 (uint16 storage) unless noted; parameters are fp32 masters
 *     with bf16 working copies; boxes are [x_min, y_min, x_max, y_max] fp32;
 *   - return value: 0 = ok, <0 = error (see frcnn_last_error(), thread-local).
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* frcnn_stream_t;
typedef uint16_t frcnn_bf16;
typedef uint8_t frcnn_fp8;     /* OCP e4m3fn (gfx950's fp8: 4 exponent bits, bias 7, max 448, no infinities) */

#define FRCNN_OK 0
#define FRCNN_EINVAL (-1)      /* bad argument / unsupported shape */
#define FRCNN_ELAUNCH (-2)     /* kernel launch failed */

/* Version of this header's structs and signatures.  Bumped whenever a struct grows or a signature changes (2: frcnn_conv_desc
 * gained workspace / workspace_bytes, frcnn_bn_bwd_apply_fused gained count / param_grad_scale; 3: the fp8 entry points; 5: frcnn_fp8_update_scales gained limit / status,
 * frcnn_losses_head_grad gained bias_grad, FRCNN_CONV_WGRAD_ACCUMULATE / _STEM_UNPACK, the fused launches of round 4; 6: frcnn_conv2d_fprop_bnin; 7: frcnn_conv2d_fprop_bnin on the
 * patch-resident 3x3 and 1x1 tile forms, frcnn_sgd_fused.arrive may be NULL).  A
 * binding must compare frcnn_abi_version() with the FRCNN_ABI_VERSION it was written against and refuse any other library:
 * an older build would read the descriptor past the caller's struct. */
#define FRCNN_ABI_VERSION 7
int frcnn_abi_version(void);
const char* frcnn_last_error(void);
/* sha1 (12 hex digits) over the kernel sources and this header the library was built from, or "unknown" (csrc/build.py passes it):
 * the build entry point rebuilds a library whose hash is not the tree's, and bench.py stamps its rocprof profiles with the same hash */
const char* frcnn_source_hash(void);

/* ------------------------------------------------------------------ dense tensor ops */

/* Implicit-GEMM convolution, bf16 MFMA, fp32 accumulate.
 * Replaces every Conv2D of the Keras ResNet50 graph (models/feature_extractor.py:8-10), the RPN
 * convs (models/detectors/rpn_detector.py:26-58,79-86), the Dense heads
 * (fast_rcnn_detector.py:25-41,62-65, as 1x1 "convs" over RoI rows) and -- with transposed,
 * tap-flipped weights -- their data gradients (tape.gradient, models/faster_rcnn.py:103).
 *
 * out[(n,oy,ox), co] = sum_{kh,kw,ci} x[n, oy*stride-pad_h+kh, ox*stride-pad_w+kw, ci] * w[co,kh,kw,ci]
 * x rows outside the input are zero.  `cin` must be a multiple of 32 (64 unless kh*kw*cin/32 is odd),
 * `cout` a multiple of 8.  in_pix_stride is the element distance between input pixels (== cin
 * for ordinary NHWC; the stem reads its 4-channel padded image with in_pix_stride 4, cin 32).
 * The output pixel (n,oy,ox) is stored at row ((n*out_h + oy*out_scatter)*out_w + ox*out_scatter)
 * (out_scatter 2 scatters a stride-2 1x1 data gradient into the larger input grid). */
#define FRCNN_CONV_BIAS       1   /* add bias[cout] */
#define FRCNN_CONV_RELU       2   /* max(.,0) */
#define FRCNN_CONV_OUT_F32    4   /* y is fp32 instead of bf16 */
#define FRCNN_CONV_ADD_RES    8   /* y = conv + res (res bf16, same addressing as y; may alias y) */
#define FRCNN_CONV_STATS      16  /* accumulate (float atomics) the column sum / sum-of-squares of the bf16-rounded
                                     output into stats_partial, f64 [FRCNN_STAT_SLOTS][2][cout], which must be pre-zeroed
                                     (f64: the arrival order of the atomics does not show in the statistics) */
#define FRCNN_STAT_SLOTS      16
#define FRCNN_CONV_SPLITK_ATOMIC 32 /* y (fp32, pre-zeroed) accumulated with atomics over split_k K-slices */
#define FRCNN_CONV_WGRAD_ACCUMULATE 64 /* frcnn_conv2d_wgrad / _wgrad_fp8 only (the forward entry points ignore it): dw is shared with
                                     other launches (e.g. the RPN weights over the levels of a feature pyramid), so this launch must ADD
                                     with float atomics even when it has a single pixel split -- without the flag a one-split launch
                                     stores plainly on the assumption that it is the only writer of a zeroed dw */
#define FRCNN_CONV_WGRAD_STEM_UNPACK 128 /* frcnn_conv2d_wgrad only, on the stem's packed descriptor (kh 7, kw 1, cin 32 = 8 pixels x 4 channels,
                                     in_pix_stride 4): dw is the UN-padded Keras kernel gradient [cout][7][7][3] (models/feature_extractor.py:8,
                                     conv1_conv) -- the 7 x 3 real values of every tap row are stored there directly, the padding taps dropped
                                     (replaces a packed scratch gradient + frcnn_stem_unpack_grad) */
typedef struct {
    int n, hi, wi, in_pix_stride, cin;
    int kh, kw, stride, pad_h, pad_w;
    int ho, wo, cout;
    int out_h, out_w, out_scatter;
    int flags, split_k;
    /* Optional scratch for layers with fewer output tiles than CUs and a long K (frcnn_conv2d_workspace_bytes(d) > 0): the K range of
     * a tile is then split over two workgroups that meet in this buffer (fp32 partial tiles + one arrival counter per tile).
     * Device memory, 16-byte aligned, ZEROED once by the caller (the kernel leaves the counters zero; see
     * frcnn_conv2d_workspace_counter_bytes for callers that re-zero them per launch); a descriptor in flight on two
     * streams at once needs two workspaces.  NULL: the one-workgroup-per-tile form. */
    void* workspace;
    size_t workspace_bytes;
} frcnn_conv_desc;
/* Data gradient fused with the BatchNorm-backward REDUCE of the layer that consumes it: gx = conv(dz, w_t) [+ res] is the
 * gradient g arriving at a BatchNorm layer whose raw input was z; while storing gx the kernel accumulates that layer's
 * partial[slot][0][c] += sum g*m and partial[slot][1][c] += sum g*m*xhat (m = relu_mask bit or 1, xhat = (z-mean)*invstd),
 * i.e. exactly what frcnn_bn_bwd_reduce(gout = gx, ...) would add -- one launch and one read of gx, z and the mask less per
 * layer (reference: the tf.GradientTape backward of keras BatchNormalization + Conv2D, models/faster_rcnn.py:95-107).
 * res_mask (optional, with FRCNN_CONV_ADD_RES): bit mask [M][C/8] applied to the residual before the add -- the residual
 * branch of a ResNet block passes the block-output gradient and the block's ReLU mask instead of a materialised product. */
typedef struct frcnn_bn_reduce {
    const frcnn_bf16* z;          /* [M][Cout of this conv = channels of the BN layer] */
    const uint8_t* relu_mask;     /* [M][C/8] or NULL (no ReLU) */
    const float* mean;            /* [C] */
    const float* invstd;          /* [C] */
    float* partial;               /* [FRCNN_STAT_SLOTS][2][C], pre-zeroed, accumulated with float atomics */
} frcnn_bn_reduce;
int frcnn_conv2d_dgrad_bnreduce(const frcnn_conv_desc* d, const frcnn_bf16* dz, const frcnn_bf16* w_t, const frcnn_bf16* res,
                                const uint8_t* res_mask, frcnn_bf16* gx, const frcnn_bn_reduce* red, frcnn_stream_t stream);
/* Forward convolution that applies the training-mode BatchNorm + ReLU of its INPUT layer itself (round 4; == frcnn_bn_train_apply(z_in ...
 * -> act, relu_mask, mean, invstd, moving statistics) followed by frcnn_conv2d_fprop(act ...): the same bits in act, relu_mask, mean,
 * invstd, the moving statistics and y; the statistics of y in a different summation order).  z_in: the raw [M][cin] output of the previous
 * convolution, whose forward statistics lie in bn->stats_partial.  Only for the shapes frcnn_conv2d_bnin_supported(d) accepts (3x3 / stride
 * 1 / pad 1, 64 input channels, the sizes that run on the weights-resident kernel: conv2's layers at the benchmark's batch); d->flags:
 * BIAS, STATS.  The reference applies the BatchNorm as its own Keras layer (tf.keras.applications ResNet50, models/feature_extractor.py:4-11). */
typedef struct frcnn_bn_in {
    const double* stats_partial;  /* [FRCNN_STAT_SLOTS][2][cin] f64: sum, sum of squares of z_in */
    const float* gamma;           /* [cin] */
    const float* beta;
    float* moving_mean;           /* [cin], updated as frcnn_bn_train_apply does */
    float* moving_var;
    float momentum, eps;
    int64_t count;                /* pixels the statistics were taken over (all ranks) */
    frcnn_bf16* act;              /* out [M][cin]: ReLU(BN(z_in)) -- the weight gradient's x operand */
    uint8_t* relu_mask;           /* out [M][cin / 8] */
    float* mean;                  /* out [cin] */
    float* invstd;
} frcnn_bn_in;
int frcnn_conv2d_fprop_bnin(const frcnn_conv_desc* d, const frcnn_bf16* z_in, const frcnn_bf16* w, const float* bias, frcnn_bf16* y,
                            double* stats_partial, const frcnn_bn_in* bn, frcnn_stream_t stream);
int frcnn_conv2d_bnin_supported(const frcnn_conv_desc* d);   /* 1 / 0; host logic, no device */
/* Diagnostics: name, template arguments and grid of the MFMA conv kernel(s) the calling thread launched last through
 * frcnn_conv2d_fprop / _dgrad_bnreduce / _wgrad / _wgrad_grouped (thread-local; "" before the first launch).  The parity tests
 * assert with it that a shape really dispatched to the instantiation they mean to cover. */
const char* frcnn_last_conv_instantiation(void);
/* The same string for the launch frcnn_conv2d_fprop (with_bn_reduce = 0) or frcnn_conv2d_dgrad_bnreduce (1) WOULD make for
 * this descriptor -- host logic only, nothing is launched and no device is needed; NULL (see frcnn_last_error) when the
 * descriptor is not supported. */
const char* frcnn_conv2d_describe(const frcnn_conv_desc* d, int with_bn_reduce);
/* likewise for frcnn_conv2d_wgrad (group_table_host == NULL) or for frcnn_conv2d_wgrad_grouped on a planned host table */
const char* frcnn_conv2d_wgrad_describe(const frcnn_conv_desc* d, int with_row_index, const void* group_table_host);
/* rows of the stats_partial buffer [rows][2][cout] (== FRCNN_STAT_SLOTS) */
/* Bytes of frcnn_conv_desc.workspace with which frcnn_conv2d_fprop / frcnn_conv2d_dgrad_bnreduce run this descriptor in the split-K
 * fix-up form; 0 when the dispatcher would not use it (the answer does not depend on d->workspace).  No device needed.  The answer is
 * the tile kernel's, which serves the bf16 AND the fp8 entry points of one descriptor: a 3x3 layer that the bf16 entry points run on the
 * patch-resident kernel (frcnn_conv2d_describe says "conv3x3_patch") simply ignores the workspace. */
size_t frcnn_conv2d_workspace_bytes(const frcnn_conv_desc* d);
/* The arrival counters are the LAST frcnn_conv2d_workspace_counter_bytes(d) bytes of that workspace (a multiple of 16).  The kernel
 * leaves them zero after a completed launch; a caller that may abort launches (or replays captured graphs after an error) zeroes
 * this tail before every launch -- cheaply, together with its other accumulation targets (frcnn_fill_zero_multi) -- so that a
 * counter left at 1 cannot make both halves of a later launch believe they arrived first. */
size_t frcnn_conv2d_workspace_counter_bytes(const frcnn_conv_desc* d);
int frcnn_conv2d_stat_tiles(const frcnn_conv_desc* d);
int frcnn_conv2d_fprop(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* w, const float* bias,
                       const frcnn_bf16* res, void* y, double* stats_partial, frcnn_stream_t stream);

/* ------------------------------------------------------------------ fp8 (e4m3) MFMA convolution path
 * BASELINE.json configs[4] ("fp8 weights ... CDNA4 fp8 MFMA conv path").  The reference is fp32 throughout
 * (models/feature_extractor.py:5-9); this is the same Conv2D contraction with both operands stored as OCP e4m3 bytes and
 * multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulation, unit block scales), twice the bf16 matrix rate and half
 * the operand bytes per FLOP:
 *   x ~= x8 * x_scale[0]   (per tensor; device scalar, so that a captured graph follows a scale that changes from step to step)
 *   w[co] ~= w8[co] * w_scale[co]   (per output channel)
 *   y[(n,oy,ox), co] = bf16( x_scale * w_scale[co] * sum_k x8 * w8  + bias[co] )        (+ FRCNN_CONV_STATS as the bf16 form)
 * Same descriptor as frcnn_conv2d_fprop with cin and in_pix_stride multiples of 128; flags BIAS / RELU / STATS; bf16 output.  The
 * sum is EXACT in fp32 terms (products of two e4m3 values are exact, accumulation is fp32): against an fp32 convolution of the
 * dequantised operands only the accumulation order and the bf16 rounding of y differ. */
int frcnn_conv2d_fprop_fp8(const frcnn_conv_desc* d, const frcnn_fp8* x8, const frcnn_fp8* w8, const float* x_scale,
                           const float* w_scale, const float* bias, frcnn_bf16* y, double* stats_partial, frcnn_stream_t stream);
/* the instantiation frcnn_conv2d_fprop_fp8 would launch (host logic, as frcnn_conv2d_describe) */
const char* frcnn_conv2d_describe_fp8(const frcnn_conv_desc* d);
/* Data gradient on fp8 operands (the backward of the layers above: tf.GradientTape over Conv2D, models/faster_rcnn.py:103):
 * gx = bf16(dz_scale * w_scale[ci] * conv(dz8, w_t8)) [+ res (* res_mask bits)], optionally fused with the BatchNorm-backward
 * reduce of the layer that consumes gx (red != NULL; semantics of frcnn_conv2d_dgrad_bnreduce).  dz8: the incoming gradient as
 * OCP e5m2 bytes (5 exponent bits: the gradient format of Micikevicius et al., "FP8 formats for deep learning", 2022) with a
 * per-tensor scale; w_t8: the tap-flipped transposed weights [cin][kh][kw][cout] as e4m3 bytes with one scale per row (per input
 * channel of the forward layer).  Descriptor as for frcnn_conv2d_dgrad_bnreduce with its cin (the forward layer's cout) a
 * multiple of 128. */
int frcnn_conv2d_dgrad_fp8(const frcnn_conv_desc* d, const frcnn_fp8* dz8, const frcnn_fp8* w_t8, const float* dz_scale,
                           const float* w_scale, const frcnn_bf16* res, const uint8_t* res_mask, frcnn_bf16* gx,
                           const struct frcnn_bn_reduce* red, frcnn_stream_t stream);
const char* frcnn_conv2d_describe_dgrad_fp8(const frcnn_conv_desc* d, int with_bn_reduce);
/* Quantisers.  out8[i] = e4m3_rne(clamp(x[i] * qscale[0], -448, 448)); amax (optional, FRCNN_FP8_AMAX_SLOTS device floats,
 * pre-zeroed per step): max |x[i]| as fp32, one slot per wave of the launch (slot = wave index mod FRCNN_FP8_AMAX_SLOTS; an atomic max,
 * but practically uncontended: atomics of many waves on one word serialise at ~0.1 us each -- 64 slots still cost a BatchNorm
 * launch 9 us, measured) -- the maximum over the slots is the input of the delayed scaling rule below.  n a multiple of 8. */
#define FRCNN_FP8_AMAX_SLOTS 8192
int frcnn_quantize_fp8(const frcnn_bf16* x, int64_t n, const float* qscale, frcnn_fp8* out8, float* amax, int e5m2 /* 0: e4m3, 1: e5m2 (clamp 57344) */,
                       frcnn_stream_t stream);
/* Weights, several layers in one launch: table int64 [n][8] = {source (rows of K values, row-major), fp8 destination, float
 * scale[rows] destination, rows, K, first workgroup, source type (0: fp32 masters, 1: bf16 -- the tap-flipped transposes the
 * data gradients read), 0}; one workgroup per row: scale = max|w| / 448 (1 for an all-zero row),
 * w8 = e4m3_rne(w * (1 / scale)), all in fp32.  Serves the forward weights [cout][kh*kw*cin] and, given transposed masters, any other row layout. */
int frcnn_quantize_weights_fp8_batched(const int64_t* table, int n, int64_t total_rows, frcnn_stream_t stream);
/* Delayed scaling (one amax of history): for i < n: a = max over amax[i][0..FRCNN_FP8_AMAX_SLOTS) (this step's maximum); if a > 0:
 * scale[i] = margin * a / 448, qscale[i] = 1 / scale[i]; a == 0 (tensor not produced this step) leaves both unchanged.
 * The rule has one step of history, so a step whose tensor outgrows margin x last step's amax CLAMPS silently in the quantiser; this
 * entry point is where that becomes visible: limit (optional, device float [n]: the clamp of tensor i's format, 448 for e4m3, 57344
 * for e5m2; NULL = 448 everywhere) and status (optional, device int32 [2], never reset here): status[0] += 1 for every tensor
 * whose amax of THIS step exceeded limit[i] * scale[i] (its twin held clamped values); status[1] += 1 for every tensor whose amax is
 * Inf / NaN -- such a tensor keeps its previous scale (scale = Inf would turn every later dequantisation into 0 * Inf = NaN). */
int frcnn_fp8_update_scales(const float* amax, float* scale, float* qscale, int n, float margin, const float* limit, int32_t* status,
                            frcnn_stream_t stream);
/* Optional fp8 twin of a BatchNorm kernel's output (frcnn_bn_train_apply / _dual): the kernel that writes the bf16 activation also
 * writes out8 = e4m3(clamp(bf16 value * qscale[0])) and folds max|value| into the amax slots -- the next convolution reads 1 byte per
 * element instead of 2 and no separate quantise pass exists. */
typedef struct frcnn_fp8_out {
    frcnn_fp8* out8;        /* [M][C] */
    const float* qscale;    /* device scalar: 1 / dequantisation scale */
    float* amax;            /* FRCNN_FP8_AMAX_SLOTS device floats or NULL */
} frcnn_fp8_out;

/* Weight gradient: dw[co,kh,kw,ci] (fp32, accumulated with atomics into a pre-zeroed buffer) =
 * sum_pixels dz[(n,oy,ox), co] * x[n, oy*stride-pad_h+kh, ox*stride-pad_w+kw, ci].
 * Same descriptor as the forward conv (out_* / flags ignored).  `row_index` (optional int32[M])
 * replaces the im2col row of output pixel p by input row row_index[p] (1x1 only; used for the Dense
 * heads, whose gradient only involves the sampled RoI rows).  dz row stride is dz_stride elements. */
int frcnn_conv2d_wgrad(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* dz, int dz_stride,
                       const int32_t* row_index, float* dw, frcnn_stream_t stream);

/* The same from fp8 operands (round 3: the third fp8 convolution of a training step, after fprop and dgrad): x8 is the e4m3
 * twin of the layer's input, dz8 the e5m2 twin of the output gradient, one byte per element (dz_stride, in_pix_stride in
 * elements = bytes; channels multiples of 64), x_scale / dz_scale the tensors' dequantisation scales (device scalars).  The
 * pixel contraction runs through v_mfma_scale_f32_16x16x128_f8f6f4 on 128-pixel slices that ds_read_b64_tr_b8 transposes out
 * of the pixel-major LDS image; dw += x_scale * dz_scale * sum_p dz8[p, co] * x8[im2col(p, tap), ci] in fp32. */
int frcnn_conv2d_wgrad_fp8(const frcnn_conv_desc* d, const frcnn_fp8* x8, const frcnn_fp8* dz8, int dz_stride,
                           const float* x_scale, const float* dz_scale, float* dw, frcnn_stream_t stream);
const char* frcnn_conv2d_wgrad_describe_fp8(const frcnn_conv_desc* d);

/* Grouped weight gradients: several layers of one backward stage in ONE launch per addressing mode, with one pixel split
 * chosen for the whole group (just enough workgroups for ~2 per CU; none -- no float atomics, every dw element stored
 * once -- when the group's 64x64 tiles fill the chip, e.g. conv4 at 375x1242: 19 layers, 1728 tiles).
 * frcnn_conv2d_wgrad_group_plan fills a HOST table (frcnn_wgrad_group_bytes() bytes) from n items (cin, cout multiples of
 * 64; per item the semantics of frcnn_conv2d_wgrad: dw pre-zeroed); the caller uploads the table once and launches with
 * both copies. */
typedef struct frcnn_wgrad_item {
    const frcnn_conv_desc* desc;
    const void* x;              /* bf16, or fp8 e4m3 when the scales below are set */
    const void* dz;             /* bf16, or fp8 e5m2 */
    float* dw;
    int32_t dz_stride;
    int32_t reserved;
    const float* x_scale;       /* device scalars: dequantisation scales of the fp8 operands (frcnn_conv2d_wgrad_fp8), */
    const float* dz_scale;      /* or both NULL for bf16 operands */
} frcnn_wgrad_item;
size_t frcnn_wgrad_group_bytes(void);
int frcnn_conv2d_wgrad_group_plan(const frcnn_wgrad_item* items, int n, void* table_host, size_t table_bytes);
int frcnn_conv2d_wgrad_grouped(const void* table_host, const void* table_dev, frcnn_stream_t stream);

/* w_t[ci][KH-1-kh][KW-1-kw][co] (bf16) = w[co][kh][kw][ci] (fp32 master) : data-gradient weights. */
int frcnn_weights_transpose_flip(const float* w, frcnn_bf16* w_t, int cout, int kh, int kw, int cin,
                                 frcnn_stream_t stream);
/* the same for n layers in ONE launch: table (device, int64[n][8]) rows = {w ptr, w_t ptr, cout, kh, kw, cin,
 * first flat output index (prefix sum of the layer sizes), 0}; total = sum of the layer sizes. */
int frcnn_weights_transpose_flip_batched(const int64_t* table, int n, int64_t total, frcnn_stream_t stream);
/* plain fp32 -> bf16 cast of n elements */
int frcnn_cast_f32_bf16(const float* src, frcnn_bf16* dst, int64_t n, frcnn_stream_t stream);
/* device-to-device copy of nbytes (16-byte aligned pointers) at HBM speed: feeds a plan's static input buffers */
int frcnn_copy_bytes(const void* src, void* dst, int64_t nbytes, frcnn_stream_t stream);
/* n (1..4) such copies in one launch: srcs / dsts / nbytes are HOST arrays of n device pointers (16-byte aligned) and sizes.
 * The train step feeds its three inputs (reference faster_rcnn.py:83 train_step(images, gt_labels, gt_boxes)) this way. */
int frcnn_copy_bytes_multi(const void* const* srcs, void* const* dsts, const int64_t* nbytes, int n, frcnn_stream_t stream);
/* zero n device buffers (16-byte aligned, sizes multiples of 16 bytes) in one launch.  table (device, int64 [n + 1][2]): row i =
 * {pointer, first 16-byte chunk of buffer i in the concatenation of all buffers}, row n = {0, total_chunks}.  Replaces the
 * per-buffer zero fills in front of the atomically accumulated buffers of a train step (flat gradient, BatchNorm partial
 * sums, scatter / split-K targets). */
int frcnn_fill_zero_multi(const int64_t* table, int n, int64_t total_chunks, frcnn_stream_t stream);
/* stem weights: master [64][7][7][3] fp32 <-> padded GEMM form [64][7][8][4]
 * (pack: -> bf16; unpack_grad: padded fp32 grad -> compact fp32 grad, overwriting). */
int frcnn_stem_pack_weights(const float* w, frcnn_bf16* w_packed, int cout, frcnn_stream_t stream);
int frcnn_stem_unpack_grad(const float* dw_packed, float* dw, int cout, frcnn_stream_t stream);

/* uint8 RGB [B,H,W,3] -> bf16 BGR minus caffe mean, zero-padded to [B,Hp,Wp,4] with the image at
 * offset (pad,pad); channel 3 = 0.  models/feature_extractor.py:6-7 (+ the ZeroPadding2D(3)
 * of Keras ResNet50). */
int frcnn_preprocess_u8_bgr_mean(const uint8_t* images, frcnn_bf16* out, int b, int h, int w, int hp, int wp,
                                 int pad, frcnn_stream_t stream);

/* BatchNormalization (Keras ResNet50 BN layers, training and inference mode).
 * finalize_train: reduce conv stats partials [tiles][2][c] -> mean/invstd, fused scale/shift
 * (scale = gamma*invstd, shift = beta - mean*scale), and update the moving averages
 * (moving = moving*momentum + batch*(1-momentum), variance unbiased as FusedBatchNormV3). */
int frcnn_bn_finalize_train(const double* stats_partial, int tiles, int c, int64_t count, const float* gamma,
                            const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                            float* scale, float* shift, float* mean, float* invstd, frcnn_stream_t stream);
int frcnn_bn_finalize_eval(int c, const float* gamma, const float* beta, const float* moving_mean,
                           const float* moving_var, float eps, float* scale, float* shift, frcnn_stream_t stream);
/* out = [relu]( z*scale + shift [+ res] )  over m rows of c channels (c % 8 == 0) */
int frcnn_bn_apply(const frcnn_bf16* z, const float* scale, const float* shift, const frcnn_bf16* res, int relu,
                   frcnn_bf16* out, int64_t m, int c, frcnn_stream_t stream);
/* backward: g = gout * (act > 0) if act != NULL else gout;  xhat = (z-mean)*invstd
 * reduce  : partial[slot][0][c] += sum g, partial[slot][1][c] += sum g*xhat over row blocks (float atomics into a
 *           PRE-ZEROED [frcnn_bn_bwd_blocks(m)][2][c] buffer; the row count is FRCNN_STAT_SLOTS)
 * finalize: dgamma = sum g*xhat, dbeta = sum g, c1 = dbeta/m, c2 = dgamma/m
 * apply   : dz = gamma*invstd*(g - c1 - xhat*c2);  gpre (optional) = g */
int frcnn_bn_bwd_blocks(int64_t m);
int frcnn_bn_bwd_reduce(const frcnn_bf16* gout, const frcnn_bf16* act, const uint8_t* relu_mask, const frcnn_bf16* z, const float* mean,
                        const float* invstd, float* partial, int64_t m, int c, frcnn_stream_t stream);
int frcnn_bn_bwd_finalize(const float* partial, int blocks, int c, int64_t m, float* dgamma, float* dbeta,
                          float* c1, float* c2, frcnn_stream_t stream);
int frcnn_bn_bwd_apply(const frcnn_bf16* gout, const frcnn_bf16* act, const frcnn_bf16* z, const float* mean,
                       const float* invstd, const float* gamma, const float* c1, const float* c2,
                       frcnn_bf16* dz, frcnn_bf16* gpre, int64_t m, int c, frcnn_stream_t stream);
/* Fused forms used by the training step (reference: keras BatchNormalization(training=True) inside
 * models/feature_extractor.py's ResNet50; same arithmetic as finalize_train + apply / bwd_finalize + bwd_apply, one launch
 * each: every workgroup reduces the partial sums of its own 64 channels).
 * train_apply    : out = [relu](z*scale + shift [+ res]); writes mean / invstd, updates the moving statistics; optionally
 *                  writes relu_mask [m][c/8]: bit e of byte (row, c/8) = (out[row][8*(c/8)+e] > 0).  The backward kernels take
 *                  EITHER the activation tensor (act) OR that bit mask (relu_mask) as the ReLU mask, or neither (no ReLU)
 * bwd_apply_fused: dz = gamma*invstd*(g - c1 - xhat*c2), gpre (optional) = g; writes dgamma / dbeta.
 * Synchronised BatchNorm over data-parallel ranks (the reference is ONE device whose BatchNorm sees the whole batch,
 * models/faster_rcnn.py:50): the caller SUM-all-reduces stats_partial (forward) / partial (backward) over the ranks between
 * the kernel that accumulates them and these kernels, and passes count = m * world (rows of the GLOBAL batch; 0 = m) and
 * param_grad_scale = 1 / world (each rank publishes its share of dgamma / dbeta, which the gradient all-reduce sums again;
 * 1 on a single rank). */
int frcnn_bn_train_apply(const frcnn_bf16* z, const double* stats_partial, int slots, int64_t count, const float* gamma,
                         const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                         const frcnn_bf16* res, int relu, frcnn_bf16* out, uint8_t* relu_mask, float* mean, float* invstd,
                         int64_t m, int c, const struct frcnn_fp8_out* f8 /* NULL: no fp8 twin; with a twin `out` may be NULL (the bf16
                         activation is then not stored: ReLU bit mask and twin only) */, frcnn_stream_t stream);
int frcnn_bn_bwd_apply_fused(const frcnn_bf16* gout, const frcnn_bf16* act, const uint8_t* relu_mask, const frcnn_bf16* z,
                             const float* mean, const float* invstd, const float* gamma, const float* partial, int slots,
                             float* dgamma, float* dbeta, frcnn_bf16* dz, frcnn_bf16* gpre, int64_t m, int c, int64_t count,
                             float param_grad_scale, const struct frcnn_fp8_out* f8 /* NULL, or the e5m2 twin of dz; with a twin dz may be
                             NULL: the bf16 tensor is then not stored (every consumer reads the twin) */, frcnn_stream_t stream);
/* frcnn_bn_bwd_apply_fused (ReLU bit mask form, no gpre) that ALSO runs the backward reduce of a second BatchNorm receiving the same
 * masked gradient (red2: its z / mean / invstd / partial; red2->relu_mask is ignored -- the mask is this launch's): the shortcut
 * BatchNorm of a stage's first block beside the block-final one (Keras conv<N>_block1_0_bn / _3_bn,
 * models/feature_extractor.py:8).  == frcnn_bn_bwd_apply_fused + frcnn_bn_bwd_reduce(gout, relu_mask, red2...), the slot partials up to
 * the order of their float atomics.  c % 64 == 0.  (ABI 7) */
int frcnn_bn_bwd_apply_fused_red2(const frcnn_bf16* gout, const uint8_t* relu_mask, const frcnn_bf16* z, const float* mean,
                                  const float* invstd, const float* gamma, const float* partial, int slots, float* dgamma, float* dbeta,
                                  frcnn_bf16* dz, int64_t m, int c, int64_t count, float param_grad_scale,
                                  const struct frcnn_fp8_out* f8, const struct frcnn_bn_reduce* red2, frcnn_stream_t stream);
/* g_out = g * (act > 0): ReLU backward without BN (RPN intermediate layer) */
int frcnn_relu_bwd(const frcnn_bf16* g, const frcnn_bf16* act, frcnn_bf16* out, int64_t n, frcnn_stream_t stream);
/* per-channel column sum of a bf16 [m,c] matrix ADDED (float atomics) to fp32 out[c] (bias gradients;
 * the flat gradient buffer is zeroed at the start of a step) */
int frcnn_colsum_bf16(const frcnn_bf16* x, int64_t m, int c, int ld, float* out, frcnn_stream_t stream);
/* The same column sums fused with the pass that produces the matrix (one launch and one read of it less; [m][c] row-major, c % 8 == 0,
 * 16-byte aligned; colsum accumulates, as frcnn_colsum_bf16):
 *   frcnn_cast_colsum:      dst = bf16(src);  colsum[col] += sum_rows dst   -- the RPN head gradient scattered in fp32 and its bias gradient
 *   frcnn_relu_bwd_colsum:  out = act > 0 ? g : 0;  colsum[col] += sum_rows out   -- the ReLU backward of the RPN's 3x3 layer
 *                           (models/detectors/rpn_detector.py:26-34) and that layer's bias gradient */
int frcnn_cast_colsum(const float* src, frcnn_bf16* dst, int64_t m, int c, float* colsum, frcnn_stream_t stream);
int frcnn_relu_bwd_colsum(const frcnn_bf16* g, const frcnn_bf16* act, frcnn_bf16* out, int64_t m, int c, float* colsum, frcnn_stream_t stream);

/* out = [ReLU](BN(z) + BN2(z2)) with batch statistics for both BatchNorm layers in one pass (the block-final BatchNorm of a
 * stage's first bottleneck block and the BatchNorm of its shortcut convolution, reference Keras ResNet50 conv*_block1_0_bn /
 * _3_bn / _add / _out): what frcnn_bn_train_apply(z2 -> tmp) followed by frcnn_bn_train_apply(z, res = tmp) computes, bit for
 * bit, without tmp.  Both layers share slots, count, momentum, eps; each publishes its own mean / invstd / moving statistics. */
int frcnn_bn_train_apply_dual(const frcnn_bf16* z, const double* stats_partial, const float* gamma, const float* beta,
                              float* moving_mean, float* moving_var, float* mean, float* invstd, const frcnn_bf16* z2,
                              const double* stats_partial2, const float* gamma2, const float* beta2, float* moving_mean2,
                              float* moving_var2, float* mean2, float* invstd2, int slots, int64_t count, float momentum, float eps,
                              int relu, frcnn_bf16* out, uint8_t* relu_mask, int64_t m, int c, const struct frcnn_fp8_out* f8,
                              frcnn_stream_t stream);
/* The ResNet stem's BatchNorm (batch statistics, as frcnn_bn_train_apply) + ReLU + 3x3 / stride-2 / pad-1 max pool
 * (reference models/feature_extractor.py:8-10: conv1_bn, conv1_relu, pool1_pad, pool1_pool) in one pass over z [n,h,w,c]:
 * pooled [n,ho,wo,c], argmax and relu_mask ([n*h*w, c/8] bits of (activation > 0), may be NULL) are bit-identical to
 * frcnn_bn_train_apply followed by frcnn_maxpool3x3s2_fwd, but the activation between them is never written.  mean / invstd /
 * moving statistics as frcnn_bn_train_apply. */
int frcnn_bn_train_apply_maxpool(const frcnn_bf16* z, const double* stats_partial, int slots, int64_t count, const float* gamma,
                                 const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                                 frcnn_bf16* pooled, uint8_t* argmax, uint8_t* relu_mask, float* mean, float* invstd, int n, int h,
                                 int w, int c, int ho, int wo, frcnn_stream_t stream);
/* ZeroPadding2D(1) + MaxPool 3x3/2 valid of Keras ResNet50 (pool1_pad/pool1_pool); input >= 0.
 * argmax (uint8, 0..8 window position, first max wins) feeds the backward gather. */
int frcnn_maxpool3x3s2_fwd(const frcnn_bf16* x, frcnn_bf16* y, uint8_t* argmax, int n, int h, int w, int c,
                           int ho, int wo, frcnn_stream_t stream);
int frcnn_maxpool3x3s2_bwd(const frcnn_bf16* gy, const uint8_t* argmax, frcnn_bf16* gx, int n, int h, int w, int c,
                           int ho, int wo, frcnn_stream_t stream);
/* frcnn_maxpool3x3s2_bwd fused with the BatchNorm-backward reduce of the layer whose (ReLU) activation the pool read -- the ResNet
 * stem (models/feature_extractor.py:8-10: conv1_bn -> conv1_relu -> pool1_pool): gx is written as by frcnn_maxpool3x3s2_bwd, and
 * red->partial receives what frcnn_bn_bwd_reduce(gout = gx, relu_mask = red->relu_mask, z = red->z, ...) would add (c = 64). */
int frcnn_maxpool3x3s2_bwd_bnreduce(const frcnn_bf16* gy, const uint8_t* argmax, frcnn_bf16* gx, int n, int h, int w, int c,
                                    int ho, int wo, const struct frcnn_bn_reduce* red, frcnn_stream_t stream);

/* Keras SGD(momentum) step on a flat parameter range (train_faster_rcnn.py:109-112,
 * models/faster_rcnn.py:104) fused with the L2 kernel regulariser gradient 2*l2*w
 * (models/faster_rcnn.py:101) and the bf16 working-copy refresh:
 *   g' = g*grad_scale + 2*l2*w ; v = momentum*v - lr*g' ; w += v ; w_bf16 = bf16(w)
 * lr = values[i] for the first i with step <= boundaries[i], values[nb] beyond the last boundary (values has nb+1 entries):
 * tf.keras.optimizers.schedules.PiecewiseConstantDecay evaluated on the device step counter. */
int frcnn_sgd_momentum(float* w, const float* g, float* v, frcnn_bf16* w_bf16, int64_t n, float momentum, float l2,
                       float grad_scale, const int64_t* step, const int64_t* boundaries, const float* values, int nb,
                       frcnn_stream_t stream);
int frcnn_step_increment(int64_t* step, frcnn_stream_t stream);
/* The whole optimizer step of a training plan in ONE launch (was: one frcnn_sgd_momentum per decay range + frcnn_stem_pack_weights
 * + frcnn_step_increment): frcnn_sgd_momentum over the n elements of the flat buffers, where
 *   - elements [0, decay_end) take the regulariser l2, the rest none (the L2-regularised kernels are registered first);
 *   - stem (optional): the 7x7x3 stem kernel occupies [stem_begin, stem_begin + stem_cout * 147) of w; every updated value of it
 *     is also written, as bf16, to its place in the packed [stem_cout][7][8][4] tap layout the stem convolution reads
 *     (padding taps stay as they are: zero);
 *   - *step is incremented by 1 AFTER every workgroup has read it: each workgroup bumps *arrive (device uint32, zero before the
 *     launch) when it is done and the last one resets it to zero and increments the step -- same arithmetic as the separate launches,
 *     element for element. */
typedef struct frcnn_sgd_fused {
    int64_t decay_end;
    float l2;
    int64_t stem_begin;           /* -1: no stem re-pack */
    int stem_cout;
    frcnn_bf16* stem_packed;
    uint32_t* arrive;             /* NULL (ABI 7): the launch covers PART of the parameters and leaves the step counter alone */
} frcnn_sgd_fused;
int frcnn_sgd_momentum_fused(float* w, const float* g, float* v, frcnn_bf16* w_bf16, int64_t n, float momentum, float grad_scale,
                             int64_t* step, const int64_t* boundaries, const float* values, int nb, const frcnn_sgd_fused* f,
                             frcnn_stream_t stream);

/* ------------------------------------------------------------------ boxes / RPN / NMS */

/* models/detectors/rpn_detector.py:162-199: anchors[(y*gw+x)*A + k] , k = ratio-major. */
int frcnn_anchors_generate(float* anchors, int gh, int gw, const float* scales /*host*/, int ns,
                           const float* ratios /*host*/, int nr, float base_h, float base_w, float stride_h,
                           float stride_w, frcnn_stream_t stream);
/* rpn_detector.py:81-91 after the two 1x1 convs: head [B*gh*gw, ld] fp32 with columns
 * [0,2A) = cls logits (k*2+{bg,fg}) and [2A,6A) = deltas (k*4+..).  For each kept anchor index
 * keep[i] (or all anchors when keep == NULL): pair softmax -> scores[B,n,2]; deltas[B,n,4]. */
int frcnn_rpn_head_post(const float* head, int ld, int b, int num_anchors_total, int a_per_loc, const int32_t* keep,
                        int n, float* scores, float* deltas, frcnn_stream_t stream);
/* The same, and in the same launch the decode step of proposal NMS (frcnn_decode_boxes with the n shared `regions` [n,4] and
 * C = 1): decoded [B,n,4] relative boxes.  decoded == NULL: plain frcnn_rpn_head_post. */
int frcnn_rpn_head_post_decode(const float* head, int ld, int b, int num_anchors_total, int a_per_loc, const int32_t* keep,
                               int n, float* scores, float* deltas, const float* regions, float* decoded, float img_w,
                               float img_h, frcnn_stream_t stream);
/* utils/boxes.py:4-17 */
int frcnn_clip_to_window(const float* boxes, float* out, int64_t n, float x0, float y0, float x1, float y1,
                         frcnn_stream_t stream);
/* utils/post_processing.py:39-49: decode(pred_boxes, tiled regions) / [W,H,W,H].
 * regions [R,4] (regions_per_image = 0) or [B,R,4]; deltas [B,R,C,4] -> out [B,R,C,4]. */
int frcnn_decode_boxes(const float* regions, int regions_per_image, const float* deltas, float* out, int b, int r,
                       int c, float img_w, float img_h, frcnn_stream_t stream);
/* utils/boxes.py:44-73: out[b,r,c] = encode(boxes[b,r,c], regions[r] or regions[b,r]) = [(centre - centre_ref) / size_ref,
 * log(size / size_ref)]; the inverse of frcnn_decode_boxes with img_w = img_h = 1. */
int frcnn_encode_boxes(const float* boxes, const float* regions, int regions_per_image, float* out, int b, int r, int c,
                       frcnn_stream_t stream);
/* utils/boxes.py:86-93: out = in / [w, h, w, h] (true division, as tf.divide) over n boxes */
int frcnn_boxes_divide(const float* in, float* out, int64_t n, float w, float h, frcnn_stream_t stream);
/* tf.image.combined_non_max_suppression as called at utils/post_processing.py:53-55.
 * boxes [B,N,q,4] (q = 1 or C), scores [B,N,*] with row stride score_stride, class c at column
 * score_offset + c.  Outputs [B,T,4], [B,T], int32 [B,T], int32 [B].
 * Workspace: frcnn_nms_workspace_bytes() bytes, reusable from call to call; zero it once after allocating it.  (Single-class
 * lists of more than 512 candidates run with eight workgroups per image that hand a suppression matrix over through the
 * workspace; their arrival words count launches and need no per-call reset, but a never-used buffer should not hold
 * arbitrary bytes.  Two calls that share a workspace must not overlap in time.) */
size_t frcnn_nms_workspace_bytes(int b, int n, int c, int max_per_class, int max_total);
int frcnn_nms_combined(const float* boxes, const float* scores, int b, int n, int q, int c, int score_stride,
                       int score_offset, int max_per_class, int max_total, float iou_thr, float score_thr,
                       float* out_boxes, float* out_scores, int32_t* out_classes, int32_t* out_valid,
                       void* workspace, size_t workspace_bytes, frcnn_stream_t stream);
/* The same, and in the same launch the kept boxes once more in absolute image coordinates: out_boxes_abs [B,T,4] =
 * out_boxes * [scale_x, scale_y, scale_x, scale_y] -- to_absolute of the proposals (utils/boxes.py:76-83 as called at
 * fast_rcnn_detector.py:67), which the Fast-RCNN stage needs as its `regions` (replaces a frcnn_boxes_scale launch). */
int frcnn_nms_combined_abs(const float* boxes, const float* scores, int b, int n, int q, int c, int score_stride,
                           int score_offset, int max_per_class, int max_total, float iou_thr, float score_thr,
                           float* out_boxes, float* out_scores, int32_t* out_classes, int32_t* out_valid,
                           void* workspace, size_t workspace_bytes, float* out_boxes_abs, float scale_x, float scale_y,
                           frcnn_stream_t stream);

/* ------------------------------------------------------------------ RoI pooling + heads */

/* fast_rcnn_detector.py:154-175: crop_and_resize(14x14 bilinear, extrapolation 0) + MaxPool 2x2,
 * fused; rois [B,P,4] relative [x1,y1,x2,y2]; pooled [B*P, ps*ps*C] in (h,w,c) order; argmax
 * uint8 (0..ks*ks-1) per pooled element.  `row_index` (optional int32[nrows]) selects which RoI rows
 * (b*P+p) to process and writes them densely (used to re-pool only sampled rows). */
int frcnn_roi_crop_pool_fwd(const frcnn_bf16* feat, const float* rois, int b, int p, int hf, int wf, int c, int ps,
                            int ks, frcnn_bf16* pooled, uint8_t* argmax, frcnn_stream_t stream);
/* gradient w.r.t. the feature map only (CropAndResizeGradImage o MaxPoolGrad): for each listed
 * row r (RoI rows[r] = b*P+p) scatter-add gpooled[r] into gfeat (fp32, pre-zeroed). */
int frcnn_roi_crop_pool_bwd(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows,
                            int nrows, int p, int hf, int wf, int c, int ps, int ks, float* gfeat,
                            frcnn_stream_t stream);
/* Same gradient, complete and in bf16: EVERY element of gfeat [B,hf,wf,c] is written exactly once (no pre-zeroing, no
 * accumulation into existing contents).  One workgroup per (image, feature row, channel slab) gathers the contributions
 * of the listed rows in LDS -- no global atomics.  c % 64 == 0, wf * 256 B <= 64 KiB. */
int frcnn_roi_crop_pool_bwd_bf16(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows,
                                 int nrows, int b, int p, int hf, int wf, int c, int ps, int ks, frcnn_bf16* gfeat,
                                 frcnn_stream_t stream);
/* The same gradient ADDED to what gfeat already holds (every element read once, written once): gfeat = bf16(gfeat + RoI-branch gradient).
 * For a map whose other consumer wrote its gradient first -- the RPN's data gradient, which then can run beside the proposal NMS
 * instead of after the RoI backward pass.  red (optional): gfeat is complete with this launch and arrives at a BatchNorm(+ReLU) layer;
 * its backward sums are accumulated while the rows are written, exactly what frcnn_bn_bwd_reduce(gout = gfeat, ...) would add. */
int frcnn_roi_crop_pool_bwd_bf16_add(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows,
                                     int nrows, int b, int p, int hf, int wf, int c, int ps, int ks, frcnn_bf16* gfeat,
                                     const struct frcnn_bn_reduce* red, frcnn_stream_t stream);
/* fast_rcnn_detector.py:62-65 after the GEMM: logits [R, ld] fp32 (+bias): softmax over the first
 * nc1 columns -> scores [R,nc1]; columns [nc1, nc1+4*(nc1-1)) -> deltas. */
int frcnn_rcnn_head_post(const float* logits, int ld, const float* bias, int r, int nc1, float* scores, float* deltas,
                         frcnn_stream_t stream);
/* The same, and in the same launch the decode step of detection NMS (frcnn_decode_boxes with regions_per_image = 1 on the deltas
 * just computed): decoded [R, nc1-1, 4] = decode(deltas, regions [R,4] absolute) / [W,H,W,H] (utils/post_processing.py:39-49). */
int frcnn_rcnn_head_post_decode(const float* logits, int ld, const float* bias, int r, int nc1, float* scores, float* deltas,
                                const float* regions, float* decoded, float img_w, float img_h, frcnn_stream_t stream);
/* rois_abs = rois_rel * [W,H,W,H]  (utils/boxes.py:76-83, fast_rcnn_detector.py:67) */
int frcnn_boxes_scale(const float* in, float* out, int64_t n, float sx, float sy, frcnn_stream_t stream);

/* ------------------------------------------------------------------ feature pyramid (BASELINE.json configs[4]) */
/* The reference has no FPN (models/faster_rcnn.py:25-34 wires one conv4 map); these restate Lin et al., "Feature Pyramid Networks for
 * Object Detection", CVPR 2017 (oracle/fpn.py).  The neck's convolutions are frcnn_conv2d_* launches.
 * sec. 3, top-down merge: out[b,y,x,:] = lat[b,y,x,:] + top[b, (y*ht)/h, (x*wt)/w, :]   (nearest-neighbour upsampling; out may alias lat)
 * and its gradient w.r.t. top: gtop[b,ys,xs,:] = (accumulate ? gtop : 0) + sum of g over the fine pixels that read (ys, xs). */
int frcnn_upsample_add(const frcnn_bf16* top, int ht, int wt, const frcnn_bf16* lat, frcnn_bf16* out, int b, int h, int w, int c,
                       frcnn_stream_t stream);
int frcnn_upsample_add_bwd(const frcnn_bf16* g, int h, int w, frcnn_bf16* gtop, int b, int ht, int wt, int c, int accumulate,
                           frcnn_stream_t stream);
/* sec. 4.1, the extra RPN level: y[b,i,j,:] = x[b,2i,2j,:] (y is [b, ceil(h/2), ceil(w/2), c]); backward: gx[b,2i,2j,:] += gy[b,i,j,:]. */
int frcnn_subsample2(const frcnn_bf16* x, frcnn_bf16* y, int b, int h, int w, int c, frcnn_stream_t stream);
int frcnn_subsample2_bwd_add(const frcnn_bf16* gy, frcnn_bf16* gx, int b, int h, int w, int c, frcnn_stream_t stream);
/* sec. 4.2, eq. (1): level of an RoI, k = floor(4 + log2(sqrt(w h) / 224)) clamped to [2, 4], evaluated as
 * 2 + [w h >= 112^2] + [w h >= 224^2] (w, h in input pixels from the relative box [x1,y1,x2,y2]); levels int32 [n]. */
int frcnn_roi_assign_levels(const float* rois_rel, int64_t n, float img_w, float img_h, int32_t* levels, frcnn_stream_t stream);
/* frcnn_roi_crop_pool_fwd / _bwd_bf16 restricted to the RoIs whose level equals `level` (rows of other levels are neither read nor
 * written by the forward launch; the backward launch writes the complete gradient of THIS level's map from this level's rows). */
int frcnn_roi_crop_pool_fwd_level(const frcnn_bf16* feat, const float* rois, int b, int p, int hf, int wf, int c, int ps, int ks,
                                  frcnn_bf16* pooled, uint8_t* argmax, const int32_t* levels, int level, frcnn_stream_t stream);
int frcnn_roi_crop_pool_bwd_bf16_level(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows, int nrows,
                                       int b, int p, int hf, int wf, int c, int ps, int ks, frcnn_bf16* gfeat, const int32_t* levels,
                                       int level, frcnn_stream_t stream);
/* frcnn_rpn_head_post_decode / frcnn_rpn_head_grad for ONE level of a pyramid whose anchors are concatenated per image: the level's n
 * kept anchors occupy rows [offset, offset + n) of the n_total rows of an image (regions: this level's [n,4] slice). */
int frcnn_rpn_head_post_level(const float* head, int ld, int b, int num_anchors_level, int a_per_loc, const int32_t* keep, int n,
                              float* scores, float* deltas, const float* regions, float* decoded, float img_w, float img_h, int n_total,
                              int offset, frcnn_stream_t stream);
int frcnn_rpn_head_grad_level(const float* dlogits_s, const float* ddeltas_s, const int32_t* indices, const int32_t* keep, int b, int s,
                              int num_anchors_level, int a_per_loc, float* dhead, int ld, int offset, int n, frcnn_stream_t stream);

/* ------------------------------------------------------------------ targets / sampling / losses */

/* utils/training.py:7-77 (+ rpn_detector.py:141 objectness conversion when objectness != 0).
 * regions [R,4] (regions_per_image = 0) or [B,R,4] absolute; gt_labels [B,G,C1g], gt_boxes [B,G,4]
 * relative.  Writes target_labels [B,R,C1] and target_boxes [B,R,C1-1,4] (C1 = 2 when objectness). */
int frcnn_assign_targets(const float* regions, int regions_per_image, const float* gt_labels, const float* gt_boxes,
                         int b, int r, int g, int c1g, int objectness, float img_w, float img_h, float fg_lo,
                         float fg_hi, float bg_lo, float bg_hi, float* target_labels, float* target_boxes,
                         frcnn_stream_t stream);
/* utils/training.py:80-120 with a counter-based RNG (Philox4x32-10; counter = (i, image_base + image, step,
 * stream_base + {0 fg, 1 bg}), key = seed).  indices int32 [B,S].  status[0] |= 1 on an empty
 * background set (the reference raises there).  workspace: int32 [B, 2*R].  image_base: index of this call's first image in
 * the GLOBAL batch (data parallelism: rank * per-rank batch), so that N ranks x b images draw exactly the samples one device
 * draws for N*b images. */
int frcnn_sample_indices(const float* target_labels, int b, int r, int c1, int num_samples, float fg_proportion,
                         uint64_t seed, const int64_t* step, int stream_base, int32_t* indices, int32_t* workspace,
                         int32_t* status, int image_base, frcnn_stream_t stream);
/* utils/losses.py + gathers of get_training_samples (rpn_detector.py:156-159,
 * fast_rcnn_detector.py:126-129) + their gradients.
 * scores [B,R,C1] (probabilities), deltas [B,R,C,4], targets as written by frcnn_assign_targets,
 * indices [B,S].  losses[0] = CCE mean, losses[1] = Huber sum.  If dlogits_s / ddeltas_s != NULL
 * they receive the PER-SAMPLE gradient of (cls_scale*CCE + reg_scale*Huber) w.r.t. the
 * PRE-softmax logits [B,S,C1] and the deltas [B,S,C,4] of the sampled rows (duplicated samples
 * stay separate rows; no atomics, deterministic). */
int frcnn_losses(const float* scores, const float* deltas, const float* target_labels, const float* target_boxes,
                 const int32_t* indices, int b, int r, int c1, int s, float cls_scale, float reg_scale, float* losses,
                 float* dlogits_s, float* ddeltas_s, frcnn_stream_t stream);
/* frcnn_losses followed by frcnn_rcnn_head_grad in ONE launch (the Fast-RCNN training chain: every kernel boundary on it
 * costs ~4 us with the chip idle): additionally writes dhead_s [B*S, ld] (bf16 gradient rows, zero padded) and rows_out
 * [B*S] exactly as frcnn_rcnn_head_grad would from dlogits_s / ddeltas_s (which may be NULL here). */
int frcnn_losses_head_grad(const float* scores, const float* deltas, const float* target_labels, const float* target_boxes,
                           const int32_t* indices, int b, int r, int c1, int s, float cls_scale, float reg_scale,
                           float* losses, float* dlogits_s, float* ddeltas_s, frcnn_bf16* dhead_s, int ld, int32_t* rows_out,
                           float* bias_grad /* optional, ld <= 64: += the column sums of dhead_s, exactly what
                           frcnn_colsum_bf16(dhead_s, B*S, ld, ld, bias_grad) adds (the Dense heads' bias gradient) */,
                           frcnn_stream_t stream);
/* frcnn_losses (C1 = 2) followed by frcnn_rpn_head_grad in ONE launch: the per-sample gradients are scatter-ADDED into dhead
 * [B*gh*gw, ld] (pre-zeroed) as frcnn_rpn_head_grad would; dlogits_s / ddeltas_s may be NULL. */
int frcnn_losses_rpn_head_grad(const float* scores, const float* deltas, const float* target_labels,
                               const float* target_boxes, const int32_t* indices, int b, int r, int s, float cls_scale,
                               float reg_scale, float* losses, float* dlogits_s, float* ddeltas_s, const int32_t* keep,
                               int num_anchors_total, int a_per_loc, float* dhead, int ld, frcnn_stream_t stream);
/* RPN: scatter-ADD the per-sample gradients (dlogits_s [B,S,2], ddeltas_s [B,S,4]) into the dense
 * fp32 head-gradient matrix dhead [B*gh*gw, ld] (pre-zeroed).  Sample (b,s) refers to kept anchor
 * indices[b,s], i.e. anchor keep[indices[b,s]] (keep == NULL: identity). */
int frcnn_rpn_head_grad(const float* dlogits_s, const float* ddeltas_s, const int32_t* indices, const int32_t* keep,
                        int b, int s, int num_anchors_total, int a_per_loc, float* dhead, int ld,
                        frcnn_stream_t stream);
/* RCNN: per-sample gradient rows -> dhead_s [B*S, ld] (bf16, zero padded to ld columns);
 * rows_out[b*S+s] = b*R + indices[b,s] (the RoI row each sample came from). */
int frcnn_rcnn_head_grad(const float* dlogits_s, const float* ddeltas_s, const int32_t* indices, int b, int r, int c1,
                         int s, frcnn_bf16* dhead_s, int ld, int32_t* rows_out, frcnn_stream_t stream);

/* Host helper (no device work): CRC-32C of data[0..n) continuing from crc (0 = fresh), as used by the TFRecord
 * framing of the reference's data files (data/build_tf_records.py:128-133, data/input_pipeline.py:31). */
uint32_t frcnn_crc32c(uint32_t crc, const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
