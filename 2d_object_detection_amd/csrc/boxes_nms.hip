// Anchors, RPN head post-processing, box decode and the batched per-class NMS
// (tf.image.combined_non_max_suppression as used by the reference's utils/post_processing.py).
//
// Box arithmetic is compiled with -ffp-contract=off so that IoU comparisons are bit-identical
// to the C oracle (no FMA contraction).
//
// NMS design (wavefront primitives, one 1024-thread workgroup per (image, class); details at nms_class_kernel):
//   1. composite keys (score bits << 32 | ~index) for score > threshold; the candidates are NOT sorted as a whole -- greedy
//      NMS stops long before the list ends.  In rounds, the next 1024 best unvisited keys are SELECTED (MSB-first radix
//      select, one per-wave LDS histogram pass per byte, ending as soon as the boundary does not split a digit; 32-bit score
//      keys staged in LDS, or re-read from global when they do not fit beside the kept list), compacted with one returning
//      atomic per wave and sorted by an in-LDS bitonic network whose stages with pair distance <= 64 are wave-local;
//   2. the sorted round is resolved in 256-candidate chunks: every candidate is tested against the kept list (4 threads
//      per candidate), the survivors' upper-triangular suppression rows are built with wave ballots (column candidates in
//      registers) and wave 0 walks them in score order, appending whole runs of non-suppressing survivors to the kept list.
//      Visit order = exact descending composite order, so the result is bit-identical to a full sort.  The loop ends when
//      max_per_class boxes are kept or candidates run out -- no host sync;
//   3. one class: the workgroup writes the final padded outputs itself; several classes: one workgroup per image merges the
//      per-class kept lists (nms_merge_kernel: bitonic sort of <= C*max_per_class (score, class, slot) keys), top max_total
//      written clipped to [0,1], remainder zero.
#include "common.h"

#pragma clang fp contract(off)

namespace {

// ---------------------------------------------------------------- anchors
struct AnchorShapes { float w[64]; float h[64]; int na; };

__global__ void anchors_kernel(float* __restrict__ out, int gh, int gw, AnchorShapes sh, float stride_h, float stride_w) {
    const int total = gh * gw * sh.na;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i % sh.na;
        const int loc = i / sh.na;
        const int x = loc % gw, y = loc / gw;
        const float xc = (float)x * stride_w, yc = (float)y * stride_h;
        const float hw = 0.5f * sh.w[k], hh = 0.5f * sh.h[k];
        f32x4 b = {xc - hw, yc - hh, xc + hw, yc + hh};
        *reinterpret_cast<f32x4*>(out + (int64_t)i * 4) = b;
    }
}

// utils/boxes.py:20-41 + :86-93: region + regression deltas -> box, relative to the image
__device__ __forceinline__ f32x4 decode_one(const f32x4 ref, const f32x4 d, const float W, const float H) {
    const float cxr = (ref[2] + ref[0]) / 2.0f, cyr = (ref[3] + ref[1]) / 2.0f;
    const float wr = ref[2] - ref[0], hr = ref[3] - ref[1];
    const float cx = d[0] * wr + cxr, cy = d[1] * hr + cyr;
    const float w = expf(d[2]) * wr, h = expf(d[3]) * hr;
    return f32x4{(cx - 0.5f * w) / W, (cy - 0.5f * h) / H, (cx + 0.5f * w) / W, (cy + 0.5f * h) / H};
}

// ---------------------------------------------------------------- RPN head post (+ optionally the decode of proposal NMS)
__global__ void rpn_head_post_kernel(const float* __restrict__ head, int ld, int B, int A_total, int apl, const int* __restrict__ keep,
                                     int n, float* __restrict__ scores, float* __restrict__ deltas, const float* __restrict__ regions,
                                     float* __restrict__ decoded, float W, float H, int out_stride, int out_off) {
    // out_stride / out_off: the n rows of image b land at rows [b * out_stride + out_off, ... + n) of the outputs -- one pyramid
    // level's window of the concatenated per-image anchor list (out_stride = n, out_off = 0: a single feature map)
    const int total = B * n;
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += gridDim.x * blockDim.x) {
        const int b = i0 / n, j = i0 - b * n;
        const int64_t i = (int64_t)b * out_stride + out_off + j;
        const int a = keep ? keep[j] : j;
        const int loc = a / apl, k = a - loc * apl;
        const int locs = A_total / apl;
        const float* row = head + ((int64_t)b * locs + loc) * ld;
        const float l0 = row[2 * k], l1 = row[2 * k + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float inv = 1.f / (e0 + e1);
        scores[(int64_t)i * 2] = e0 * inv;
        scores[(int64_t)i * 2 + 1] = e1 * inv;
        const float* d = row + 2 * apl + 4 * k;
        const f32x4 dv = {d[0], d[1], d[2], d[3]};
        *reinterpret_cast<f32x4*>(deltas + (int64_t)i * 4) = dv;
        if (decoded) *reinterpret_cast<f32x4*>(decoded + (int64_t)i * 4) = decode_one(*reinterpret_cast<const f32x4*>(regions + (int64_t)j * 4), dv, W, H);
    }
}

__global__ void clip_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, float x0, float y0, float x1, float y1) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 4);
        b[0] = fmaxf(fminf(b[0], x1), x0);
        b[1] = fmaxf(fminf(b[1], y1), y0);
        b[2] = fmaxf(fminf(b[2], x1), x0);
        b[3] = fmaxf(fminf(b[3], y1), y0);
        *reinterpret_cast<f32x4*>(out + i * 4) = b;
    }
}

__global__ void scale_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, float sx, float sy) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 4);
        b[0] *= sx; b[1] *= sy; b[2] *= sx; b[3] *= sy;
        *reinterpret_cast<f32x4*>(out + i * 4) = b;
    }
}

// utils/boxes.py:86-93 (true division, as tf.divide)
__global__ void divide_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, float w, float h) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 4);
        b[0] /= w; b[1] /= h; b[2] /= w; b[3] /= h;
        *reinterpret_cast<f32x4*>(out + i * 4) = b;
    }
}

// utils/boxes.py:44-73: t = [(c - c_ref) / size_ref, log(size / size_ref)] (zero-size references divide by zero as the reference does)
__global__ void encode_kernel(const float* __restrict__ boxes, const float* __restrict__ regions, int rpi, float* __restrict__ out, int B, int R,
                              int C) {
    const int64_t total = (int64_t)B * R * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t br = i / C;
        const int r = (int)(br % R);
        const int b = (int)(br / R);
        const f32x4 ref = *reinterpret_cast<const f32x4*>(regions + ((rpi ? (int64_t)b * R : 0) + r) * 4);
        const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + i * 4);
        const float cxr = (ref[2] + ref[0]) / 2.0f, cyr = (ref[3] + ref[1]) / 2.0f;
        const float wr = ref[2] - ref[0], hr = ref[3] - ref[1];
        const float cx = (bx[2] + bx[0]) / 2.0f, cy = (bx[3] + bx[1]) / 2.0f;
        const float w = bx[2] - bx[0], h = bx[3] - bx[1];
        f32x4 o = {(cx - cxr) / wr, (cy - cyr) / hr, logf(w / wr), logf(h / hr)};
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

__global__ void decode_kernel(const float* __restrict__ regions, int rpi, const float* __restrict__ deltas, float* __restrict__ out, int B,
                              int R, int C, float W, float H) {
    const int64_t total = (int64_t)B * R * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t br = i / C;
        const int r = (int)(br % R);
        const int b = (int)(br / R);
        const f32x4 ref = *reinterpret_cast<const f32x4*>(regions + ((rpi ? (int64_t)b * R : 0) + r) * 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(deltas + i * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = decode_one(ref, d, W, H);
    }
}

// RCNN head post (bias + softmax / split: the arithmetic of rcnn_head_post_kernel, targets_losses.hip -- one wave per row, lane =
// column, the exponentials summed in ascending class order) and, in the same launch, the decode step of detection NMS
// (decode_kernel on this row's freshly computed deltas and its absolute region): frcnn_rcnn_head_post + frcnn_decode_boxes.
__global__ __launch_bounds__(256) void rcnn_head_post_decode_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ bias, int R,
                                                                    int C1, float* __restrict__ scores, float* __restrict__ deltas,
                                                                    const float* __restrict__ regions, float* __restrict__ decoded, float W, float H) {
    const int nreg = 4 * (C1 - 1);
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = logits + (int64_t)r * ld;
    const float l = lane < C1 ? row[lane] + bias[lane] : -INFINITY;
    float mx = l;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
    const float e = lane < C1 ? expf(l - mx) : 0.f;
    float s = 0.f;
    for (int c = 0; c < C1; ++c) s += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), c));
    const float inv = 1.f / s;
    if (lane < C1) scores[(int64_t)r * C1 + lane] = e * inv;
    for (int c = lane; c < nreg; c += 64) deltas[(int64_t)r * nreg + c] = row[C1 + c] + bias[C1 + c];
    const f32x4 ref = *reinterpret_cast<const f32x4*>(regions + (int64_t)r * 4);
    for (int c = lane; c < C1 - 1; c += 64) {
        const f32x4 d = {row[C1 + 4 * c] + bias[C1 + 4 * c], row[C1 + 4 * c + 1] + bias[C1 + 4 * c + 1], row[C1 + 4 * c + 2] + bias[C1 + 4 * c + 2],
                         row[C1 + 4 * c + 3] + bias[C1 + 4 * c + 3]};
        *reinterpret_cast<f32x4*>(decoded + ((int64_t)r * (C1 - 1) + c) * 4) = decode_one(ref, d, W, H);
    }
}

// ---------------------------------------------------------------- NMS
__device__ __forceinline__ float nms_iou(const f32x4 a, const f32x4 b) {
    const float y0i = fminf(a[0], a[2]), x0i = fminf(a[1], a[3]);
    const float y1i = fmaxf(a[0], a[2]), x1i = fmaxf(a[1], a[3]);
    const float y0j = fminf(b[0], b[2]), x0j = fminf(b[1], b[3]);
    const float y1j = fmaxf(b[0], b[2]), x1j = fmaxf(b[1], b[3]);
    const float area_i = (y1i - y0i) * (x1i - x0i);
    const float area_j = (y1j - y0j) * (x1j - x0j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iy0 = fmaxf(y0i, y0j), ix0 = fmaxf(x0i, x0j);
    const float iy1 = fminf(y1i, y1j), ix1 = fminf(x1i, x1j);
    const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
    return inter / (area_i + area_j - inter);
}

// The same IoU on corner-normalised boxes (y0 <= y1, x0 <= x1) with precomputed areas: bit-identical to nms_iou (same
// operations on the same values), but the division is skipped when the boxes do not intersect (0 / union = 0 <= thr).
__device__ __forceinline__ f32x4 nms_norm(const f32x4 a) {
    return f32x4{fminf(a[0], a[2]), fminf(a[1], a[3]), fmaxf(a[0], a[2]), fmaxf(a[1], a[3])};
}
__device__ __forceinline__ float nms_area(const f32x4 n) { return (n[2] - n[0]) * (n[3] - n[1]); }
// v_max_f32 / v_min_f32 as written (fmaxf / fminf put a canonicalising v_max x, x in front of every operand that comes from memory:
// 8 of this test's ~45 instructions; for non-NaN operands the results are the same bits)
__device__ __forceinline__ float vmax(const float a, const float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(const float a, const float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

struct IouThr {                 // thr, fl(thr (1 + 4e-6)), fl(thr (1 - 4e-6))
    float thr, hi, lo;
};
__device__ __forceinline__ IouThr iou_thr_of(const float thr) { return IouThr{thr, thr * (1.0f + 4e-6f), thr * (1.0f - 4e-6f)}; }

__device__ __forceinline__ bool nms_over(const f32x4 a, const float area_a, const f32x4 b, const float area_b, const IouThr th) {
    const float iy0 = vmax(a[0], b[0]), ix0 = vmax(a[1], b[1]);
    const float iy1 = vmin(a[2], b[2]), ix1 = vmin(a[3], b[3]);
    const float inter = vmax(iy1 - iy0, 0.0f) * vmax(ix1 - ix0, 0.0f);
    const bool live = inter > 0.0f && area_a > 0.0f && area_b > 0.0f;           // thr >= 0: a zero IoU never suppresses
    // The quotient is only needed where it could round to either side of thr.  With u = area_a + area_b - inter (the divisor's own
    // value): inter > fl(u thr (1 + 4e-6)) implies inter / u > thr (1 + 3.8e-6) (two roundings of 2^-24 each against the 4e-6), whose
    // correctly rounded value is still above thr; inter < fl(u thr (1 - 4e-6)) likewise stays below -- provided the products are
    // normal numbers (`sure`).  Only the sliver between (or a NaN / tiny product) takes the IEEE division -- behind a wave-uniform
    // branch, so a wave without such a lane never issues it (the division was a third of this function's VALU work, and the two
    // callers are VALU-bound on ONE CU).  Same truth value as the division for every non-NaN input
    // (tests/test_host_logic.py::test_nms_threshold_predicate_equals_the_division restates the arithmetic).
    const float u = area_a + area_b - inter;
    const float hi = u * th.hi, lo = u * th.lo;
    const bool sure = lo > 1e-30f;
    bool over = live && sure && inter > hi;
    const bool sliver = live && !over && !(sure && inter < lo);
    if (__builtin_amdgcn_ballot_w64(sliver) != 0ull) {
        if (sliver) over = inter / u > th.thr;
    }
    return over;
}

// order-preserving float -> uint (handles negatives too)
__device__ __forceinline__ unsigned int float_key(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned int k) {
    const unsigned int u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

constexpr int NMS_T = 1024;
constexpr int NMS_LDS_KEYS = 16384;          // merge kernel: C * max_per_class keys sorted in LDS
constexpr int NMS_RK = 1024;                 // candidates selected, sorted and resolved per round
constexpr int NMS_HC = NMS_T / 64;            // copies of the digit histogram: one per wave
constexpr int NMS_CH = 256;                  // candidates resolved per iteration of the greedy loop

// descending bitonic sort of n_pad (power of two) u64 keys by NMS_T threads.  Pair t of a stage with distance j is
// (i, i + j), i = ((t >> lj) << (lj + 1)) + (t & (j - 1)); a wave's 64 consecutive pairs of a stage with j <= 64 lie in ONE
// 128-key block, the same block in every such stage, so those stages need no workgroup barrier: the LDS unit executes a
// wave's instructions in order (the wave-level fence only stops the compiler from keeping keys in registers).  Only the
// stages with j >= 128 are followed by a barrier: 6 instead of 55 for 1024 keys, 15 instead of 78 for 4096.
__device__ void bitonic_desc(unsigned long long* keys, int n_pad) {
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1, lj = 31 - __clz(k >> 1); j > 0; j >>= 1, --lj) {
            for (int t = threadIdx.x; t < (n_pad >> 1); t += blockDim.x) {
                const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1));     // lower index of the pair (j = 1 << lj)
                const int l = i + j;
                const bool desc = ((i & k) == 0);
                const unsigned long long a = keys[i], b = keys[l];
                if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
            }
            if (j > 64 || (j == 1 && k >= 128)) __syncthreads();          // (after the last wave-local stage of a phase: the next
            else {                                                        //  phase opens with j = k >= 128, across waves)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
}

// The same network for n_pad <= NMS_T keys with ONE KEY PER THREAD, held in a register: a stage with pair distance j < 64 is a lane
// exchange (two ds_bpermute, no LDS store / wait / reload between dependent stages -- the in-memory form above spends ~500 cycles on
// each of its 45-55 dependent stages), only the stages with j >= 64 (6 of 45 for 512 keys, 10 of 55 for 1024) go through keys[].
// Thread t's key ends in keys[t]: descending.
__device__ void bitonic_desc_reg(unsigned long long* keys, const int n_pad) {
    const int t = threadIdx.x;
    unsigned long long key = t < n_pad ? keys[t] : 0ull;
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long other;
            if (j >= 64) {                                    // (uniform)
                __syncthreads();                              // the previous cross-wave stage's readers are done
                if (t < n_pad) keys[t] = key;
                __syncthreads();
                other = t < n_pad ? keys[t ^ j] : 0ull;
            } else {
                other = __shfl_xor(key, j);
            }
            const bool take_max = ((t & j) == 0) == ((t & k) == 0);      // lower index of the pair in a descending block, or upper in an ascending one
            const bool gt = key > other;
            key = (take_max == gt) ? key : other;
        }
    }
    __syncthreads();
    if (t < n_pad) keys[t] = key;
    __syncthreads();
}

struct NmsParams {
    const float* boxes; const float* scores;
    int N, q, C, score_stride, score_offset, max_per_class, lds_keys;
    float iou_thr, score_thr;
    unsigned long long* kept_keys;   // [B][C*max_per_class]: (score key << 32 | ~(class*max_per_class + slot)) or 0
    int* kept_idx;                   // [B][C*max_per_class] box index
    // C == 1 only (proposal NMS): the class list IS the image's merged list, so this kernel also writes the final outputs
    // (no nms_merge_kernel launch); null otherwise
    float* out_boxes; float* out_scores; int* out_classes; int* out_valid; int max_total;
    float* out_abs; float sx, sy;    // optional: the final boxes once more, scaled by [sx, sy, sx, sy] (frcnn_nms_combined_abs)
    // team mode (nms_class_kernel<., true>): `team` workgroups per (image, class); the first round's suppression matrix is built by all
    // of them (rows split), exchanged through team_mat [B*C][NMS_TEAM_N][NMS_TEAM_N / 64] and the arrival words team_sync [B*C][16]
    int team; unsigned long long* team_mat; unsigned int* team_sync;
    // split mode (nms_class_kernel<false, true, true>: lists whose score keys do not fit one workgroup's LDS): every workgroup of the team
    // owns N / NMS_TEAM consecutive candidates; counts, digit histograms and the round's keys are exchanged through team_xch
    int split_len; unsigned int* team_xflag; unsigned int* team_xch;
};

constexpr int NMS_XSTAGES = 10;              // exchanges of a split launch: valid counts, up to 8 select passes, the round's keys
constexpr size_t NMS_XCH_WORDS = 8 /*counts*/ + (size_t)8 * 8 * 256 /*[pass][workgroup][digit]*/ + 8 /*key counts*/ + (size_t)8 * 512 * 2 /*[workgroup][key] as word pairs*/;

constexpr int NMS_TEAM = 8;                  // workgroups per (image, class) in team mode
constexpr int NMS_TEAM_N = 512;              // candidates of the team round (NMS_RK / 2: the first round's goal)

// LDS of nms_class_kernel; *lds_keys: the 32-bit score keys of the N candidates are staged in LDS (else re-read from global)
static size_t nms_class_lds(int n, int max_per_class, bool team, bool* lds_keys, int* split_len) {
    const size_t cap = team ? NMS_TEAM_N : NMS_CH;            // candidates whose boxes / matrix rows are resident at once
    const size_t fixed = (size_t)max_per_class * 28 + 16 + cap * (16 + 4 + 4 + 4) + cap * (cap / 64) * 8 + 64 + 16 + (size_t)NMS_HC * 256 * 4 + (size_t)NMS_RK * 8 + 16;
    *lds_keys = fixed + (size_t)n * 4 <= 150 * 1024;
    *split_len = 0;
    if (*lds_keys) return fixed + (size_t)n * 4;
    const int slice = ((n + 8 - 1) / 8 + 3) & ~3;              // a team member's candidates in split mode (NMS_TEAM = 8)
    if (team && fixed + (size_t)slice * 4 <= 150 * 1024) {
        *split_len = slice;
        return fixed + (size_t)slice * 4;
    }
    return fixed;
}

// One workgroup per (image, class).  Greedy NMS visits candidates in descending score order and usually stops long before the
// list ends (max_per_class boxes kept), so the candidates are NOT sorted as a whole: in rounds, the next NMS_RK best ones are
// SELECTED (MSB-first radix select of the composite key (score bits << 32 | ~index): one histogram pass per byte, ending as
// soon as the boundary does not split a digit), compacted, sorted with an in-LDS bitonic network and resolved:
//   1. every candidate of a 256-chunk is tested against the boxes kept so far (4 threads per candidate);
//   2. for the survivors only, the chunk's upper-triangular suppression matrix is built with wave ballots
//      (row i, word w: which later candidates 64w..64w+63 box i would suppress);
//   3. one wave walks the survivors in score order -- ctz over the alive mask, one v_readlane per matrix word -- so the
//      sequential part costs one step per KEPT box, not per candidate.
// The order of visits is exactly the descending composite-key order (every key of a round is larger than every key of the
// next; inside a round the sort is total): bit-identical results to a full sort.  No host sync, no library sort.
#ifdef FRCNN_NMS_STAMPS
// kernel-development build (tools/nms_stamps.py): thread 0 of the first workgroups accumulates the constant-rate clock (100 MHz)
// over the kernel's phases -- 0 key staging, 1 radix select, 2 compaction, 3 sort, 4 chunk load + kept-list test, 5 survivor
// compaction + suppression rows, 6 walk, 7 tail; 8 rounds, 9 chunks, 10 kept, 11-13 parts of 5, 14 select passes, 15 the histogram sweeps of 1 -- into a device symbol read by frcnn_debug_nms_stamps
__device__ unsigned long long g_nms_stamps[2][8][16];
#define NMS_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_prev; st_prev = now_; } } while (0)
#define NMS_COUNT(i) do { if (threadIdx.x == 0) st_acc[i] += 1ull; } while (0)
#else
#define NMS_STAMP(i) do { } while (0)
#define NMS_COUNT(i) do { } while (0)
#endif

// TEAM (proposal NMS: few (image, class) lists, each long): NMS_TEAM workgroups per list.  Each of them stages the keys and runs the FIRST
// round's select / compaction / sort itself (identical results, no exchange, on CUs that would idle otherwise); then the round's whole
// suppression matrix -- up to 512 x 512 pair tests, the VALU-bound bulk of the one-workgroup form (kept-list test + per-chunk matrices:
// 40 of its 67 us at 8768 -> 300) -- is built once, rows split over the team, and handed to workgroup 0 of the team through global
// memory (release: __threadfence + arrival counter; acquire: the counter, then __threadfence).  Workgroup 0 walks the round in score
// order from that matrix and, if the list does not end inside the round, continues alone with the rounds of the one-workgroup form.
// Same visits in the same order: same results.  The helpers wait for nothing, so the counter is always reached.
// SPLIT (TEAM, lists too long for LDSK: the pyramid's 82 k candidates, 57 k at 600 x 1987): one workgroup streaming such a list once takes
// 20 us, and the select + compaction of a round stream it three or four times.  Every team member stages only ITS N / 8 candidates (in
// LDS), sweeps only those, and the members exchange what the round needs -- valid counts, one 256-bin histogram per select pass, the
// round's keys -- through global memory, all-to-all, with one arrival word per (exchange, member) holding the launch number (the hand-off
// form of the matrix below).  Every member computes the same digit, threshold and key list from the same sums; from the sort on the
// team round is unchanged.  Rounds after the team round (workgroup 0 alone) re-read the scores from global memory.
template <bool LDSK, bool TEAM, bool SPLIT = false>
__global__ __launch_bounds__(NMS_T) void nms_class_kernel(const NmsParams p) {
    static_assert(!SPLIT || (TEAM && !LDSK), "split mode: a team without whole-list LDS keys");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef FRCNN_NMS_STAMPS
    unsigned long long st_acc[16] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif
    // (every pointer is smem + a byte offset: an alignment step on the POINTER's integer value -- the form up to round 5 -- made the
    // compiler lose the LDS address space, and the select histogram, the round's keys and the staged score keys were reached with
    // flat_load / flat_store / flat_atomic instructions: the bitonic network's 45-55 dependent stages each paid a flat round trip)
    constexpr size_t CAP = TEAM ? NMS_TEAM_N : NMS_CH;        // (the chunk loop uses the first NMS_CH entries of these arrays)
    const size_t o_kept_key = (size_t)p.max_per_class * 16;
    const size_t o_chunk_box = (o_kept_key + (size_t)p.max_per_class * 8 + 15) & ~(size_t)15;     // (16-byte vectors: an odd max_per_class must not misalign them)
    const size_t o_sup = o_chunk_box + CAP * 16;
    const size_t o_kept_area = o_sup + CAP * (CAP / 64) * 8;
    const size_t o_chunk_area = o_kept_area + (size_t)p.max_per_class * 4;                       // (4-byte entries from here to o_hist, which is re-aligned)
    const size_t o_dead = o_chunk_area + CAP * 4;
    const size_t o_rows = o_dead + CAP * 4;
    const size_t o_misc = o_rows + CAP * 4;
    const size_t o_hist = (o_misc + 64 + 15) & ~(size_t)15;
    const size_t o_rkeys = o_hist + (size_t)NMS_HC * 256 * 4;
    const size_t o_skeys = o_rkeys + (size_t)NMS_RK * 8;
    f32x4* kept_box = reinterpret_cast<f32x4*>(smem);                                       // [max_per_class] corner-normalised
    unsigned long long* kept_key = reinterpret_cast<unsigned long long*>(smem + o_kept_key);   // [max_per_class] composite keys of the kept
    f32x4* chunk_box = reinterpret_cast<f32x4*>(smem + o_chunk_box);                        // [NMS_CH] corner-normalised
    unsigned long long* sup_of = reinterpret_cast<unsigned long long*>(smem + o_sup);       // [NMS_CH][4] later candidates suppressed by row
    float* kept_area = reinterpret_cast<float*>(smem + o_kept_area);                        // [max_per_class]
    float* chunk_area = reinterpret_cast<float*>(smem + o_chunk_area);                      // [NMS_CH]
    int* dead = reinterpret_cast<int*>(smem + o_dead);                                      // [NMS_CH] invalid or suppressed by the kept list
    int* rows = reinterpret_cast<int*>(smem + o_rows);                                      // [NMS_CH] compacted list of surviving rows
    int* misc = reinterpret_cast<int*>(smem + o_misc);                                      // [16] scalars shared through LDS
    int* hist = reinterpret_cast<int*>(smem + o_hist);                                      // [NMS_HC][256], 16-byte aligned
    unsigned long long* rkeys = reinterpret_cast<unsigned long long*>(smem + o_rkeys);      // [NMS_RK]
    unsigned int* skeys = reinterpret_cast<unsigned int*>(smem + o_skeys);                  // [N] (LDSK)

    const int unit = TEAM ? (int)blockIdx.x / NMS_TEAM : (int)blockIdx.x;        // the (image, class) list
    const int team_rank = TEAM ? (int)blockIdx.x % NMS_TEAM : 0;
    const int b = unit / p.C, c = unit % p.C;
    const int bc = (p.q == 1) ? 0 : c;
    const float* boxes = p.boxes + (int64_t)b * p.N * p.q * 4;
    const float* scores = p.scores + (int64_t)b * p.N * p.score_stride + p.score_offset + c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const IouThr thr = iou_thr_of(p.iou_thr);

    unsigned int gen0 = 0u;                                   // team mode: this launch's number (see the hand-off below)
    if (TEAM) gen0 = __hip_atomic_load(p.team_sync + (size_t)unit * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // split mode: this workgroup's candidates [lo, hi) while split_phase (the team round); afterwards, and in every other mode, all of them
    bool split_phase = SPLIT;
    const int s0 = SPLIT ? min(team_rank * p.split_len, p.N) : 0, s1 = SPLIT ? min(s0 + p.split_len, p.N) : p.N;
    int lo = s0, hi = s1;
    auto score_key = [&](const int i) -> unsigned int {       // 0: not a candidate (score <= threshold)
        if (LDSK) return skeys[i];
        if (SPLIT && split_phase) return skeys[i - s0];
        const float s = scores[(int64_t)i * p.score_stride];
        return s > p.score_thr ? float_key(s) : 0u;
    };
    // all-to-all exchange `stage` of a split launch: every member has published its part of team_xch (agent-scope relaxed atomic stores)
    unsigned int* xch = SPLIT ? p.team_xch + (size_t)unit * NMS_XCH_WORDS : nullptr;
    auto exchange = [&](const int stage) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned int* flags = p.team_xflag + ((size_t)unit * NMS_XSTAGES + stage) * NMS_TEAM;
        if (threadIdx.x == 0) __hip_atomic_store(flags + team_rank, gen0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < NMS_TEAM) {
            long spins = 0;
            while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen0 + 1u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1l << 24)) __builtin_trap();   // (seconds: a member that never ran -- abort loudly rather than hang)
            }
        }
        __syncthreads();
    };
    auto composite = [](const unsigned int sk, const int i) -> unsigned long long {
        return ((unsigned long long)sk << 32) | (unsigned int)(~(unsigned int)i);
    };

    if (threadIdx.x < 16) misc[threadIdx.x] = 0;
    __syncthreads();
    {
        int valid = 0;
        for (int i = lo + threadIdx.x; i < hi; i += NMS_T) {
            const float s = scores[(int64_t)i * p.score_stride];
            const unsigned int sk = s > p.score_thr ? float_key(s) : 0u;
            if (LDSK) skeys[i] = sk;
            if (SPLIT) skeys[i - s0] = sk;
            valid += sk != 0u;
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) valid += __shfl_xor(valid, sh);
        if (lane == 0 && valid) atomicAdd(&misc[2], valid);
    }
    __syncthreads();
    if (SPLIT) {                                              // exchange 0: the members' valid counts
        if (threadIdx.x == 0) __hip_atomic_store(xch + team_rank, (unsigned int)misc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        exchange(0);
        if (threadIdx.x == 0) {
            int total = 0;
            for (int g = 0; g < NMS_TEAM; ++g) total += (int)__hip_atomic_load(xch + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            misc[2] = total;
        }
        __syncthreads();
    }
    NMS_STAMP(0);
    int remaining = misc[2];
    unsigned long long prev = ~0ull;                          // candidates with a composite key below prev are still unvisited
    int kept = 0;
    bool first_round = true;
    while (remaining > 0 && kept < p.max_per_class) {
        NMS_COUNT(8);
        // a round takes the next K_goal best candidates -- or, as soon as a digit boundary leaves at least half of that above it, the ones
        // above that boundary: the rounds only have to partition the keys into descending ranges, not into equal ones, and every pass of
        // the select that is not run saves a sweep over all N candidates.  The first round aims at 512: a list that suppresses little
        // (max_per_class = 300 of an untrained RPN's 8768) ends inside it, one that suppresses much pays one short round more.
        const int K_goal = first_round ? NMS_RK / 2 : NMS_RK;
        const bool team_round = TEAM && first_round;
        first_round = false;
        int K = remaining < K_goal ? remaining : K_goal;
        unsigned long long T = 1ull;                          // this round takes the keys in [T, prev)
        if (remaining > K_goal) {
            // ---- radix select: T = the K-th largest unvisited composite key (or an earlier digit boundary, see above)
            unsigned long long prefix = 0ull;
            int want = K;
            bool found = false;
            for (int pass = 0; pass < 8 && !found; ++pass) {
                const int shift = 56 - 8 * pass;
                for (int t = threadIdx.x; t < NMS_HC * 256; t += NMS_T) hist[t] = 0;
                __syncthreads();
                NMS_COUNT(14);
                NMS_STAMP(1);
                for (int i4 = lo; i4 < hi; i4 += 4 * NMS_T) {     // (all lanes stay in the loop: ballots below)
                  // four keys per thread in flight (a key per trip left every trip waiting for its own LDS / global round trip:
                  // N / 1024 dependent latencies per pass -- 80 global ones for the pyramid's 82 k candidates)
                  unsigned int sk4[4];
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                      const int i = i4 + u * NMS_T + threadIdx.x;
                      sk4[u] = i < hi ? score_key(i) : 0u;
                  }
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                    if (i4 + u * NMS_T >= hi) break;              // (uniform)
                    const int i = i4 + u * NMS_T + threadIdx.x;
                    const unsigned int sk = sk4[u];
                    const unsigned long long k = composite(sk, i);
                    const bool in = sk != 0u && k < prev && (pass == 0 || (k >> (shift + 8)) == prefix);
                    const int dg = (int)((k >> shift) & 255ull);
                    // one histogram per wave (NMS_HC = 16 copies = 16 waves): no atomic ever meets another wave's; equal digits
                    // inside a wave instruction (scores cluster: an untrained RPN puts every objectness near 0.5) are
                    // serialised by the LDS unit itself, a few cycles each
#ifndef FRCNN_NMS_NOPEEL
                    // ... unless most of the wave shares a digit (the exponent byte of scores that all lie in [0.5, 1): a 64-way same-
                    // address atomic occupies the LDS unit for 64+ cycles, and there are N / 64 of them per pass): up to two groups of
                    // 16+ equal digits are counted by one lane each
                    unsigned long long todo = __ballot(in);
                    bool mine = in;
#pragma unroll
                    for (int peel = 0; peel < 2; ++peel) {
                        if (todo == 0ull) break;
                        const int leader = __builtin_ctzll(todo);
                        const int d0 = __builtin_amdgcn_readlane(dg, leader);
                        const unsigned long long same = __ballot(mine && dg == d0);
                        if (__popcll(same) < 16) break;
                        if (lane == leader) atomicAdd(&hist[wave * 256 + d0], __popcll(same));
                        mine = mine && dg != d0;
                        todo &= ~same;
                    }
                    if (mine) atomicAdd(&hist[wave * 256 + dg], 1);
#else
                    if (in) atomicAdd(&hist[wave * 256 + dg], 1);
#endif
                  }
                }
                __syncthreads();
                if (SPLIT && split_phase) {                   // exchange 1 + pass: every member's 256 digit counts of this pass
                    unsigned int* xh = xch + 8 + (size_t)pass * NMS_TEAM * 256;
                    if (threadIdx.x < 256) {
                        int sum = 0;
#pragma unroll
                        for (int h = 0; h < NMS_HC; ++h) sum += hist[h * 256 + threadIdx.x];
                        __hip_atomic_store(xh + team_rank * 256 + threadIdx.x, (unsigned int)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    exchange(1 + pass);
                    for (int t = threadIdx.x; t < NMS_HC * 256; t += NMS_T)      // the members' counts as the first 8 of the 16 copies
                        hist[t] = t < NMS_TEAM * 256 ? (int)__hip_atomic_load(xh + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    __syncthreads();
                }
                NMS_STAMP(15);
                if (threadIdx.x < 64) {                       // the digit d with  #(digits above d) < want <= #(digits >= d)
                    // (copy-major layout hist[wave][256]: a wave's atomics spread over the banks by digit, and a lane's four digits of
                    // one copy are one 16-byte read, consecutive over the lanes -- the digit-major layout put a wave's atomics on two
                    // banks and this loop's 64 reads on one)
                    int cnt[4] = {0, 0, 0, 0}, mine = 0;
#pragma unroll
                    for (int h = 0; h < NMS_HC; ++h) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(&hist[h * 256 + lane * 4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) cnt[e] += (int)v[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) mine += cnt[e];
                    int incl = mine;                          // inclusive suffix sum over the lanes (Hillis-Steele, doubling)
#pragma unroll
                    for (int sh = 1; sh < 64; sh <<= 1) {
                        const int v = __shfl_down(incl, sh);
                        if (lane + sh < 64) incl += v;
                    }
                    int running = incl - mine;                // candidates in the digits of higher lanes
#pragma unroll
                    for (int e = 3; e >= 0; --e) {
                        if (running < want && running + cnt[e] >= want) {
                            misc[3] = lane * 4 + e;
                            misc[4] = running;
                            misc[5] = cnt[e];
                        }
                        running += cnt[e];
                    }
                }
                __syncthreads();
                const unsigned long long above = ((prefix << 8) + (unsigned long long)misc[3] + 1ull) << shift;   // first key above this digit
                prefix = (prefix << 8) | (unsigned long long)misc[3];
                want -= misc[4];
                if (want == misc[5] || pass == 7) {           // the boundary does not split this digit: every key with this prefix is taken
                    T = prefix << shift;
                    found = true;
                } else if (K - want >= K_goal / 2) {          // enough candidates lie above the digit the boundary falls into: take those
                    T = above;                                // (K - want > 0: some digit above misc[3] is occupied, so `above` did not wrap)
                    K -= want;
                    found = true;
                }
                __syncthreads();
            }
        }
        NMS_STAMP(1);
        // ---- compact the round's keys, pad, sort
        if (threadIdx.x == 0) misc[6] = 0;
        const int n_sort = K <= NMS_CH ? NMS_CH : K <= 512 ? 512 : NMS_RK;
        for (int t = threadIdx.x; t < n_sort; t += NMS_T) rkeys[t] = 0ull;
        __syncthreads();
        for (int i4 = lo; i4 < hi; i4 += 4 * NMS_T) {
            unsigned int sk4[4];                              // (four keys in flight, as in the select's sweeps)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i4 + u * NMS_T + threadIdx.x;
                sk4[u] = i < hi ? score_key(i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i4 + u * NMS_T + threadIdx.x;
                const unsigned long long k = composite(sk4[u], i);
                const bool take = sk4[u] != 0u && k >= T && k < prev;
                const unsigned long long m = __ballot(take);  // one returning atomic per wave, not per candidate
                if (m != 0ull) {
                    const int leader = __builtin_ctzll(m);
                    int base = 0;
                    if (lane == leader) base = atomicAdd(&misc[6], __popcll(m));
                    base = __builtin_amdgcn_readlane(base, leader);
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (take && pos < NMS_RK) rkeys[pos] = k; // (pos < K always; the guard keeps a logic error from corrupting LDS)
                }
            }
        }
        __syncthreads();
        if (SPLIT && split_phase) {                           // exchange 9: every member's keys of the round
            unsigned int* xk = xch + 8 + (size_t)8 * NMS_TEAM * 256;
            const int mine_n = misc[6];
            if (threadIdx.x == 0) __hip_atomic_store(xk + team_rank, (unsigned int)mine_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int t = threadIdx.x; t < mine_n; t += NMS_T) {
                const unsigned long long k = rkeys[t];
                __hip_atomic_store(xk + 8 + ((size_t)team_rank * NMS_TEAM_N + t) * 2, (unsigned int)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(xk + 8 + ((size_t)team_rank * NMS_TEAM_N + t) * 2 + 1, (unsigned int)(k >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            exchange(9);
            for (int t = threadIdx.x; t < n_sort; t += NMS_T) rkeys[t] = 0ull;
            __syncthreads();
            int off = 0;
            for (int g = 0; g < NMS_TEAM; ++g) {              // (uniform: every thread reads the eight counts)
                const int ng = (int)__hip_atomic_load(xk + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int t = threadIdx.x; t < ng && off + t < NMS_RK; t += NMS_T) {
                    const unsigned int klo = __hip_atomic_load(xk + 8 + ((size_t)g * NMS_TEAM_N + t) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned int khi = __hip_atomic_load(xk + 8 + ((size_t)g * NMS_TEAM_N + t) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rkeys[off + t] = ((unsigned long long)khi << 32) | klo;
                }
                off += ng;
            }
            __syncthreads();
        }
        NMS_STAMP(2);
        bitonic_desc_reg(rkeys, n_sort);
        NMS_STAMP(3);

        if (TEAM && team_round) {
            // ================= the team round: n_sort <= NMS_TEAM_N candidates, kept == 0
            constexpr int NBM = NMS_TEAM_N / 64;                  // matrix words per row (capacity)
            const int NB = n_sort >> 6;                           // ... in use
            for (int t = threadIdx.x; t < n_sort; t += NMS_T) {
                const unsigned long long k = rkeys[t];
                f32x4 bx = {0.f, 0.f, 0.f, 0.f};
                if (k != 0ull) {
                    const unsigned int idx = ~(unsigned int)(k & 0xFFFFFFFFull);
                    bx = nms_norm(*reinterpret_cast<const f32x4*>(boxes + ((int64_t)idx * p.q + bc) * 4));
                }
                chunk_box[t] = bx;
                chunk_area[t] = nms_area(bx);
                dead[t] = (k == 0ull) ? 1 : 0;
            }
            __syncthreads();
            NMS_STAMP(4);
            unsigned long long* mat = p.team_mat + (size_t)unit * NMS_TEAM_N * NBM;
            {
                // this workgroup's rows: i = team_rank + NMS_TEAM * (wave, wave + 16, ...); a lane's column candidates (j = 64 w + lane) in
                // registers; lane w of the row's wave ends with word w and the row leaves as one 64-byte store
                const int rpw = n_sort / NMS_TEAM;
#pragma unroll 1
                for (int h = 0; h * 4 < NB; ++h) {                // four column words at a time (all eight in registers: spills)
                    f32x4 cb[4];
                    float ca[4];
                    bool cdead[4];
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) {
                        const int j = (h * 4 + w4) * 64 + lane;   // (NB is 4 or 8: a started group of four is complete)
                        cb[w4] = chunk_box[j];
                        ca[w4] = chunk_area[j];
                        cdead[w4] = dead[j] != 0;
                    }
                    for (int ri = wave; ri < rpw; ri += NMS_T / 64) {
                        // COLUMN form: word w of candidate i = which EARLIER candidates 64 w .. 64 w + 63 would suppress it (the overlap
                        // test is symmetric, so this is the row form's test with the roles swapped): the walk below then needs no
                        // cross-lane traffic at all -- a lane asks "does anybody kept so far suppress MY candidate"
                        const int i = ri * NMS_TEAM + team_rank;  // (wave-uniform; interleaved: a candidate late in the round has up to 8 words
                                                                  //  to test, an early one 1 -- every member and wave gets a mix)
                        const f32x4 rb = chunk_box[i];
                        const float ra = chunk_area[i];
                        const bool rdead = dead[i] != 0;
                        unsigned long long mine = 0ull;
#pragma unroll
                        for (int w4 = 0; w4 < 4; ++w4) {
                            const int w = h * 4 + w4;
                            if (w > (i >> 6)) continue;           // (uniform) every candidate of this word comes after i
                            const int j = w * 64 + lane;
                            const bool sgt = !rdead && j < i && !cdead[w4] && nms_over(cb[w4], ca[w4], rb, ra, thr);
                            const unsigned long long m = __ballot(sgt);
                            if (lane == w) mine = m;
                        }
                        if ((lane >> 2) == h) __hip_atomic_store(mat + (size_t)i * NBM + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            // hand-off as in conv_tile.hip's split-K fix-up (the form MI355X_MICROARCH.md lists as measured-valid on gfx950): payload by
            // agent-scope relaxed atomic stores (write-through), drained by every wave's vmcnt(0); a workgroup barrier; ONE lane's
            // agent-scope counter add.  The reader loads the payload with agent-scope relaxed atomic loads behind its own barrier.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // Arrival words that need no zeroing (a memset in front of the kernel is a 5 us launch of its own): word 0 of the list's 16 is
            // its launch count `gen`, read by every workgroup of the team at the start of the kernel and advanced by workgroup 0 once
            // all helpers have arrived (so after all of them have read it); helper g arrives by storing gen + 1 into word g.  Words
            // left by earlier launches hold values <= gen; a buffer that was never used works whatever it holds, unless a word
            // happens to equal its neighbour's garbage + 1 (2^-32 per word, on the first launch only -- frcnn_nms_workspace_bytes asks
            // callers to zero the buffer once for that reason).
            unsigned int* sync = p.team_sync + (size_t)unit * 16;
            if (team_rank != 0) {
                if (threadIdx.x == 0) __hip_atomic_store(sync + team_rank, gen0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;                                           // the helpers are done
            }
            NMS_STAMP(5);
            if (threadIdx.x < NMS_TEAM && threadIdx.x > 0) {
                long spins = 0;
                while (__hip_atomic_load(sync + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen0 + 1u) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1l << 24)) __builtin_trap();   // (seconds: a helper that never ran -- abort loudly rather than hang)
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(sync, gen0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            NMS_STAMP(11);
            for (int t = threadIdx.x; t < n_sort * NBM; t += NMS_T) sup_of[t] = __hip_atomic_load(mat + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            NMS_STAMP(12);
            // ---- resolve the round in score order (wave 0), one block of 64 candidates at a time, from the column form: lane l of block r
            // knows which earlier candidates of the block suppress its own (word r of its column).  "Kept" inside the block is the fixed
            // point of  K <- alive & ~(somebody in K suppresses me)  started from K = alive: after t rounds the first t candidates of
            // the block are final (a candidate depends on earlier ones only), so at most 64 rounds and in practice the depth of the
            // longest suppression chain -- one ballot each, no cross-lane reads.  The kept set then strikes the later blocks' candidates
            // the same way (one ballot per block).  (The row-form walk this replaces took one dependent scalar chain of ~150 cycles per
            // KEPT candidate: 15 of the kernel's 50 us.)
            if (threadIdx.x < 64) {
                unsigned long long alive[NBM];
                unsigned long long* kmask_of = reinterpret_cast<unsigned long long*>(rows);     // [NBM] which candidates of a block are kept (`rows` is idle in this round)
#pragma unroll
                for (int r = 0; r < NBM; ++r) {
                    alive[r] = __ballot(r < NB && dead[(r < NB ? r : 0) * 64 + lane] == 0);
                    if (lane == 0) kmask_of[r] = 0ull;
                }
                int k_now = 0;
#pragma unroll
                for (int r = 0; r < NBM; ++r) {
                    if (r >= NB || alive[r] == 0ull || k_now >= p.max_per_class) continue;      // (uniform)
                    unsigned long long col[NBM];                  // word r of the columns of this lane's candidates in blocks w >= r
#pragma unroll
                    for (int w = r; w < NBM; ++w) col[w] = w < NB ? sup_of[(size_t)(w * 64 + lane) * NBM + r] : 0ull;
                    const bool mine_alive = (alive[r] >> lane) & 1ull;
                    unsigned long long K = alive[r];
                    for (int it = 0; it < 64; ++it) {
                        const unsigned long long Kn = __ballot(mine_alive && (col[r] & K) == 0ull);
                        if (Kn == K) break;
                        K = Kn;
                    }
                    int cnt = __popcll(K);
                    if (k_now + cnt > p.max_per_class) {          // the cap falls inside the block: its first (cap - kept) members
                        unsigned long long t = K, sel = 0ull;
                        for (int need = p.max_per_class - k_now; need > 0; --need) {
                            sel |= t & (~t + 1ull);
                            t &= t - 1ull;
                        }
                        K = sel;
                        cnt = __popcll(K);
                    }
                    k_now += cnt;
                    if (lane == 0) kmask_of[r] = K;
#pragma unroll
                    for (int w = r + 1; w < NBM; ++w) alive[w] &= ~__ballot((col[w] & K) != 0ull);
                }
                int before = 0;
                for (int r = 0; r < NB; ++r) {
                    const unsigned long long km = kmask_of[r];        // (written by lane 0 of this wave: the LDS unit keeps a wave's order)
                    if ((km >> lane) & 1ull) {
                        const int slot = before + __popcll(km & ((1ull << lane) - 1ull));
                        kept_box[slot] = chunk_box[r * 64 + lane];
                        kept_area[slot] = chunk_area[r * 64 + lane];
                        const unsigned long long k = rkeys[r * 64 + lane];
                        kept_key[slot] = k;
                        if (!p.out_boxes) {                   // (a single class writes the final outputs itself, from kept_key: no merge launch reads these)
                            const int64_t o = (int64_t)b * p.C * p.max_per_class + (int64_t)c * p.max_per_class + slot;
                            p.kept_keys[o] = (k & 0xFFFFFFFF00000000ull) | (unsigned int)(~(unsigned int)(c * p.max_per_class + slot));
                            p.kept_idx[o] = (int)(~(unsigned int)(k & 0xFFFFFFFFull));
                        }
                    }
                    before += __popcll(km);
                }
                if (lane == 0) misc[0] = k_now;
            }
            __syncthreads();
            NMS_STAMP(6);
            kept = misc[0];
            prev = T;
            remaining -= K;
            if (SPLIT) {                                          // later rounds (workgroup 0 alone): all N candidates, from global memory
                split_phase = false;
                lo = 0;
                hi = p.N;
            }
            __syncthreads();
            continue;
        }

        for (int base = 0; base < n_sort; base += NMS_CH) {
            NMS_COUNT(9);
            // ---- 1. load the chunk
            if (threadIdx.x < NMS_CH) {
                const unsigned long long k = rkeys[base + threadIdx.x];
                f32x4 bx = {0.f, 0.f, 0.f, 0.f};
                if (k != 0ull) {
                    const unsigned int idx = ~(unsigned int)(k & 0xFFFFFFFFull);
                    bx = nms_norm(*reinterpret_cast<const f32x4*>(boxes + ((int64_t)idx * p.q + bc) * 4));
                }
                chunk_box[threadIdx.x] = bx;
                chunk_area[threadIdx.x] = nms_area(bx);
                dead[threadIdx.x] = (k == 0ull) ? 1 : 0;
            }
            __syncthreads();
            if (rkeys[base] == 0ull) break;       // sorted: nothing valid left in this round (uniform: same value for all threads)
            // ---- 2. test against the kept list: 4 threads per candidate
            {
                const int cand = threadIdx.x >> 2, sub = threadIdx.x & 3;
                const f32x4 cb = chunk_box[cand];
                const float ca = chunk_area[cand];
                int hit = 0;
                for (int j0 = sub; j0 < kept && !hit; j0 += 16) {          // four independent tests per trip (LDS latency overlaps)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + 4 * u;
                        if (j < kept && nms_over(cb, ca, kept_box[j], kept_area[j], thr)) hit = 1;
                    }
                }
                hit |= __shfl_xor(hit, 1);
                hit |= __shfl_xor(hit, 2);
                if (sub == 0 && hit) dead[cand] = 1;
            }
            __syncthreads();
            NMS_STAMP(4);
            // ---- 3. compact the survivors (wave 0), then build their rows of the suppression matrix
            if (threadIdx.x < 64) {
                int n_alive = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool al = dead[r * 64 + lane] == 0;
                    const unsigned long long m = __ballot(al);
                    if (al) rows[n_alive + __popcll(m & ((1ull << lane) - 1ull))] = r * 64 + lane;
                    n_alive += __popcll(m);
                }
                if (lane == 0) misc[1] = n_alive;
            }
            __syncthreads();
            NMS_STAMP(11);
            const int n_alive = misc[1];
            {
                // a lane's four column candidates (j = 64 w + lane) stay in registers for all the rows of its wave; a row costs
                // one broadcast read of its box and four independent overlap tests (the per-(row, word) loop with its chain of
                // dependent LDS reads was the largest single piece of this kernel)
                f32x4 cb[4];
                float ca[4];
                bool cdead[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    cb[w] = chunk_box[w * 64 + lane];
                    ca[w] = chunk_area[w * 64 + lane];
                    cdead[w] = dead[w * 64 + lane] != 0;
                }
                NMS_STAMP(12);
                for (int ri = wave; ri < n_alive; ri += NMS_T / 64) {
                    const int i = rows[ri];
                    const f32x4 rb = chunk_box[i];
                    const float ra = chunk_area[i];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (w < (i >> 6)) continue;           // (wave-uniform) every column of this word precedes row i: never read
                        const int j = w * 64 + lane;
                        const bool sgt = j > i && !cdead[w] && nms_over(cb[w], ca[w], rb, ra, thr);
                        const unsigned long long m = __ballot(sgt);
                        if (lane == 0) sup_of[i * 4 + w] = m;
                    }
                }
                NMS_STAMP(13);
            }
            __syncthreads();
            NMS_STAMP(5);
            // ---- 4. walk the survivors in score order (wave 0): one step per kept box
            if (threadIdx.x < 64) {
                unsigned long long alive[4], kmask[4];
                unsigned int slo[4][4], shi[4][4];              // row r*64+lane, word w (only w >= r is ever read)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool al = dead[r * 64 + lane] == 0;
                    alive[r] = __ballot(al);
                    kmask[r] = 0ull;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned long long v = (w >= r && al) ? sup_of[(r * 64 + lane) * 4 + w] : 0ull;
                        slo[r][w] = (unsigned int)v;
                        shi[r][w] = (unsigned int)(v >> 32);
                    }
                }
                // rows that suppress nobody (the common case once overlapping boxes are gone) need no matrix lookup: a whole run
                // of such survivors is kept in one step; only a survivor with a non-empty row costs the 2 x (4 - r) v_readlane
                unsigned long long nz[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    unsigned int any = 0u;
#pragma unroll
                    for (int w = 0; w < 4; ++w) any |= slo[r][w] | shi[r][w];
                    nz[r] = __ballot(any != 0u);
                }
                int k_now = kept;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    while (alive[r] != 0ull && k_now < p.max_per_class) {
                        const unsigned long long az = alive[r] & nz[r];
                        const int pos = az != 0ull ? __builtin_ctzll(az) : 64;
                        unsigned long long run = pos < 64 ? (alive[r] & ((1ull << pos) - 1ull)) : alive[r];
                        int c = __popcll(run);
                        if (k_now + c > p.max_per_class) {               // the cap falls inside the run: its first (cap - kept) members
                            unsigned long long t = run, sel = 0ull;
                            for (int need = p.max_per_class - k_now; need > 0; --need) {
                                sel |= t & (~t + 1ull);
                                t &= t - 1ull;
                            }
                            run = sel;
                            c = __popcll(run);
                        }
                        kmask[r] |= run;
                        k_now += c;
                        alive[r] &= ~run;
                        if (pos == 64 || k_now >= p.max_per_class) continue;
                        kmask[r] |= 1ull << pos;
                        ++k_now;
                        alive[r] &= ~(1ull << pos);
#pragma unroll
                        for (int w = r; w < 4; ++w) {
                            const unsigned long long sp = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)shi[r][w], pos) << 32) |
                                                          (unsigned int)__builtin_amdgcn_readlane((int)slo[r][w], pos);
                            alive[w] &= ~sp;
                        }
                    }
                }
                int before = kept;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((kmask[r] >> lane) & 1ull) {
                        const int slot = before + __popcll(kmask[r] & ((1ull << lane) - 1ull));
                        kept_box[slot] = chunk_box[r * 64 + lane];
                        kept_area[slot] = chunk_area[r * 64 + lane];
                        const unsigned long long k = rkeys[base + r * 64 + lane];
                        kept_key[slot] = k;
                        if (!p.out_boxes) {                   // (a single class writes the final outputs itself, from kept_key: no merge launch reads these)
                            const int64_t o = (int64_t)b * p.C * p.max_per_class + (int64_t)c * p.max_per_class + slot;
                            p.kept_keys[o] = (k & 0xFFFFFFFF00000000ull) | (unsigned int)(~(unsigned int)(c * p.max_per_class + slot));
                            p.kept_idx[o] = (int)(~(unsigned int)(k & 0xFFFFFFFFull));
                        }
                    }
                    before += __popcll(kmask[r]);
                }
                if (lane == 0) misc[0] = k_now;
            }
            __syncthreads();
            NMS_STAMP(6);
            kept = misc[0];
            if (kept >= p.max_per_class) break;
        }
        prev = T;
        remaining -= K;
        __syncthreads();
    }
    if (TEAM && team_rank != 0) return;                      // (a list without candidates: no round was run)
    // zero the unused kept slots of this (image, class)
    for (int s = kept + threadIdx.x; s < p.max_per_class && !p.out_boxes; s += blockDim.x) {
        const int64_t o = (int64_t)b * p.C * p.max_per_class + (int64_t)c * p.max_per_class + s;
        p.kept_keys[o] = 0ull;
        p.kept_idx[o] = 0;
    }
    if (p.out_boxes) {
        // single class: slots are already in the merged order (descending score, earlier slot first among equals); what
        // nms_merge_kernel would emit, from this workgroup's own kept list (written by wave 0 above, in LDS: barrier)
        __syncthreads();
        const int nv = kept < p.max_total ? kept : p.max_total;
        for (int t = threadIdx.x; t < p.max_total; t += blockDim.x) {
            f32x4 bx = {0.f, 0.f, 0.f, 0.f};
            float s = 0.f;
            if (t < nv) {                                     // (from the LDS copy of the kept keys: no global store -> load round trip)
                const unsigned long long k = kept_key[t];
                const int idx = (int)(~(unsigned int)(k & 0xFFFFFFFFull));
                bx = *reinterpret_cast<const f32x4*>(boxes + ((int64_t)idx * p.q + bc) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) bx[e] = fminf(fmaxf(bx[e], 0.0f), 1.0f);
                s = key_float((unsigned int)(k >> 32));
            }
            *reinterpret_cast<f32x4*>(p.out_boxes + ((int64_t)b * p.max_total + t) * 4) = bx;
            if (p.out_abs) *reinterpret_cast<f32x4*>(p.out_abs + ((int64_t)b * p.max_total + t) * 4) = f32x4{bx[0] * p.sx, bx[1] * p.sy, bx[2] * p.sx, bx[3] * p.sy};
            p.out_scores[(int64_t)b * p.max_total + t] = s;
            p.out_classes[(int64_t)b * p.max_total + t] = 0;
        }
        if (threadIdx.x == 0) p.out_valid[b] = nv;
    }
#ifdef FRCNN_NMS_STAMPS
    __syncthreads();
    NMS_STAMP(7);
    if (threadIdx.x == 0 && unit < 8) {
        st_acc[10] = (unsigned long long)kept;
        for (int i = 0; i < 16; ++i) g_nms_stamps[p.C == 1 ? 0 : 1][unit][i] = st_acc[i];
    }
#endif
}
#undef NMS_STAMP
#undef NMS_COUNT

struct MergeParams {
    const float* boxes; const unsigned long long* kept_keys; const int* kept_idx;
    int N, q, C, max_per_class, max_total, m_pad;
    float* out_boxes; float* out_scores; int* out_classes; int* out_valid;
    float* out_abs; float sx, sy;
};

__global__ __launch_bounds__(NMS_T) void nms_merge_kernel(const MergeParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    int& nvalid = *reinterpret_cast<int*>(keys + p.m_pad);
    const int b = blockIdx.x;
    const int total = p.C * p.max_per_class;
    if (threadIdx.x == 0) nvalid = 0;
    for (int i = threadIdx.x; i < p.m_pad; i += blockDim.x) keys[i] = i < total ? p.kept_keys[(int64_t)b * total + i] : 0ull;
    __syncthreads();
    bitonic_desc(keys, p.m_pad);
    int cnt = 0;
    for (int t = threadIdx.x; t < p.max_total; t += blockDim.x) {
        const unsigned long long k = keys[t];
        f32x4 bx = {0.f, 0.f, 0.f, 0.f};
        float s = 0.f;
        int cls = 0;
        if (k != 0ull) {
            const unsigned int cs = ~(unsigned int)(k & 0xFFFFFFFFull);          // class*max_per_class + slot
            cls = (int)(cs / (unsigned)p.max_per_class);
            const int idx = p.kept_idx[(int64_t)b * total + cs];
            const int bc = (p.q == 1) ? 0 : cls;
            bx = *reinterpret_cast<const f32x4*>(p.boxes + (((int64_t)b * p.N + idx) * p.q + bc) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) bx[e] = fminf(fmaxf(bx[e], 0.0f), 1.0f);
            s = key_float((unsigned int)(k >> 32));
            ++cnt;
        }
        *reinterpret_cast<f32x4*>(p.out_boxes + ((int64_t)b * p.max_total + t) * 4) = bx;
        if (p.out_abs) *reinterpret_cast<f32x4*>(p.out_abs + ((int64_t)b * p.max_total + t) * 4) = f32x4{bx[0] * p.sx, bx[1] * p.sy, bx[2] * p.sx, bx[3] * p.sy};
        p.out_scores[(int64_t)b * p.max_total + t] = s;
        p.out_classes[(int64_t)b * p.max_total + t] = cls;
    }
    if (cnt) atomicAdd(&nvalid, cnt);
    __syncthreads();
    if (threadIdx.x == 0) p.out_valid[b] = nvalid;
}

int next_pow2(int v) { int p = 64; while (p < v) p <<= 1; return p; }

}  // namespace

#define S_(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int frcnn_anchors_generate(float* anchors, int gh, int gw, const float* scales, int ns, const float* ratios, int nr,
                                      float base_h, float base_w, float stride_h, float stride_w, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(anchors && scales && ratios && ns * nr <= 64 && ns > 0 && nr > 0, "anchors_generate: bad arguments");
    AnchorShapes sh;
    sh.na = ns * nr;
    for (int r = 0; r < nr; ++r)
        for (int s = 0; s < ns; ++s) {
            // rpn_detector.py:182-184 in fp32: heights = scales / sqrt(ratio) * base[0]; widths = scales * sqrt(ratio) * base[1]
            const float rs = sqrtf(ratios[r]);
            volatile float hq = scales[s] / rs;
            volatile float wq = scales[s] * rs;
            sh.h[r * ns + s] = hq * base_h;
            sh.w[r * ns + s] = wq * base_w;
        }
    const int total = gh * gw * sh.na;
    hipLaunchKernelGGL(anchors_kernel, dim3(cdiv(total, 256)), dim3(256), 0, S_(stream), anchors, gh, gw, sh, stride_h, stride_w);
    FRCNN_CHECK_LAUNCH("anchors_generate");
    return FRCNN_OK;
}

extern "C" int frcnn_rpn_head_post_decode(const float* head, int ld, int b, int num_anchors_total, int a_per_loc, const int32_t* keep, int n,
                                          float* scores, float* deltas, const float* regions, float* decoded, float img_w, float img_h,
                                          frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(head && scores && deltas && ld >= 6 * a_per_loc && num_anchors_total % a_per_loc == 0 && n > 0, "rpn_head_post: bad arguments");
    FRCNN_CHECK_ARG(!decoded || regions, "rpn_head_post_decode: decoded boxes need the regions");
    hipLaunchKernelGGL(rpn_head_post_kernel, dim3(cdiv((int64_t)b * n, 256)), dim3(256), 0, S_(stream), head, ld, b, num_anchors_total,
                       a_per_loc, keep, n, scores, deltas, regions, decoded, img_w, img_h, n, 0);
    FRCNN_CHECK_LAUNCH("rpn_head_post");
    return FRCNN_OK;
}

extern "C" int frcnn_rpn_head_post_level(const float* head, int ld, int b, int num_anchors_level, int a_per_loc, const int32_t* keep, int n,
                                         float* scores, float* deltas, const float* regions, float* decoded, float img_w, float img_h,
                                         int n_total, int offset, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(head && scores && deltas && ld >= 6 * a_per_loc && num_anchors_level % a_per_loc == 0 && n > 0 && offset >= 0 && offset + n <= n_total,
                    "rpn_head_post_level: bad arguments");
    FRCNN_CHECK_ARG(!decoded || regions, "rpn_head_post_level: decoded boxes need the regions");
    hipLaunchKernelGGL(rpn_head_post_kernel, dim3(cdiv((int64_t)b * n, 256)), dim3(256), 0, S_(stream), head, ld, b, num_anchors_level, a_per_loc, keep,
                       n, scores, deltas, regions, decoded, img_w, img_h, n_total, offset);
    FRCNN_CHECK_LAUNCH("rpn_head_post_level");
    return FRCNN_OK;
}

extern "C" int frcnn_rpn_head_post(const float* head, int ld, int b, int num_anchors_total, int a_per_loc, const int32_t* keep, int n,
                                   float* scores, float* deltas, frcnn_stream_t stream) {
    return frcnn_rpn_head_post_decode(head, ld, b, num_anchors_total, a_per_loc, keep, n, scores, deltas, nullptr, nullptr, 1.f, 1.f, stream);
}

extern "C" int frcnn_rcnn_head_post_decode(const float* logits, int ld, const float* bias, int r, int nc1, float* scores, float* deltas,
                                           const float* regions, float* decoded, float img_w, float img_h, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(logits && bias && scores && deltas && regions && decoded && nc1 >= 2 && nc1 <= 64 && ld >= nc1 + 4 * (nc1 - 1) && r > 0,
                    "rcnn_head_post_decode: bad arguments");
    hipLaunchKernelGGL(rcnn_head_post_decode_kernel, dim3(cdiv(r, 4)), dim3(256), 0, S_(stream), logits, ld, bias, r, nc1, scores, deltas, regions,
                       decoded, img_w, img_h);
    FRCNN_CHECK_LAUNCH("rcnn_head_post_decode");
    return FRCNN_OK;
}

extern "C" int frcnn_clip_to_window(const float* boxes, float* out, int64_t n, float x0, float y0, float x1, float y1, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(boxes && out, "clip_to_window: null pointer");
    hipLaunchKernelGGL(clip_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(stream), boxes, out, n, x0, y0, x1, y1);
    FRCNN_CHECK_LAUNCH("clip_to_window");
    return FRCNN_OK;
}

extern "C" int frcnn_boxes_scale(const float* in, float* out, int64_t n, float sx, float sy, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(in && out, "boxes_scale: null pointer");
    hipLaunchKernelGGL(scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(stream), in, out, n, sx, sy);
    FRCNN_CHECK_LAUNCH("boxes_scale");
    return FRCNN_OK;
}

extern "C" int frcnn_boxes_divide(const float* in, float* out, int64_t n, float w, float h, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(in && out && n >= 0, "boxes_divide: bad arguments");
    if (n == 0) return FRCNN_OK;
    hipLaunchKernelGGL(divide_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(stream), in, out, n, w, h);
    FRCNN_CHECK_LAUNCH("boxes_divide");
    return FRCNN_OK;
}

extern "C" int frcnn_encode_boxes(const float* boxes, const float* regions, int regions_per_image, float* out, int b, int r, int c,
                                  frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(boxes && regions && out && b > 0 && r > 0 && c > 0, "encode_boxes: bad arguments");
    hipLaunchKernelGGL(encode_kernel, dim3(cdiv((int64_t)b * r * c, 256)), dim3(256), 0, S_(stream), boxes, regions, regions_per_image, out, b, r, c);
    FRCNN_CHECK_LAUNCH("encode_boxes");
    return FRCNN_OK;
}

extern "C" int frcnn_decode_boxes(const float* regions, int regions_per_image, const float* deltas, float* out, int b, int r, int c,
                                  float img_w, float img_h, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(regions && deltas && out && b > 0 && r > 0 && c > 0, "decode_boxes: bad arguments");
    const int64_t total = (int64_t)b * r * c;
    hipLaunchKernelGGL(decode_kernel, dim3(cdiv(total, 256)), dim3(256), 0, S_(stream), regions, regions_per_image, deltas, out, b, r, c,
                       img_w, img_h);
    FRCNN_CHECK_LAUNCH("decode_boxes");
    return FRCNN_OK;
}

// team mode (see nms_class_kernel): single-class lists longer than the team round, few enough for every team to have its own CUs
static bool nms_team_mode(int b, int n, int c) { return c == 1 && n > NMS_TEAM_N && b * NMS_TEAM <= 128; }
static size_t nms_kept_bytes(int b, int c, int max_per_class) {
    return ((size_t)b * c * max_per_class * (sizeof(unsigned long long) + sizeof(int)) + 255) & ~(size_t)255;
}

extern "C" size_t frcnn_nms_workspace_bytes(int b, int n, int c, int max_per_class, int max_total) {
    (void)max_total;
    size_t bytes = nms_kept_bytes(b, c, max_per_class);
    if (nms_team_mode(b, n, c)) {
        bytes += (size_t)b * 64 + (size_t)b * NMS_TEAM_N * (NMS_TEAM_N / 64) * sizeof(unsigned long long);    // arrival words + matrices
        bool lds_keys;
        int split_len;
        nms_class_lds(n, max_per_class, true, &lds_keys, &split_len);
        if (split_len) bytes += (size_t)b * (NMS_XSTAGES * NMS_TEAM + NMS_XCH_WORDS) * sizeof(unsigned int);     // split mode: arrival words + exchange buffers
    }
    return bytes;
}

static int nms_combined_impl(const float* boxes, const float* scores, int b, int n, int q, int c, int score_stride, int score_offset,
                             int max_per_class, int max_total, float iou_thr, float score_thr, float* out_boxes, float* out_scores,
                             int32_t* out_classes, int32_t* out_valid, void* workspace, size_t workspace_bytes, float* out_abs, float sx, float sy,
                             frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(boxes && scores && out_boxes && out_scores && out_classes && out_valid && workspace, "nms_combined: null pointer");
    FRCNN_CHECK_ARG(b > 0 && n > 0 && c > 0 && (q == 1 || q == c) && max_per_class > 0 && max_total > 0, "nms_combined: bad sizes");
    FRCNN_CHECK_ARG(workspace_bytes >= frcnn_nms_workspace_bytes(b, n, c, max_per_class, max_total), "nms_combined: workspace too small");
    FRCNN_CHECK_ARG(max_per_class <= 4096, "nms_combined: max_output_size_per_class=%d > 4096 unsupported", max_per_class);
    FRCNN_CHECK_ARG(n <= (1 << 24), "nms_combined: N too large");
    FRCNN_CHECK_ARG(iou_thr >= 0.0f, "nms_combined: negative IoU threshold");
    const int m_pad = next_pow2(c * max_per_class);
    FRCNN_CHECK_ARG(m_pad <= NMS_LDS_KEYS, "nms_combined: C*max_per_class=%d too large", c * max_per_class);

    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    NmsParams p;
    p.boxes = boxes; p.scores = scores; p.N = n; p.q = q; p.C = c; p.score_stride = score_stride; p.score_offset = score_offset;
    p.max_per_class = max_per_class; p.iou_thr = iou_thr; p.score_thr = score_thr;
    p.kept_keys = reinterpret_cast<unsigned long long*>(ws);
    p.kept_idx = reinterpret_cast<int*>(ws + (size_t)b * c * max_per_class * sizeof(unsigned long long));
    const bool fused_merge = c == 1;
    p.out_boxes = fused_merge ? out_boxes : nullptr; p.out_scores = out_scores; p.out_classes = out_classes; p.out_valid = out_valid;
    p.max_total = max_total;
    p.out_abs = out_abs; p.sx = sx; p.sy = sy;
    bool team = nms_team_mode(b, n, c);
#ifdef FRCNN_SWEEP
    if (getenv("FRCNN_NMS_NO_TEAM")) team = false;            // (A/B switches of the kernel-development build: production reads no environment)
#endif
    p.team = team ? NMS_TEAM : 1;
    p.team_sync = nullptr; p.team_mat = nullptr;
    if (team) {
        p.team_sync = reinterpret_cast<unsigned int*>(ws + nms_kept_bytes(b, c, max_per_class));
        p.team_mat = reinterpret_cast<unsigned long long*>(ws + nms_kept_bytes(b, c, max_per_class) + (size_t)b * 64);
    }
    bool lds_keys;
    int split_len;
    const size_t smem = nms_class_lds(n, max_per_class, team, &lds_keys, &split_len);
#ifdef FRCNN_SWEEP
    if (getenv("FRCNN_NMS_NO_SPLIT")) split_len = 0;
#endif
    p.lds_keys = lds_keys ? 1 : 0;
    p.split_len = split_len; p.team_xflag = nullptr; p.team_xch = nullptr;
    if (split_len) {
        unsigned char* x = reinterpret_cast<unsigned char*>(p.team_mat) + (size_t)b * NMS_TEAM_N * (NMS_TEAM_N / 64) * sizeof(unsigned long long);
        p.team_xflag = reinterpret_cast<unsigned int*>(x);
        p.team_xch = p.team_xflag + (size_t)b * NMS_XSTAGES * NMS_TEAM;
    }
#define FRCNN_NMS_LAUNCH(LDSK, TEAM, GRID)                                                                                                  \
    do {                                                                                                                                  \
        FRCNN_CHECK_ARG(frcnn_allow_big_lds(reinterpret_cast<const void*>(nms_class_kernel<LDSK, TEAM>), smem) == 0,                      \
                        "nms_combined: cannot reserve %zu B of LDS", smem);                                                               \
        hipLaunchKernelGGL((nms_class_kernel<LDSK, TEAM>), dim3(GRID), dim3(NMS_T), smem, S_(stream), p);                                  \
    } while (0)
    if (team) {
        if (lds_keys) FRCNN_NMS_LAUNCH(true, true, b * c * NMS_TEAM);
        else if (split_len) {
            FRCNN_CHECK_ARG(frcnn_allow_big_lds(reinterpret_cast<const void*>(nms_class_kernel<false, true, true>), smem) == 0,
                            "nms_combined: cannot reserve %zu B of LDS", smem);
            hipLaunchKernelGGL((nms_class_kernel<false, true, true>), dim3(b * c * NMS_TEAM), dim3(NMS_T), smem, S_(stream), p);
        } else FRCNN_NMS_LAUNCH(false, true, b * c * NMS_TEAM);
    } else {
        if (lds_keys) FRCNN_NMS_LAUNCH(true, false, b * c);
        else FRCNN_NMS_LAUNCH(false, false, b * c);
    }
#undef FRCNN_NMS_LAUNCH
    FRCNN_CHECK_LAUNCH("nms_combined(class)");
    if (fused_merge) return FRCNN_OK;

    MergeParams m;
    m.boxes = boxes; m.kept_keys = p.kept_keys; m.kept_idx = p.kept_idx; m.N = n; m.q = q; m.C = c; m.max_per_class = max_per_class;
    m.max_total = max_total; m.m_pad = m_pad; m.out_boxes = out_boxes; m.out_scores = out_scores; m.out_classes = out_classes;
    m.out_valid = out_valid;
    m.out_abs = out_abs; m.sx = sx; m.sy = sy;
    hipLaunchKernelGGL(nms_merge_kernel, dim3(b), dim3(NMS_T), (size_t)m_pad * 8 + 16, S_(stream), m);
    FRCNN_CHECK_LAUNCH("nms_combined(merge)");
    return FRCNN_OK;
}

extern "C" int frcnn_nms_combined(const float* boxes, const float* scores, int b, int n, int q, int c, int score_stride, int score_offset,
                                  int max_per_class, int max_total, float iou_thr, float score_thr, float* out_boxes, float* out_scores,
                                  int32_t* out_classes, int32_t* out_valid, void* workspace, size_t workspace_bytes,
                                  frcnn_stream_t stream) {
    return nms_combined_impl(boxes, scores, b, n, q, c, score_stride, score_offset, max_per_class, max_total, iou_thr, score_thr, out_boxes,
                             out_scores, out_classes, out_valid, workspace, workspace_bytes, nullptr, 1.f, 1.f, stream);
}

extern "C" int frcnn_nms_combined_abs(const float* boxes, const float* scores, int b, int n, int q, int c, int score_stride, int score_offset,
                                      int max_per_class, int max_total, float iou_thr, float score_thr, float* out_boxes, float* out_scores,
                                      int32_t* out_classes, int32_t* out_valid, void* workspace, size_t workspace_bytes, float* out_boxes_abs,
                                      float scale_x, float scale_y, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(out_boxes_abs, "nms_combined_abs: null pointer");
    return nms_combined_impl(boxes, scores, b, n, q, c, score_stride, score_offset, max_per_class, max_total, iou_thr, score_thr, out_boxes,
                             out_scores, out_classes, out_valid, workspace, workspace_bytes, out_boxes_abs, scale_x, scale_y, stream);
}

#ifdef FRCNN_NMS_STAMPS
extern "C" __attribute__((visibility("default"))) int frcnn_debug_nms_stamps(unsigned long long* host) {          // [2][8][16]; development builds only (not in the header)
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_nms_stamps), sizeof(unsigned long long) * 2 * 8 * 16);
}
#endif
