// HBM-bound elementwise / reduction kernels of the backbone: preprocess, BatchNorm (apply,
// backward reduce / apply), ReLU backward, max-pool, SGD, weight re-layouts.
// All activations are NHWC bf16 and are moved as 16-byte vectors (8 channels per lane), rows
// grid-strided over <= 2048 workgroups.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";
void frcnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* frcnn_last_error(void) { return g_err; }
extern "C" int frcnn_abi_version(void) { return FRCNN_ABI_VERSION; }

namespace {

constexpr int kMaxBlocks = 2048;
inline int grid_for(int64_t work_items, int threads) {
    int64_t b = (work_items + threads - 1) / threads;
    if (b < 1) b = 1;
    return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

// ---------------------------------------------------------------- preprocess
__global__ void preprocess_kernel(const uint8_t* __restrict__ img, bf16_t* __restrict__ out, int B, int H, int W, int Hp,
                                  int Wp, int pad) {
    const int64_t total = (int64_t)B * Hp * Wp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const int64_t t = i / Wp;
        const int yp = (int)(t % Hp);
        const int b = (int)(t / Hp);
        const int x = xp - pad, y = yp - pad;
        u32x2 v = {0u, 0u};
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const uint8_t* px = img + (((int64_t)b * H + y) * W + x) * 3;
            const float bl = (float)px[2] - 103.939f, g = (float)px[1] - 116.779f, r = (float)px[0] - 123.68f;
            v[0] = (unsigned)f32_to_bf16_bits(bl) | ((unsigned)f32_to_bf16_bits(g) << 16);
            v[1] = (unsigned)f32_to_bf16_bits(r);
        }
        *reinterpret_cast<u32x2*>(out + i * 4) = v;
    }
}

// ---------------------------------------------------------------- BN finalize
__global__ void bn_finalize_train_kernel(const double* __restrict__ part, int tiles, int C, float inv_count, float unbias,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ mm, float* __restrict__ mv, float momentum, float eps,
                                         float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o,
                                         float* __restrict__ invstd_o) {
    // block = 64 channels x 4 slot lanes
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, ss = 0.0;
    if (c < C)
        for (int t = sl; t < tiles; t += 4) {
            s += part[((int64_t)t * 2) * C + c];
            ss += part[((int64_t)t * 2 + 1) * C + c];
        }
    red[0][sl][cl] = s;
    red[1][sl][cl] = ss;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    ss = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    const double mean = s * inv_count;
    double var = ss * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    mean_o[c] = (float)mean;
    invstd_o[c] = invstd;
    mm[c] = mm[c] * momentum + (float)mean * (1.f - momentum);
    mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
}

__global__ void bn_finalize_eval_kernel(int C, const float* gamma, const float* beta, const float* mm, const float* mv,
                                        float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(mv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
}

// ---------------------------------------------------------------- BN apply (+res, +relu)
__global__ void bn_apply_kernel(const bf16_t* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
                                const bf16_t* __restrict__ res, int relu, bf16_t* __restrict__ out, int64_t nvec, int C8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        float v[8], r[8];
        unpack8(*reinterpret_cast<const u32x4*>(z + i * 8), v);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c), s1 = *reinterpret_cast<const f32x4*>(scale + c + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + c), h1 = *reinterpret_cast<const f32x4*>(shift + c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = v[e] * s0[e] + h0[e];
            v[e + 4] = v[e + 4] * s1[e] + h1[e];
        }
        if (res) {
            unpack8(*reinterpret_cast<const u32x4*>(res + i * 8), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8(v);
    }
}

// ---------------------------------------------------------------- BN backward
// reduce: grid = (64-channel strips, row chunks), 256 threads = 8 channel vectors x 32 row lanes.  A thread keeps mean /
// invstd of its 8 channels in registers and streams its rows with independent 16-byte loads (unrolled: 12 loads in flight per
// thread); the 32 row lanes meet in an LDS tree and the workgroup adds its 2 x 64 partial sums to one of
// FRCNN_STAT_SLOTS pre-zeroed slots with coalesced float atomics (consecutive lanes: consecutive channels).
// (strip, row chunk) of a workgroup of the strip kernels.  1-D grid of strips * 8 * ceil(chunks / 8) workgroups; consecutive
// workgroup ids go to consecutive XCDs, so id = (slot << 3) | xcd with slot = group * strips + strip puts ALL strips of a row chunk
// on one XCD, a few dispatches apart: the 128-byte pieces of a row (and the 8-byte pieces of its ReLU-mask bytes, which otherwise
// reach memory as partial sectors from four different L2s) meet in one L2.  Workgroups with chunk >= chunks have nothing to do.
__device__ __forceinline__ void strip_chunk(const int strips, int& strip, int& chunk) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    strip = slot % strips;
    chunk = (slot / strips) * 8 + xcd;
}

__device__ __forceinline__ u32x4 load_stream(const bf16_t* p) {      // last use of these 16 bytes: do not keep them in the caches
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}

template <int MASK>    // 0: no ReLU, 1: mask from the activation tensor, 2: mask from the forward pass's bit mask
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ gout, const void* __restrict__ act,
                                                            const bf16_t* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ part,
                                                            int64_t M, int C, int rows_per_block, int strips, int chunks) {
    __shared__ float red[32][8][17];             // [row lane][vector][16 sums + pad]
    int strip, chunk;
    strip_chunk(strips, strip, chunk);
    if (chunk >= chunks) return;
    const int c0 = strip * 64;
    const int v = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int C8 = C / 8, cv = c0 / 8 + v;
    float sg[8], sgx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sg[e] = sgx[e] = 0.f;
    if (cv < C8) {
        float mu[8], is[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean[cv * 8 + e]; is[e] = invstd[cv * 8 + e]; }
        const int64_t row_begin = (int64_t)chunk * rows_per_block;
        const int64_t row_end = min(M, row_begin + (int64_t)rows_per_block);
#pragma unroll 4
        for (int64_t r = row_begin + rl; r < row_end; r += 32) {
            const int64_t i = r * C8 + cv;
            float g[8], zz[8];
            unpack8(*reinterpret_cast<const u32x4*>(gout + i * 8), g);
            unpack8(*reinterpret_cast<const u32x4*>(z + i * 8), zz);
            if (MASK == 1) {
                float a[8];
                unpack8(*reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(act) + i * 8), a);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
            } else if (MASK == 2) {
                const unsigned m = reinterpret_cast<const uint8_t*>(act)[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = ((m >> e) & 1u) ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sg[e] += g[e];
                sgx[e] += g[e] * ((zz[e] - mu[e]) * is[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[rl][v][e] = sg[e]; red[rl][v][8 + e] = sgx[e]; }
    __syncthreads();
    for (int s = 16; s > 0; s >>= 1) {
        if (rl < s) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[rl][v][e] += red[rl + s][v][e];
        }
        __syncthreads();
    }
    if (threadIdx.x < 128) {
        const int stat = threadIdx.x >> 6, cl = threadIdx.x & 63;
        const int c = c0 + cl;
        if (c < C) {
            const int slot = (chunk * strips + strip) & (FRCNN_STAT_SLOTS - 1);
            atomicAdd(part + ((int64_t)slot * 2 + stat) * C + c, red[0][cl >> 3][stat * 8 + (cl & 7)]);
        }
    }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int blocks, int C, float inv_m, float* dgamma,
                                       float* dbeta, float* c1, float* c2) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, sx = 0.0;
    if (c < C)
        for (int t = sl; t < blocks; t += 4) {
            s += (double)part[((int64_t)t * 2) * C + c];
            sx += (double)part[((int64_t)t * 2 + 1) * C + c];
        }
    red[0][sl][cl] = s;
    red[1][sl][cl] = sx;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    sx = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    dbeta[c] = (float)s;
    dgamma[c] = (float)sx;
    c1[c] = (float)(s * inv_m);
    c2[c] = (float)(sx * inv_m);
}

// ---------------------------------------------------------------- fused BN finalize + apply (training)
// grid = (64-channel strips, row chunks), 256 threads.  Every workgroup first reduces the conv kernel's partial sums of
// ITS 64 channels (slots x 2 x 64 floats, L2 resident) to scale / shift -- no separate finalize launch -- and keeps each
// thread's 8 channels in registers for all its rows, so the streaming loop issues one 16-byte load per stream and one
// 16-byte store, no parameter loads.  The row-chunk-0 workgroups also publish mean / invstd (for the backward pass) and
// update the moving statistics.
// The second BatchNorm of the DUAL form (the shortcut branch of a stage's first bottleneck block): its conv output, statistics and
// parameters.  out = ReLU(BN(z) + BN2(z2)) in one pass: the shortcut's normalised output -- 60 MB at conv2, written by one kernel
// and read back by the next -- is never stored (the backward pass needs z2, not BN2(z2)).  BN2(z2) is rounded to bf16 before the
// addition, as it was when it went through memory: bit-identical results.
// The 8 fp8 bytes of a thread's channel vector, stored 16 bytes at a time: neighbouring lanes (channel vectors v, v ^ 1 of one row: same
// control flow) pool their halves and the even lane stores both -- as many store instructions per wave, but full 16-byte lanes
// (the strip kernels are bound by memory instructions in flight, not bytes: 8-byte stores of the twin cost as much as the
// 16-byte stores of the bf16 tensor)
__device__ __forceinline__ void store_fp8_pair(uint8_t* dst, const u32x2 q, const int v, const bool pair_ok, const bool ok = true) {
#ifndef FRCNN_FP8_STORE8
    if (pair_ok) {                               // (uniform; the shuffle runs on every lane, ok only gates the store)
        const unsigned p0 = __shfl_xor(q[0], 1), p1 = __shfl_xor(q[1], 1);
        if (!(v & 1) && ok) *reinterpret_cast<u32x4*>(dst) = u32x4{q[0], q[1], p0, p1};
        return;
    }
#endif
    if (ok) *reinterpret_cast<u32x2*>(dst) = q;
}

// Sum of the slot partials of one channel handled by slot lane `sl` (slots sl, sl + 4, ...), both statistics.  With the usual
// FRCNN_STAT_SLOTS slots the 2 x 4 loads are issued TOGETHER: as a runtime loop they were four dependent L2 round trips -- most of the
// ~2 us during which a strip workgroup streams nothing (the additions keep the loop's order: same bits).
template <typename T>
__device__ __forceinline__ void slot_sums(const T* __restrict__ part, const int slots, const int sl, const int C, const int c, double& s0, double& s1) {
    s0 = s1 = 0.0;
#ifndef FRCNN_BN_SLOT_LOOP                       // (A/B builds: FRCNN_DEFINES=FRCNN_BN_SLOT_LOOP keeps the runtime loop)
    if (slots == FRCNN_STAT_SLOTS) {
        T a[FRCNN_STAT_SLOTS / 4], b[FRCNN_STAT_SLOTS / 4];
#pragma unroll
        for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) {
            a[k] = part[((int64_t)(sl + 4 * k) * 2) * C + c];
            b[k] = part[((int64_t)(sl + 4 * k) * 2 + 1) * C + c];
        }
#pragma unroll
        for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) { s0 += (double)a[k]; s1 += (double)b[k]; }
        return;
    }
#endif
    for (int t = sl; t < slots; t += 4) {
        s0 += (double)part[((int64_t)t * 2) * C + c];
        s1 += (double)part[((int64_t)t * 2 + 1) * C + c];
    }
}

#ifndef FRCNN_BN_U
// rows per thread and round of the strip kernels' streaming loops, all their loads issued before the first is consumed.  Measured in
// the step on one box (round 4, tools/ab_lib.sh, FRCNN_DEFINES=FRCNN_BN_U=n): 1: 4.215 ms, 4: 4.315 ms -- more bytes in flight per
// wave cost more (118 / 157 VGPRs, fewer resident waves) than they bring: these kernels are not bound by the latency of one row
#define FRCNN_BN_U 1
#endif

struct Bn2 {
    const bf16_t* z; const double* part; const float* gamma; const float* beta;
    float* mm; float* mv; float* mean_o; float* invstd_o;
    // optional fp8 twin of the output (frcnn_fp8_out): e4m3 bytes of the bf16-rounded activation times *qscale, max |value| into *amax
    uint8_t* out8; const float* qscale; float* amax;
};

template <int VAR, bool DUAL = false>   // VAR 0: production; 4: round-1 form (2-D placement, cached loads) for FRCNN_SWEEP A/B runs (tools/ab_lib.sh)
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const bf16_t* __restrict__ z, const double* __restrict__ part, int slots,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mm, float* __restrict__ mv, float momentum, float eps,
                                                             float inv_count, float unbias, const bf16_t* __restrict__ res, int relu,
                                                             bf16_t* __restrict__ out, uint8_t* __restrict__ relu_mask,
                                                             float* __restrict__ mean_o, float* __restrict__ invstd_o, int64_t M, int C,
                                                             int rows_per_block, int strips, int chunks, const Bn2 b2) {
    __shared__ double red[2][4][64];
    __shared__ float s_scale[64], s_shift[64];
    __shared__ float s_scale2[DUAL ? 64 : 1], s_shift2[DUAL ? 64 : 1];
    int strip, chunk;
    if (VAR == 4) { strip = blockIdx.x % strips; chunk = blockIdx.x / strips; }      // (the round-1 placement, for A/B runs)
    else strip_chunk(strips, strip, chunk);
    if (chunk >= chunks) return;
    const int c0 = strip * 64;
    const int v = threadIdx.x & 7, rl = threadIdx.x >> 3;        // 8 channel vectors x 32 row lanes
    const int C8 = C / 8, cv = c0 / 8 + v;
    const bool live = cv < C8;
    const int64_t row_begin = (int64_t)chunk * rows_per_block;
    const int64_t row_end = min(M, row_begin + (int64_t)rows_per_block);
    // z and the residual are read for the last time before the backward pass: streamed past the caches (measured on the
    // 375x1242 batch-4 shapes, tools/bn_bench.py: 46 -> 42 us on conv2's 256-channel layers with z warm, 62 -> 45 cold)
    constexpr bool NT = VAR != 4;
    // (The statistics prologue below is ~2 us of every launch during which a workgroup streams nothing.  Requesting each
    // thread's first four rows BEFORE it was tried twice, in both rounds' forms of this kernel: 0.05 ms per step SLOWER.)
    {
        const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int c = c0 + cl;
        float ga = 0.f, be = 0.f;
        if (sl == 0 && c < C) { ga = gamma[c]; be = beta[c]; }   // (requested with the partial sums, not after the barrier)
        double s = 0.0, ss = 0.0;
        if (c < C) slot_sums(part, slots, sl, C, c, s, ss);
        red[0][sl][cl] = s;
        red[1][sl][cl] = ss;
        __syncthreads();
        if (sl == 0 && c < C) {
            s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
            ss = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
            const double mean = s * inv_count;
            double var = ss * inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = ga * invstd;
            s_scale[cl] = sc;
            s_shift[cl] = be - (float)mean * sc;
            if (chunk == 0) {
                mean_o[c] = (float)mean;
                invstd_o[c] = invstd;
                mm[c] = mm[c] * momentum + (float)mean * (1.f - momentum);
                mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
            }
        }
        __syncthreads();
        if (DUAL) {                              // the same for the second BatchNorm (its loads overlap the first one's arithmetic)
            float ga2 = 0.f, be2 = 0.f;
            if (sl == 0 && c < C) { ga2 = b2.gamma[c]; be2 = b2.beta[c]; }
            s = ss = 0.0;
            if (c < C) slot_sums(b2.part, slots, sl, C, c, s, ss);
            red[0][sl][cl] = s;
            red[1][sl][cl] = ss;
            __syncthreads();
            if (sl == 0 && c < C) {
                s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
                ss = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
                const double mean = s * inv_count;
                double var = ss * inv_count - mean * mean;
                if (var < 0.0) var = 0.0;
                const float invstd = (float)(1.0 / sqrt(var + (double)eps));
                const float sc2 = ga2 * invstd;
                s_scale2[cl] = sc2;
                s_shift2[cl] = be2 - (float)mean * sc2;
                if (chunk == 0) {
                    b2.mean_o[c] = (float)mean;
                    b2.invstd_o[c] = invstd;
                    b2.mm[c] = b2.mm[c] * momentum + (float)mean * (1.f - momentum);
                    b2.mv[c] = b2.mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
                }
            }
            __syncthreads();
        }
    }
    if (!live) return;
    float sc[8], sh[8], sc2[DUAL ? 8 : 1], sh2[DUAL ? 8 : 1];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s_scale[v * 8 + e]; sh[e] = s_shift[v * 8 + e]; }
    if (DUAL) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc2[e] = s_scale2[v * 8 + e]; sh2[e] = s_shift2[v * 8 + e]; }
    }
    const bf16_t* second = DUAL ? b2.z : res;
    const float f8_qs = b2.out8 ? *b2.qscale : 0.f;
    float f8_max = 0.f;
    // ok: this thread's row exists.  The cross-lane pooling below runs on every lane (its partners are the 8 channel-vector lanes of
    // the SAME row: ok is uniform over them); only the stores are predicated.
    auto finish = [&](const int64_t i, const u32x4 zraw, const u32x4 qraw, const bool ok) {
        float x[8];
        unpack8(zraw, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = x[e] * sc[e] + sh[e];
        if (DUAL) {
            float q[8];
            unpack8(qraw, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += bf16_round(q[e] * sc2[e] + sh2[e]);
        } else if (res) {
            float q[8];
            unpack8(qraw, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += q[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        const u32x4 pk = pack8(x);
        if (out && ok) *reinterpret_cast<u32x4*>(out + i * 8) = pk;      // (NULL: every consumer of the activation reads the e4m3 twin below)
        if (b2.out8) {                           // fp8 twin of the STORED (bf16-rounded) activation
            float xr[8];
            unpack8(pk, xr);
            store_fp8_pair(b2.out8 + i * 8, pack8_fp8(xr, f8_qs), v, (C & 15) == 0, ok);
            if (ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f8_max = amax_fold(f8_max, xr[e]);
            }
        }
        if (relu_mask) {
            // one bit per element: (stored bf16 activation > 0), i.e. exactly what the backward pass would derive from the
            // activation itself -- it reads this byte instead of the 16-byte activation vector (twice)
            unsigned m = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                m |= ((pk[q] & 0x7FFFu) != 0u && !(pk[q] & 0x8000u)) ? (1u << (2 * q)) : 0u;
                m |= ((pk[q] & 0x7FFF0000u) != 0u && !(pk[q] & 0x80000000u)) ? (1u << (2 * q + 1)) : 0u;
            }
            // the 8 channel-vector lanes of a row (consecutive lanes) pool their bytes: one 8-byte store per row and strip
            unsigned lo = m << (8 * (v & 3));
            lo |= __shfl_xor(lo, 1);
            lo |= __shfl_xor(lo, 2);
            const unsigned hi = __shfl_xor(lo, 4);
            if (C8 % 8 == 0) {
                if (v == 0 && ok) *reinterpret_cast<u32x2*>(relu_mask + i) = u32x2{lo, hi};      // i = r*C8 + cv, cv % 8 == 0 here
            } else if (ok) {
                relu_mask[i] = (uint8_t)m;
            }
        }
    };
    // U rows per thread and round, all their loads issued before the first is consumed (FRCNN_BN_U; uniform loop bound: `#pragma
    // unroll` on the per-thread form r = row_begin + rl; r < row_end is refused by the compiler -- a divergent trip count around the
    // cross-lane pooling of finish() -- so rounds 1-3 ran one row at a time without saying so).  Measured: U = 4 is SLOWER than U = 1
    // in the step (4.315 vs 4.215 ms on one box): the kernel is not bound by the latency of a single row per wave.
    constexpr int U = VAR == 4 ? 1 : FRCNN_BN_U;
    for (int64_t rb = row_begin; rb < row_end; rb += 32 * U) {
        u32x4 zraw[U], qraw[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + u * 32 + rl;
            ok[u] = r < row_end;
            const int64_t i = r * C8 + cv;
            zraw[u] = qraw[u] = u32x4{0u, 0u, 0u, 0u};
            if (ok[u]) {
                zraw[u] = NT ? load_stream(z + i * 8) : *reinterpret_cast<const u32x4*>(z + i * 8);
                if (second) qraw[u] = NT ? load_stream(second + i * 8) : *reinterpret_cast<const u32x4*>(second + i * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) finish((rb + u * 32 + rl) * C8 + cv, zraw[u], qraw[u], ok[u]);
    }
    if (b2.out8 && b2.amax) atomic_amax(b2.amax, f8_max);
}

// BatchNorm (batch statistics) + ReLU + 3x3 / stride-2 / pad-1 max pool in one pass (the ResNet stem: reference
// models/feature_extractor.py:8-10, conv1_bn -> conv1_relu -> pool1_pad -> pool1_pool): the activation between them -- 60 MB at
// 375x1242, batch 4, written once and read once -- never exists.  Same statistics prologue, same arithmetic and the same bf16
// rounding of the activation as bn_train_apply_kernel followed by maxpool_fwd_kernel (zero padding takes part in the max; the
// first maximum in window order wins): bit-identical pooled values, arg-max bytes and ReLU bit mask (which the backward pass
// needs: the gradient of the activation is still materialised by maxpool_bwd_kernel -- gathering it through the pool inside the
// BatchNorm backward kernels was built and measured: 82 + 80 us against 34 + 25 + 31).  Grid as the strip kernels; a workgroup
// owns 64 channels of a run of pooled cells, 8 channel vectors x 32 cell lanes.
__global__ __launch_bounds__(256) void bn_train_apply_pool_kernel(const bf16_t* __restrict__ z, const double* __restrict__ part, int slots,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ mm, float* __restrict__ mv, float momentum, float eps,
                                                                  float inv_count, float unbias, bf16_t* __restrict__ pool,
                                                                  uint8_t* __restrict__ amax, uint8_t* __restrict__ relu_mask,
                                                                  float* __restrict__ mean_o,
                                                                  float* __restrict__ invstd_o, int N, int H, int W, int C, int Ho, int Wo,
                                                                  int cells_per_block, int strips, int chunks) {
    __shared__ double red[2][4][64];
    __shared__ float s_scale[64], s_shift[64];
    int strip, chunk;
    strip_chunk(strips, strip, chunk);
    if (chunk >= chunks) return;
    const int c0 = strip * 64;
    {
        const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int c = c0 + cl;
        float ga = 0.f, be = 0.f;
        if (sl == 0 && c < C) { ga = gamma[c]; be = beta[c]; }
        double s = 0.0, ss = 0.0;
        if (c < C) slot_sums(part, slots, sl, C, c, s, ss);
        red[0][sl][cl] = s;
        red[1][sl][cl] = ss;
        __syncthreads();
        if (sl == 0 && c < C) {
            s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
            ss = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
            const double mean = s * inv_count;
            double var = ss * inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = ga * invstd;
            s_scale[cl] = sc;
            s_shift[cl] = be - (float)mean * sc;
            if (chunk == 0) {
                mean_o[c] = (float)mean;
                invstd_o[c] = invstd;
                mm[c] = mm[c] * momentum + (float)mean * (1.f - momentum);
                mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
            }
        }
        __syncthreads();
    }
    const int v = threadIdx.x & 7, cell_lane = threadIdx.x >> 3;
    const int C8 = C / 8, cv = c0 / 8 + v;
    if (cv >= C8) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s_scale[v * 8 + e]; sh[e] = s_shift[v * 8 + e]; }
    const int64_t cells = (int64_t)N * Ho * Wo;
    const int64_t cell_begin = (int64_t)chunk * cells_per_block;
    const int64_t cell_end = min(cells, cell_begin + (int64_t)cells_per_block);
    for (int64_t cell = cell_begin + cell_lane; cell < cell_end; cell += 32) {
        const int ox = (int)(cell % Wo);
        const int64_t t = cell / Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        u32x4 raw[9];
        bool in[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {                               // all nine loads in flight
            const int iy = oy * 2 - 1 + k / 3, ix = ox * 2 - 1 + k % 3;
            in[k] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int64_t src = ((((int64_t)n * H + (in[k] ? iy : 0)) * W + (in[k] ? ix : 0)) * C8 + cv) * 8;
            raw[k] = *reinterpret_cast<const u32x4*>(z + src);
        }
        float best[8];
        unsigned char arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -1.f; arg[e] = 0; }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float x[8];
            unpack8(raw[k], x);
            unsigned mbits = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // the activation as bn_train_apply_kernel stores it (bf16), or the zero padding
                const float a = in[k] ? bf16_round(fmaxf(x[e] * sc[e] + sh[e], 0.f)) : 0.f;
                if (a > best[e]) { best[e] = a; arg[e] = (unsigned char)k; }
                mbits |= a > 0.f ? (1u << e) : 0u;
            }
            // the ReLU bit mask of the activation (read by the BatchNorm backward kernels): every pixel is written by the one cell
            // that owns it -- window positions 4, 5, 7, 8 = pixels (2oy, 2ox), (2oy, 2ox+1), (2oy+1, 2ox), (2oy+1, 2ox+1)
            if (relu_mask && (k == 4 || k == 5 || k == 7 || k == 8) && in[k]) {
                const int iy = oy * 2 - 1 + k / 3, ix = ox * 2 - 1 + k % 3;
                relu_mask[(((int64_t)n * H + iy) * W + ix) * C8 + cv] = (uint8_t)mbits;
            }
        }
        const int64_t o = cell * C8 + cv;
        *reinterpret_cast<u32x4*>(pool + o * 8) = pack8(best);
        u32x2 a;
        a[0] = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
        a[1] = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
        *reinterpret_cast<u32x2*>(amax + o * 8) = a;
    }
}

// The same operation, tiled: a workgroup owns 4 x 16 pooled cells of a 64-channel strip.  Its 9 x 33 input pixels are normalised ONCE
// (the per-cell form above normalises every pixel of a 3 x 3 / 2 window again for each of the 2.25 cells that see it, and spends ~12
// instructions per tap and channel on compare-and-select bookkeeping), stored as bf16 in LDS, and pooled from there with ONE v_max_u32 per
// tap and channel: the activation is non-negative, so its bf16 pattern orders like an unsigned integer, and the key
// (pattern << 16 | 15 - tap) makes the maximum of the keys the FIRST maximum of the window -- value and arg-max together.  Same bits as the
// per-cell form (tests/test_gpu_kernels.py::test_stem_bn_relu_maxpool_fused) except that a -0.0 activation is stored as +0.0.
constexpr int POOL_TH = 4, POOL_TW = 16, POOL_PH = 2 * POOL_TH + 1, POOL_PW = 2 * POOL_TW + 1, POOL_PITCH = 144;   // bytes per patch pixel (128 + pad)
__global__ __launch_bounds__(256) void bn_train_apply_pool_tiled_kernel(const bf16_t* __restrict__ z, const double* __restrict__ part, int slots,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        float* __restrict__ mm, float* __restrict__ mv, float momentum, float eps,
                                                                        float inv_count, float unbias, bf16_t* __restrict__ pool,
                                                                        uint8_t* __restrict__ amax, uint8_t* __restrict__ relu_mask,
                                                                        float* __restrict__ mean_o, float* __restrict__ invstd_o, int N, int H,
                                                                        int W, int C, int Ho, int Wo, int strips, int tiles_x, int tiles_y) {
    __shared__ double red[2][4][64];
    __shared__ float s_scale[64], s_shift[64];
    __shared__ __attribute__((aligned(16))) unsigned char patch[POOL_PH * POOL_PW * POOL_PITCH];
    const int strip = blockIdx.x % strips;
    int tile = blockIdx.x / strips;
    const int tx0 = (tile % tiles_x) * POOL_TW;
    tile /= tiles_x;
    const int ty0 = (tile % tiles_y) * POOL_TH, n = tile / tiles_y;
    const int c0 = strip * 64;
    {
        const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int c = c0 + cl;
        float ga = 0.f, be = 0.f;
        if (sl == 0 && c < C) { ga = gamma[c]; be = beta[c]; }
        double s = 0.0, ss = 0.0;
        if (c < C) slot_sums(part, slots, sl, C, c, s, ss);
        red[0][sl][cl] = s;
        red[1][sl][cl] = ss;
        __syncthreads();
        if (sl == 0) {
            float sc = 0.f, shf = 0.f;
            if (c < C) {
                s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
                ss = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
                const double mean = s * inv_count;
                double var = ss * inv_count - mean * mean;
                if (var < 0.0) var = 0.0;
                const float invstd = (float)(1.0 / sqrt(var + (double)eps));
                sc = ga * invstd;
                shf = be - (float)mean * sc;
                if (blockIdx.x / strips == 0) {
                    mean_o[c] = (float)mean;
                    invstd_o[c] = invstd;
                    mm[c] = mm[c] * momentum + (float)mean * (1.f - momentum);
                    mv[c] = mv[c] * momentum + (float)(var * unbias) * (1.f - momentum);
                }
            }
            s_scale[cl] = sc;
            s_shift[cl] = shf;
        }
        __syncthreads();
    }
    const int C8 = C / 8;
    const int v = threadIdx.x & 7;
    const int cv = c0 / 8 + v;
    const bool v_ok = cv < C8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s_scale[v * 8 + e]; sh[e] = s_shift[v * 8 + e]; }
    // ---- 1. the patch: pixel rows 2 ty0 - 1 .. 2 (ty0 + TH - 1) + 1, columns likewise; activation bf16 -> LDS, ReLU mask of the owned pixels
    const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;
    for (int it = threadIdx.x >> 3; it < POOL_PH * POOL_PW; it += 32) {
        const int py = it / POOL_PW, px = it - py * POOL_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool in = v_ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        u32x4 out = {0u, 0u, 0u, 0u};                                   // zero padding (and the channels beyond C)
        if (in) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(z + ((((int64_t)n * H + iy) * W + ix) * C8 + cv) * 8);
            float x[8];
            unpack8(raw, x);
            unsigned mbits = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x2 a2;
                a2[0] = fmaxf(x[2 * q] * sc[2 * q] + sh[2 * q], 0.f);
                a2[1] = fmaxf(x[2 * q + 1] * sc[2 * q + 1] + sh[2 * q + 1], 0.f);
                const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(a2, bf16x2_t)) & 0x7FFF7FFFu;   // (RNE; -0.0 -> +0.0)
                out[q] = bits;
                mbits |= ((bits & 0xFFFFu) ? 1u : 0u) << (2 * q);
                mbits |= ((bits >> 16) ? 1u : 0u) << (2 * q + 1);
            }
            // every pixel's mask byte is written by the tile that OWNS it: pooled cell (iy / 2, ix / 2) -- the patch's first row and column
            // belong to the neighbouring tiles, and so do rows / columns past this tile's cells
            if (relu_mask && py >= 1 && px >= 1 && py <= 2 * POOL_TH && px <= 2 * POOL_TW)
                relu_mask[(((int64_t)n * H + iy) * W + ix) * C8 + cv] = (uint8_t)mbits;
        }
        *reinterpret_cast<u32x4*>(patch + it * POOL_PITCH + v * 16) = out;
    }
    __syncthreads();
    // ---- 2. pooling from LDS: one key maximum per tap and channel
    for (int cell = threadIdx.x >> 3; cell < POOL_TH * POOL_TW; cell += 32) {
        const int ty = cell / POOL_TW, tx = cell - ty * POOL_TW;
        const int oy = ty0 + ty, ox = tx0 + tx;
        if (oy >= Ho || ox >= Wo || !v_ok) continue;
        unsigned key[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) key[e] = 0u;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(patch + ((2 * ty + k / 3) * POOL_PW + 2 * tx + k % 3) * POOL_PITCH + v * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned klo = (t[q] << 16) | (unsigned)(15 - k), khi = (t[q] & 0xFFFF0000u) | (unsigned)(15 - k);
                key[2 * q] = key[2 * q] > klo ? key[2 * q] : klo;
                key[2 * q + 1] = key[2 * q + 1] > khi ? key[2 * q + 1] : khi;
            }
        }
        const int64_t o = ((int64_t)(n * Ho + oy) * Wo + ox) * C8 + cv;
        u32x4 pv;
        u32x2 av;
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[q] = (key[2 * q] >> 16) | (key[2 * q + 1] & 0xFFFF0000u);
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)
            av[hlf] = (15u - (key[4 * hlf] & 15u)) | ((15u - (key[4 * hlf + 1] & 15u)) << 8) | ((15u - (key[4 * hlf + 2] & 15u)) << 16) |
                      ((15u - (key[4 * hlf + 3] & 15u)) << 24);
        *reinterpret_cast<u32x4*>(pool + o * 8) = pv;
        *reinterpret_cast<u32x2*>(amax + o * 8) = av;
    }
}

// fused BN backward finalize + apply: c1 = sum(g)/m, c2 = sum(g*xhat)/m of the workgroup's 64 channels from the reduce
// kernel's slot partials; the row-chunk-0 workgroups publish dgamma / dbeta.
// RED2 (frcnn_bn_bwd_apply_fused_red2): the same launch ALSO runs the backward reduce of a SECOND BatchNorm that receives the same masked
// gradient -- the shortcut branch of a stage's first block (conv<N>_block1_0_bn) next to the block-final BatchNorm (..._3_bn): sum g*m and
// sum g*m*xhat2 with xhat2 from the second layer's z / mean / invstd, per thread over the rows it streams anyway (the rows, their order
// and the workgroup's reduction tree are bn_bwd_reduce_kernel's: the slot partials are the same numbers), one more 16-byte load per row
// vector instead of a launch that re-reads g, z2 and the mask.
struct BnRed2 {
    const bf16_t* z; const float* mean; const float* invstd; float* part;
};
template <int MASK, int LEGACY = 0, bool RED2 = false>          // LEGACY 1: round-1 placement and cached loads (FRCNN_SWEEP A/B runs only)
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const bf16_t* __restrict__ gout, const void* __restrict__ act,
                                                                 const bf16_t* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ part, int slots, float inv_m, float pscale,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 bf16_t* __restrict__ dz, bf16_t* __restrict__ gpre, int64_t M, int C,
                                                                 int rows_per_block, int strips, int chunks, uint8_t* __restrict__ dz8,
                                                                 const float* __restrict__ dz8_qscale, float* __restrict__ dz8_amax, const BnRed2 r2) {
    __shared__ double red[2][4][64];
    __shared__ float red2[RED2 ? 32 : 1][8][17];  // RED2: [row lane][vector][16 sums + pad], as bn_bwd_reduce_kernel
    __shared__ float s_par[4][64];                // gamma*invstd, mean, invstd (xhat), c1, c2 folded: a, mu, is, k1, k2
    __shared__ float s_c2[64];
    int strip, chunk;
    if (LEGACY == 1) { strip = blockIdx.x % strips; chunk = blockIdx.x / strips; }
    else strip_chunk(strips, strip, chunk);
    if (chunk >= chunks) return;
    const int c0 = strip * 64;
    {
        const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int c = c0 + cl;
        float is = 0.f, ga_c = 0.f, mu_c = 0.f;                 // (requested with the partial sums, not after the barrier)
        if (sl == 0 && c < C) { is = invstd[c]; ga_c = gamma[c]; mu_c = mean[c]; }
        double s = 0.0, sx = 0.0;
        if (c < C) slot_sums(part, slots, sl, C, c, s, sx);
        red[0][sl][cl] = s;
        red[1][sl][cl] = sx;
        __syncthreads();
        if (sl == 0 && c < C) {
            s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
            sx = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
            s_par[0][cl] = ga_c * is;
            s_par[1][cl] = mu_c;
            s_par[2][cl] = is;
            s_par[3][cl] = (float)(s * inv_m);
            s_c2[cl] = (float)(sx * inv_m);
            if (chunk == 0) {
                dbeta[c] = (float)s * pscale;      // (synchronised BN: s, sx are sums over ALL ranks; every rank publishes its 1/world share)
                dgamma[c] = (float)sx * pscale;
            }
        }
        __syncthreads();
    }
    const int v = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int C8 = C / 8, cv = c0 / 8 + v;
    if (cv >= C8) return;                        // (RED2: the host requires C % 64 == 0 -- nobody leaves before the reduction's barriers)
    float ga[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ga[e] = s_par[0][v * 8 + e]; mu[e] = s_par[1][v * 8 + e]; is[e] = s_par[2][v * 8 + e];
        k1[e] = s_par[3][v * 8 + e]; k2[e] = s_c2[v * 8 + e];
    }
    float mu2[RED2 ? 8 : 1], is2[RED2 ? 8 : 1], sg2[RED2 ? 8 : 1], sgx2[RED2 ? 8 : 1];
    if (RED2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu2[e] = r2.mean[cv * 8 + e]; is2[e] = r2.invstd[cv * 8 + e]; sg2[e] = sgx2[e] = 0.f; }
    }
    const int64_t row_begin = (int64_t)chunk * rows_per_block;
    const int64_t row_end = min(M, row_begin + (int64_t)rows_per_block);
    // dz is stored in bf16, and sum_rows(dz) is exactly zero in exact arithmetic.  Plain round-to-nearest breaks that by far more
    // than a random walk: the incoming gradient g is itself bf16, so per channel dz = a*(g - c1 - xhat*c2) takes ~1000 distinct
    // values a*g shifted by the tiny c1, each with ITS fixed rounding error, repeated over 10^5 rows -- the errors add coherently
    // (measured at 375x1242, batch 4: |sum dz| = 2.4 where a random walk gives 0.02), and the weight gradient sum(dz * x) picks
    // that up multiplied by mean(x): 3.6 % of the gradient of the 1x1 convolutions that read the max-pool output.  First-order
    // error feedback along each thread's chain of rows (the residual of one rounding is added to the next element of the same
    // channel) makes the column sums exact to an ulp per chain; an element is off by at most half an ulp of itself plus half
    // an ulp of its predecessor.  Deterministic, 2 VALU per element.
    float carry[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) carry[e] = 0.f;
    // optional fp8 twin of dz for the fp8 data-gradient convolution: e5m2 bytes of the STORED bf16 value times *dz8_qscale
    const float f8_qs = dz8 ? *dz8_qscale : 0.f;
    float f8_max = 0.f;
    // U rows per thread and round (FRCNN_BN_U, see bn_train_apply_kernel); the rows of a thread are consumed in order, so the
    // error-feedback chain does not depend on U.
    constexpr int U = LEGACY == 1 ? 1 : FRCNN_BN_U;
    for (int64_t rb = row_begin; rb < row_end; rb += 32 * U) {
        u32x4 graw[U], zraw[U], araw[U], z2raw[RED2 ? U : 1];
        unsigned mraw[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + u * 32 + rl;
            ok[u] = r < row_end;
            const int64_t i = r * C8 + cv;
            graw[u] = zraw[u] = araw[u] = u32x4{0u, 0u, 0u, 0u};
            mraw[u] = 0u;
            if (ok[u]) {
                graw[u] = LEGACY ? *reinterpret_cast<const u32x4*>(gout + i * 8) : load_stream(gout + i * 8);   // (last use of the incoming
                zraw[u] = LEGACY == 1 ? *reinterpret_cast<const u32x4*>(z + i * 8) : load_stream(z + i * 8);    //  gradient and of z: streamed)
                if (MASK == 1) araw[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(act) + i * 8);
                if (MASK == 2) mraw[u] = reinterpret_cast<const uint8_t*>(act)[i];
                if (RED2) z2raw[u] = *reinterpret_cast<const u32x4*>(r2.z + i * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = (rb + u * 32 + rl) * C8 + cv;
            float g[8], zz[8], o[8];
            unpack8(graw[u], g);
            unpack8(zraw[u], zz);
            if (MASK == 1) {
                float a[8];
                unpack8(araw[u], a);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
            } else if (MASK == 2) {
                const unsigned m = mraw[u];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = ((m >> e) & 1u) ? g[e] : 0.f;
            }
            if (RED2 && ok[u]) {                 // (g is the masked gradient here: bn_bwd_reduce_kernel's terms)
                float z2[8];
                unpack8(z2raw[u], z2);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sg2[e] += g[e];
                    sgx2[e] += g[e] * ((z2[e] - mu2[e]) * is2[e]);
                }
            }
            if (ok[u]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (zz[e] - mu[e]) * is[e];
                    const float v = ga[e] * (g[e] - k1[e] - xh * k2[e]) + carry[e];
                    o[e] = bf16_round(v);
                    carry[e] = v - o[e];
                }
                if (dz) *reinterpret_cast<u32x4*>(dz + i * 8) = pack8(o);      // (NULL: every consumer of dz reads the e5m2 twin below)
                if (gpre) *reinterpret_cast<u32x4*>(gpre + i * 8) = pack8(g);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = 0.f;
            }
            if (dz8) {
                store_fp8_pair(dz8 + i * 8, pack8_bf8(o, f8_qs), v, (C & 15) == 0, ok[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) f8_max = amax_fold(f8_max, o[e]);
            }
        }
    }
    if (dz8 && dz8_amax) atomic_amax(dz8_amax, f8_max);
    if (RED2) {                                  // workgroup reduction + slot atomics of bn_bwd_reduce_kernel
#pragma unroll
        for (int e = 0; e < 8; ++e) { red2[rl][v][e] = sg2[e]; red2[rl][v][8 + e] = sgx2[e]; }
        __syncthreads();
        for (int s = 16; s > 0; s >>= 1) {
            if (rl < s) {
#pragma unroll
                for (int e = 0; e < 16; ++e) red2[rl][v][e] += red2[rl + s][v][e];
            }
            __syncthreads();
        }
        if (threadIdx.x < 128) {
            const int stat = threadIdx.x >> 6, cl = threadIdx.x & 63;
            const int slot = (chunk * strips + strip) & (FRCNN_STAT_SLOTS - 1);
            atomicAdd(r2.part + ((int64_t)slot * 2 + stat) * C + c0 + cl, red2[0][cl >> 3][stat * 8 + (cl & 7)]);
        }
    }
}

template <bool MASK>
__global__ void bn_bwd_apply_kernel(const bf16_t* __restrict__ gout, const bf16_t* __restrict__ act, const bf16_t* __restrict__ z,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ c1, const float* __restrict__ c2,
                                    bf16_t* __restrict__ dz, bf16_t* __restrict__ gpre, int64_t nvec, int C8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        float g[8], zz[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(gout + i * 8), g);
        unpack8(*reinterpret_cast<const u32x4*>(z + i * 8), zz);
        if (MASK) {
            float a[8];
            unpack8(*reinterpret_cast<const u32x4*>(act + i * 8), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float is = invstd[c + e];
            const float xh = (zz[e] - mean[c + e]) * is;
            o[e] = gamma[c + e] * is * (g[e] - c1[c + e] - xh * c2[c + e]);
        }
        *reinterpret_cast<u32x4*>(dz + i * 8) = pack8(o);
        if (gpre) *reinterpret_cast<u32x4*>(gpre + i * 8) = pack8(g);
    }
}

__global__ void relu_bwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ act, bf16_t* __restrict__ out, int64_t nvec) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const u32x4*>(g + i * 8), a);
        unpack8(*reinterpret_cast<const u32x4*>(act + i * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = b[e] > 0.f ? a[e] : 0.f;
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8(a);
    }
}

// column sums of a bf16 matrix [m, ld] (first c columns).  grid = (column groups of 64, row chunks of 256), 256 threads = 8 column
// vectors x 32 row lanes: a thread owns 8 adjacent columns and reads them with one 16-byte load per row, its 8 rows issued back to
// back (the one-element-per-load form spent 20-30 us per call waiting on 64 dependent 2-byte loads).  One atomic per column and
// workgroup.  ld % 8 == 0 and a 16-byte aligned base take the vector path; anything else the scalar one.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t m, int c, int ld, float* __restrict__ out, int vec) {
    __shared__ float red[32][65];
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = min(m, r0 + 256);
    if (vec) {
        const int v = threadIdx.x & 7, rl = threadIdx.x >> 3;
        const int col0 = blockIdx.x * 64 + v * 8;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (col0 < c) {
#pragma unroll 8
            for (int64_t r = r0 + rl; r < r1; r += 32) {
                float q[8];
                unpack8(*reinterpret_cast<const u32x4*>(x + r * ld + col0), q);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += q[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl][v * 8 + e] = acc[e];
    } else {
        const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
        const int col = blockIdx.x * 64 + cl;
        float s = 0.f;
        if (col < c)
            for (int64_t r = r0 + rl; r < r1; r += 4) s += bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(x + r * ld + col));
        for (int k = rl; k < 32; k += 4) red[k][cl] = k == rl ? s : 0.f;
    }
    __syncthreads();
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && col < c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
        atomicAdd(out + col, s);
    }
}

// The column sums fused with the pass that PRODUCES the bf16 matrix (same grid, same per-thread row order and LDS tree as colsum_kernel's
// vector path: the sums are the ones colsum_kernel would compute from the stored values):
//   SRC 1: dst = bf16(src32), out[col] += sum_rows dst        (frcnn_cast_f32_bf16 + frcnn_colsum_bf16: the RPN head gradient and its bias gradient)
//   SRC 2: dst = act > 0 ? g : 0, out[col] += sum_rows dst     (frcnn_relu_bwd + frcnn_colsum_bf16: the RPN 3x3 layer's dz and bias gradient)
// [m][c] row-major, c % 8 == 0, 16-byte aligned.
template <int SRC>
__global__ __launch_bounds__(256) void colsum_produce_kernel(const float* __restrict__ src32, const bf16_t* __restrict__ g, const bf16_t* __restrict__ act,
                                                             bf16_t* __restrict__ dst, int64_t m, int c, float* __restrict__ out) {
    __shared__ float red[32][65];
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = min(m, r0 + 256);
    const int v = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col0 = blockIdx.x * 64 + v * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (col0 < c) {
#pragma unroll 8
        for (int64_t r = r0 + rl; r < r1; r += 32) {
            float q[8];
            u32x4 pk;
            if (SRC == 1) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src32 + r * c + col0), b = *reinterpret_cast<const f32x4*>(src32 + r * c + col0 + 4);
                const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                pk = pack8(f);
            } else {
                float gg[8], aa[8];
                unpack8(*reinterpret_cast<const u32x4*>(g + r * c + col0), gg);
                unpack8(*reinterpret_cast<const u32x4*>(act + r * c + col0), aa);
#pragma unroll
                for (int e = 0; e < 8; ++e) gg[e] = aa[e] > 0.f ? gg[e] : 0.f;
                pk = pack8(gg);
            }
            *reinterpret_cast<u32x4*>(dst + r * c + col0) = pk;
            unpack8(pk, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += q[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][v * 8 + e] = acc[e];
    __syncthreads();
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && col < c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
        atomicAdd(out + col, s);
    }
}

// ---------------------------------------------------------------- max pool 3x3 / 2 with zero pad 1
__global__ void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, uint8_t* __restrict__ amax, int N, int H,
                                   int W, int C8, int Ho, int Wo) {
    const int64_t total = (int64_t)N * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % C8);
        int64_t t = i / C8;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best[8];
        unsigned char arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -1.f; arg[e] = 0; }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int iy = oy * 2 - 1 + k / 3, ix = ox * 2 - 1 + k % 3;
            float v[8];
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                unpack8(*reinterpret_cast<const u32x4*>(x + ((((int64_t)n * H + iy) * W + ix) * C8 + cv) * 8), v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;     // ZeroPadding2D: the pad value takes part in the max
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (v[e] > best[e]) { best[e] = v[e]; arg[e] = (unsigned char)k; }
        }
        *reinterpret_cast<u32x4*>(y + i * 8) = pack8(best);
        u32x2 a;
        a[0] = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
        a[1] = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
        *reinterpret_cast<u32x2*>(amax + i * 8) = a;
    }
}

// gather form: input pixel (iy,ix) receives gy of every window whose argmax points at it (deterministic)
__global__ void maxpool_bwd_kernel(const bf16_t* __restrict__ gy, const uint8_t* __restrict__ amax, bf16_t* __restrict__ gx, int N,
                                   int H, int W, int C8, int Ho, int Wo) {
    const int64_t total = (int64_t)N * H * W * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % C8);
        int64_t t = i / C8;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // windows oy with oy*2-1 <= iy <= oy*2+1
        const int oy_lo = (iy) / 2, oy_hi = (iy + 1) / 2;
        const int ox_lo = (ix) / 2, ox_hi = (ix + 1) / 2;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (oy >= Ho) continue;
            const int ky = iy - (oy * 2 - 1);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ox >= Wo) continue;
                const int kx = ix - (ox * 2 - 1);
                const unsigned k = (unsigned)(ky * 3 + kx);
                const int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * C8 + cv;
                const u32x2 a = *reinterpret_cast<const u32x2*>(amax + o * 8);
                float g[8];
                unpack8(*reinterpret_cast<const u32x4*>(gy + o * 8), g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned ak = (a[e >> 2] >> ((e & 3) * 8)) & 0xFFu;
                    if (ak == k) acc[e] += g[e];
                }
            }
        }
        *reinterpret_cast<u32x4*>(gx + i * 8) = pack8(acc);
    }
}

// maxpool_bwd_kernel fused with the BatchNorm-backward REDUCE of the layer whose activation the pool read (the ResNet stem: conv1_bn ->
// conv1_relu -> pool1): while the gradient of the activation is being written, its masked sums  sum g*m  and  sum g*m*xhat  (m: the
// layer's ReLU bit mask, xhat = (z - mean) * invstd; g as STORED, i.e. rounded to bf16) go to the backward partial slots -- what
// bn_bwd_reduce_kernel<2> would add after re-reading g, z and the mask (130 MB at 375x1242, batch 4, and a launch).  64 channels:
// 8 channel vectors x 32 pixel lanes per workgroup, a run of pixels per workgroup.
__global__ __launch_bounds__(256) void maxpool_bwd_bnreduce_kernel(const bf16_t* __restrict__ gy, const uint8_t* __restrict__ amax, bf16_t* __restrict__ gx,
                                                                   int N, int H, int W, int Ho, int Wo, const bf16_t* __restrict__ z,
                                                                   const uint8_t* __restrict__ relu_mask, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, float* __restrict__ part, int pix_per_block) {
    __shared__ float red[32][8][17];
    constexpr int C8 = 8, C = 64;
    const int v = threadIdx.x & 7, rl = threadIdx.x >> 3;
    float mu[8], is[8], sg[8], sgx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = mean[v * 8 + e]; is[e] = invstd[v * 8 + e]; sg[e] = sgx[e] = 0.f; }
    const int64_t total = (int64_t)N * H * W;
    const int64_t p_begin = (int64_t)blockIdx.x * pix_per_block, p_end = min(total, p_begin + (int64_t)pix_per_block);
    for (int64_t px = p_begin + rl; px < p_end; px += 32) {
        const int64_t i = px * C8 + v;
        const unsigned m = relu_mask[i];
        u32x4 zraw = *reinterpret_cast<const u32x4*>(z + i * 8);
        int64_t t = px;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // the (at most four) pooled cells whose window holds this pixel: cell rows iy / 2 and (iy + 1) / 2, columns likewise.  All eight
        // loads are issued before any of them is used (the nested loops with their early `continue`s waited for each cell's pair in turn:
        // the kernel ran at 2.5 TB/s of its 146 MB); a cell that does not exist, or is the same as its neighbour, is read at a clamped
        // index and masked out.  Same additions in the same order.
        const int oy_c[2] = {iy / 2, (iy + 1) / 2}, ox_c[2] = {ix / 2, (ix + 1) / 2};
        u32x2 a4[4];
        u32x4 g4[4];
        unsigned k4[4];
        bool ok4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = oy_c[q >> 1], ox = ox_c[q & 1];
            ok4[q] = oy < Ho && ox < Wo && !((q >> 1) && oy_c[1] == oy_c[0]) && !((q & 1) && ox_c[1] == ox_c[0]);
            const int oyc = oy < Ho ? oy : Ho - 1, oxc = ox < Wo ? ox : Wo - 1;
            k4[q] = (unsigned)((iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1)));
            const int64_t o = (((int64_t)n * Ho + oyc) * Wo + oxc) * C8 + v;
            a4[q] = *reinterpret_cast<const u32x2*>(amax + o * 8);
            g4[q] = *reinterpret_cast<const u32x4*>(gy + o * 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float g[8];
            unpack8(g4[q], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned ak = (a4[q][e >> 2] >> ((e & 3) * 8)) & 0xFFu;
                if (ok4[q] && ak == k4[q]) acc[e] += g[e];
            }
        }
        const u32x4 pk = pack8(acc);
        *reinterpret_cast<u32x4*>(gx + i * 8) = pk;
        float g[8], zz[8];
        unpack8(pk, g);
        unpack8(zraw, zz);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gm = ((m >> e) & 1u) ? g[e] : 0.f;
            sg[e] += gm;
            sgx[e] += gm * ((zz[e] - mu[e]) * is[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[rl][v][e] = sg[e]; red[rl][v][8 + e] = sgx[e]; }
    __syncthreads();
    for (int s_ = 16; s_ > 0; s_ >>= 1) {
        if (rl < s_) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[rl][v][e] += red[rl + s_][v][e];
        }
        __syncthreads();
    }
    if (threadIdx.x < 128) {
        const int stat = threadIdx.x >> 6, cl = threadIdx.x & 63;
        const int slot = blockIdx.x & (FRCNN_STAT_SLOTS - 1);
        atomicAdd(part + ((int64_t)slot * 2 + stat) * C + cl, red[0][cl >> 3][stat * 8 + (cl & 7)]);
    }
}

// ---------------------------------------------------------------- SGD + bf16 refresh
__global__ void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, bf16_t* __restrict__ wb,
                           int64_t n, float momentum, float l2x2, float gscale, const int64_t* __restrict__ step,
                           const int64_t* __restrict__ bounds, const float* __restrict__ values, int nb) {
    const int64_t st = *step;
    int k = 0;
    while (k < nb && st > bounds[k]) ++k;            // Keras: values[k] while step <= boundaries[k]
    const float lr = values[k];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        const float gi = g[i] * gscale + l2x2 * wi;
        const float vi = momentum * v[i] - lr * gi;
        const float wn = wi + vi;
        v[i] = vi;
        w[i] = wn;
        if (wb) wb[i] = (bf16_t)wn;
    }
}
__global__ void step_inc_kernel(int64_t* step) { *step += 1; }

// the optimizer step of a training plan in one launch: both decay ranges, the stem's packed bf16 taps, the step counter
__global__ void sgd_fused_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, bf16_t* __restrict__ wb,
                                 int64_t n, float momentum, float gscale, int64_t* __restrict__ step, const int64_t* __restrict__ bounds,
                                 const float* __restrict__ values, int nb, const frcnn_sgd_fused f) {
    const int64_t st = *step;
    int k = 0;
    while (k < nb && st > bounds[k]) ++k;            // Keras: values[k] while step <= boundaries[k]
    const float lr = values[k];
    const float l2x2 = 2.f * f.l2;
    const int64_t stem_end = f.stem_begin >= 0 ? f.stem_begin + (int64_t)f.stem_cout * 147 : -1;
    bf16_t* wp = reinterpret_cast<bf16_t*>(f.stem_packed);
    // A thread owns at most ONE element of the stem kernel (the grid has at least as many threads as the stem has elements: checked by
    // the host): it is only remembered inside the streaming loop and re-laid-out after it.  (With the index arithmetic -- four integer
    // divisions -- inside the loop the compiler computed it for every element, speculatively: the launch took 130 us instead of 52.)
    int stem_j = -1;
    float stem_v = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        // the gradient and the momentum are touched once per step: streamed past the caches (the masters are read again by the next step's
        // weight re-layouts, the bf16 copies by its convolutions); same-box A/B 3.805 -> 3.783, 3.808 -> 3.805
        const float gi = __builtin_nontemporal_load(g + i) * gscale + (i < f.decay_end ? l2x2 : 0.f) * wi;
        const float vi = momentum * __builtin_nontemporal_load(v + i) - lr * gi;
        const float wn = wi + vi;
        __builtin_nontemporal_store(vi, v + i);
        w[i] = wn;
        if (wb) wb[i] = (bf16_t)wn;
        const bool in_stem = i >= f.stem_begin && i < stem_end;
        stem_j = in_stem ? (int)(i - f.stem_begin) : stem_j;
        stem_v = in_stem ? wn : stem_v;
    }
    if (stem_j >= 0) {                               // [co][kh][kw][c] of the 7x7x3 kernel -> [co][kh][8][4]
        const int c = stem_j % 3, kw = (stem_j / 3) % 7, kh = (stem_j / 21) % 7, co = stem_j / 147;
        wp[((co * 7 + kh) * 8 + kw) * 4 + c] = (bf16_t)stem_v;
    }
    // every wave of every workgroup has read *step (it used the rate) before its workgroup arrives here: the last arriver may move it.
    // (No fence: nothing but the counter itself is handed over, and a fence would hold the workgroup until its stores have landed.)
    // (arrive == NULL: a launch over PART of the buffer -- a gradient bucket updated early, under the rest of the backward pass; the launch
    // over the last part carries the counter)
    __syncthreads();
    if (f.arrive && threadIdx.x == 0) {
        if (atomicAdd(f.arrive, 1u) == gridDim.x - 1) {
            *f.arrive = 0u;
            *step = st + 1;
        }
    }
}

__global__ void cast_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (bf16_t)s[i];
}

// w[co][kh][kw][ci] fp32 -> wt[ci][KH-1-kh][KW-1-kw][co] bf16
__global__ void transpose_flip_kernel(const float* __restrict__ w, bf16_t* __restrict__ wt, int Cout, int KH, int KW, int Cin) {
    const int64_t total = (int64_t)Cout * KH * KW * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the OUTPUT (coalesced writes): [ci][kh'][kw'][co]
        const int co = (int)(i % Cout);
        int64_t t = i / Cout;
        const int kwp = (int)(t % KW);
        t /= KW;
        const int khp = (int)(t % KH);
        const int ci = (int)(t / KH);
        const int kh = KH - 1 - khp, kw = KW - 1 - kwp;
        wt[i] = (bf16_t)w[(((int64_t)co * KH + kh) * KW + kw) * Cin + ci];
    }
}

// batched form: table[i] = {w ptr, wt ptr, cout, kh, kw, cin, first tile index, unused}; one launch for all layers.
// One workgroup per 32 x 32 (co, ci) tile of one filter tap, transposed through LDS: both the fp32 reads (ci contiguous)
// and the bf16 writes (co contiguous) are coalesced; the layer is found by a per-workgroup (scalar) binary search.
__global__ __launch_bounds__(256) void transpose_flip_batched_kernel(const long long* __restrict__ table, int n, long long total_tiles) {
    __shared__ float tile[32][33];
    const long long tidx = blockIdx.x;
    int lo = 0, hi = n - 1;                           // last layer whose first tile <= tidx
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 6] <= tidx) lo = mid; else hi = mid - 1;
    }
    const long long* d = table + lo * 8;
    const float* w = reinterpret_cast<const float*>(d[0]);
    bf16_t* wt = reinterpret_cast<bf16_t*>(d[1]);
    const int Cout = (int)d[2], KH = (int)d[3], KW = (int)d[4], Cin = (int)d[5];
    const int tci = (Cin + 31) >> 5, tco = (Cout + 31) >> 5;
    int t = (int)(tidx - d[6]);                       // ((tap * tco) + co_tile) * tci + ci_tile
    const int ci0 = (t % tci) * 32;
    t /= tci;
    const int co0 = (t % tco) * 32;
    const int tap = t / tco;                          // source tap kh*KW + kw; destination tap is the flipped one
    const int taps = KH * KW;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = co0 + ty + 8 * k, ci = ci0 + tx;
        tile[ty + 8 * k][tx] = (co < Cout && ci < Cin) ? w[((long long)co * taps + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
    const int tap_dst = taps - 1 - tap;               // (KH-1-kh)*KW + (KW-1-kw)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = ci0 + ty + 8 * k, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((long long)ci * taps + tap_dst) * Cout + co] = (bf16_t)tile[tx][ty + 8 * k];
    }
}

__global__ void stem_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp, int Cout) {
    const int total = Cout * 7 * 8 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i & 3, kw = (i >> 2) & 7, kh = (i >> 5) % 7, co = i / (7 * 32);
        float v = 0.f;
        if (c < 3 && kw < 7) v = w[((co * 7 + kh) * 7 + kw) * 3 + c];
        wp[i] = (bf16_t)v;
    }
}
__global__ void stem_unpack_grad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout) {
    const int total = Cout * 7 * 7 * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i % 3, kw = (i / 3) % 7, kh = (i / 21) % 7, co = i / 147;
        dw[i] = dwp[((co * 7 + kh) * 8 + kw) * 4 + c];
    }
}

}  // namespace

#define S_(stream) reinterpret_cast<hipStream_t>(stream)
#define BF(p) reinterpret_cast<bf16_t*>(p)
#define CBF(p) reinterpret_cast<const bf16_t*>(p)

extern "C" int frcnn_preprocess_u8_bgr_mean(const uint8_t* images, frcnn_bf16* out, int b, int h, int w, int hp, int wp, int pad,
                                            frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(images && out && b > 0 && hp >= h + pad && wp >= w + pad, "preprocess: bad arguments");
    const int64_t total = (int64_t)b * hp * wp;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(total, 256)), dim3(256), 0, S_(stream), images, BF(out), b, h, w, hp, wp, pad);
    FRCNN_CHECK_LAUNCH("preprocess");
    return FRCNN_OK;
}

extern "C" int frcnn_bn_finalize_train(const double* stats_partial, int tiles, int c, int64_t count, const float* gamma,
                                       const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                                       float* scale, float* shift, float* mean, float* invstd, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(stats_partial && gamma && beta && moving_mean && moving_var && scale && shift && mean && invstd && count > 0,
                    "bn_finalize_train: bad arguments");
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(cdiv(c, 64)), dim3(256), 0, S_(stream), stats_partial, tiles, c,
                       (float)(1.0 / (double)count), unbias, gamma, beta, moving_mean, moving_var, momentum, eps, scale, shift,
                       mean, invstd);
    FRCNN_CHECK_LAUNCH("bn_finalize_train");
    return FRCNN_OK;
}

extern "C" int frcnn_bn_finalize_eval(int c, const float* gamma, const float* beta, const float* moving_mean,
                                      const float* moving_var, float eps, float* scale, float* shift, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift, "bn_finalize_eval: null pointer");
    hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3(cdiv(c, 64)), dim3(64), 0, S_(stream), c, gamma, beta, moving_mean,
                       moving_var, eps, scale, shift);
    FRCNN_CHECK_LAUNCH("bn_finalize_eval");
    return FRCNN_OK;
}

extern "C" int frcnn_bn_apply(const frcnn_bf16* z, const float* scale, const float* shift, const frcnn_bf16* res, int relu,
                              frcnn_bf16* out, int64_t m, int c, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(z && scale && shift && out && c % 8 == 0, "bn_apply: bad arguments (c %% 8 != 0?)");
    const int64_t nvec = m * (c / 8);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(nvec, 256)), dim3(256), 0, S_(stream), CBF(z), scale, shift, CBF(res), relu,
                       BF(out), nvec, c / 8);
    FRCNN_CHECK_LAUNCH("bn_apply");
    return FRCNN_OK;
}

// rows per workgroup of the strip kernels: ~1024 workgroups, at least 64 rows (the per-workgroup statistics prologue
// must stay small next to the streamed rows)
static int strip_rows_per_block(int64_t m, int c) {
    const int strips = (c + 63) / 64;
    int64_t wgs = 1024;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_BN_WGS")) wgs = atoll(e);
#endif
    int64_t chunks = wgs / strips;
    if (chunks < 1) chunks = 1;
    int64_t rows = (m + chunks - 1) / chunks;
    if (rows < 64) rows = 64;
    return (int)rows;
}

extern "C" int frcnn_bn_train_apply(const frcnn_bf16* z, const double* stats_partial, int slots, int64_t count, const float* gamma,
                                    const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                                    const frcnn_bf16* res, int relu, frcnn_bf16* out, uint8_t* relu_mask, float* mean, float* invstd,
                                    int64_t m, int c, const frcnn_fp8_out* f8, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(!f8 || (f8->out8 && f8->qscale), "bn_train_apply: fp8 output without buffer / scale");
    Bn2 extra{};
    if (f8) { extra.out8 = f8->out8; extra.qscale = f8->qscale; extra.amax = f8->amax; }
    FRCNN_CHECK_ARG(z && stats_partial && gamma && beta && moving_mean && moving_var && (out || f8) && mean && invstd && count > 0 &&
                        slots > 0 && c % 8 == 0,
                    "bn_train_apply: bad arguments (out may be NULL only with an fp8 twin)");
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    const int rows = strip_rows_per_block(m, c);
    const int strips = (c + 63) / 64, chunks = (int)((m + rows - 1) / rows);
    const dim3 grid((unsigned)(strips * 8 * ((chunks + 7) / 8)));
#define FRCNN_BN_LAUNCH(V)                                                                                                          \
    hipLaunchKernelGGL(bn_train_apply_kernel<V>, grid, dim3(256), 0, S_(stream), CBF(z), stats_partial, slots, gamma, beta, moving_mean, \
                       moving_var, momentum, eps, (float)(1.0 / (double)count), unbias, CBF(res), relu, BF(out), relu_mask, mean, invstd, \
                       m, c, rows, strips, chunks, extra)
#ifdef FRCNN_SWEEP
    const char* ev = getenv("FRCNN_BN_VAR");
    const int var = ev ? atoi(ev) : 0;
    if (var == 4) FRCNN_BN_LAUNCH(4);
    else
#endif
    FRCNN_BN_LAUNCH(0);
#undef FRCNN_BN_LAUNCH
    FRCNN_CHECK_LAUNCH("bn_train_apply");
    return FRCNN_OK;
}

static int bn_bwd_apply_fused_impl(const frcnn_bf16* gout, const frcnn_bf16* act, const uint8_t* relu_mask, const frcnn_bf16* z,
                                   const float* mean, const float* invstd, const float* gamma, const float* partial, int slots,
                                   float* dgamma, float* dbeta, frcnn_bf16* dz, frcnn_bf16* gpre, int64_t m, int c, int64_t count,
                                   float param_grad_scale, const frcnn_fp8_out* f8, const frcnn_bn_reduce* red2, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gout && z && mean && invstd && gamma && partial && dgamma && dbeta && (dz || f8) && m > 0 && slots > 0 && c % 8 == 0 &&
                        !(act && relu_mask),
                    "bn_bwd_apply_fused: bad arguments (dz may be NULL only with an fp8 twin)");
    BnRed2 r2 = {nullptr, nullptr, nullptr, nullptr};
    if (red2) {
        FRCNN_CHECK_ARG(red2->z && red2->mean && red2->invstd && red2->partial && c % 64 == 0 && relu_mask,
                        "bn_bwd_apply_fused_red2: second reduce needs z / mean / invstd / partial, a ReLU bit mask and c %% 64 == 0");
        r2.z = CBF(red2->z); r2.mean = red2->mean; r2.invstd = red2->invstd; r2.part = red2->partial;
    }
    FRCNN_CHECK_ARG(!f8 || (f8->out8 && f8->qscale), "bn_bwd_apply_fused: fp8 output without buffer / scale");
    const int rows = strip_rows_per_block(m, c);
    const int strips = (c + 63) / 64, chunks = (int)((m + rows - 1) / rows);
    const dim3 grid((unsigned)(strips * 8 * ((chunks + 7) / 8)));
    const float inv_m = (float)(1.0 / (double)(count > 0 ? count : m));
#define FRCNN_LAUNCH(MODE, PTR, LEG)                                                                                                      \
    hipLaunchKernelGGL((bn_bwd_apply_fused_kernel<MODE, LEG>), grid, dim3(256), 0, S_(stream), CBF(gout), (const void*)(PTR), CBF(z), mean, \
                       invstd, gamma, partial, slots, inv_m, param_grad_scale, dgamma, dbeta, BF(dz), BF(gpre), m, c, rows, strips, chunks,   \
                       f8 ? f8->out8 : (uint8_t*)nullptr, f8 ? f8->qscale : (const float*)nullptr, f8 ? f8->amax : (float*)nullptr, r2)
    if (red2) {
        hipLaunchKernelGGL((bn_bwd_apply_fused_kernel<2, 0, true>), grid, dim3(256), 0, S_(stream), CBF(gout), (const void*)relu_mask, CBF(z), mean,
                           invstd, gamma, partial, slots, inv_m, param_grad_scale, dgamma, dbeta, BF(dz), BF(gpre), m, c, rows, strips, chunks,
                           f8 ? f8->out8 : (uint8_t*)nullptr, f8 ? f8->qscale : (const float*)nullptr, f8 ? f8->amax : (float*)nullptr, r2);
        FRCNN_CHECK_LAUNCH("bn_bwd_apply_fused_red2");
        return FRCNN_OK;
    }
#ifdef FRCNN_SWEEP
    const char* ev = getenv("FRCNN_BN_VAR");
    if (ev && atoi(ev) == 4) {
        if (relu_mask) FRCNN_LAUNCH(2, relu_mask, 1);
        else if (act) FRCNN_LAUNCH(1, act, 1);
        else FRCNN_LAUNCH(0, nullptr, 1);
    } else
#endif
    if (relu_mask) FRCNN_LAUNCH(2, relu_mask, 0);
    else if (act) FRCNN_LAUNCH(1, act, 0);
    else FRCNN_LAUNCH(0, nullptr, 0);
#undef FRCNN_LAUNCH
    FRCNN_CHECK_LAUNCH("bn_bwd_apply_fused");
    return FRCNN_OK;
}

extern "C" int frcnn_bn_bwd_apply_fused(const frcnn_bf16* gout, const frcnn_bf16* act, const uint8_t* relu_mask, const frcnn_bf16* z,
                                        const float* mean, const float* invstd, const float* gamma, const float* partial, int slots,
                                        float* dgamma, float* dbeta, frcnn_bf16* dz, frcnn_bf16* gpre, int64_t m, int c, int64_t count,
                                        float param_grad_scale, const frcnn_fp8_out* f8, frcnn_stream_t stream) {
    return bn_bwd_apply_fused_impl(gout, act, relu_mask, z, mean, invstd, gamma, partial, slots, dgamma, dbeta, dz, gpre, m, c, count, param_grad_scale, f8,
                                   nullptr, stream);
}

extern "C" int frcnn_bn_bwd_apply_fused_red2(const frcnn_bf16* gout, const uint8_t* relu_mask, const frcnn_bf16* z, const float* mean,
                                             const float* invstd, const float* gamma, const float* partial, int slots, float* dgamma,
                                             float* dbeta, frcnn_bf16* dz, int64_t m, int c, int64_t count, float param_grad_scale,
                                             const frcnn_fp8_out* f8, const frcnn_bn_reduce* red2, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(red2, "bn_bwd_apply_fused_red2: null frcnn_bn_reduce");
    return bn_bwd_apply_fused_impl(gout, nullptr, relu_mask, z, mean, invstd, gamma, partial, slots, dgamma, dbeta, dz, nullptr, m, c, count,
                                   param_grad_scale, f8, red2, stream);
}

extern "C" int frcnn_bn_bwd_blocks(int64_t m) { (void)m; return FRCNN_STAT_SLOTS; }

extern "C" int frcnn_bn_bwd_reduce(const frcnn_bf16* gout, const frcnn_bf16* act, const uint8_t* relu_mask, const frcnn_bf16* z,
                                   const float* mean, const float* invstd, float* partial, int64_t m, int c, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gout && z && mean && invstd && partial && c % 8 == 0 && m > 0 && !(act && relu_mask), "bn_bwd_reduce: bad arguments");
    const int rows = strip_rows_per_block(m, c);
    const int strips = (c + 63) / 64, chunks = (int)((m + rows - 1) / rows);
    const dim3 grid((unsigned)(strips * 8 * ((chunks + 7) / 8)));
#define FRCNN_LAUNCH(MODE, PTR)                                                                                                    \
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<MODE>, grid, dim3(256), 0, S_(stream), CBF(gout), (const void*)(PTR), CBF(z), mean, invstd, \
                       partial, m, c, rows, strips, chunks)
    if (relu_mask) FRCNN_LAUNCH(2, relu_mask);
    else if (act) FRCNN_LAUNCH(1, act);
    else FRCNN_LAUNCH(0, nullptr);
#undef FRCNN_LAUNCH
    FRCNN_CHECK_LAUNCH("bn_bwd_reduce");
    return FRCNN_OK;
}

extern "C" int frcnn_bn_train_apply_dual(const frcnn_bf16* z, const double* stats_partial, const float* gamma, const float* beta,
                                         float* moving_mean, float* moving_var, float* mean, float* invstd, const frcnn_bf16* z2,
                                         const double* stats_partial2, const float* gamma2, const float* beta2, float* moving_mean2,
                                         float* moving_var2, float* mean2, float* invstd2, int slots, int64_t count, float momentum,
                                         float eps, int relu, frcnn_bf16* out, uint8_t* relu_mask, int64_t m, int c, const frcnn_fp8_out* f8,
                                         frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(!f8 || (f8->out8 && f8->qscale), "bn_train_apply_dual: fp8 output without buffer / scale");
    FRCNN_CHECK_ARG(z && stats_partial && gamma && beta && moving_mean && moving_var && mean && invstd && z2 && stats_partial2 && gamma2 &&
                        beta2 && moving_mean2 && moving_var2 && mean2 && invstd2 && out && count > 0 && slots > 0 && c % 8 == 0,
                    "bn_train_apply_dual: bad arguments");
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    const int rows = strip_rows_per_block(m, c);
    const int strips = (c + 63) / 64, chunks = (int)((m + rows - 1) / rows);
    Bn2 b2{};
    if (f8) { b2.out8 = f8->out8; b2.qscale = f8->qscale; b2.amax = f8->amax; }
    b2.z = CBF(z2); b2.part = stats_partial2; b2.gamma = gamma2; b2.beta = beta2; b2.mm = moving_mean2; b2.mv = moving_var2;
    b2.mean_o = mean2; b2.invstd_o = invstd2;
    hipLaunchKernelGGL((bn_train_apply_kernel<0, true>), dim3((unsigned)(strips * 8 * ((chunks + 7) / 8))), dim3(256), 0, S_(stream), CBF(z),
                       stats_partial, slots, gamma, beta, moving_mean, moving_var, momentum, eps, (float)(1.0 / (double)count), unbias,
                       (const bf16_t*)nullptr, relu, BF(out), relu_mask, mean, invstd, m, c, rows, strips, chunks, b2);
    FRCNN_CHECK_LAUNCH("bn_train_apply_dual");
    return FRCNN_OK;
}

// ---- the stem's BatchNorm + ReLU + max pool without the activation tensor
extern "C" int frcnn_bn_train_apply_maxpool(const frcnn_bf16* z, const double* stats_partial, int slots, int64_t count, const float* gamma,
                                            const float* beta, float* moving_mean, float* moving_var, float momentum, float eps,
                                            frcnn_bf16* pooled, uint8_t* argmax, uint8_t* relu_mask, float* mean, float* invstd, int n,
                                            int h, int w, int c, int ho, int wo, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(z && stats_partial && gamma && beta && moving_mean && moving_var && pooled && argmax && mean && invstd && count > 0 &&
                        slots > 0 && c % 8 == 0 && n > 0 && ho == (h + 2 - 3) / 2 + 1 && wo == (w + 2 - 3) / 2 + 1,
                    "bn_train_apply_maxpool: bad arguments");
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    const int64_t cells = (int64_t)n * ho * wo;
    const int per = strip_rows_per_block(cells, c);
    const int strips = (c + 63) / 64, chunks = (int)((cells + per - 1) / per);
#ifndef FRCNN_POOL_PER_CELL
    {
        const int tiles_x = (wo + POOL_TW - 1) / POOL_TW, tiles_y = (ho + POOL_TH - 1) / POOL_TH;
        const long long wgs = (long long)n * tiles_x * tiles_y * strips;
        FRCNN_CHECK_ARG(wgs < (1ll << 31), "bn_train_apply_maxpool: grid too large");
        hipLaunchKernelGGL(bn_train_apply_pool_tiled_kernel, dim3((unsigned)wgs), dim3(256), 0, S_(stream), CBF(z), stats_partial, slots, gamma, beta,
                           moving_mean, moving_var, momentum, eps, (float)(1.0 / (double)count), unbias, BF(pooled), argmax, relu_mask, mean, invstd, n,
                           h, w, c, ho, wo, strips, tiles_x, tiles_y);
        FRCNN_CHECK_LAUNCH("bn_train_apply_maxpool");
        return FRCNN_OK;
    }
#endif
    hipLaunchKernelGGL(bn_train_apply_pool_kernel, dim3((unsigned)(strips * 8 * ((chunks + 7) / 8))), dim3(256), 0, S_(stream), CBF(z),
                       stats_partial, slots, gamma, beta, moving_mean, moving_var, momentum, eps, (float)(1.0 / (double)count), unbias,
                       BF(pooled), argmax, relu_mask, mean, invstd, n, h, w, c, ho, wo, per, strips, chunks);
    FRCNN_CHECK_LAUNCH("bn_train_apply_maxpool");
    return FRCNN_OK;
}

extern "C" int frcnn_bn_bwd_finalize(const float* partial, int blocks, int c, int64_t m, float* dgamma, float* dbeta, float* c1,
                                     float* c2, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(partial && dgamma && dbeta && c1 && c2 && m > 0, "bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, 64)), dim3(256), 0, S_(stream), partial, blocks, c,
                       (float)(1.0 / (double)m), dgamma, dbeta, c1, c2);
    FRCNN_CHECK_LAUNCH("bn_bwd_finalize");
    return FRCNN_OK;
}

extern "C" int frcnn_bn_bwd_apply(const frcnn_bf16* gout, const frcnn_bf16* act, const frcnn_bf16* z, const float* mean,
                                  const float* invstd, const float* gamma, const float* c1, const float* c2, frcnn_bf16* dz,
                                  frcnn_bf16* gpre, int64_t m, int c, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gout && z && mean && invstd && gamma && c1 && c2 && dz && c % 8 == 0, "bn_bwd_apply: bad arguments");
    const int64_t nvec = m * (c / 8);
    if (act)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid_for(nvec, 256)), dim3(256), 0, S_(stream), CBF(gout), CBF(act), CBF(z),
                           mean, invstd, gamma, c1, c2, BF(dz), BF(gpre), nvec, c / 8);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(nvec, 256)), dim3(256), 0, S_(stream), CBF(gout), CBF(act), CBF(z),
                           mean, invstd, gamma, c1, c2, BF(dz), BF(gpre), nvec, c / 8);
    FRCNN_CHECK_LAUNCH("bn_bwd_apply");
    return FRCNN_OK;
}

extern "C" int frcnn_relu_bwd(const frcnn_bf16* g, const frcnn_bf16* act, frcnn_bf16* out, int64_t n, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(g && act && out && n % 8 == 0, "relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, S_(stream), CBF(g), CBF(act), BF(out), n / 8);
    FRCNN_CHECK_LAUNCH("relu_bwd");
    return FRCNN_OK;
}

extern "C" int frcnn_colsum_bf16(const frcnn_bf16* x, int64_t m, int c, int ld, float* out, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(x && out && c > 0 && ld >= c, "colsum: bad arguments");
    // (vector path: whole 8-column groups inside [0, c), 16-byte aligned rows)
    const int vec = (ld % 8 == 0 && c % 8 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(c, 64), cdiv(m, 256)), dim3(256), 0, S_(stream), CBF(x), m, c, ld, out, vec);
    FRCNN_CHECK_LAUNCH("colsum");
    return FRCNN_OK;
}

extern "C" int frcnn_cast_colsum(const float* src, frcnn_bf16* dst, int64_t m, int c, float* colsum, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(src && dst && colsum && m > 0 && c > 0 && c % 8 == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0,
                    "cast_colsum: bad arguments (c %% 8, 16-byte aligned)");
    hipLaunchKernelGGL(colsum_produce_kernel<1>, dim3(cdiv(c, 64), cdiv(m, 256)), dim3(256), 0, S_(stream), src, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, BF(dst), m, c, colsum);
    FRCNN_CHECK_LAUNCH("cast_colsum");
    return FRCNN_OK;
}

extern "C" int frcnn_relu_bwd_colsum(const frcnn_bf16* g, const frcnn_bf16* act, frcnn_bf16* out, int64_t m, int c, float* colsum,
                                     frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(g && act && out && colsum && m > 0 && c > 0 && c % 8 == 0 &&
                        ((reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(act) | reinterpret_cast<size_t>(out)) & 15) == 0,
                    "relu_bwd_colsum: bad arguments (c %% 8, 16-byte aligned)");
    hipLaunchKernelGGL(colsum_produce_kernel<2>, dim3(cdiv(c, 64), cdiv(m, 256)), dim3(256), 0, S_(stream), (const float*)nullptr, CBF(g), CBF(act),
                       BF(out), m, c, colsum);
    FRCNN_CHECK_LAUNCH("relu_bwd_colsum");
    return FRCNN_OK;
}

extern "C" int frcnn_maxpool3x3s2_fwd(const frcnn_bf16* x, frcnn_bf16* y, uint8_t* argmax, int n, int h, int w, int c, int ho,
                                      int wo, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(x && y && argmax && c % 8 == 0 && ho == (h + 2 - 3) / 2 + 1 && wo == (w + 2 - 3) / 2 + 1, "maxpool_fwd: bad arguments");
    const int64_t total = (int64_t)n * ho * wo * (c / 8);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, S_(stream), CBF(x), BF(y), argmax, n, h, w, c / 8,
                       ho, wo);
    FRCNN_CHECK_LAUNCH("maxpool_fwd");
    return FRCNN_OK;
}

extern "C" int frcnn_maxpool3x3s2_bwd(const frcnn_bf16* gy, const uint8_t* argmax, frcnn_bf16* gx, int n, int h, int w, int c, int ho,
                                      int wo, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gy && gx && argmax && c % 8 == 0, "maxpool_bwd: bad arguments");
    const int64_t total = (int64_t)n * h * w * (c / 8);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, S_(stream), CBF(gy), argmax, BF(gx), n, h, w, c / 8,
                       ho, wo);
    FRCNN_CHECK_LAUNCH("maxpool_bwd");
    return FRCNN_OK;
}

extern "C" int frcnn_maxpool3x3s2_bwd_bnreduce(const frcnn_bf16* gy, const uint8_t* argmax, frcnn_bf16* gx, int n, int h, int w, int c, int ho,
                                               int wo, const struct frcnn_bn_reduce* red, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gy && argmax && gx && red && red->z && red->relu_mask && red->mean && red->invstd && red->partial,
                    "maxpool_bwd_bnreduce: null pointer (the reduce needs z, the ReLU bit mask, mean, invstd and the partial slots)");
    FRCNN_CHECK_ARG(c == 64 && ho == (h + 2 - 3) / 2 + 1 && wo == (w + 2 - 3) / 2 + 1, "maxpool_bwd_bnreduce: 64 channels (the ResNet stem), 3x3 / 2 / pad 1 geometry");
    const int64_t total = (int64_t)n * h * w;
    int64_t per = (total + 1023) / 1024;         // ~1024 workgroups, whole 32-pixel rounds
    per = (per + 31) / 32 * 32;
    if (per < 32) per = 32;
    hipLaunchKernelGGL(maxpool_bwd_bnreduce_kernel, dim3((unsigned)((total + per - 1) / per)), dim3(256), 0, S_(stream), CBF(gy), argmax, BF(gx), n, h, w,
                       ho, wo, CBF(red->z), red->relu_mask, red->mean, red->invstd, red->partial, (int)per);
    FRCNN_CHECK_LAUNCH("maxpool_bwd_bnreduce");
    return FRCNN_OK;
}

extern "C" int frcnn_sgd_momentum(float* w, const float* g, float* v, frcnn_bf16* w_bf16, int64_t n, float momentum, float l2,
                                  float grad_scale, const int64_t* step, const int64_t* boundaries, const float* values, int nb,
                                  frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(w && g && v && step && values && (nb == 0 || boundaries), "sgd_momentum: null pointer");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, S_(stream), w, g, v, BF(w_bf16), n, momentum, 2.f * l2,
                       grad_scale, step, boundaries, values, nb);
    FRCNN_CHECK_LAUNCH("sgd_momentum");
    return FRCNN_OK;
}

extern "C" int frcnn_sgd_momentum_fused(float* w, const float* g, float* v, frcnn_bf16* w_bf16, int64_t n, float momentum, float grad_scale,
                                        int64_t* step, const int64_t* boundaries, const float* values, int nb, const frcnn_sgd_fused* f,
                                        frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(w && g && v && step && values && (nb == 0 || boundaries) && f, "sgd_momentum_fused: null pointer");
    FRCNN_CHECK_ARG(f->decay_end >= 0 && f->decay_end <= n, "sgd_momentum_fused: decay_end outside [0, n]");
    FRCNN_CHECK_ARG(f->stem_begin < 0 || (f->stem_packed && f->stem_cout > 0 && f->stem_begin + (int64_t)f->stem_cout * 147 <= n),
                    "sgd_momentum_fused: stem range outside the buffer or no packed destination");
    // (one stem element per thread at most: the grid is min(n, 2048 * 256) threads)
    FRCNN_CHECK_ARG(f->stem_begin < 0 || (int64_t)f->stem_cout * 147 <= (int64_t)grid_for(n, 256) * 256, "sgd_momentum_fused: stem kernel larger than the grid");
    hipLaunchKernelGGL(sgd_fused_kernel, dim3(grid_for(n, 256)), dim3(256), 0, S_(stream), w, g, v, BF(w_bf16), n, momentum, grad_scale, step,
                       boundaries, values, nb, *f);
    FRCNN_CHECK_LAUNCH("sgd_momentum_fused");
    return FRCNN_OK;
}

extern "C" int frcnn_step_increment(int64_t* step, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(step, "step_increment: null pointer");
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, S_(stream), step);
    FRCNN_CHECK_LAUNCH("step_increment");
    return FRCNN_OK;
}

// Device-to-device copy of nbytes (16-byte aligned pointers) with enough workgroups to run at HBM speed: the runtime's blit
// kernel moves the 5.6 MB image batch into the plan's static input buffer in 24 us (256 workgroups), this one in ~3.
__global__ __launch_bounds__(256) void copy_bytes_kernel(const u32x4* __restrict__ s, u32x4* __restrict__ d, int64_t n16, const unsigned char* __restrict__ st,
                                                         unsigned char* __restrict__ dt, int tail) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dt[threadIdx.x] = st[threadIdx.x];
}

extern "C" int frcnn_copy_bytes(const void* src, void* dst, int64_t nbytes, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(src && dst && nbytes >= 0, "copy_bytes: bad arguments");
    FRCNN_CHECK_ARG((reinterpret_cast<size_t>(src) & 15) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0, "copy_bytes: pointers must be 16-byte aligned");
    if (nbytes == 0) return FRCNN_OK;
    const int64_t n16 = nbytes / 16;
    const int tail = (int)(nbytes - n16 * 16);
    const int64_t want = (n16 + 255) / 256;
    const int grid = (int)(want < 1 ? 1 : want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid), dim3(256), 0, S_(stream), reinterpret_cast<const u32x4*>(src), reinterpret_cast<u32x4*>(dst), n16,
                       reinterpret_cast<const unsigned char*>(src) + n16 * 16, reinterpret_cast<unsigned char*>(dst) + n16 * 16, tail);
    FRCNN_CHECK_LAUNCH("copy_bytes");
    return FRCNN_OK;
}

// Up to 4 independent copies in ONE launch (blockIdx.y = copy): a training step's three inputs (image batch, ground-truth labels
// and boxes) enter the plan's static buffers with one kernel boundary instead of three.
struct CopyMulti {
    const unsigned char* src[4]; unsigned char* dst[4]; int64_t nbytes[4];
};

__global__ __launch_bounds__(256) void copy_bytes_multi_kernel(const CopyMulti c) {
    const int k = blockIdx.y;
    const int64_t n16 = c.nbytes[k] / 16;
    const u32x4* s = reinterpret_cast<const u32x4*>(c.src[k]);
    u32x4* d = reinterpret_cast<u32x4*>(c.dst[k]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
    const int tail = (int)(c.nbytes[k] - n16 * 16);
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) c.dst[k][n16 * 16 + threadIdx.x] = c.src[k][n16 * 16 + threadIdx.x];
}

extern "C" int frcnn_copy_bytes_multi(const void* const* srcs, void* const* dsts, const int64_t* nbytes, int n, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(srcs && dsts && nbytes && n >= 1 && n <= 4, "copy_bytes_multi: 1..4 copies per call");
    CopyMulti c;
    int64_t most = 0;
    for (int k = 0; k < 4; ++k) {
        const int j = k < n ? k : 0;
        FRCNN_CHECK_ARG(srcs[j] && dsts[j] && nbytes[j] >= 0, "copy_bytes_multi: bad arguments");
        FRCNN_CHECK_ARG((reinterpret_cast<size_t>(srcs[j]) & 15) == 0 && (reinterpret_cast<size_t>(dsts[j]) & 15) == 0,
                        "copy_bytes_multi: pointers must be 16-byte aligned");
        c.src[k] = reinterpret_cast<const unsigned char*>(srcs[j]);
        c.dst[k] = reinterpret_cast<unsigned char*>(dsts[j]);
        c.nbytes[k] = k < n ? nbytes[j] : 0;
        if (c.nbytes[k] > most) most = c.nbytes[k];
    }
    if (most == 0) return FRCNN_OK;
    const int64_t want = (most / 16 + 255) / 256;
    const int grid = (int)(want < 1 ? 1 : want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(copy_bytes_multi_kernel, dim3(grid, n), dim3(256), 0, S_(stream), c);
    FRCNN_CHECK_LAUNCH("copy_bytes_multi");
    return FRCNN_OK;
}

// Zero n buffers in ONE launch.  table (device, int64 [n + 1][2]): row i = {pointer, first 16-byte chunk of buffer i in the concatenated
// chunk space}; row n = {0, total chunks}.  A training step pre-zeroes seven accumulation targets (flat gradient, BatchNorm sums,
// scatter targets, split-K outputs): one fill at HBM speed instead of seven launches of the framework's fill kernel.
template <bool NT>
__global__ __launch_bounds__(256) void fill_zero_multi_kernel(const int64_t* __restrict__ table, int n, int64_t chunks_per_block) {
    const int64_t total = table[2 * n + 1];
    int64_t c0 = (int64_t)blockIdx.x * chunks_per_block;
    const int64_t c1 = c0 + chunks_per_block < total ? c0 + chunks_per_block : total;
    int seg = 0;
    while (seg + 1 < n && table[2 * (seg + 1) + 1] <= c0) ++seg;              // (uniform: scalar loads)
    const u32x4 z = {0u, 0u, 0u, 0u};
    while (c0 < c1) {
        const int64_t seg_end = table[2 * (seg + 1) + 1] < c1 ? table[2 * (seg + 1) + 1] : c1;
        u32x4* base = reinterpret_cast<u32x4*>(table[2 * seg]) - table[2 * seg + 1];
        for (int64_t i = c0 + threadIdx.x; i < seg_end; i += 256) {
            if (NT) __builtin_nontemporal_store(z, base + i);       // (a large fill whose buffers are first touched a millisecond later)
            else base[i] = z;
        }
        c0 = seg_end;
        ++seg;
    }
}

extern "C" int frcnn_fill_zero_multi(const int64_t* table, int n, int64_t total_chunks, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(table && n > 0 && total_chunks >= 0, "fill_zero_multi: bad arguments");
    if (total_chunks == 0) return FRCNN_OK;
    int64_t per = 2048;                                       // 32 KiB per workgroup
    int64_t grid = (total_chunks + per - 1) / per;
    if (grid > 16384) { per = (total_chunks + 16383) / 16384; grid = (total_chunks + per - 1) / per; }
    // (a fill of more than 32 MB -- the train plan's late fill of the backward pass's accumulation targets -- streams past the caches;
    // same-box A/B 3.802 -> 3.798, 3.798 -> 3.791)
    if (total_chunks * 16 > (32ll << 20)) hipLaunchKernelGGL(fill_zero_multi_kernel<true>, dim3((int)grid), dim3(256), 0, S_(stream), table, n, per);
    else hipLaunchKernelGGL(fill_zero_multi_kernel<false>, dim3((int)grid), dim3(256), 0, S_(stream), table, n, per);
    FRCNN_CHECK_LAUNCH("fill_zero_multi");
    return FRCNN_OK;
}

extern "C" int frcnn_cast_f32_bf16(const float* src, frcnn_bf16* dst, int64_t n, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(src && dst, "cast: null pointer");
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n, 256)), dim3(256), 0, S_(stream), src, BF(dst), n);
    FRCNN_CHECK_LAUNCH("cast");
    return FRCNN_OK;
}

extern "C" int frcnn_weights_transpose_flip(const float* w, frcnn_bf16* w_t, int cout, int kh, int kw, int cin, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(w && w_t, "weights_transpose_flip: null pointer");
    const int64_t total = (int64_t)cout * kh * kw * cin;
    hipLaunchKernelGGL(transpose_flip_kernel, dim3(grid_for(total, 256)), dim3(256), 0, S_(stream), w, BF(w_t), cout, kh, kw, cin);
    FRCNN_CHECK_LAUNCH("weights_transpose_flip");
    return FRCNN_OK;
}

extern "C" int frcnn_weights_transpose_flip_batched(const int64_t* table, int n, int64_t total, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(table && n > 0 && total > 0, "weights_transpose_flip_batched: bad arguments");
    FRCNN_CHECK_ARG(total < (1ll << 31), "weights_transpose_flip_batched: too many tiles");
    hipLaunchKernelGGL(transpose_flip_batched_kernel, dim3((unsigned)total), dim3(256), 0, S_(stream),
                       reinterpret_cast<const long long*>(table), n, (long long)total);
    FRCNN_CHECK_LAUNCH("weights_transpose_flip_batched");
    return FRCNN_OK;
}

extern "C" int frcnn_stem_pack_weights(const float* w, frcnn_bf16* w_packed, int cout, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(w && w_packed, "stem_pack_weights: null pointer");
    hipLaunchKernelGGL(stem_pack_kernel, dim3(cdiv(cout * 224, 256)), dim3(256), 0, S_(stream), w, BF(w_packed), cout);
    FRCNN_CHECK_LAUNCH("stem_pack_weights");
    return FRCNN_OK;
}

extern "C" int frcnn_stem_unpack_grad(const float* dw_packed, float* dw, int cout, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(dw_packed && dw, "stem_unpack_grad: null pointer");
    hipLaunchKernelGGL(stem_unpack_grad_kernel, dim3(cdiv(cout * 147, 256)), dim3(256), 0, S_(stream), dw_packed, dw, cout);
    FRCNN_CHECK_LAUNCH("stem_unpack_grad");
    return FRCNN_OK;
}
