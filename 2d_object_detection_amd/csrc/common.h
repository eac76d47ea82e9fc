// Shared device/host helpers for the gfx950 Faster-RCNN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "../../include/frcnn_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

void frcnn_set_error(const char* fmt, ...);
void frcnn_note_instantiation(const char* s);      // conv_tile.hip: what frcnn_last_conv_instantiation() returns

#define FRCNN_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            frcnn_set_error(__VA_ARGS__);          \
            return FRCNN_EINVAL;                   \
        }                                          \
    } while (0)

#define FRCNN_CHECK_LAUNCH(name)                                                   \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            frcnn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return FRCNN_ELAUNCH;                                                  \
        }                                                                          \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Dynamic LDS above 64 KiB must be opted into per kernel (gfx950 has 160 KiB per CU).  Not a
// stream operation and idempotent, so it is safe under graph capture.
static inline int frcnn_allow_big_lds(const void* func, size_t bytes) {
    if (bytes <= 65536) return 0;
    if (bytes > 163840) return -1;
    return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    bf16_t h = (bf16_t)f;                       // v_cvt_pk_bf16_f32: RNE, NaN-preserving
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_round(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

// unpack 8 bf16 (one 16-byte vector) to floats and back
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = (unsigned int)f32_to_bf16_bits(f[2 * i]) | ((unsigned int)f32_to_bf16_bits(f[2 * i + 1]) << 16);
    return v;
}

// 8 floats -> 8 OCP e4m3 bytes (RNE; gfx950's v_cvt_pk_fp8_f32 converts to the OCP format).  Inputs are clamped to the finite
// range first: e4m3fn has no infinity and an out-of-range value would otherwise become NaN.
__device__ __forceinline__ u32x2 pack8_fp8(const float* f, const float qs) {
    u32x2 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int w = 0;
        const float a = __builtin_amdgcn_fmed3f(f[4 * h] * qs, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(f[4 * h + 1] * qs, -448.f, 448.f);
        const float c = __builtin_amdgcn_fmed3f(f[4 * h + 2] * qs, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(f[4 * h + 3] * qs, -448.f, 448.f);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
        r[h] = (unsigned)w;
    }
    return r;
}
// 8 floats -> 8 OCP e5m2 bytes ("bf8": the gradient format -- 5 exponent bits, max 57344), clamped to the finite range
__device__ __forceinline__ u32x2 pack8_bf8(const float* f, const float qs) {
    u32x2 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int w = 0;
        const float a = __builtin_amdgcn_fmed3f(f[4 * h] * qs, -57344.f, 57344.f), b = __builtin_amdgcn_fmed3f(f[4 * h + 1] * qs, -57344.f, 57344.f);
        const float c = __builtin_amdgcn_fmed3f(f[4 * h + 2] * qs, -57344.f, 57344.f), d = __builtin_amdgcn_fmed3f(f[4 * h + 3] * qs, -57344.f, 57344.f);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
        r[h] = (unsigned)w;
    }
    return r;
}
// wave-wide max of non-negative floats folded into one of the FRCNN_FP8_AMAX_SLOTS slots of dst (device floats compared as their bit
// patterns: order-preserving for x >= 0).  One slot per wave of a launch (8192 slots: more than the waves of any launch here),
// because atomics on ONE address serialise (~0.1 us each: the 4096 waves of a BatchNorm launch on one word cost ~40 us, on 64
// words still 9 us of an 18 us launch -- tools/bn_bench.py, BN_F8=1 against BN_F8=noamax).
// Callable after per-thread early returns: with every lane of the wave active the reduction is a butterfly and lane 0 writes; with
// some lanes gone (a channel count that is not a multiple of 64, a tail row) the exited lanes' registers are undefined to a shuffle,
// so the maximum is collected from the ACTIVE lanes one by one (v_readlane on the ballot's set bits) and the first active lane writes.
// per-thread accumulation for atomic_amax: m <- max(m, |v|) on bit patterns (m >= 0), so that a NaN stays visible (see below)
__device__ __forceinline__ float amax_fold(float m, float v) {
    const unsigned a = __float_as_uint(m), b = __float_as_uint(v) & 0x7fffffffu;
    return __uint_as_float(a > b ? a : b);
}
// The reduction runs on the BIT PATTERNS of |v| (unsigned integer max: the order of non-negative floats, with +Inf above every finite
// value and every NaN above +Inf): fmaxf would drop a NaN operand, and a tensor holding NaN but no Inf would leave a finite amax behind --
// fp8_update_scales_kernel's `!(v <= 3e38f)` test then never fired for it and FasterRCNN.fp8_status() reported a healthy tensor.
__device__ __forceinline__ void atomic_amax(float* dst, float v) {
    const unsigned long long act = __ballot(1);
    unsigned u = __float_as_uint(v) & 0x7fffffffu;
    int writer = 0;
    if (act == ~0ull) {
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, sh));
    } else {
        unsigned m = 0u;
        for (unsigned long long rem = act; rem; rem &= rem - 1)
            m = max(m, (unsigned)__builtin_amdgcn_readlane(u, __ffsll((long long)rem) - 1));
        u = m;
        writer = __ffsll((long long)act) - 1;
    }
    if ((int)(threadIdx.x & 63) == writer && u > 0u) {
        const unsigned slot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (FRCNN_FP8_AMAX_SLOTS - 1);
        atomicMax(reinterpret_cast<unsigned*>(dst) + slot, u);
    }
}

// Philox4x32-10 (shared by the sampler; oracle/philox.py is the numpy twin)
__device__ __forceinline__ unsigned int philox_first(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                                     unsigned int k0, unsigned int k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0;
        const unsigned long long p1 = 0xCD9E8D57ull * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (unsigned int)p1; c2 = n2; c3 = (unsigned int)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
