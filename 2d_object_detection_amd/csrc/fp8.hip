// fp8 (OCP e4m3) support of the convolution path: quantisers and the delayed-scaling update (include/frcnn_hip.h, "fp8 (e4m3) MFMA
// convolution path").  The convolution itself is conv_tile_kernel<..., F8> (conv_tile.hip); the BatchNorm kernels write the fp8 twin of
// their output themselves (elementwise.hip, frcnn_fp8_out).  All HBM-bound byte movers: 16-byte loads, 8-byte stores.
#include "common.h"

namespace {

#define S_(s) reinterpret_cast<hipStream_t>(s)

// out8 = e4m3(clamp(x * qscale)), |x| max folded into *amax
template <bool E5M2>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const bf16_t* __restrict__ x, int64_t nvec, const float* __restrict__ qscale,
                                                           uint8_t* __restrict__ out8, float* __restrict__ amax) {
    const float qs = *qscale;
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), f);
        *reinterpret_cast<u32x2*>(out8 + i * 8) = E5M2 ? pack8_bf8(f, qs) : pack8_fp8(f, qs);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = amax_fold(mx, f[e]);
    }
    if (amax) atomic_amax(amax, mx);
}

// one workgroup per weight row: amax over the row, then the scaled conversion.  K is a multiple of 8.
__global__ __launch_bounds__(256) void quantize_weights_fp8_kernel(const long long* __restrict__ table, int n, long long total_rows) {
    __shared__ float red[4];
    const long long row_id = blockIdx.x;
    if (row_id >= total_rows) return;
    int lo = 0, hi = n - 1;                       // layer by prefix search over the table's "first workgroup" column
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 5] <= row_id) lo = mid; else hi = mid - 1;
    }
    const long long* t = table + lo * 8;
    uint8_t* w8 = reinterpret_cast<uint8_t*>(t[1]);
    float* scale = reinterpret_cast<float*>(t[2]);
    const long long K = t[4];
    const long long r = row_id - t[5];
    const bool src_bf16 = t[6] != 0;
    const float* src = reinterpret_cast<const float*>(t[0]) + r * K;
    const bf16_t* srcb = reinterpret_cast<const bf16_t*>(t[0]) + r * K;
    auto load8 = [&](const long long k, float* f) {
        if (src_bf16) {
            unpack8(*reinterpret_cast<const u32x4*>(srcb + k), f);
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + k), b = *reinterpret_cast<const f32x4*>(src + k + 4);
            f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
        }
    };
    float mx = 0.f;
    for (long long k = (long long)threadIdx.x * 8; k < K; k += 256 * 8) {
        float f[8];
        load8(k, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(f[e]));
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = mx > 0.f ? mx * (1.f / 448.f) : 1.f;
    const float qs = 1.f / sc;
    if (threadIdx.x == 0) scale[r] = sc;
    for (long long k = (long long)threadIdx.x * 8; k < K; k += 256 * 8) {
        float f[8];
        load8(k, f);
        *reinterpret_cast<u32x2*>(w8 + r * K + k) = pack8_fp8(f, qs);
    }
}

// one workgroup per tensor: maximum over its amax slots, then the next step's scales.  A non-finite amax (an overflowed activation)
// keeps the previous scale -- scale = Inf would dequantise 0 * Inf = NaN from the next step on -- and is counted in status[1]; an
// amax beyond what this step's scale could represent (limit x scale: the twin's values were clamped) is counted in status[0].
__global__ __launch_bounds__(256) void fp8_update_scales_kernel(const float* __restrict__ amax, float* __restrict__ scale, float* __restrict__ qscale, int n, float margin,
                                                                const float* __restrict__ limit, int* __restrict__ status) {
    __shared__ float red[4];
    __shared__ int bad[4];
    const int i = blockIdx.x;
    const f32x4* src = reinterpret_cast<const f32x4*>(amax + (long long)i * FRCNN_FP8_AMAX_SLOTS);
    float a = 0.f;
    int nonfinite = 0;
    for (int k = threadIdx.x; k < FRCNN_FP8_AMAX_SLOTS / 4; k += 256) {
        const f32x4 v = src[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (fmaxf drops a NaN operand: test the slots themselves.  The slots hold max |x| as float bits >= 0.)
            if (!(v[e] <= 3.0e38f)) nonfinite = 1; else a = fmaxf(a, v[e]);
        }
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        a = fmaxf(a, __shfl_xor(a, sh));
        nonfinite |= __shfl_xor(nonfinite, sh);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; bad[threadIdx.x >> 6] = nonfinite; }
    __syncthreads();
    a = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    nonfinite = bad[0] | bad[1] | bad[2] | bad[3];
    if (threadIdx.x == 0) {
        if (nonfinite) {
            if (status) atomicAdd(status + 1, 1);
        } else if (a > 0.f) {
            const float lim = limit ? limit[i] : 448.f;
            if (status && a > lim * scale[i] * 1.0001f) atomicAdd(status, 1);
            const float sc = margin * a * (1.f / 448.f);
            scale[i] = sc;
            qscale[i] = 1.f / sc;
        }
    }
}

}  // namespace

extern "C" int frcnn_quantize_fp8(const frcnn_bf16* x, int64_t n, const float* qscale, frcnn_fp8* out8, float* amax, int e5m2, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(x && qscale && out8 && n > 0 && n % 8 == 0, "quantize_fp8: bad arguments (n must be a multiple of 8)");
    const int64_t nvec = n / 8;
    const int blocks = (int)((nvec + 255) / 256 < 2048 ? (nvec + 255) / 256 : 2048);
    if (e5m2) hipLaunchKernelGGL(quantize_fp8_kernel<true>, dim3(blocks), dim3(256), 0, S_(stream), reinterpret_cast<const bf16_t*>(x), nvec, qscale, out8, amax);
    else hipLaunchKernelGGL(quantize_fp8_kernel<false>, dim3(blocks), dim3(256), 0, S_(stream), reinterpret_cast<const bf16_t*>(x), nvec, qscale, out8, amax);
    FRCNN_CHECK_LAUNCH("quantize_fp8");
    return FRCNN_OK;
}

extern "C" int frcnn_quantize_weights_fp8_batched(const int64_t* table, int n, int64_t total_rows, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(table && n > 0 && total_rows > 0 && total_rows < (1ll << 31), "quantize_weights_fp8_batched: bad arguments");
    hipLaunchKernelGGL(quantize_weights_fp8_kernel, dim3((unsigned)total_rows), dim3(256), 0, S_(stream), reinterpret_cast<const long long*>(table), n,
                       (long long)total_rows);
    FRCNN_CHECK_LAUNCH("quantize_weights_fp8_batched");
    return FRCNN_OK;
}

extern "C" int frcnn_fp8_update_scales(const float* amax, float* scale, float* qscale, int n, float margin, const float* limit, int32_t* status,
                                       frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(amax && scale && qscale && n > 0 && margin > 0.f, "fp8_update_scales: bad arguments");
    hipLaunchKernelGGL(fp8_update_scales_kernel, dim3(n), dim3(256), 0, S_(stream), amax, scale, qscale, n, margin, limit, status);
    FRCNN_CHECK_LAUNCH("fp8_update_scales");
    return FRCNN_OK;
}
