// v_pk_fma_f32 against v_fma_f32 issue rate with no MFMA beside them (the RoI forward kernel's interpolation): the same number of
// fp32 FMAs per lane as 8 packed or 16 scalar independent chains.  build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_rate tools/probes/pk_rate.hip && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    f32x2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PK) {
                acc[i] = __builtin_elementwise_fma(acc[i], f32x2{a, a}, f32x2{b, b});
            } else {
                float x = acc[i][0], y = acc[i][1];
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));
                acc[i] = f32x2{x, y};
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4 * sizeof(float));
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
            else hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fma = 1024.0 * 256 * 16 * iters;
            printf("%s: %.3f ms  %.1f TFLOP/s fp32 (2 flop per fma)\n", pk ? "v_pk_fma_f32 x 8" : "v_fma_f32 x 16  ", ms, 2 * fma / ms / 1e9);
        }
    }
    return 0;
}
