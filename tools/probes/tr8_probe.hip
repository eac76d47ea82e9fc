// Probe of ds_read_b64_tr_b8 (gfx950): which (source lane, source byte) lands in (lane, result byte).
// Every lane reads 8 bytes at LDS offset 8*lane; pass 0 encodes (lane % 16) * 8 + byte, pass 1 encodes lane / 16.
// build: hipcc --offload-arch=gfx950 tools/probes/tr8_probe.hip -o gpurun_out/tr8_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned char* out, int pass) {
    __shared__ __attribute__((aligned(16))) unsigned char s[512];
    const int lane = threadIdx.x;
    for (int j = 0; j < 8; ++j) s[lane * 8 + j] = pass == 0 ? (unsigned char)((lane % 16) * 8 + j) : (unsigned char)(lane / 16);
    __syncthreads();
    typedef __attribute__((address_space(3))) v2i lv;
    const v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lv*)(s + lane * 8));
    for (int j = 0; j < 8; ++j) out[lane * 8 + j] = (unsigned char)((j < 4 ? (unsigned)r[0] >> (8 * j) : (unsigned)r[1] >> (8 * (j - 4))) & 0xFF);
}
int main() {
    unsigned char* d;
    unsigned char h[2][512];
    hipMalloc(&d, 512);
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pass);
        hipMemcpy(h[pass], d, 512, hipMemcpyDeviceToHost);
    }
    for (int lane = 0; lane < 64; ++lane) {
        printf("lane %2d:", lane);
        for (int j = 0; j < 8; ++j) printf("  b%d<-(blk %d lane %2d byte %d)", j, h[1][lane * 8 + j], h[0][lane * 8 + j] / 8, h[0][lane * 8 + j] % 8);
        printf("\n");
    }
    return 0;
}
