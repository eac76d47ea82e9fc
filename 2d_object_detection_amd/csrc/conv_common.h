// Shared by the implicit-GEMM convolution kernels (conv_igemm.hip: persistent split-K / fp32-output kernel,
// conv_tile.hip: one-tile-per-workgroup bf16 kernel): launch parameters, LDS swizzle, LDS helpers.
#pragma once
#include "common.h"

#define FRCNN_ENOTSUP (-100)      /* internal: no instantiation of the one-tile kernel fits, use the general one */

namespace {

struct ConvParams {
    const bf16_t* x;
    const bf16_t* w;
    const float* bias;
    const bf16_t* res;
    void* y;
    double* stats;                         // [FRCNN_STAT_SLOTS][2][Cout] f64: cross-workgroup sums are order-independent to ~1e-16
    int Hi, Wi, in_pix_stride, Cin, KW, stride, pad_h, pad_w;
    int Ho, Wo, Cout, out_h, out_w, out_scatter, flags;
    int M, Ktot, k_tiles, k_tiles_per_split, split, taps, linear_a;
    int in_row_stride32;                    // in_row_stride (elements); the whole tensor stays below 2 GiB
    unsigned x_bytes, w_bytes, y_bytes;     // buffer descriptor sizes (x incl. the leading halo shift)
    int tap_mask;                           // taps <= 32: per-row tap validity bit masks
    int direct_out;                         // bf16 output row == GEMM row and the tensor stays below 4 GiB: buffer-store epilogue
    int tiles_m, tiles_n, items;            // items = tiles_m * tiles_n * split
    int tiles_per_block;                    // conv_tile.hip: consecutive m-tiles per workgroup
    // fused BatchNorm-backward reduce of the layer that CONSUMES this data gradient (conv_tile.hip, SMODE 2): its raw conv
    // output z, its ReLU bit mask (or NULL), batch mean / invstd, and its [FRCNN_STAT_SLOTS][2][Cout] fp32 partial sums
    const bf16_t* red_z;
    const unsigned char* red_mask;
    const float* red_mean;
    const float* red_invstd;
    float* red_part;
    const unsigned char* res_mask;          // ADD_RES: bit mask applied to the residual (the ReLU mask of the block output whose
                                            // gradient the residual is) or NULL
    long long in_row_stride, in_img_stride;
    float* fix_partial;                     // split-K fix-up form (conv_tile.hip, FIX): [tiles][2][128*64] fp32 partial tiles and
    unsigned* fix_counter;                  // [tiles] arrival counters (zero between launches) in the caller's workspace
    const float* f8_x_scale;                // fp8 operands (conv_tile.hip, F8): dequantisation scale of x (device scalar) and of every
    const float* f8_w_scale;                //   output channel's weight row (device [Cout]); NULL: bf16 operands
    int f8_fmt;                             // 1: x is e4m3 (activations), 2: x is e5m2 (gradients); the weights are always e4m3
    // frcnn_conv2d_fprop_bnin (conv3x3_wres_kernel<.., BNIN>): the BatchNorm of the INPUT layer, applied by this convolution; NULL part: plain input
    const double* bnin_part;
    const float* bnin_gamma;
    const float* bnin_beta;
    float* bnin_mm;
    float* bnin_mv;
    float* bnin_mean;
    float* bnin_invstd;
    bf16_t* bnin_act;
    unsigned char* bnin_mask;
    float bnin_momentum, bnin_eps, bnin_inv_count, bnin_unbias;
    int dry_run;                            // host only: stop before the launch (frcnn_conv2d_describe)
    unsigned long long* dbg;                // FRCNN_STAMPS builds: per-workgroup phase stamps (NULL otherwise)
};

template <int BK>
__device__ __forceinline__ int swz(int chunk, int row) {
    if (BK == 128) return chunk ^ (row & 15);
    if (BK == 64) return chunk ^ (row & 7);
    return chunk ^ ((4 - ((row >> 2) & 3)) & 3);
}

__device__ __forceinline__ long long out_row_of(const ConvParams& p, int m) {
    if (p.out_scatter == 1 && p.out_h == p.Ho && p.out_w == p.Wo) return m;
    const int hw = p.Ho * p.Wo;
    const int n = m / hw;
    const int rem = m - n * hw;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    return ((long long)n * p.out_h + (long long)oy * p.out_scatter) * p.out_w + (long long)ox * p.out_scatter;
}

// Epilogue LDS writes go through inline asm: hipcc orders every DS *write/atomic* it emits behind ALL pending
// LDS-DMA (s_waitcnt vmcnt(0)), although staging tile / statistics array and DMA ring never overlap; that would
// drain the prefetch ring once per tile.  The asm forms are invisible to that pass; their completion is awaited
// explicitly (s_waitcnt lgkmcnt(0)) before the barrier that publishes them.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_write_b64(unsigned addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_add_f32(unsigned addr, float v) {
    asm volatile("ds_add_f32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// ReLU bit mask of 8 stored bf16 activations (bit e: element e > 0), as bn_train_apply_kernel derives it
__device__ __forceinline__ unsigned relu_bits8(const u32x4 pk) {
    unsigned m = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
        m |= ((pk[w2] & 0x7FFFu) != 0u && !(pk[w2] & 0x8000u)) ? (1u << (2 * w2)) : 0u;
        m |= ((pk[w2] & 0x7FFF0000u) != 0u && !(pk[w2] & 0x80000000u)) ? (1u << (2 * w2 + 1)) : 0u;
    }
    return m;
}
// Eight CONSECUTIVE lanes hold the mask bytes of the eight channel vectors of one row (this lane: vector c8): the row's 8 mask bytes as two
// dwords, valid in all eight lanes -- one 8-byte store per row instead of eight byte stores (partial-line byte writes are what the
// BatchNorm kernels avoid the same way).  Every lane of the group must call it.
__device__ __forceinline__ u32x2 pool_mask8(const unsigned m, const int c8) {
    unsigned lo = c8 < 4 ? m << (8 * c8) : 0u, hi = c8 >= 4 ? m << (8 * (c8 - 4)) : 0u;
#pragma unroll
    for (int sh = 1; sh < 8; sh <<= 1) {
        lo |= (unsigned)__shfl_xor((int)lo, sh);
        hi |= (unsigned)__shfl_xor((int)hi, sh);
    }
    return u32x2{lo, hi};
}

constexpr unsigned kOob = 0xFFFFFFF0u;       // voffset beyond every buffer: the hardware range check returns zeros

inline int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace
