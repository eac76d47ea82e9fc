// Convolution weight gradient for gfx950: dW[co][tap][ci] += sum_p dz[p][co] * x[im2col(p,tap)][ci]
//
// GEMM view with the PIXEL index as the contraction dimension: both operands are stored pixel-major
// ([pixel][channel]), i.e. K is the strided dimension of both.  Tiles are staged pixel-major in LDS exactly
// as they lie in HBM and the MFMA operand fragments (8 consecutive pixels of one channel per lane) are
// produced by the CDNA4 transposing LDS read ds_read_b64_tr_b16 (two reads per fragment) -- no transpose
// pass.  v2 structure, as the forward kernel: 8 waves per workgroup (2 per SIMD), an S-deep LDS ring filled
// by LDS-DMA (buffer_load_dwordx4 ... lds: out-of-range lanes -- halo, pixel tail, channel tail -- are
// zero-filled by the buffer range check), counted s_waitcnt vmcnt + one raw s_barrier per 64-pixel slice.
// The LDS image is lane-linear, so the 32-byte-chunk XOR swizzle that keeps the transposing reads
// conflict-free is applied to the per-lane SOURCE column and to the read address.
// One workgroup owns one (co tile, tap, ci tile) output tile for a contiguous pixel range (gridDim.z
// ranges); the partial tile is staged through LDS and added to the fp32 gradient with whole-row
// (256-byte contiguous) float atomics.
#include "common.h"

namespace {

struct WgradParams {
    const bf16_t* x;
    const bf16_t* dz;
    const int* row_index;
    float* dw;
    int Hi, Wi, in_pix_stride, Cin, KW, stride, pad_h, pad_w;
    int Ho, Wo, Cout, taps, dz_stride;
    int M, p_tiles, p_tiles_per_split;
    int tiles_co, tiles_ci, linear_x;
    int in_row_stride32;
    unsigned x_bytes, dz_bytes;
};

constexpr unsigned kOob = 0xFFFFFFF0u;

// 32-byte-chunk XOR swizzle of a pixel-major tile of W channels: the 4x16 blocks fetched by one
// ds_read_b64_tr_b16 half-wave (pixel rows {0-3, 8-11} or {4-7, 12-15}) land on distinct banks.
template <int W>
__device__ __forceinline__ int fsw(int r) {
    if (W == 128) return (r & 3) | (((r >> 3) & 1) << 2);
    if (W == 64) return ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    return (r >> 3) & 1;   // W == 32
}
template <int W>
__device__ __forceinline__ int tile_off(int r, int c) {   // byte offset of element (r, c)
    return r * (2 * W) + ((((c >> 4) ^ fsw<W>(r)) << 5) | ((c & 15) << 1));
}

// fragment of 8 consecutive pixels (k = 8*(lane>>4) + 0..7, from row kp0) of channel c0 + (lane&15)
template <int W>
__device__ __forceinline__ bf16x8 load_frag_tr(const unsigned char* tile, int kp0, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int r1 = kp0 + 8 * g + q;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tile_off<W>(r1, c0 + 4 * pp)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + tile_off<W>(r1 + 4, c0 + 4 * pp)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM /*co*/, int BN /*ci*/, int S>
__global__ __launch_bounds__(512) void wgrad_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = 8, T = 512, BKP = 64;
    constexpr int WM = 2, WN = 4;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = (WTM + 15) / 16, NI = (WTN + 15) / 16;
    constexpr bool N_SPLIT = WTN >= 16;                       // BN = 32: only waves with wn < 2 own columns
    constexpr int Z_BYTES = BKP * BM * 2, X_BYTES = BKP * BN * 2, STAGE_BYTES = Z_BYTES + X_BYTES;
    constexpr int Z_RPI = 1024 / (BM * 2), X_RPI = 1024 / (BN * 2);          // pixel rows per 1-KiB DMA instruction
    constexpr int Z_INSTR = BKP / Z_RPI, X_INSTR = BKP / X_RPI;
    constexpr int Z_IT = (Z_INSTR + NW - 1) / NW, X_IT = (X_INSTR + NW - 1) / NW;
    static_assert(Z_INSTR % NW == 0 || Z_INSTR < NW, "tile");
    constexpr int SROW = BN * 4 + 16;                        // staging pitch (bytes)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                // [S][Z tile | X tile]; the epilogue staging re-uses it after the loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    int bid = blockIdx.x;
    const int tile_ci = bid % p.tiles_ci;
    bid /= p.tiles_ci;
    const int tap = bid % p.taps;
    const int tile_co = bid / p.taps;
    const int co0 = tile_co * BM, ci0 = tile_ci * BN;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const int pt_begin = blockIdx.z * p.p_tiles_per_split;
    const int pt_end = min(p.p_tiles, pt_begin + p.p_tiles_per_split);
    const int n_slices = pt_end - pt_begin;

    // descriptors: x shifted back by the halo so that the per-lane pixel offset and the scalar tap offset are >= 0
    const long long halo = (long long)p.pad_h * p.in_row_stride32 + (long long)p.pad_w * p.in_pix_stride;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - halo), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.dz_bytes, 0x00020000);
    const unsigned soff_x = (unsigned)((kh * p.in_row_stride32 + kw * p.in_pix_stride + ci0) * 2);
    const unsigned soff_z = (unsigned)(co0 * 2);

    // per-lane DMA state: this lane's pixel row inside a slice and its swizzled source column
    int z_row[Z_IT], x_row[X_IT];
    unsigned z_col[Z_IT], x_col[X_IT];        // byte offset of the 16-byte source chunk inside the channel row, or kOob
    int x_n[X_IT], x_oy[X_IT], x_ox[X_IT];    // im2col walk of the lane's pixel (advanced by BKP per slice)
#pragma unroll
    for (int i = 0; i < Z_IT; ++i) {
        const int ins = wave + NW * i;
        const int lanes_per_row = (BM * 2) / 16;
        const int r = ins * Z_RPI + lane / lanes_per_row, s16 = lane % lanes_per_row;
        const int col = (((s16 >> 1) ^ fsw<BM>(r)) << 4) + (s16 & 1) * 8;          // element column that LDS slot s16 of row r must hold
        z_row[i] = r;
        z_col[i] = (ins < Z_INSTR && co0 + col < p.Cout) ? (unsigned)(col * 2) : kOob;
    }
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
        const int ins = wave + NW * i;
        const int lanes_per_row = (BN * 2) / 16;
        const int r = ins * X_RPI + lane / lanes_per_row, s16 = lane % lanes_per_row;
        const int col = (((s16 >> 1) ^ fsw<BN>(r)) << 4) + (s16 & 1) * 8;
        x_row[i] = r;
        x_col[i] = (ins < X_INSTR && ci0 + col < p.Cin) ? (unsigned)(col * 2) : kOob;
        const int m = pt_begin * BKP + r;
        x_n[i] = m / hw;
        const int rem = m - x_n[i] * hw;
        x_oy[i] = rem / p.Wo;
        x_ox[i] = rem - x_oy[i] * p.Wo;
    }
    const int Lw = [&]() {
        int l = 0;
        for (int i = 0; i < Z_IT; ++i) l += (wave + NW * i < Z_INSTR) ? 1 : 0;
        for (int i = 0; i < X_IT; ++i) l += (wave + NW * i < X_INSTR) ? 1 : 0;
        return l;
    }();

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    int ld_pix0 = pt_begin * BKP;
    auto issue_slice = [&](const int slot) {
        unsigned char* sz = ring + slot * STAGE_BYTES;
        unsigned char* sx = sz + Z_BYTES;
#pragma unroll
        for (int i = 0; i < Z_IT; ++i) {
            if (wave + NW * i < Z_INSTR) {
                const int m = ld_pix0 + z_row[i];
                const unsigned vo = (m < p.M && z_col[i] != kOob) ? (unsigned)m * (unsigned)(p.dz_stride * 2) + z_col[i] : kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_z, (lds_ptr_t)(sz + (wave + NW * i) * 1024), 16, vo, soff_z, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            if (wave + NW * i < X_INSTR) {
                const int m = ld_pix0 + x_row[i];
                unsigned vo = kOob;
                if (m < p.M && x_col[i] != kOob) {
                    if (p.row_index) {
                        vo = (unsigned)p.row_index[m] * (unsigned)(p.in_pix_stride * 2) + x_col[i];
                    } else if (p.linear_x) {
                        vo = (unsigned)m * (unsigned)(p.in_pix_stride * 2) + x_col[i];
                    } else {
                        const int iy = x_oy[i] * p.stride - p.pad_h + kh, ix = x_ox[i] * p.stride - p.pad_w + kw;
                        if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                            vo = (unsigned)(((x_n[i] * p.Hi + x_oy[i] * p.stride) * p.Wi + x_ox[i] * p.stride) * p.in_pix_stride * 2) + x_col[i];
                    }
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sx + (wave + NW * i) * 1024), 16, vo, soff_x, 0, 0);
                if (!p.linear_x && !p.row_index) {           // advance this lane's pixel by BKP (division-free)
                    x_ox[i] += BKP;
                    while (x_ox[i] >= p.Wo) { x_ox[i] -= p.Wo; ++x_oy[i]; }
                    while (x_oy[i] >= p.Ho) { x_oy[i] -= p.Ho; ++x_n[i]; }
                }
            }
        }
        ld_pix0 += BKP;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int issued = 0, ld_slot = 0, cp_slot = 0;
    for (int s = 0; s < S - 1; ++s) {
        if (issued < n_slices) {
            issue_slice(ld_slot);
            ld_slot = ld_slot + 1 == S ? 0 : ld_slot + 1;
            ++issued;
        }
    }
    const bool owns_cols = N_SPLIT || wn * 16 < BN;
    for (int q = 0; q < n_slices; ++q) {
        const int younger = issued - q - 1;
        if (younger >= 2) { if (Lw == 4) wait_vmcnt_imm<8>(); else if (Lw == 3) wait_vmcnt_imm<6>(); else if (Lw == 2) wait_vmcnt_imm<4>(); else wait_vmcnt_imm<2>(); }
        else if (younger == 1) { if (Lw == 4) wait_vmcnt_imm<4>(); else if (Lw == 3) wait_vmcnt_imm<3>(); else if (Lw == 2) wait_vmcnt_imm<2>(); else wait_vmcnt_imm<1>(); }
        else wait_vmcnt_imm<0>();
        __builtin_amdgcn_s_barrier();
        if (issued < n_slices) {
            issue_slice(ld_slot);
            ld_slot = ld_slot + 1 == S ? 0 : ld_slot + 1;
            ++issued;
        }
        const unsigned char* cZ = ring + cp_slot * STAGE_BYTES;
        const unsigned char* cX = cZ + Z_BYTES;
        cp_slot = cp_slot + 1 == S ? 0 : cp_slot + 1;
        if (owns_cols) {
#pragma unroll
            for (int kk = 0; kk < BKP / 32; ++kk) {
                bf16x8 zf[MI], xf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) zf[i] = load_frag_tr<BM>(cZ, kk * 32, wm * WTM + i * 16, lane);
#pragma unroll
                for (int j = 0; j < NI; ++j) xf[j] = load_frag_tr<BN>(cX, kk * 32, (N_SPLIT ? wn * WTN : wn * 16) + j * 16, lane);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[j], zf[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    // D rows = ci (4 consecutive per lane), cols = co (lane & 15): stage [co][ci] fp32 in the (now idle) ring, then
    // whole-row float atomics (64 lanes x 4 B = 256 contiguous bytes per wave instruction)
    __syncthreads();
    unsigned char* stage = smem;
    if (owns_cols) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int co_l = wm * WTM + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int ci_l = (N_SPLIT ? wn * WTN : wn * 16) + j * 16 + (lane >> 4) * 4;
                *reinterpret_cast<f32x4*>(stage + co_l * SROW + ci_l * 4) = acc[i][j];
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * BN; idx += T) {
        const int r = idx / BN, c = idx - r * BN;
        const int co = co0 + r, ci = ci0 + c;
        if (co < p.Cout && ci < p.Cin) {
            const float v = *reinterpret_cast<const float*>(stage + r * SROW + c * 4);
            atomicAdd(p.dw + ((long long)co * p.taps + tap) * p.Cin + ci, v);
        }
    }
#endif
}

int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int BM, int BN, int S>
int launch(const WgradParams& p, int split, hipStream_t s) {
    constexpr int ring_bytes = S * 64 * (BM + BN) * 2;
    constexpr int stage_bytes = BM * (BN * 4 + 16);
    constexpr int smem = ring_bytes > stage_bytes ? ring_bytes : stage_bytes;
    static_assert(smem <= 163840, "LDS budget");
    if (frcnn_allow_big_lds(reinterpret_cast<const void*>(&wgrad_kernel<BM, BN, S>), smem) != 0) {
        frcnn_set_error("frcnn_conv2d_wgrad: cannot reserve %d B of LDS", smem);
        return FRCNN_EINVAL;
    }
    dim3 grid(p.tiles_co * p.taps * p.tiles_ci, 1, split);
    hipLaunchKernelGGL((wgrad_kernel<BM, BN, S>), grid, dim3(512), smem, s, p);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_wgrad");
    return FRCNN_OK;
}

}  // namespace

extern "C" int frcnn_conv2d_wgrad(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* dz, int dz_stride,
                                  const int32_t* row_index, float* dw, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(d && x && dz && dw, "conv2d_wgrad: null pointer");
    FRCNN_CHECK_ARG(d->cin % 8 == 0 && d->cout % 8 == 0 && dz_stride % 8 == 0, "conv2d_wgrad: channels must be multiples of 8");
    FRCNN_CHECK_ARG(!row_index || (d->kh == 1 && d->kw == 1), "conv2d_wgrad: row_index only for 1x1");
    FRCNN_CHECK_ARG(d->in_pix_stride % 4 == 0 && (d->kw == 1 || d->in_pix_stride % 8 == 0) &&
                        (d->stride * d->in_pix_stride) % 8 == 0 && (d->pad_w * d->in_pix_stride) % 8 == 0 &&
                        ((long long)d->wi * d->in_pix_stride) % 8 == 0,
                    "conv2d_wgrad: pixel addressing breaks 16-byte alignment");
    WgradParams p;
    p.x = reinterpret_cast<const bf16_t*>(x);
    p.dz = reinterpret_cast<const bf16_t*>(dz);
    p.row_index = row_index;
    p.dw = dw;
    p.Hi = d->hi; p.Wi = d->wi; p.in_pix_stride = d->in_pix_stride; p.Cin = d->cin; p.KW = d->kw;
    p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w; p.Ho = d->ho; p.Wo = d->wo; p.Cout = d->cout;
    p.taps = d->kh * d->kw;
    p.dz_stride = dz_stride;
    const long long M = (long long)d->n * d->ho * d->wo;
    FRCNN_CHECK_ARG(M < (1ll << 31), "conv2d_wgrad: M too large");
    p.M = (int)M;
    const long long in_row_stride = (long long)d->wi * d->in_pix_stride;
    p.in_row_stride32 = (int)in_row_stride;
    p.linear_x = (!row_index && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->ho == d->hi &&
                  d->wo == d->wi) ? 1 : 0;
    {
        const long long halo = (long long)d->pad_h * in_row_stride + (long long)d->pad_w * d->in_pix_stride;
        // with row_index the x extent is unknown here: the caller's rows are trusted (checked upstream), use the 4 GiB cap
        const long long x_elems = row_index ? 0x7FFF0000ll : (long long)d->n * d->hi * in_row_stride + (long long)d->kw * d->in_pix_stride + 64;
        const long long xb = (x_elems + halo) * 2, zb = M * dz_stride * 2;
        FRCNN_CHECK_ARG(xb < 0xFFFF0000ll && zb < 0xFFFF0000ll, "conv2d_wgrad: operand larger than 4 GiB (32-bit buffer offsets)");
        p.x_bytes = (unsigned)xb;
        p.dz_bytes = (unsigned)zb;
    }

    const int bm = d->cout >= 128 ? 128 : 64;
    const int bn = d->cin >= 128 ? 128 : (d->cin >= 64 ? 64 : 32);
    constexpr int BKP = 64;
    p.tiles_co = (d->cout + bm - 1) / bm;
    p.tiles_ci = (d->cin + bn - 1) / bn;
    p.p_tiles = (int)((M + BKP - 1) / BKP);
    const int blocks_mn = p.tiles_co * p.taps * p.tiles_ci;
    // every split adds one fp32 tile of float atomics (chip-wide ~1.3 TB/s): aim for ~1.5 workgroups per CU, no more
    int split = (num_cus() * 3 / 2 + blocks_mn - 1) / blocks_mn;
    if (split > p.p_tiles) split = p.p_tiles;
    if (split < 1) split = 1;
    p.p_tiles_per_split = (p.p_tiles + split - 1) / split;
    split = (p.p_tiles + p.p_tiles_per_split - 1) / p.p_tiles_per_split;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define FRCNN_DISPATCH(BM_, BN_, S_) \
    if (bm == BM_ && bn == BN_) return launch<BM_, BN_, S_>(p, split, s);
    FRCNN_DISPATCH(128, 128, 3)
    FRCNN_DISPATCH(128, 64, 4)
    FRCNN_DISPATCH(128, 32, 4)
    FRCNN_DISPATCH(64, 128, 4)
    FRCNN_DISPATCH(64, 64, 4)
    FRCNN_DISPATCH(64, 32, 4)
#undef FRCNN_DISPATCH
    frcnn_set_error("conv2d_wgrad: no tile configuration");
    return FRCNN_EINVAL;
}
