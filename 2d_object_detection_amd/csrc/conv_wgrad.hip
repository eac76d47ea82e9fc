// Convolution weight gradient for gfx950: dW[co][tap][ci] += sum_p dz[p][co] * x[im2col(p,tap)][ci]
//
// GEMM view with the PIXEL index as the contraction dimension: both operands are stored
// pixel-major ([pixel][channel]), i.e. K is the strided dimension of both.  Tiles are staged
// pixel-major in LDS exactly as they sit in HBM (coalesced 16-byte loads) and the MFMA operand
// fragments (8 consecutive pixels of one channel per lane) are produced by the CDNA4 transposing
// LDS read ds_read_b64_tr_b16 (two reads per fragment) -- no explicit transpose pass.
// The pixel range is split over gridDim.z; partial results are staged through LDS and added to
// the fp32 gradient with whole-row (256-byte contiguous) float atomics.
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
    int tiles_co, tiles_ci;
    long long in_row_stride, in_img_stride;
};

// 32-byte-chunk XOR swizzle of a pixel-major tile of W channels (makes the 4x16
// blocks fetched by one ds_read_b64_tr_b16 half-wave land on distinct banks).
template <int W>
__device__ __forceinline__ int fsw(int r) {
    if (W == 128) return (r & 3) | (((r >> 3) & 1) << 2);
    if (W == 64) return ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    return (r >> 3) & 1;   // W == 32
}
template <int W>
__device__ __forceinline__ int tile_off(int r, int c) {   // byte offset of element (r, c), c % 4 == 0 for vector access
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

template <int BM /*co*/, int BN /*ci*/, int BKP /*pixels per stage*/>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr int T = 256;
    constexpr int WM = 2, WN = 2;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int Z_BYTES = BKP * BM * 2, X_BYTES = BKP * BN * 2;
    constexpr int ZC = BM / 8, XC = BN / 8;                  // 16-byte chunks per row
    constexpr int Z_IT = (BKP * ZC + T - 1) / T, X_IT = (BKP * XC + T - 1) / T;
    constexpr int SROW = BN * 4 + 16;                        // staging pitch (bytes)
    static_assert(MI >= 1 && NI >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sZ = smem;                  // [2][Z_BYTES]
    unsigned char* sX = smem + 2 * Z_BYTES;    // [2][X_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int bid = blockIdx.x;
    const int tile_ci = bid % p.tiles_ci;
    bid /= p.tiles_ci;
    const int tap = bid % p.taps;
    const int tile_co = bid / p.taps;
    const int co0 = tile_co * BM, ci0 = tile_ci * BN;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const int pt_begin = blockIdx.z * p.p_tiles_per_split;
    const int pt_end = min(p.p_tiles, pt_begin + p.p_tiles_per_split);
    const int hw = p.Ho * p.Wo;

    u32x4 zreg[Z_IT], xreg[X_IT];
    auto load_tile = [&](int pt) {
        const int pix0 = pt * BKP;
#pragma unroll
        for (int i = 0; i < Z_IT; ++i) {
            const int idx = tid + i * T;
            const int r = idx / ZC, c8 = idx - r * ZC;
            const int m = pix0 + r;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < BKP * ZC && m < p.M && co0 + c8 * 8 < p.Cout)
                v = *reinterpret_cast<const u32x4*>(p.dz + (long long)m * p.dz_stride + co0 + c8 * 8);
            zreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            const int idx = tid + i * T;
            const int r = idx / XC, c8 = idx - r * XC;
            const int m = pix0 + r;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < BKP * XC && m < p.M && ci0 + c8 * 8 < p.Cin) {
                if (p.row_index) {
                    const long long row = p.row_index[m];
                    v = *reinterpret_cast<const u32x4*>(p.x + row * p.in_pix_stride + ci0 + c8 * 8);
                } else {
                    const int n = m / hw;
                    const int rem = m - n * hw;
                    const int oy = rem / p.Wo;
                    const int ox = rem - oy * p.Wo;
                    const int iy = oy * p.stride - p.pad_h + kh, ix = ox * p.stride - p.pad_w + kw;
                    if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                        v = *reinterpret_cast<const u32x4*>(p.x + (long long)n * p.in_img_stride + (long long)iy * p.in_row_stride +
                                                            (long long)ix * p.in_pix_stride + ci0 + c8 * 8);
                }
            }
            xreg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < Z_IT; ++i) {
            const int idx = tid + i * T;
            const int r = idx / ZC, c8 = idx - r * ZC;
            if (idx < BKP * ZC) *reinterpret_cast<u32x4*>(sZ + buf * Z_BYTES + tile_off<BM>(r, c8 * 8)) = zreg[i];
        }
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            const int idx = tid + i * T;
            const int r = idx / XC, c8 = idx - r * XC;
            if (idx < BKP * XC) *reinterpret_cast<u32x4*>(sX + buf * X_BYTES + tile_off<BN>(r, c8 * 8)) = xreg[i];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (pt_begin < pt_end) {
        load_tile(pt_begin);
        store_tile(0);
    }
    __syncthreads();

    for (int pt = pt_begin; pt < pt_end; ++pt) {
        const int cur = (pt - pt_begin) & 1;
        const bool more = pt + 1 < pt_end;
        if (more) load_tile(pt + 1);
        const unsigned char* cZ = sZ + cur * Z_BYTES;
        const unsigned char* cX = sX + cur * X_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKP / 32; ++kk) {
            bf16x8 zf[MI], xf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) zf[i] = load_frag_tr<BM>(cZ, kk * 32, wm * WTM + i * 16, lane);
#pragma unroll
            for (int j = 0; j < NI; ++j) xf[j] = load_frag_tr<BN>(cX, kk * 32, wn * WTN + j * 16, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[j], zf[i], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // D rows = ci (4 consecutive per lane), cols = co (lane & 15): stage [co][ci] fp32, then row atomics
    unsigned char* stage = smem;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int co_l = wm * WTM + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int ci_l = wn * WTN + j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<f32x4*>(stage + co_l * SROW + ci_l * 4) = acc[i][j];
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
}

template <int BM, int BN, int BKP>
int launch(const WgradParams& p, int split, hipStream_t s) {
    constexpr int loop_bytes = 2 * BKP * (BM + BN) * 2;
    constexpr int stage_bytes = BM * (BN * 4 + 16);
    constexpr int smem = loop_bytes > stage_bytes ? loop_bytes : stage_bytes;
    dim3 grid(p.tiles_co * p.taps * p.tiles_ci, 1, split);
    if (frcnn_allow_big_lds(reinterpret_cast<const void*>(&wgrad_kernel<BM, BN, BKP>), smem) != 0) { frcnn_set_error("frcnn_conv2d_wgrad: cannot reserve %d B of LDS", smem); return FRCNN_EINVAL; }
    hipLaunchKernelGGL((wgrad_kernel<BM, BN, BKP>), grid, dim3(256), smem, s, p);
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
    p.in_row_stride = (long long)d->wi * d->in_pix_stride;
    p.in_img_stride = (long long)d->hi * p.in_row_stride;

    const int bm = d->cout >= 128 ? 128 : 64;
    const int bn = d->cin >= 128 ? 128 : (d->cin >= 64 ? 64 : 32);
    constexpr int BKP = 64;
    p.tiles_co = (d->cout + bm - 1) / bm;
    p.tiles_ci = (d->cin + bn - 1) / bn;
    p.p_tiles = (int)((M + BKP - 1) / BKP);
    const int blocks_mn = p.tiles_co * p.taps * p.tiles_ci;
    // every split adds one fp32 tile of float atomics (chip-wide ~1.3 TB/s): aim for ~1.5 workgroups per CU, no more
    int split = (384 + blocks_mn - 1) / blocks_mn;
    if (split > p.p_tiles) split = p.p_tiles;
    if (split < 1) split = 1;
    p.p_tiles_per_split = (p.p_tiles + split - 1) / split;
    split = (p.p_tiles + p.p_tiles_per_split - 1) / p.p_tiles_per_split;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define FRCNN_DISPATCH(BM_, BN_) \
    if (bm == BM_ && bn == BN_) return launch<BM_, BN_, BKP>(p, split, s);
    FRCNN_DISPATCH(128, 128)
    FRCNN_DISPATCH(128, 64)
    FRCNN_DISPATCH(128, 32)
    FRCNN_DISPATCH(64, 128)
    FRCNN_DISPATCH(64, 64)
    FRCNN_DISPATCH(64, 32)
#undef FRCNN_DISPATCH
    frcnn_set_error("conv2d_wgrad: no tile configuration");
    return FRCNN_EINVAL;
}
