// Implicit-GEMM convolution for gfx950 (CDNA4): bf16 MFMA 16x16x32, fp32 accumulate.
//
// GEMM view: C[M = N*Ho*Wo pixels][Cout] = A[M][K = KH*KW*Cin] * W[Cout][K]^T, A gathered on the
// fly from the NHWC input (im2col never materialised).  One workgroup = 4 waves computes a
// BM x BN tile; K is walked in BK-wide slices that never straddle a filter tap (Cin % BK == 0),
// so every A-tile row is one contiguous 64/128-byte run of an input pixel (or zeros in the
// padding halo).  Tiles are staged global -> registers -> LDS (16 B per lane, XOR-swizzled so the
// ds_read_b128 fragment reads are bank-conflict free), double buffered with one barrier per
// K-slice: the next slice's global loads are issued before the MFMAs of the current one and
// written to the other LDS buffer after them.
//
// The MFMA is issued with swapped operands (A-op = weights, B-op = pixels) so that each lane
// ends up with 4 consecutive output channels of one pixel: the epilogue packs them to bf16,
// stages the tile in LDS and writes full 16-byte/lane coalesced rows, optionally fusing bias,
// ReLU, a residual add, a stride-2 scatter (data gradient of strided 1x1 convs) and the
// per-tile BatchNorm statistics (column sum / sum of squares of the bf16-rounded outputs).
#include "common.h"

namespace {

struct ConvParams {
    const bf16_t* x;
    const bf16_t* w;
    const float* bias;
    const bf16_t* res;
    void* y;
    float* stats;
    int Hi, Wi, in_pix_stride, Cin, KW, stride, pad_h, pad_w;
    int Ho, Wo, Cout, out_h, out_w, out_scatter, flags;
    int M, Ktot, k_tiles, k_tiles_per_split;
    int tiles_m, tiles_n;
    long long in_row_stride, in_img_stride;
};

template <int BK>
__device__ __forceinline__ int swz(int chunk, int row) {
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

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const ConvParams p) {
    constexpr int T = WM * WN * 64;
    constexpr int CPR = BK / 8;               // 16-byte chunks per tile row
    constexpr int RPP = T / CPR;              // rows loaded per pass
    constexpr int A_IT = BM / RPP;
    constexpr int B_IT = BN / RPP;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int ROWB = BN * 2 + 16;         // epilogue staging row pitch (bytes)
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the thread count");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                 // [2][A_BYTES]
    unsigned char* sB = smem + 2 * A_BYTES;   // [2][B_BYTES]

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (its L2); give each XCD a contiguous
    // run of tiles, N-tiles fastest, so the gathered A rows are re-used out of that L2.
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.tiles_n;
    const int tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int chunk = tid % CPR, row0 = tid / CPR;

    // per-thread im2col row state
    long long a_base[A_IT];
    int a_iy0[A_IT], a_ix0[A_IT];
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + row0 + i * RPP;
        if (m < p.M) {
            const int n = m / hw;
            const int rem = m - n * hw;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad_h;
            a_ix0[i] = ox * p.stride - p.pad_w;
            a_base[i] = (long long)n * p.in_img_stride + (long long)a_iy0[i] * p.in_row_stride +
                        (long long)a_ix0[i] * p.in_pix_stride + chunk * 8;
        } else {
            a_iy0[i] = -0x40000000;           // never valid
            a_ix0[i] = 0;
            a_base[i] = 0;
        }
    }
    long long b_base[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + row0 + i * RPP;
        b_ok[i] = n < p.Cout;
        b_base[i] = (long long)n * p.Ktot + chunk * 8;
    }

    const int kt_begin = blockIdx.z * p.k_tiles_per_split;
    const int kt_end = min(p.k_tiles, kt_begin + p.k_tiles_per_split);

    u32x4 areg[A_IT], breg[B_IT];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int tap = k0 / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int kh = tap / p.KW;
        const int kw = tap - kh * p.KW;
        const long long tap_off = (long long)kh * p.in_row_stride + (long long)kw * p.in_pix_stride + c0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(p.x + a_base[i] + tap_off);
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (b_ok[i]) v = *reinterpret_cast<const u32x4*>(p.w + b_base[i] + k0);
            breg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = row0 + i * RPP;
            *reinterpret_cast<u32x4*>(sA + buf * A_BYTES + r * (BK * 2) + swz<BK>(chunk, r) * 16) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = row0 + i * RPP;
            *reinterpret_cast<u32x4*>(sB + buf * B_BYTES + r * (BK * 2) + swz<BK>(chunk, r) * 16) = breg[i];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = kt + 1 < kt_end;
        if (more) load_tile(kt + 1);
        const unsigned char* cA = sA + cur * A_BYTES;
        const unsigned char* cB = sB + cur * B_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * WTM + i * 16 + frow;
                af[i] = *reinterpret_cast<const bf16x8*>(cA + r * (BK * 2) + swz<BK>(kk * 4 + fchunk, r) * 16);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = wn * WTN + j * 16 + frow;
                bfr[j] = *reinterpret_cast<const bf16x8*>(cB + r * (BK * 2) + swz<BK>(kk * 4 + fchunk, r) * 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // lane holds, for tile (i,j): pixel = wm*WTM + i*16 + (lane&15); couts = wn*WTN + j*16 + (lane>>4)*4 + 0..3
    const int flags = p.flags;
    if (flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC)) {
        float* y = reinterpret_cast<float*>(p.y);
        const bool add_bias = (flags & FRCNN_CONV_BIAS) && blockIdx.z == 0;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * WTM + i * 16 + frow;
            if (m >= p.M) continue;
            const long long orow = out_row_of(p, m);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int c = n0 + wn * WTN + j * 16 + fchunk * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c + e >= p.Cout) continue;
                    float v = acc[i][j][e];
                    if (add_bias) v += p.bias[c + e];
                    if (flags & FRCNN_CONV_RELU) v = fmaxf(v, 0.f);
                    if (flags & FRCNN_CONV_SPLITK_ATOMIC)
                        atomicAdd(y + orow * p.Cout + c + e, v);
                    else
                        y[orow * p.Cout + c + e] = v;
                }
            }
        }
        return;
    }

    unsigned char* stage = smem;              // [BM][ROWB]; all waves are past the last barrier
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wm * WTM + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int cl = wn * WTN + j * 16 + fchunk * 4;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[i][j][e];
                if (flags & FRCNN_CONV_BIAS) v[e] += (n0 + cl + e < p.Cout) ? p.bias[n0 + cl + e] : 0.f;
                if (flags & FRCNN_CONV_RELU) v[e] = fmaxf(v[e], 0.f);
            }
            u32x2 pk;
            pk[0] = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
            pk[1] = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
            *reinterpret_cast<u32x2*>(stage + r * ROWB + cl * 2) = pk;
        }
    }
    __syncthreads();

    bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
    constexpr int C8 = BN / 8;
    for (int idx = tid; idx < BM * C8; idx += T) {
        const int r = idx / C8, c8 = idx - r * C8;
        const int m = m0 + r, c = n0 + c8 * 8;
        if (m >= p.M || c >= p.Cout) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * ROWB + c8 * 16);
        const long long off = out_row_of(p, m) * p.Cout + c;
        if (flags & FRCNN_CONV_ADD_RES) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(p.res + off);
            float a[8], b[8];
            unpack8(v, a);
            unpack8(rv, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
            v = pack8(a);
        }
        *reinterpret_cast<u32x4*>(y + off) = v;
    }

    if (flags & FRCNN_CONV_STATS) {
        constexpr int PARTS = T / BN;
        constexpr int RPART = BM / PARTS;
        const int col = tid % BN, part = tid / BN;
        if (n0 + col < p.Cout) {
            float s = 0.f, ss = 0.f;
            const int rbeg = part * RPART;
            const int rend = min(rbeg + RPART, p.M - m0);
            for (int r = rbeg; r < rend; ++r) {
                const float v = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(stage + r * ROWB + col * 2));
                s += v;
                ss += v * v;
            }
            // 64 accumulation slots (pre-zeroed by the caller) keep the finalize pass short; a slot sees
            // tiles/64 float atomics per address, each wave instruction adding 256 contiguous bytes
            float* dst = p.stats + ((long long)((tile_m * PARTS + part) & (FRCNN_STAT_SLOTS - 1)) * 2) * p.Cout + n0 + col;
            atomicAdd(dst, s);
            atomicAdd(dst + p.Cout, ss);
        }
    }
}

struct TileCfg { int bm, bn, bk; };

TileCfg pick_tile(const frcnn_conv_desc* d) {
    TileCfg t;
    t.bk = (d->cin % 64 == 0) ? 64 : 32;
    t.bn = d->cout >= 128 ? 128 : 64;
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long blocks128 = ((M + 127) / 128) * ((d->cout + t.bn - 1) / t.bn) * (d->split_k > 1 ? d->split_k : 1);
    t.bm = blocks128 >= 512 ? 128 : 64;
    return t;
}

template <int BM, int BN, int BK, int WM, int WN>
int launch(const ConvParams& p, int split, hipStream_t s) {
    constexpr int loop_bytes = 2 * (BM + BN) * BK * 2;
    constexpr int stage_bytes = BM * (BN * 2 + 16);
    constexpr int smem = loop_bytes > stage_bytes ? loop_bytes : stage_bytes;
    dim3 grid(p.tiles_m * p.tiles_n, 1, split);
    if (frcnn_allow_big_lds(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, BK, WM, WN>), smem) != 0) { frcnn_set_error("frcnn_conv2d_fprop: cannot reserve %d B of LDS", smem); return FRCNN_EINVAL; }
    hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, WM, WN>), grid, dim3(WM * WN * 64), smem, s, p);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_fprop");
    return FRCNN_OK;
}

int stats_parts(const TileCfg& t) { return 256 / t.bn; }

}  // namespace

extern "C" int frcnn_conv2d_stat_tiles(const frcnn_conv_desc* d) {
    if (!d) return FRCNN_EINVAL;
    return FRCNN_STAT_SLOTS;
}

extern "C" int frcnn_conv2d_fprop(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* w, const float* bias,
                                  const frcnn_bf16* res, void* y, float* stats_partial, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(d && x && w && y, "conv2d_fprop: null pointer");
    FRCNN_CHECK_ARG(d->cin > 0 && d->cin % 32 == 0, "conv2d_fprop: cin=%d must be a multiple of 32", d->cin);
    FRCNN_CHECK_ARG(d->cout > 0 && d->cout % 8 == 0, "conv2d_fprop: cout=%d must be a multiple of 8", d->cout);
    FRCNN_CHECK_ARG(d->in_pix_stride % 4 == 0 && (d->kw == 1 || d->in_pix_stride % 8 == 0),
                    "conv2d_fprop: in_pix_stride=%d breaks 16-byte alignment", d->in_pix_stride);
    FRCNN_CHECK_ARG(d->stride >= 1 && d->kh >= 1 && d->kw >= 1 && d->n >= 1 && d->ho >= 1 && d->wo >= 1,
                    "conv2d_fprop: bad geometry");
    FRCNN_CHECK_ARG(((long long)d->wi * d->in_pix_stride) % 8 == 0, "conv2d_fprop: input row pitch not 16-byte aligned");
    FRCNN_CHECK_ARG((d->stride * d->in_pix_stride) % 8 == 0 && (d->pad_w * d->in_pix_stride) % 8 == 0,
                    "conv2d_fprop: pixel addressing breaks 16-byte alignment");
    const int flags = d->flags;
    FRCNN_CHECK_ARG(!(flags & FRCNN_CONV_BIAS) || bias, "conv2d_fprop: BIAS without bias pointer");
    FRCNN_CHECK_ARG(!(flags & FRCNN_CONV_ADD_RES) || res, "conv2d_fprop: ADD_RES without res pointer");
    FRCNN_CHECK_ARG(!(flags & FRCNN_CONV_STATS) || stats_partial, "conv2d_fprop: STATS without buffer");
    FRCNN_CHECK_ARG(!((flags & FRCNN_CONV_STATS) && (flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC | FRCNN_CONV_ADD_RES))),
                    "conv2d_fprop: STATS only with plain bf16 output");
    FRCNN_CHECK_ARG(!((flags & FRCNN_CONV_ADD_RES) && (flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC))),
                    "conv2d_fprop: ADD_RES only with bf16 output");
    const int split = d->split_k > 1 ? d->split_k : 1;
    FRCNN_CHECK_ARG(split == 1 || (flags & FRCNN_CONV_SPLITK_ATOMIC), "conv2d_fprop: split_k needs SPLITK_ATOMIC");
    FRCNN_CHECK_ARG(!(flags & FRCNN_CONV_SPLITK_ATOMIC) || !(flags & FRCNN_CONV_RELU), "conv2d_fprop: no ReLU with split-K");
    FRCNN_CHECK_ARG(d->out_scatter >= 1 && (d->ho - 1) * d->out_scatter < d->out_h && (d->wo - 1) * d->out_scatter < d->out_w,
                    "conv2d_fprop: scatter target out of range");

    const TileCfg t = pick_tile(d);
    ConvParams p;
    p.x = reinterpret_cast<const bf16_t*>(x);
    p.w = reinterpret_cast<const bf16_t*>(w);
    p.bias = bias;
    p.res = reinterpret_cast<const bf16_t*>(res);
    p.y = y;
    p.stats = stats_partial;
    p.Hi = d->hi; p.Wi = d->wi; p.in_pix_stride = d->in_pix_stride; p.Cin = d->cin; p.KW = d->kw;
    p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.Ho = d->ho; p.Wo = d->wo; p.Cout = d->cout; p.out_h = d->out_h; p.out_w = d->out_w; p.out_scatter = d->out_scatter;
    p.flags = flags;
    const long long M = (long long)d->n * d->ho * d->wo;
    FRCNN_CHECK_ARG(M < (1ll << 31), "conv2d_fprop: M too large");
    p.M = (int)M;
    p.Ktot = d->kh * d->kw * d->cin;
    FRCNN_CHECK_ARG(p.Ktot % t.bk == 0, "conv2d_fprop: K=%d not a multiple of %d", p.Ktot, t.bk);
    p.k_tiles = p.Ktot / t.bk;
    p.k_tiles_per_split = (p.k_tiles + split - 1) / split;
    p.tiles_m = (int)((M + t.bm - 1) / t.bm);
    p.tiles_n = (d->cout + t.bn - 1) / t.bn;
    p.in_row_stride = (long long)d->wi * d->in_pix_stride;
    p.in_img_stride = (long long)d->hi * p.in_row_stride;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);

#define FRCNN_DISPATCH(BM_, BN_, BK_, WM_, WN_) \
    if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_) return launch<BM_, BN_, BK_, WM_, WN_>(p, split, s);
    FRCNN_DISPATCH(128, 128, 64, 2, 2)
    FRCNN_DISPATCH(128, 64, 64, 2, 2)
    FRCNN_DISPATCH(64, 128, 64, 2, 2)
    FRCNN_DISPATCH(64, 64, 64, 2, 2)
    FRCNN_DISPATCH(128, 128, 32, 2, 2)
    FRCNN_DISPATCH(128, 64, 32, 2, 2)
    FRCNN_DISPATCH(64, 128, 32, 2, 2)
    FRCNN_DISPATCH(64, 64, 32, 2, 2)
#undef FRCNN_DISPATCH
    frcnn_set_error("conv2d_fprop: no tile configuration for bm=%d bn=%d bk=%d", t.bm, t.bn, t.bk);
    return FRCNN_EINVAL;
}
