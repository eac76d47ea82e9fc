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
#include "conv_common.h"
#include <stdlib.h>

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
    int dev;                                    // FRCNN_SWEEP builds: timing experiments (wrong results), else 0
    int sp_tx, sp_ty;                           // wgrad3_body: spatial tiles (4 x 32 pixels) per image row / column; p_tiles = all of them
    int in_row_stride32;
    unsigned x_bytes, dz_bytes;
    int plain_store;                            // one pixel split: every dw element is written once -> plain stores, no float atomics
    int stem_unpack;                            // FRCNN_CONV_WGRAD_STEM_UNPACK: dw is the un-padded [cout][7][7][3] stem kernel gradient
    const float* f8_x_scale;                    // fp8 operands (x: e4m3, dz: e5m2, one byte per element): device scalars, the
    const float* f8_z_scale;                    // dequantisation scales of the two tensors; NULL for bf16 operands
    // taps folded into the input-channel axis (kw == 1 filters with few channels per tap: the stem's 7 x 32): Cin = fold_taps * fold_cin,
    // taps = 1; column c of the x operand is channel c % fold_cin of tap row c / fold_cin.  A tile then spans several taps and the dz
    // slices it streams are shared by them (per-tap tiles re-read dz once per tap: 7 x 60 MB for the stem).  0: off
    int fold_cin, fold_taps;
};

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

// fp8 operands (one byte per element, 128-pixel slices): ds_read_b64_tr_b8 hands lane i of a 16-lane block byte k = the byte
// (i & 7) that source lane 2 k + (i >> 3) addressed -- source lane s points at row s >> 1, byte column 8 (s & 1) of an
// 8-row x 16-byte block and gets back channel i of the block for its 8 rows (tools/probes/tr8_probe.hip).  One half-wave
// fetches 16 rows x 16 bytes: the 16-byte-chunk XOR below puts them on 16 distinct bank groups.
template <int W>
__device__ __forceinline__ int fsw8(int r) {
    return W == 128 ? (r >> 1) & 7 : (r >> 2) & 3;   // W == 64
}
template <int W>
__device__ __forceinline__ int tile_off8(int r, int c) {   // byte offset of byte column c of pixel row r
    return r * W + ((((c >> 4) ^ fsw8<W>(r)) << 4) | (c & 15));
}

enum { X_LINEAR = 0, X_ROWIDX = 1, X_GENERAL = 2 };   // how the x operand's pixel rows are addressed

// Lanes whose channel chunk lies outside the tensor start here: the per-slice pixel advance keeps them inside
// [2 GiB, 4 GiB), beyond every descriptor (operands are checked to stay below 2 GiB), so the range check zero-fills.
constexpr unsigned kColOob = 0x80000000u;

// Workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous chunk of the logical workgroup list.
__device__ __forceinline__ int xcd_chunk(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// bid: logical workgroup index inside this layer's (pixel split, tile) list, tile fastest
template <int BM /*co*/, int BN /*ci*/, int S, int MODE, int OCC, bool F8>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, int bid) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ES = F8 ? 1 : 2;                             // bytes per operand element
    constexpr int NW = 8, T = 512, BKP = F8 ? 128 : 64, KK = 2;
#ifndef FRCNN_WGRAD_WM
#define FRCNN_WGRAD_WM 2
#endif
    // wave grid: WM x WN = 8 waves.  (A/B builds: FRCNN_DEFINES=FRCNN_WGRAD_WM=4 gives 128 x 64 tiles 32 x 32 wave tiles -- 8 instead of 10
    // transposing reads per 4 MFMAs.)
    constexpr int WM = (BM == 128 && BN == 64) ? FRCNN_WGRAD_WM : 2, WN = 8 / WM;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = (WTM + 15) / 16, NI = (WTN + 15) / 16;
    constexpr bool N_SPLIT = WTN >= 16;                       // BN = 32: only waves with wn < 2 own columns
    constexpr int Z_BYTES = BKP * BM * ES, X_BYTES = BKP * BN * ES;
    constexpr int Z_RPI = 1024 / (BM * ES), X_RPI = 1024 / (BN * ES);        // pixel rows per 1-KiB DMA instruction
    constexpr int Z_INSTR = BKP / Z_RPI, X_INSTR = BKP / X_RPI;
    constexpr int Z_IT = (Z_INSTR + NW - 1) / NW, X_IT = (X_INSTR + NW - 1) / NW;
    constexpr bool Z_UNI = Z_INSTR % NW == 0, X_UNI = X_INSTR % NW == 0;
    constexpr int LC = Z_INSTR / NW + X_INSTR / NW;
    static_assert(S == 2 || (S == 3 && Z_UNI && X_UNI), "counted waits need a uniform DMA split");
    constexpr int SROW = BN * 4 + 16;                        // staging pitch (bytes)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                // [S Z tiles][S X tiles]; the epilogue staging re-uses it after the loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    // (pixel split, tile) list, tile fastest: all tiles of one pixel range run on the same XCD at about the same time, and both
    // operands of that range come out of HBM once instead of once per XCD
    const int tiles_all = p.tiles_co * p.taps * p.tiles_ci;
    const int split_idx = bid / tiles_all;
    bid -= split_idx * tiles_all;
    const int tile_ci = bid % p.tiles_ci;
    bid /= p.tiles_ci;
    const int tap = bid % p.taps;
    const int tile_co = bid / p.taps;
    const int co0 = tile_co * BM, ci0 = tile_ci * BN;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const int pt_begin = split_idx * p.p_tiles_per_split;
    const int pt_end = min(p.p_tiles, pt_begin + p.p_tiles_per_split);
    const int n_slices = pt_end - pt_begin;
    const int pix0 = pt_begin * BKP;

    // descriptors: x shifted back by the halo so that the per-lane pixel offset and the scalar tap offset are >= 0
    const long long halo = (long long)p.pad_h * p.in_row_stride32 + (long long)p.pad_w * p.in_pix_stride;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(p.x) - halo * ES), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.dz_bytes, 0x00020000);
    const unsigned soff_x = p.fold_cin ? 0u : (unsigned)((kh * p.in_row_stride32 + kw * p.in_pix_stride + ci0) * ES);   // (folded taps: in x_col)
    const unsigned soff_z = (unsigned)(co0 * ES);

    // per-lane DMA state.  dz (and x when its rows are linear in the pixel index): a byte offset that advances by a
    // constant per slice; pixel rows beyond M fall outside the descriptor and read zeros.
    unsigned z_vo[Z_IT], x_vo[X_IT], x_col[X_IT];
    int x_row[X_IT], x_n[X_IT], x_oy[X_IT], x_ox[X_IT], x_kh[X_IT];    // im2col walk of the lane's pixel (X_GENERAL); its tap row
    const unsigned z_step = (unsigned)(BKP * p.dz_stride * ES), x_step = (unsigned)(BKP * p.in_pix_stride * ES);
#pragma unroll
    for (int i = 0; i < Z_IT; ++i) {
        const int ins = wave + NW * i;
        const int lanes_per_row = (BM * ES) / 16;
        const int r = ins * Z_RPI + lane / lanes_per_row, s16 = lane % lanes_per_row;
        const int col = F8 ? (s16 ^ fsw8<BM>(r)) << 4                               // element column that LDS slot s16 of row r must hold
                           : (((s16 >> 1) ^ fsw<BM>(r)) << 4) + (s16 & 1) * 8;
        z_vo[i] = (ins < Z_INSTR && co0 + col < p.Cout) ? (unsigned)(pix0 + r) * (unsigned)(p.dz_stride * ES) + (unsigned)(col * ES) : kColOob;
    }
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
        const int ins = wave + NW * i;
        const int lanes_per_row = (BN * ES) / 16;
        const int r = ins * X_RPI + lane / lanes_per_row, s16 = lane % lanes_per_row;
        const int col = F8 ? (s16 ^ fsw8<BN>(r)) << 4 : (((s16 >> 1) ^ fsw<BN>(r)) << 4) + (s16 & 1) * 8;
        const bool col_ok = ins < X_INSTR && ci0 + col < p.Cin;
        x_row[i] = r;
        x_col[i] = col_ok ? (unsigned)(col * ES) : kColOob;
        x_kh[i] = kh;
        if (MODE == X_GENERAL && p.fold_cin) {     // folded taps: this lane's 16-byte chunk lies in tap row (ci0 + col) / fold_cin
            const int gc = ci0 + col, t = gc / p.fold_cin;
            x_kh[i] = t;
            x_col[i] = col_ok ? (unsigned)((t * p.in_row_stride32 + (gc - t * p.fold_cin)) * ES) : kColOob;
        }
        x_vo[i] = col_ok ? (unsigned)(pix0 + r) * (unsigned)(p.in_pix_stride * ES) + (unsigned)(col * ES) : kColOob;
        if (MODE == X_GENERAL) {
            const int m = pix0 + r;
            x_n[i] = m / hw;
            const int rem = m - x_n[i] * hw;
            x_oy[i] = rem / p.Wo;
            x_ox[i] = rem - x_oy[i] * p.Wo;
        }
    }

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    int ld_pix0 = pix0;
    auto issue_slice = [&](const int slot) {
        unsigned char* sz = ring + slot * Z_BYTES;
        unsigned char* sx = ring + S * Z_BYTES + slot * X_BYTES;
#pragma unroll
        for (int i = 0; i < Z_IT; ++i) {
            if (Z_UNI || wave + NW * i < Z_INSTR) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_z, (lds_ptr_t)(sz + (wave + NW * i) * 1024), 16, z_vo[i], soff_z, 0, 0);
                z_vo[i] += z_step;
            }
        }
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            if (X_UNI || wave + NW * i < X_INSTR) {
                unsigned vo;
                if (MODE == X_LINEAR) {
                    vo = x_vo[i];
                    x_vo[i] += x_step;
                } else if (MODE == X_ROWIDX) {
                    const int m = ld_pix0 + x_row[i];
                    vo = (m < p.M && x_col[i] != kColOob) ? (unsigned)p.row_index[m] * (unsigned)(p.in_pix_stride * ES) + x_col[i] : kOob;
                } else {
                    const int iy = x_oy[i] * p.stride - p.pad_h + x_kh[i], ix = x_ox[i] * p.stride - p.pad_w + kw;
                    const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && x_n[i] * hw < p.M && x_col[i] != kColOob;
                    vo = ok ? (unsigned)(((x_n[i] * p.Hi + x_oy[i] * p.stride) * p.Wi + x_ox[i] * p.stride) * p.in_pix_stride * ES) + x_col[i] : kOob;
                    x_ox[i] += BKP;                          // advance this lane's pixel by BKP (division-free)
                    while (x_ox[i] >= p.Wo) { x_ox[i] -= p.Wo; ++x_oy[i]; }
                    while (x_oy[i] >= p.Ho) { x_oy[i] -= p.Ho; ++x_n[i]; }
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sx + (wave + NW * i) * 1024), 16, vo, soff_x, 0, 0);
            }
        }
        ld_pix0 += BKP;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing fragment reads: the swizzle term of a lane is the same for every 32-pixel step and for both 4-row
    // halves, so one byte offset per channel fragment is computed once; slot / step / half are ds_read immediates
    const bool owns_cols = N_SPLIT || wn * 16 < BN;
    unsigned z_foff[MI], x_foff[NI];
    {
        const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
        const int r1 = 8 * g + q;
        const int r8 = 8 * g + (li >> 1), c8 = (li & 1) * 8;   // fp8: the lane's source row / byte column inside its block's 8 x 16 bytes
#pragma unroll
        for (int i = 0; i < MI; ++i)
            z_foff[i] = F8 ? (unsigned)tile_off8<BM>(r8, wm * WTM + i * 16 + c8) : (unsigned)tile_off<BM>(r1, wm * WTM + i * 16 + 4 * pp);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int cb = (N_SPLIT ? wn * WTN : wn * 16) + j * 16;
            x_foff[j] = F8 ? (unsigned)tile_off8<BN>(r8, cb + c8) : (unsigned)tile_off<BN>(r1, cb + 4 * pp);
        }
    }
    // The transposing reads go through inline asm: hipcc orders every ds_read_b64_tr_b16 it emits itself behind ALL pending
    // LDS-DMA (s_waitcnt vmcnt(0) right after the DMA issue -- no prefetch left).  The asm forms are invisible to that pass;
    // their completion is awaited by a counted s_waitcnt lgkmcnt followed by empty asm statements that take the destination
    // registers as in/out operands, so no consumer (not even a register copy) can be scheduled before the data has landed.
    const unsigned lds_base = lds_addr(ring);
    unsigned z_addr[MI], x_addr[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) z_addr[i] = lds_base + z_foff[i];
#pragma unroll
    for (int j = 0; j < NI; ++j) x_addr[j] = lds_base + S * Z_BYTES + x_foff[j];
    constexpr int NR = 2 * (MI + NI);            // reads per 32-pixel step
    static_assert(NR <= 15, "lgkmcnt is a 4-bit counter");
    auto read_step = [&](auto slot_c, auto kk_c, u32x2 (&zl)[MI], u32x2 (&zh)[MI], u32x2 (&xl)[NI], u32x2 (&xh)[NI]) {
        constexpr int slot = decltype(slot_c)::value, kk = decltype(kk_c)::value;
        // bf16: step kk = pixel rows 32 kk .. 32 kk + 31, lo / hi = the lane's rows q and q + 4 of its block's 8.
        // fp8: step kk = pixel rows 64 kk .. 64 kk + 63, lo / hi = rows 8 g + (0..7) and 32 + 8 g + (0..7) of them (the K order
        // inside the 128-pixel MFMA is a permutation shared by both operands).
        constexpr int zo = slot * Z_BYTES + kk * (F8 ? 64 : 32) * (BM * ES), xo = slot * X_BYTES + kk * (F8 ? 64 : 32) * (BN * ES);
        constexpr int zh_o = zo + (F8 ? 32 : 4) * (BM * ES), xh_o = xo + (F8 ? 32 : 4) * (BN * ES);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const unsigned za = z_addr[i];       // (asm operands cannot name a captured variable of a generic lambda directly)
            if (F8) {
                asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(zl[i]) : "v"(za), "n"(zo) : "memory");
                asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(zh[i]) : "v"(za), "n"(zh_o) : "memory");
            } else {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(zl[i]) : "v"(za), "n"(zo) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(zh[i]) : "v"(za), "n"(zh_o) : "memory");
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const unsigned xa = x_addr[j];
            if (F8) {
                asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(xl[j]) : "v"(xa), "n"(xo) : "memory");
                asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(xh[j]) : "v"(xa), "n"(xh_o) : "memory");
            } else {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xl[j]) : "v"(xa), "n"(xo) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xh[j]) : "v"(xa), "n"(xh_o) : "memory");
            }
        }
    };
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    auto frag8 = [](const u32x2 a, const u32x2 b, const u32x2 c, const u32x2 d) -> i32x8 {
        const i32x8 v = {(int)a[0], (int)a[1], (int)b[0], (int)b[1], (int)c[0], (int)c[1], (int)d[0], (int)d[1]};
        return v;
    };
    auto frag = [](const u32x2 lo, const u32x2 hi) -> bf16x8 {
        const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
        return __builtin_bit_cast(bf16x8, v);
    };
    // wait until at most `pending` younger LDS reads are outstanding (LDS returns in order) and pin the step's registers
    auto wait_step = [&](auto pending_c, u32x2 (&zl)[MI], u32x2 (&zh)[MI], u32x2 (&xl)[NI], u32x2 (&xh)[NI]) {
        constexpr int pending = decltype(pending_c)::value;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(pending) : "memory");
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(zl[i]), "+v"(zh[i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(xl[j]), "+v"(xh[j]));
    };
    auto mfma_slice = [&](auto slot_c) {
        if (!owns_cols) return;
        u32x2 zl[2][MI], zh[2][MI], xl[2][NI], xh[2][NI];
        read_step(slot_c, std::integral_constant<int, 0>{}, zl[0], zh[0], xl[0], xh[0]);
        static_assert(KK == 2, "two 32-pixel steps per 64-pixel slice");
        read_step(slot_c, std::integral_constant<int, 1>{}, zl[1], zh[1], xl[1], xh[1]);
        wait_step(std::integral_constant<int, NR>{}, zl[0], zh[0], xl[0], xh[0]);
        if (F8) {
            // one 128-pixel MFMA per accumulator: x is e4m3 (A, cbsz 0), dz is e5m2 (B, blgp 1), unit block scales
            wait_step(std::integral_constant<int, 0>{}, zl[1], zh[1], xl[1], xh[1]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag8(xl[0][j], xh[0][j], xl[1][j], xh[1][j]),
                                                                                 frag8(zl[0][i], zh[0][i], zl[1][i], zh[1][i]), acc[i][j], 0, 1, 0,
                                                                                 0x7F7F7F7F, 0, 0x7F7F7F7F);
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(xl[0][j], xh[0][j]), frag(zl[0][i], zh[0][i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);       // keep the first step's MFMAs ahead of the wait for the second step's reads
        wait_step(std::integral_constant<int, 0>{}, zl[1], zh[1], xl[1], xh[1]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(xl[1][j], xh[1][j]), frag(zl[1][i], zh[1][i]), acc[i][j], 0, 0, 0);
    };

#define FRCNN_WAIT_IMM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    {
        const int pre = n_slices < S - 1 ? n_slices : S - 1;
        for (int s = 0; s < pre; ++s) issue_slice(s);
    }
    int left = n_slices;
    int to_issue = n_slices - (n_slices < S - 1 ? n_slices : S - 1);
    auto step = [&](auto c_c) {                  // one slice with compile-time ring slots
        constexpr int c = decltype(c_c)::value;
        FRCNN_WAIT_IMM((S - 2) * LC);
        __builtin_amdgcn_s_barrier();            // slice landed for everyone; everyone's reads of the slot refilled next are done (lgkmcnt(0) in mfma_slice)
        issue_slice((c + S - 1) % S);
        mfma_slice(c_c);
    };
    while (to_issue >= S) {                      // whole trips around the ring: the ring stays full
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        if (S == 3) step(std::integral_constant<int, S - 1>{});
        to_issue -= S;
        left -= S;
    }
    int slot = 0;
    while (left > 0) {
        if (to_issue > 0) FRCNN_WAIT_IMM((S - 2) * LC);
        else FRCNN_WAIT_IMM(0);
        __builtin_amdgcn_s_barrier();
        if (to_issue > 0) {
            issue_slice(slot == 0 ? S - 1 : slot - 1);
            --to_issue;
        }
        if (slot == 0) mfma_slice(std::integral_constant<int, 0>{});
        else if (slot == 1) mfma_slice(std::integral_constant<int, 1>{});
        else mfma_slice(std::integral_constant<int, S - 1>{});
        slot = slot + 1 == S ? 0 : slot + 1;
        --left;
    }
#undef FRCNN_WAIT_IMM

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
    const float dq = F8 ? *p.f8_x_scale * *p.f8_z_scale : 1.f;       // per-tensor dequantisation of the fp8 product
    for (int idx = tid; idx < BM * BN; idx += T) {
        const int r = idx / BN, c = idx - r * BN;
        const int co = co0 + r, ci = ci0 + c;
        if (co < p.Cout && ci < p.Cin) {
            float v = *reinterpret_cast<const float*>(stage + r * SROW + c * 4);
            if (F8) v *= dq;
            long long off = ((long long)co * p.taps + tap) * p.Cin + ci;      // (folded taps: Cin = taps x channels, the same element)
            if (p.stem_unpack) {                 // ci = kw * 4 + c of the padded 8-pixel x 4-channel tap row: 7 x 3 of them are real
                const int t = p.fold_cin ? ci / p.fold_cin : tap, cc = p.fold_cin ? ci - t * p.fold_cin : ci;
                const int kw = cc >> 2, c4 = cc & 3;
                if (kw >= 7 || c4 == 3) continue;
                off = ((long long)co * (p.fold_cin ? p.fold_taps : p.taps) + t) * 21 + kw * 3 + c4;
            }
            float* dst = p.dw + off;
            if (p.plain_store) *dst = v;
            else atomicAdd(dst, v);
        }
    }
#endif
}

template <int BM, int BN, int S, int MODE, int OCC, bool F8 = false>
__global__ __launch_bounds__(512, 2 * OCC) void wgrad_kernel(const WgradParams p) {
    wgrad_body<BM, BN, S, MODE, OCC, F8>(p, xcd_chunk(blockIdx.x, gridDim.x));
}

// Grouped launch: the weight gradients of several layers (same tile shape and addressing mode) in ONE grid.  A stage of
// small-M layers (conv4: M = 7488) has enough 64 x 64 tiles in total to fill the chip several times over, so no layer needs
// a pixel split: no float atomics (110 MB per step at the memory side's 1.3 TB/s for conv4), one ramp-up and one tail
// instead of one per layer.
constexpr int kGroupMax = 32;
struct WgradGroup {
    int n, total, bm, bn, stages;              // bm x bn: (cout x cin) tile of this group's launch; ring slots
    int first[kGroupMax + 1];                  // first logical workgroup of layer i; first[n] = total
    WgradParams p[kGroupMax];
};

template <int BM, int BN, int S, int MODE, int OCC, bool F8 = false>
__global__ __launch_bounds__(512, 2 * OCC) void wgrad_group_kernel(const WgradGroup* __restrict__ g) {
    const int id = xcd_chunk(blockIdx.x, gridDim.x);
    // layer = number of layers whose first workgroup is <= id (first[] is non-decreasing).  Constant indices: the whole table comes
    // in with a few wide scalar loads issued together -- as a search loop it was up to n - 1 DEPENDENT scalar round trips per workgroup
    int layer = 0;
    const int n = g->n;
#pragma unroll
    for (int i = 1; i < kGroupMax; ++i) {
        const int f = g->first[i];               // (unconditional: the whole array lies inside the table; entries beyond n are masked)
        layer += (int)(i < n) & (int)(id >= f);
    }
    const WgradParams p = g->p[layer];
    wgrad_body<BM, BN, S, MODE, OCC, F8>(p, id - g->first[layer]);
}

#ifdef FRCNN_SWEEP
// ---------------------------------------------------------------------------------------------------- 3x3 / stride 1 / pad 1: all nine taps per workgroup
// wgrad_body gives every (cout tile, TAP, cin tile) its own workgroup: a 3x3 layer streams its dz and x pixels through the CUs' load path
// nine times per tile pair (conv4's 256 -> 256: 207 MB through L2 -> LDS for 7.7 MB of operands), and that path -- ~37 GB/s per CU,
// DESIGN.md 4.3 -- is what bounds it.  Here ONE workgroup owns all nine taps of a 64 (cout) x 64 (cin) block: 9 x 16 accumulator blocks of
// 16 x 16 spread over 8 waves (72 VGPRs per lane), and walks SPATIAL tiles of 4 x 32 output pixels.  Per tile it takes in the 6 x 34 input
// patch of its 64 input channels (26 KB) and the 128 dz pixels of its 64 output channels (16 KB) ONCE -- 42 KB for 9.4 MFLOP instead of
// 9 x 32 KB -- both staged pixel-major exactly as they lie in HBM (fragments by ds_read_b64_tr_b16, as wgrad_body).
//   * a K step is one tile row of 32 pixels, so the x fragment of (patch row R, kw) serves the taps kh = 0, 1, 2 of the tile rows R, R - 1,
//     R - 2: the tile is walked by PATCH row -- 12 x reads feed up to 18 MFMAs per wave; the four dz fragments of a tile are read once;
//   * the patch rows sit at a pitch of 48 pixel slots (34 used): a multiple of 16, so the swizzle term of a fragment address depends on
//     (kw, lane) only and the row is a ds_read immediate;
//   * pixels outside the image are zero-filled by the buffer range check -- in the patch (the padding) and in dz (tiles that overhang the
//     image contribute nothing), so the loop has no masks;
//   * three stages of (patch, dz tile) in LDS: tiles t + 1 and t + 2 are in flight while tile t is multiplied; one barrier per tile;
//   * epilogue: three passes of three taps through LDS, whole 256-byte rows of dw stored (one pixel range) or added with float atomics.
// MEASURED AND NOT USED (round 4; compiled in FRCNN_SWEEP builds only, FRCNN_WGRAD3=1 selects it; parity-green: tests/test_gpu_conv.py passed
// with it as the default).  Isolated, graph replays, us (tools/wgrad3_bench.py), per-tap kernel -> this one: conv4 256 -> 256 30.2 -> 41.1,
// conv3 128 -> 128 35.8 -> 43.4, conv2 64 -> 64 27.8 -> 46.5, RPN 1024 -> 256 71.0 -> 72.4; in the step 4.00 -> 4.055 ms.  Why: (1) a tile
// takes 2.07 us, not the 1.1 us of its 72 MFMAs per wave -- 0.52 us of that is the six LDS-DMA pieces every wave issues per tile (the
// instruction stalls its wave while the CU's load path is backed up; without them 1.55 us), and the rest is the read -> wait -> MFMA chain
// of a row; (2) a layer has only (cout / 64) x (cin / 64) blocks -- 16 for conv4, 4 for conv3, 1 for conv2 -- so filling 256 CUs needs
// pixel splits, and every split costs 9 x 64 x 64 x 4 bytes of float atomics per workgroup at the memory side's 1.3 TB/s: the per-tap
// kernel has nine times the tiles and needs no split in the grouped launches at all.  What would make it win is in DESIGN.md 0.2
// (dedicated loader waves; 32-wide cin blocks).
constexpr int kW3_TH = 4, kW3_TW = 32, kW3_PITCH = 48;
constexpr int kW3_XBUF = 6 * kW3_PITCH * 128, kW3_ZBUF = 16 * 1024, kW3_STAGE = kW3_XBUF + kW3_ZBUF, kW3_STAGES = 3;

__device__ __forceinline__ void wgrad3_body(const WgradParams& p, int bid) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = 8, T = 512, TH = kW3_TH, TW = kW3_TW, PW = TW + 2, PITCH = kW3_PITCH;
    constexpr int SROW = 64 * 4 + 16;            // staging pitch (bytes) of one tap's [64 cout][64 cin] fp32 block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 1, jb0 = (wave & 1) * 2;              // this wave's cout block and its two cin blocks (of 16)

    const int tiles_all = p.tiles_co * p.tiles_ci;
    const int split_idx = bid / tiles_all;
    bid -= split_idx * tiles_all;
    const int tile_ci = bid % p.tiles_ci, tile_co = bid / p.tiles_ci;
    const int co0 = tile_co * 64, ci0 = tile_ci * 64;
    const int t_begin = split_idx * p.p_tiles_per_split;
    const int t_end = min(p.p_tiles, t_begin + p.p_tiles_per_split);
    const int per_img = p.sp_tx * p.sp_ty;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.dz_bytes, 0x00020000);

    // ---- DMA pieces of this lane: 4 of the patch (piece i = wave + 8 k: patch row i / 5, pixel slots 8 (i % 5) .. + 7; 30 real pieces),
    // 2 of the dz tile (piece j = wave + 8 k: tile pixels 8 j .. 8 j + 7).  Lane l of a piece: pixel l / 8, 16-byte slot l % 8.
    int x_pr[4], x_pc[4];
    unsigned x_col[4], z_col[2];
    int z_ty[2], z_tx[2];
    {
        const int pl = lane >> 3, s16 = lane & 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = wave + NW * k;
            const int pr = i / 5, pc = (i - pr * 5) * 8 + pl;
            const bool ok = i < 30 && pc < PW;
            x_pr[k] = ok ? pr : -100000;                         // (never inside an image)
            x_pc[k] = pc;
            x_col[k] = (unsigned)(((((s16 >> 1) ^ fsw<64>(pc)) << 4) + (s16 & 1) * 8) * 2);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = (wave + NW * k) * 8 + pl;
            z_ty[k] = r >> 5;
            z_tx[k] = r & 31;
            z_col[k] = (unsigned)(((((s16 >> 1) ^ fsw<64>(r)) << 4) + (s16 & 1) * 8) * 2);
        }
    }
    auto issue_tile = [&](const int tile, const int stage) {
        const int img = tile / per_img, trem = tile - img * per_img;
        const int tyt = trem / p.sp_tx;
        const int oy0 = tyt * TH, ox0 = (trem - tyt * p.sp_tx) * TW;
        unsigned char* sx = smem + stage * kW3_STAGE;
        unsigned char* sz = sx + kW3_XBUF;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = wave + NW * k;
            const int iy = oy0 - 1 + x_pr[k], ix = ox0 - 1 + x_pc[k];
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const unsigned vo = ok ? (unsigned)((img * p.Hi + iy) * p.Wi + ix) * (unsigned)(p.in_pix_stride * 2) + (unsigned)(ci0 * 2) + x_col[k] : kOob;
            const int pr = i / 5;
            if (i < 30) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sx + (pr * PITCH + (i - pr * 5) * 8) * 128), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int oy = oy0 + z_ty[k], ox = ox0 + z_tx[k];
            const bool ok = oy < p.Ho && ox < p.Wo;
            const unsigned vo = ok ? (unsigned)((img * p.Ho + oy) * p.Wo + ox) * (unsigned)(p.dz_stride * 2) + (unsigned)(co0 * 2) + z_col[k] : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_z, (lds_ptr_t)(sz + (wave + NW * k) * 1024), 16, vo, 0, 0, 0);
        }
    };

    // ---- fragment addresses (stage 0; + stage * kW3_STAGE per tile).  Lane (g, q, pp) of a transposing read points at pixel row
    // base + 8 g + q (+ 4 for the second half) and channels 16 blk + 4 pp of a pixel-major tile (load_frag_tr)
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int rb = 8 * g + q;
    const unsigned lds0 = lds_addr(smem);
    unsigned x_off[3][2][2], z_off[2];           // [kw][half][cin block], [half]
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = kw + rb + 4 * h;
                x_off[kw][h][j] = lds0 + (unsigned)(r * 128 + ((((jb0 + j) ^ fsw<64>(r)) << 5) | (pp << 3)));
            }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = rb + 4 * h;
        z_off[h] = lds0 + (unsigned)(kW3_XBUF + r * 128 + (((cb ^ fsw<64>(r)) << 5) | (pp << 3)));
    }

    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto frag = [](const u32x2 lo, const u32x2 hi) -> bf16x8 {
        const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
        return __builtin_bit_cast(bf16x8, v);
    };

    int stage = 0;
    if (t_begin < t_end) issue_tile(t_begin, 0);
    if (t_begin + 1 < t_end) issue_tile(t_begin + 1, 1);
    for (int tile = t_begin; tile < t_end; ++tile, stage = stage + 1 == kW3_STAGES ? 0 : stage + 1) {
        // this tile's pieces have landed; younger than them are only the next tile's (6 per wave; 5 for the two waves without a 4th patch piece)
        if (tile + 1 < t_end) {
            if (wave < 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // the tile is complete for everyone; everyone's reads of the stage refilled next are done
#ifdef FRCNN_SWEEP
        if (!(p.dev & 1))
#endif
        if (tile + 2 < t_end) issue_tile(tile + 2, stage >= 1 ? stage - 1 : kW3_STAGES - 1);      // (stage + 2) % 3
        const unsigned sb = (unsigned)(stage * kW3_STAGE);
        // the transposing reads go through inline asm (hipcc would order its own behind the DMA just issued: vmcnt(0)); completion by
        // counted lgkmcnt waits + register pins, as wgrad_body
        u32x2 zl[4], zh[4];
#define FRCNN_W3_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
        {
            const unsigned a0 = z_off[0] + sb, a1 = z_off[1] + sb;
            FRCNN_W3_READ(zl[0], a0, 0 * 4096); FRCNN_W3_READ(zh[0], a1, 0 * 4096);
            FRCNN_W3_READ(zl[1], a0, 1 * 4096); FRCNN_W3_READ(zh[1], a1, 1 * 4096);
            FRCNN_W3_READ(zl[2], a0, 2 * 4096); FRCNN_W3_READ(zh[2], a1, 2 * 4096);
            FRCNN_W3_READ(zl[3], a0, 3 * 4096); FRCNN_W3_READ(zh[3], a1, 3 * 4096);
        }
        unsigned xa[3][2][2];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 2; ++j) xa[kw][h][j] = x_off[kw][h][j] + sb;
        u32x2 xl[2][3][2], xh[2][3][2];          // [buffer][kw][cin block]
        // (asm operands cannot name a captured variable of a generic lambda: the destination arrays travel as reference parameters)
        auto read_row = [&](auto r_c, u32x2 (&dl)[3][2], u32x2 (&dh)[3][2], const unsigned (&ad)[3][2][2]) {     // the 12 reads of patch row R
            constexpr int R = decltype(r_c)::value;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned al = ad[kw][0][j], ah = ad[kw][1][j];
                    FRCNN_W3_READ(dl[kw][j], al, R * PITCH * 128);
                    FRCNN_W3_READ(dh[kw][j], ah, R * PITCH * 128);
                }
        };
        auto pin_row = [](u32x2 (&dl)[3][2], u32x2 (&dh)[3][2]) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(dl[kw][j]), "+v"(dh[kw][j]));
        };
        read_row(std::integral_constant<int, 0>{}, xl[0], xh[0], xa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(zl[i]), "+v"(zh[i]));
        pin_row(xl[0], xh[0]);
        auto row_step = [&](auto r_c, u32x2 (&cl)[3][2], u32x2 (&ch)[3][2], u32x2 (&nl)[3][2], u32x2 (&nh)[3][2]) {
            // patch row R (fragments in cl / ch): fetch row R + 1 into nl / nh, multiply row R into the taps kh = R - ty of the tile rows ty
            constexpr int R = decltype(r_c)::value;
            if constexpr (R + 1 < TH + 2) read_row(std::integral_constant<int, R + 1>{}, nl, nh, xa);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ty = R - kh;
                if (ty < 0 || ty >= TH) continue;
                const bf16x8 zf = frag(zl[ty], zh[ty]);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[kh * 3 + kw][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(cl[kw][j], ch[kw][j]), zf, acc[kh * 3 + kw][j], 0, 0, 0);
            }
            if constexpr (R + 1 < TH + 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                pin_row(nl, nh);
            }
        };
        row_step(std::integral_constant<int, 0>{}, xl[0], xh[0], xl[1], xh[1]);
        row_step(std::integral_constant<int, 1>{}, xl[1], xh[1], xl[0], xh[0]);
        row_step(std::integral_constant<int, 2>{}, xl[0], xh[0], xl[1], xh[1]);
        row_step(std::integral_constant<int, 3>{}, xl[1], xh[1], xl[0], xh[0]);
        row_step(std::integral_constant<int, 4>{}, xl[0], xh[0], xl[1], xh[1]);
        row_step(std::integral_constant<int, 5>{}, xl[1], xh[1], xl[0], xh[0]);
#undef FRCNN_W3_READ
    }

    // ---- epilogue: D rows = cin (4 consecutive per lane), cols = cout (lane & 15); three taps per pass through the idle stages
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            unsigned char* stg = smem + tt * (64 * SROW);
            const int co_l = cb * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci_l = (jb0 + j) * 16 + (lane >> 4) * 4;
                *reinterpret_cast<f32x4*>(stg + co_l * SROW + ci_l * 4) = acc[pass * 3 + tt][j];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 3 * 64 * 64; idx += T) {
            const int tt = idx >> 12, r = (idx >> 6) & 63, c = idx & 63;
            const float v = *reinterpret_cast<const float*>(smem + tt * (64 * SROW) + r * SROW + c * 4);
            float* dst = p.dw + ((long long)(co0 + r) * 9 + pass * 3 + tt) * p.Cin + ci0 + c;
            if (p.plain_store) *dst = v;
            else atomicAdd(dst, v);
        }
        __syncthreads();
    }
#endif
}

__global__ __launch_bounds__(512, 2) void wgrad3_kernel(const WgradParams p) { wgrad3_body(p, xcd_chunk(blockIdx.x, gridDim.x)); }

__global__ __launch_bounds__(512, 2) void wgrad3_group_kernel(const WgradGroup* __restrict__ g) {
    const int id = xcd_chunk(blockIdx.x, gridDim.x);
    int layer = 0;
    const int n = g->n;
#pragma unroll
    for (int i = 1; i < kGroupMax; ++i) {
        const int f = g->first[i];
        layer += (int)(i < n) & (int)(id >= f);
    }
    const WgradParams p = g->p[layer];
    wgrad3_body(p, id - g->first[layer]);
}

#endif  // FRCNN_SWEEP (wgrad3_body)

thread_local bool t_dry_run = false;            // frcnn_conv2d_wgrad*_describe: stop before the launch

template <int BM, int BN, int S, int MODE, int OCC, bool F8 = false>
int launch_mode(const WgradParams& p, int split, hipStream_t s) {
    constexpr int ring_bytes = S * 64 * (BM + BN) * 2;        // (fp8: 128-pixel slices of one-byte elements, the same bytes)
    constexpr int stage_bytes = BM * (BN * 4 + 16);
    constexpr int smem = ring_bytes > stage_bytes ? ring_bytes : stage_bytes;
    static_assert(smem * OCC <= 163840, "LDS budget");
    if (!t_dry_run && frcnn_allow_big_lds(reinterpret_cast<const void*>(&wgrad_kernel<BM, BN, S, MODE, OCC, F8>), smem) != 0) {
        frcnn_set_error("frcnn_conv2d_wgrad: cannot reserve %d B of LDS", smem);
        return FRCNN_EINVAL;
    }
    dim3 grid(p.tiles_co * p.taps * p.tiles_ci * split, 1, 1);
    {
        char name[160];
        snprintf(name, sizeof(name), "wgrad<BM=%d,BN=%d,S=%d,MODE=%d,OCC=%d%s> grid=%d split=%d", BM, BN, S, MODE, OCC, F8 ? ",F8" : "", (int)grid.x, split);
        frcnn_note_instantiation(name);
    }
    if (t_dry_run) return FRCNN_OK;
    hipLaunchKernelGGL((wgrad_kernel<BM, BN, S, MODE, OCC, F8>), grid, dim3(512), smem, s, p);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_wgrad");
    return FRCNN_OK;
}

template <int BM, int BN, int S, int OCC>
int launch(const WgradParams& p, int split, hipStream_t s) {
    if constexpr (BN >= 64) {
        if (p.f8_x_scale) {
            if (p.linear_x) return launch_mode<BM, BN, S, X_LINEAR, OCC, true>(p, split, s);
            return launch_mode<BM, BN, S, X_GENERAL, OCC, true>(p, split, s);
        }
    }
    if (p.row_index) return launch_mode<BM, BN, S, X_ROWIDX, OCC>(p, split, s);
    if (p.linear_x) return launch_mode<BM, BN, S, X_LINEAR, OCC>(p, split, s);
    return launch_mode<BM, BN, S, X_GENERAL, OCC>(p, split, s);
}

}  // namespace

// validation + geometry of one weight gradient (tile counts and the pixel split are set by the callers)
// es: bytes per operand element (2: bf16; 1: fp8 with the two dequantisation scales)
static int wgrad_fill(const frcnn_conv_desc* d, const void* x, const void* dz, int dz_stride, const int32_t* row_index,
                      float* dw, WgradParams& p, int es = 2, const float* x_scale = nullptr, const float* dz_scale = nullptr) {
    FRCNN_CHECK_ARG(d && x && dz && dw, "conv2d_wgrad: null pointer");
    FRCNN_CHECK_ARG(d->cin % 8 == 0 && d->cout % 8 == 0 && dz_stride % 8 == 0, "conv2d_wgrad: channels must be multiples of 8");
    FRCNN_CHECK_ARG(!row_index || (d->kh == 1 && d->kw == 1), "conv2d_wgrad: row_index only for 1x1");
    const int a16 = 16 / es;                     // elements per 16 bytes
    FRCNN_CHECK_ARG(d->in_pix_stride % (a16 / 2) == 0 && (d->kw == 1 || d->in_pix_stride % a16 == 0) &&
                        (d->stride * d->in_pix_stride) % a16 == 0 && (d->pad_w * d->in_pix_stride) % a16 == 0 &&
                        ((long long)d->wi * d->in_pix_stride) % a16 == 0,
                    "conv2d_wgrad: pixel addressing breaks 16-byte alignment");
    if (es == 1) {
        FRCNN_CHECK_ARG(x_scale && dz_scale && !row_index, "conv2d_wgrad fp8: needs both dequantisation scales, no row_index");
        FRCNN_CHECK_ARG(d->cin % 64 == 0 && d->cout % 64 == 0 && dz_stride % 16 == 0 && d->in_pix_stride % 16 == 0,
                        "conv2d_wgrad fp8: channels must be multiples of 64, row strides of 16");
    }
    p.f8_x_scale = es == 1 ? x_scale : nullptr;
    p.f8_z_scale = es == 1 ? dz_scale : nullptr;
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
        // linear rows: the descriptor ends exactly at pixel M, so the pixel tail of the last slice reads zeros
        const long long xb = p.linear_x ? M * d->in_pix_stride * es : (x_elems + halo) * es, zb = M * dz_stride * es;
        FRCNN_CHECK_ARG(xb < 0xFFFF0000ll && zb < 0x7FFF0000ll && (!p.linear_x || xb < 0x7FFF0000ll),
                        "conv2d_wgrad: operand too large for 32-bit buffer offsets");
        p.x_bytes = (unsigned)xb;
        p.dz_bytes = (unsigned)zb;
    }
    p.plain_store = 0;
    p.fold_cin = p.fold_taps = 0;
    p.stem_unpack = (d->flags & FRCNN_CONV_WGRAD_STEM_UNPACK) ? 1 : 0;
    FRCNN_CHECK_ARG(!p.stem_unpack || (d->cin == 32 && d->in_pix_stride == 4 && d->kw == 1 && d->kh == 7 && es == 2),
                    "conv2d_wgrad: STEM_UNPACK is for the packed 7 x (8 px x 4 ch) stem descriptor");
    return FRCNN_OK;
}

// ---- the all-taps kernel for 3x3 / stride 1 / pad 1 (wgrad3_body): which layers, and how their pixels are split over workgroups
#ifdef FRCNN_SWEEP
static bool wgrad3_eligible(const frcnn_conv_desc* d, const int es, const int32_t* row_index) {
    bool on = es == 2 && !row_index && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->hi == d->ho && d->wi == d->wo &&
              d->cin % 64 == 0 && d->cout % 64 == 0 && d->in_pix_stride % 8 == 0 && !(d->flags & FRCNN_CONV_WGRAD_STEM_UNPACK);
    const char* e = getenv("FRCNN_WGRAD3");
    return on && e && atoi(e) != 0;
}

// One split count per LAYER (layers of one launch differ in pixels per block: conv3's blocks have 240 spatial tiles, conv4's 72), all derived
// from one target T of tiles per workgroup, chosen by a cost model: rounds of workgroups x (T tiles at ~1.1 us + ~6 us of prologue and
// epilogue) + the float atomics of the split layers at the memory side's ~1.3 TB/s (9 x 64 x 64 x 4 bytes per workgroup).
static void wgrad3_plan(WgradParams* ps, const int n, const bool* accumulate, int* first) {
    int max_sp = 1;
    for (int i = 0; i < n; ++i) {
        WgradParams& p = ps[i];
        p.sp_tx = (p.Wo + kW3_TW - 1) / kW3_TW;
        p.sp_ty = (p.Ho + kW3_TH - 1) / kW3_TH;
        p.p_tiles = (p.M / (p.Ho * p.Wo)) * p.sp_tx * p.sp_ty;
        p.tiles_co = p.Cout / 64;
        p.tiles_ci = p.Cin / 64;
        if (p.p_tiles > max_sp) max_sp = p.p_tiles;
    }
    double tile_us = 2.1, fixed_us = 6.0, atom_bpus = 1.3e6;
    if (const char* e = getenv("FRCNN_WGRAD3_MODEL")) sscanf(e, "%lf,%lf,%lf", &tile_us, &fixed_us, &atom_bpus);
    const int cus = num_cus();
    double best = 1e30;
    int best_t = max_sp;
    for (int t = 1; t <= max_sp; ++t) {
        long long wgs = 0;
        double atom = 0.0;
        int t_eff = 0;
        for (int i = 0; i < n; ++i) {
            const int sp = ps[i].p_tiles, blocks = ps[i].tiles_co * ps[i].tiles_ci;
            const int splits = (sp + t - 1) / t, per = (sp + splits - 1) / splits;
            wgs += (long long)blocks * splits;
            if (splits > 1 || accumulate[i]) atom += (double)blocks * splits * (9.0 * 64 * 64 * 4);
            if (per > t_eff) t_eff = per;
        }
        const double rounds = (double)((wgs + cus - 1) / cus);
        const double cost = rounds * (t_eff * tile_us + fixed_us) + atom / atom_bpus;
        if (cost < best) { best = cost; best_t = t; }
    }
    int total = 0;
    for (int i = 0; i < n; ++i) {
        WgradParams& p = ps[i];
        const int splits = (p.p_tiles + best_t - 1) / best_t;
        p.p_tiles_per_split = (p.p_tiles + splits - 1) / splits;
        const int eff = (p.p_tiles + p.p_tiles_per_split - 1) / p.p_tiles_per_split;
        p.plain_store = eff == 1 && !accumulate[i] ? 1 : 0;
        first[i] = total;
        total += p.tiles_co * p.tiles_ci * eff;
    }
    first[n] = total;
}

constexpr int kW3_SMEM = kW3_STAGES * kW3_STAGE;
static_assert(kW3_SMEM <= 163840 && 3 * 64 * (64 * 4 + 16) <= kW3_SMEM, "LDS budget of wgrad3_body");
#else
static bool wgrad3_eligible(const frcnn_conv_desc*, int, const int32_t*) { return false; }
#endif

static int wgrad_one(const frcnn_conv_desc* d, const void* x, const void* dz, int dz_stride, const int32_t* row_index, float* dw,
                     int es, const float* x_scale, const float* dz_scale, frcnn_stream_t stream) {
    WgradParams p;
    if (const int rc = wgrad_fill(d, x, dz, dz_stride, row_index, dw, p, es, x_scale, dz_scale)) return rc;
    p.dev = 0;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_W3_DEV")) p.dev = atoi(e);
    if (wgrad3_eligible(d, es, row_index)) {
        const bool accumulate = (d->flags & FRCNN_CONV_WGRAD_ACCUMULATE) != 0;
        int first[2];
        wgrad3_plan(&p, 1, &accumulate, first);
        if (!t_dry_run && frcnn_allow_big_lds(reinterpret_cast<const void*>(&wgrad3_kernel), kW3_SMEM) != 0) {
            frcnn_set_error("frcnn_conv2d_wgrad(3x3): cannot reserve %d B of LDS", kW3_SMEM);
            return FRCNN_EINVAL;
        }
        char name[160];
        snprintf(name, sizeof(name), "wgrad3x3_patch grid=%d split=%d", first[1], first[1] / (p.tiles_co * p.tiles_ci));
        frcnn_note_instantiation(name);
        if (t_dry_run) return FRCNN_OK;
        hipLaunchKernelGGL(wgrad3_kernel, dim3(first[1]), dim3(512), kW3_SMEM, reinterpret_cast<hipStream_t>(stream), p);
        FRCNN_CHECK_LAUNCH("frcnn_conv2d_wgrad(3x3)");
        return FRCNN_OK;
    }
#endif
    const long long M = p.M;
    const int BKP = es == 1 ? 128 : 64;          // pixels per slice
    int cin_eff = d->cin;
#ifndef FRCNN_WGRAD_NOFOLD
    if (es == 2 && !row_index && d->kw == 1 && d->kh > 1 && d->cin == 32 && d->pad_w == 0) {
        // the stem (7 tap rows of 32 packed values): 64-wide tiles over the 224 folded columns -- dz is streamed 4 times instead of 7
        p.fold_cin = d->cin;
        p.fold_taps = d->kh;
        p.Cin = cin_eff = d->kh * d->cin;
        p.taps = 1;
        p.KW = 1;
    }
#endif

    // measured on the R50-C4 layer shapes (tools/wgrad_sweep.py): 64 x 64 tiles with a 3-slot ring (three workgroups per
    // CU) win nearly everywhere -- small tiles need few pixel splits to fill the chip, and every split costs one fp32 tile
    // of float atomics
    // Long pixel streams under a multi-tap filter (the feature pyramid's 3x3 layers at stride 4 / 8: M = 58 k .. 234 k) are
    // bound by what the workgroups pull through the CUs' load path -- every (cout tile, tap, cin tile) item streams its whole
    // pixel range: 128 x 128 tiles halve that against 64 x 64 (tools/wgrad_sweep.py fpn: 305 vs 651 us at M = 233,872,
    // 90 vs 156 us at M = 58,656; below M ~ 15 k the small tiles' better fill wins again)
    long long wide_m = 24576;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_WG_WIDE_M")) wide_m = atoll(e);
#endif
    // ... or when the layer alone has >= 128 tiles of 128 x 128 (the RPN's 3x3 1024 -> 256 at M = 7,488: 144; same-box A/B in the step
    // 3.970 -> 3.948 ms): enough workgroups with a pixel split of 3, half the staged bytes per FLOP
    const bool wide = d->kh * d->kw > 1 && d->cin % 128 == 0 && d->cout % 128 == 0 &&
                      (M >= wide_m || (long long)(d->cout / 128) * d->kh * d->kw * (d->cin / 128) >= 128);
    int bm = wide ? 128 : 64;
    int bn = wide ? 128 : cin_eff >= 64 ? 64 : 32;
    // (the folded stem on 128-wide tiles -- dz streamed twice instead of four times -- measured 40.3 us against 34.9 for 64-wide and 46.0 unfolded)
    int stages = wide || bn == 32 ? 2 : 3, want_split = 0;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_WGRAD")) {                // kernel development builds: "bm,bn,stages,split"
        int a = 0, b = 0, c = 0, sp = 0;
        if (sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &sp) == 4) { bm = a; bn = b; stages = c; want_split = sp; }
    }
#endif
    p.tiles_co = (d->cout + bm - 1) / bm;
    p.tiles_ci = (cin_eff + bn - 1) / bn;
    p.p_tiles = (int)((M + BKP - 1) / BKP);
    const int blocks_mn = p.tiles_co * p.taps * p.tiles_ci;
    // ~two workgroups per CU for 1x1 filters, four for multi-tap ones (same-box A/B in the step, ms: one / two per CU: batch 4
    // 4.19 -> 4.16, pyramid fp8 batch 8 9.53 -> 9.47; three: 4.165 / 9.50)
    int one_f = 2;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_WG1_FACTOR")) one_f = atoi(e);
#endif
    int split = want_split > 0 ? want_split : wide ? 2 * num_cus() / blocks_mn : (one_f * num_cus() * (p.taps > 1 ? 2 : 1) + blocks_mn - 1) / blocks_mn;
    if (split > p.p_tiles) split = p.p_tiles;
    if (split < 1) split = 1;
    p.p_tiles_per_split = (p.p_tiles + split - 1) / split;
    split = (p.p_tiles + p.p_tiles_per_split - 1) / p.p_tiles_per_split;
    // (dw arrives zeroed: a single writer per element may store -- unless the caller says dw is shared with other launches)
    p.plain_store = split == 1 && !(d->flags & FRCNN_CONV_WGRAD_ACCUMULATE) ? 1 : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define FRCNN_DISPATCH(BM_, BN_, S_, OCC_) \
    if (bm == BM_ && bn == BN_ && stages == S_) return launch<BM_, BN_, S_, OCC_>(p, split, s);
    FRCNN_DISPATCH(64, 64, 3, 3)           // the two shapes the rule above selects
    FRCNN_DISPATCH(64, 32, 2, 3)
    FRCNN_DISPATCH(128, 128, 2, 2)
#ifdef FRCNN_SWEEP
    FRCNN_DISPATCH(128, 64, 2, 3)
    FRCNN_DISPATCH(128, 32, 2, 3)
    FRCNN_DISPATCH(64, 128, 2, 3)
    FRCNN_DISPATCH(64, 64, 2, 3)
    FRCNN_DISPATCH(128, 128, 3, 1)
    FRCNN_DISPATCH(128, 64, 3, 2)
    FRCNN_DISPATCH(64, 128, 3, 2)
#endif
#undef FRCNN_DISPATCH
    frcnn_set_error("conv2d_wgrad: no tile configuration");
    return FRCNN_EINVAL;
}

extern "C" int frcnn_conv2d_wgrad(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* dz, int dz_stride,
                                  const int32_t* row_index, float* dw, frcnn_stream_t stream) {
    return wgrad_one(d, x, dz, dz_stride, row_index, dw, 2, nullptr, nullptr, stream);
}

extern "C" int frcnn_conv2d_wgrad_fp8(const frcnn_conv_desc* d, const frcnn_fp8* x8, const frcnn_fp8* dz8, int dz_stride,
                                      const float* x_scale, const float* dz_scale, float* dw, frcnn_stream_t stream) {
    return wgrad_one(d, x8, dz8, dz_stride, nullptr, dw, 1, x_scale, dz_scale, stream);
}

// ---------------------------------------------------------------------------------------------------- grouped launches
constexpr int kWideGroupDefault = 0;             // (see frcnn_conv2d_wgrad_group_plan)
constexpr int kGroups = 5;                      // {bf16, fp8} x {x rows linear in the pixel index, general addressing}; 4: bf16 3x3 on wgrad3_body
extern "C" size_t frcnn_wgrad_group_bytes(void) { return kGroups * sizeof(WgradGroup); }

// Fill `table_host` (frcnn_wgrad_group_bytes() bytes: one group of 1x1 / stride-1 layers whose x rows are the GEMM rows, one
// group of everything else) for n weight gradients that are launched together with ONE small pixel split for the whole group
// (none when the group's tiles fill the chip: every dw element is then stored once; otherwise dw must arrive zeroed).  The caller copies the
// table to device memory once -- shapes and pointers of a training plan are static -- and passes both copies to
// frcnn_conv2d_wgrad_grouped.  Layers must have cin, cout multiples of 64 (64 x 64 tiles).
extern "C" int frcnn_conv2d_wgrad_group_plan(const frcnn_wgrad_item* items, int n, void* table_host, size_t table_bytes) {
    FRCNN_CHECK_ARG(items && table_host && n > 0 && table_bytes >= kGroups * sizeof(WgradGroup), "conv2d_wgrad_group_plan: bad arguments");
    WgradGroup* g = reinterpret_cast<WgradGroup*>(table_host);
    // tile shape of the grouped launches.  Launched alone a layer wants small tiles (few pixel splits fill the chip); a group
    // has tiles to spare, so it can afford more MFMA work per barrier.  FRCNN_WGRAD_GROUP="bm,bn" overrides (development aid)
    int gbm = 0, gbn = 64, gst = 3;            // 0: 128 output channels per tile when every layer of the group has >= 128, else 64
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_WGRAD_GROUP")) {
        int a = 0, b = 0, c = 3;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) >= 2 && (a == 0 || a == 64 || a == 128) && (b == 64 || b == 128)) { gbm = a; gbn = b; gst = c; }
    }
#endif
    int tiles[kGroups], min_p_tiles[kGroups];
    bool acc3[kGroupMax];
    bool accumulate[kGroups][kGroupMax];          // FRCNN_CONV_WGRAD_ACCUMULATE of every layer: dw shared with other launches -> never a plain store
    for (int m = 0; m < kGroups; ++m) {
        g[m].n = g[m].total = tiles[m] = 0;
        min_p_tiles[m] = 1 << 30;
    }
    for (int i = 0; i < n; ++i) {
        WgradParams p;
        const bool f8 = items[i].x_scale || items[i].dz_scale;
        if (const int rc = wgrad_fill(items[i].desc, items[i].x, items[i].dz, items[i].dz_stride, nullptr, items[i].dw, p, f8 ? 1 : 2,
                                      items[i].x_scale, items[i].dz_scale)) return rc;
        FRCNN_CHECK_ARG(p.Cin % 64 == 0 && p.Cout % 64 == 0, "conv2d_wgrad_group_plan: layer %d: channels must be multiples of 64", i);
        p.dev = 0;
        p.p_tiles = f8 ? (p.M + 127) / 128 : (p.M + 63) / 64;
        const int m = wgrad3_eligible(items[i].desc, f8 ? 1 : 2, nullptr) ? 4 : (f8 ? 2 : 0) + (p.linear_x ? 0 : 1);
        FRCNN_CHECK_ARG(g[m].n < kGroupMax, "conv2d_wgrad_group_plan: more than %d layers in one group", kGroupMax);
        if (m == 4) acc3[g[m].n] = (items[i].desc->flags & FRCNN_CONV_WGRAD_ACCUMULATE) != 0;
        accumulate[m][g[m].n] = (items[i].desc->flags & FRCNN_CONV_WGRAD_ACCUMULATE) != 0;
        g[m].p[g[m].n++] = p;
        if (p.p_tiles < min_p_tiles[m]) min_p_tiles[m] = p.p_tiles;
    }
#ifdef FRCNN_SWEEP
    if (g[4].n > 0) {
        wgrad3_plan(g[4].p, g[4].n, acc3, g[4].first);
        g[4].total = g[4].first[g[4].n];
        g[4].bm = g[4].bn = 64;
        g[4].stages = 2;
    }
#endif
    for (int m = 0; m < 4; ++m) {
        // measured (train step, batch 4): 128 x 64 tiles 0.73 ms of weight-gradient time per step, 64 x 64 0.82, 64 x 128 0.77,
        // 128 x 128 0.83 -- twice the MFMA work per barrier for 1.5x the staged bytes; not for 64-channel outputs (half a tile idle)
        int min_cout = 1 << 30;
        bool all128 = true;
        for (int i = 0; i < g[m].n; ++i) {
            min_cout = g[m].p[i].Cout < min_cout ? g[m].p[i].Cout : min_cout;
            all128 = all128 && g[m].p[i].Cout % 128 == 0 && g[m].p[i].Cin % 128 == 0;
        }
        g[m].bm = gbm ? gbm : (min_cout >= 128 ? 128 : 64);
        g[m].bn = gbn;
        g[m].stages = gst;
        // 128 x 128 tiles for a group whose every layer has both channel counts in multiples of 128 (conv3, conv4): a third less staged
        // bytes per FLOP than 128 x 64
        int wide_group = kWideGroupDefault;
#ifdef FRCNN_SWEEP
        if (const char* e = getenv("FRCNN_WGRAD_GROUP_WIDE")) wide_group = atoi(e);
#endif
        if (wide_group && all128 && g[m].n > 0 && !gbm) { g[m].bm = 128; g[m].bn = 128; g[m].stages = 2; }
        for (int i = 0; i < g[m].n; ++i) {
            WgradParams& p = g[m].p[i];
            p.tiles_co = (p.Cout + g[m].bm - 1) / g[m].bm;
            p.tiles_ci = (p.Cin + g[m].bn - 1) / g[m].bn;
            tiles[m] += p.tiles_co * p.taps * p.tiles_ci;
        }
    }
    // one pixel split for the whole group: just enough workgroups for ~4 per CU.  The float atomics of a split cost
    // split x |dw| bytes at the memory side's 1.3 TB/s; launched alone, a layer with few tiles needs a far larger split to fill
    // the chip (conv2: 64-128) than the group does (~10)
    for (int m = 0; m < 4; ++m) {
        if (g[m].n == 0) continue;
        // (same-box A/B of the workgroups-per-CU target in the step, ms: batch 4: 2 4.215, 4 4.19, 6 4.19, 8 4.215, 12 4.275; fp8 batch 8
        // 6.79 -> 6.74; pyramid 9.64 -> 9.56: more, shorter pixel ranges even out the tail; the extra float atomics cost less)
        int per_cu = 4;
#ifdef FRCNN_SWEEP
        if (const char* e = getenv("FRCNN_WG_PER_CU")) per_cu = atoi(e);
#endif
        int split = (per_cu * num_cus() + tiles[m] - 1) / tiles[m];
        if (split > min_p_tiles[m]) split = min_p_tiles[m];
        if (split < 1) split = 1;
        for (int i = 0; i < g[m].n; ++i) {
            WgradParams& p = g[m].p[i];
            p.p_tiles_per_split = (p.p_tiles + split - 1) / split;
            const int eff = (p.p_tiles + p.p_tiles_per_split - 1) / p.p_tiles_per_split;
            p.plain_store = eff == 1 && !accumulate[m][i] ? 1 : 0;
            g[m].first[i] = g[m].total;
            g[m].total += p.tiles_co * p.taps * p.tiles_ci * eff;
        }
        g[m].first[g[m].n] = g[m].total;
    }
    return FRCNN_OK;
}

extern "C" int frcnn_conv2d_wgrad_grouped(const void* table_host, const void* table_dev, frcnn_stream_t stream);
extern "C" const char* frcnn_conv2d_wgrad_describe(const frcnn_conv_desc* d, int with_row_index, const void* group_table_host) {
    // dispatch decision only (host logic, no device): of the grouped launch when group_table_host != NULL, else of one layer
    static const int32_t dummy[4] = {0};
    const frcnn_bf16* q = reinterpret_cast<const frcnn_bf16*>(dummy);
    frcnn_note_instantiation("");
    t_dry_run = true;
    const int rc = group_table_host ? frcnn_conv2d_wgrad_grouped(group_table_host, dummy, nullptr)
                                    : frcnn_conv2d_wgrad(d, q, q, d ? d->cout : 0, with_row_index ? dummy : nullptr,
                                                         const_cast<float*>(reinterpret_cast<const float*>(dummy)), nullptr);
    t_dry_run = false;
    return rc == FRCNN_OK ? frcnn_last_conv_instantiation() : nullptr;
}

extern "C" const char* frcnn_conv2d_wgrad_describe_fp8(const frcnn_conv_desc* d) {
    static const int32_t dummy[4] = {0};
    const frcnn_fp8* q = reinterpret_cast<const frcnn_fp8*>(dummy);
    const float* f = reinterpret_cast<const float*>(dummy);
    frcnn_note_instantiation("");
    t_dry_run = true;
    const int rc = frcnn_conv2d_wgrad_fp8(d, q, q, d ? d->cout : 0, f, f, const_cast<float*>(f), nullptr);
    t_dry_run = false;
    return rc == FRCNN_OK ? frcnn_last_conv_instantiation() : nullptr;
}

// A/B builds: FRCNN_DEFINES=FRCNN_WGRAD_LDS_PAD=<bytes> makes every grouped launch REQUEST at least that much LDS per workgroup (84 KB: one
// workgroup per CU), so that the kernels of another stream find room beside them (FRCNN_WGRAD_TRAIL, models/feature_extractor.py)
#ifndef FRCNN_WGRAD_LDS_PAD
#define FRCNN_WGRAD_LDS_PAD 0
#endif
extern "C" int frcnn_conv2d_wgrad_grouped(const void* table_host, const void* table_dev, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(table_host && table_dev, "conv2d_wgrad_grouped: null pointer");
    const WgradGroup* h = reinterpret_cast<const WgradGroup*>(table_host);
    const WgradGroup* dv = reinterpret_cast<const WgradGroup*>(table_dev);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int launched = 0;
    char name[384] = "";
#define FRCNN_GROUP_LAUNCH1(BM_, BN_, S_, OCC_, MODE_, IDX_, F8_)                                                                  \
    if (h[IDX_].n > 0 && h[IDX_].bm == BM_ && h[IDX_].bn == BN_ && h[IDX_].stages == S_) {                                         \
        constexpr int ring_b = S_ * 64 * (BM_ + BN_) * 2, stage_b = BM_ * (BN_ * 4 + 16);                                          \
        constexpr int smem_b = FRCNN_WGRAD_LDS_PAD > (ring_b > stage_b ? ring_b : stage_b) ? FRCNN_WGRAD_LDS_PAD : (ring_b > stage_b ? ring_b : stage_b);  \
        static_assert(smem_b <= 163840, "LDS budget");                                                                             \
        FRCNN_CHECK_ARG(t_dry_run || frcnn_allow_big_lds(reinterpret_cast<const void*>(&wgrad_group_kernel<BM_, BN_, S_, MODE_, OCC_, F8_>), smem_b) == 0, \
                        "conv2d_wgrad_grouped: cannot reserve %d B of LDS", smem_b);                                              \
        if (!t_dry_run) hipLaunchKernelGGL((wgrad_group_kernel<BM_, BN_, S_, MODE_, OCC_, F8_>), dim3(h[IDX_].total), dim3(512), smem_b, s, dv + IDX_);   \
        snprintf(name + strlen(name), sizeof(name) - strlen(name), "wgrad_group<BM=%d,BN=%d,S=%d,MODE=%d,OCC=%d%s> grid=%d; ", BM_, BN_, S_, MODE_, OCC_, \
                 F8_ ? ",F8" : "", h[IDX_].total);                                                                                 \
        if (!t_dry_run) FRCNN_CHECK_LAUNCH("frcnn_conv2d_wgrad_grouped");                                                          \
        launched |= 1 << IDX_;                                                                                                     \
    }
#define FRCNN_GROUP_LAUNCH(BM_, BN_, S_, OCC_)                    \
    FRCNN_GROUP_LAUNCH1(BM_, BN_, S_, OCC_, X_LINEAR, 0, false)   \
    FRCNN_GROUP_LAUNCH1(BM_, BN_, S_, OCC_, X_GENERAL, 1, false)  \
    FRCNN_GROUP_LAUNCH1(BM_, BN_, S_, OCC_, X_LINEAR, 2, true)    \
    FRCNN_GROUP_LAUNCH1(BM_, BN_, S_, OCC_, X_GENERAL, 3, true)
    FRCNN_GROUP_LAUNCH(64, 64, 3, 3)
    FRCNN_GROUP_LAUNCH(128, 64, 3, 2)
#ifdef FRCNN_SWEEP
    FRCNN_GROUP_LAUNCH(64, 128, 3, 2)
    FRCNN_GROUP_LAUNCH(128, 128, 2, 2)
    FRCNN_GROUP_LAUNCH(128, 64, 2, 3)
    FRCNN_GROUP_LAUNCH(64, 64, 2, 3)
#endif
#undef FRCNN_GROUP_LAUNCH
#undef FRCNN_GROUP_LAUNCH1
#ifdef FRCNN_SWEEP
    if (h[4].n > 0) {
        FRCNN_CHECK_ARG(t_dry_run || frcnn_allow_big_lds(reinterpret_cast<const void*>(&wgrad3_group_kernel), kW3_SMEM) == 0,
                        "conv2d_wgrad_grouped: cannot reserve %d B of LDS", kW3_SMEM);
        if (!t_dry_run) hipLaunchKernelGGL(wgrad3_group_kernel, dim3(h[4].total), dim3(512), kW3_SMEM, s, dv + 4);
        snprintf(name + strlen(name), sizeof(name) - strlen(name), "wgrad3x3_patch_group grid=%d layers=%d; ", h[4].total, h[4].n);
        if (!t_dry_run) FRCNN_CHECK_LAUNCH("frcnn_conv2d_wgrad_grouped");
        launched |= 1 << 4;
    }
#endif
    frcnn_note_instantiation(name);
    for (int m = 0; m < kGroups; ++m)
        FRCNN_CHECK_ARG(h[m].n == 0 || (launched & (1 << m)), "conv2d_wgrad_grouped: no kernel for tile %dx%d, %d slots", h[m].bm, h[m].bn, h[m].stages);
    return FRCNN_OK;
}
