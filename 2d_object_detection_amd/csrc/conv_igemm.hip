// Implicit-GEMM convolution for gfx950 (CDNA4): bf16 MFMA 16x16x32, fp32 accumulate.
//
// GEMM view: C[M = N*Ho*Wo pixels][Cout] = A[M][K = KH*KW*Cin] * W[Cout][K]^T, A gathered on the fly
// from the NHWC input (im2col never materialised).  K is walked in BK-wide slices that never
// straddle a filter tap (Cin % BK == 0), so every A-tile row is one contiguous 64/128-byte run of an
// input pixel (or zeros in the padding halo / M tail).
//
// Structure (v2): PERSISTENT workgroups (4 waves) walk a list of work items (tile x K-split) and
// stream (item, K-slice) pairs through an S-deep LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPR staging).  The loader runs S-1
// slices ahead of the MFMA consumer ACROSS item boundaries, so the next tile's operands are already
// in flight while the current tile's epilogue runs -- the small-K (1x1, K = 64..256) layers of
// conv2/conv3, which are HBM-bound, get the same latency hiding as the long-K 3x3 layers.
// Synchronisation per slice: counted s_waitcnt vmcnt(N) (never 0 in steady state) + one raw
// s_barrier; a ring slot is refilled only after the barrier that follows its last read.
// The LDS image is lane-linear (DMA constraint), the XOR bank swizzle is applied to the per-lane
// SOURCE address and to the ds_read_b128 fragment reads.  Rows outside the input read a 16-byte zero
// page.
//
// The MFMA is issued with swapped operands (A-op = weights, B-op = pixels) so that each lane ends
// with 4 consecutive output channels of one pixel: the epilogue adds bias / ReLU, packs to bf16,
// reduces the BatchNorm statistics (sum, sum of squares of the ROUNDED outputs) in registers with
// 16-lane butterflies, stages the tile in a dedicated LDS region and writes 16-byte/lane coalesced
// rows, optionally adding a residual and scattering with stride 2 (data gradient of strided 1x1).
#include "conv_common.h"
#include <stdlib.h>

namespace {

// gfx950 retires loads, LDS-DMA, stores and atomics through ONE in-order vmcnt: "slice q landed" is
// "at most n younger vector-memory ops are still outstanding", where n must count the DMA of the
// younger slices AND the epilogue stores issued after slice q's DMA (waiting for those acknowledgements
// would serialise every small-K tile on a store round trip).  n is data dependent, the immediate is not:
// dispatch over the possible values (an under-estimate only waits longer, never too little).
#define FRCNN_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt_at_most(int n) {
    switch (n < 8 ? n : 8) {
        FRCNN_VMCNT_CASE(0) FRCNN_VMCNT_CASE(1) FRCNN_VMCNT_CASE(2) FRCNN_VMCNT_CASE(3) FRCNN_VMCNT_CASE(4) FRCNN_VMCNT_CASE(5)
        FRCNN_VMCNT_CASE(6) FRCNN_VMCNT_CASE(7)
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
#undef FRCNN_VMCNT_CASE

// Block -> work-item schedule.  Blocks b, b+8, ... share an XCD (its L2): every XCD owns a contiguous chunk
// of the item list (item = tile * split + k-split, tile = tile_m * tiles_n + tile_n) and its blocks interleave
// inside it, so concurrently running blocks of one XCD touch neighbouring tiles (shared A rows / weight panels
// come out of that L2).  A block walks its items with a constant stride; (tile_m, tile_n, ks) advance by
// carry-propagating adds -- no division per item.
struct ItemWalk {
    int left;                 // items still to take
    int tm, tn, ks;           // current item
    int dm, dn, dk;           // per-step increments (stride decomposed)
    __device__ __forceinline__ void init(const int items, const int tiles_n, const int split, const int bid, const int nblocks) {
        const int xcd = bid & 7, local = bid >> 3;
        const int stride = (nblocks >> 3) + ((nblocks & 7) > xcd ? 1 : 0);
        const int q = items >> 3, r = items & 7;
        const int begin = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int end = begin + q + (xcd < r ? 1 : 0);
        const int first = begin + local;
        left = first < end ? (end - first + stride - 1) / stride : 0;
        int t = first / split;
        ks = first - t * split;
        tm = t / tiles_n;
        tn = t - tm * tiles_n;
        int st = stride / split;
        dk = stride - st * split;
        dm = st / tiles_n;
        dn = st - dm * tiles_n;
    }
    __device__ __forceinline__ void advance(const int tiles_n, const int split) {
        --left;
        ks += dk;
        int carry = 0;
        if (ks >= split) { ks -= split; carry = 1; }
        tn += dn + carry;
        carry = 0;
        if (tn >= tiles_n) { tn -= tiles_n; carry = 1; }
        if (tn >= tiles_n) { tn -= tiles_n; ++carry; }
        tm += dm + carry;
    }
};


template <int BM, int BN, int BK, int S, int NW>
__global__ __launch_bounds__(NW * 64) void igemm_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses device-only types/builtins (buffer resources, LDS-DMA): the host pass only needs the stub
    constexpr int T = NW * 64, WM = 2, WN = NW / 2;   // 8 waves: 2 per SIMD, partner waves overlap DMA issue / LDS reads / epilogue
    constexpr int CPR = BK / 8;                  // 16-byte chunks per tile row
    constexpr int RPI = 64 / CPR;                // rows written by one DMA wave instruction (1 KiB)
    constexpr int A_INSTR = BM / RPI, B_INSTR = BN / RPI;            // DMA instructions per slice
    constexpr int A_IT = (A_INSTR + NW - 1) / NW, B_IT = (B_INSTR + NW - 1) / NW;
    constexpr bool UNIFORM_L = (A_INSTR % NW == 0) && (B_INSTR % NW == 0);
    constexpr int LC = A_INSTR / NW + B_INSTR / NW;                    // DMA instructions per wave and slice when uniform
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int ROWB = BN * 2 + 16;            // epilogue staging row pitch (bytes)
    constexpr int C8 = BN / 8;                   // 16-byte chunks per output tile row
    constexpr int ST_IT = (BM * C8) / T;         // store passes (one 16-byte store per lane each) per tile
    static_assert((BM * C8) % T == 0, "store loop covers the tile in whole passes");
    static_assert(MI >= 1 && NI >= 1 && S >= 2 && S <= 4, "unsupported tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                              // [S A tiles][S B tiles]: slot strides stay inside the ds_read offset field
    unsigned char* stage = smem + S * STAGE_BYTES;           // [BM][ROWB] epilogue staging
    float* bias_s = reinterpret_cast<float*>(stage + BM * ROWB);                      // [Cout]    (BIAS, bf16 path)
    float* stat_s = bias_s + ((p.flags & FRCNN_CONV_BIAS) ? p.tiles_n * BN : 0);     // [2][Cout] (STATS); bias is padded to the tile grid

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int flags = p.flags;
    const bool bf16_path = !(flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC));
    const unsigned stage_a = lds_addr(stage), stat_a = lds_addr(stat_s);
    if (bf16_path) {
        // bias and the per-block BN partial sums live in LDS: no ordinary global load in the loop (the compiler would
        // drain the DMA ring with vmcnt(0) for it) and ONE global atomic flush per block instead of one per tile
        if (flags & FRCNN_CONV_BIAS)
            for (int c = tid; c < p.tiles_n * BN; c += T) bias_s[c] = c < p.Cout ? p.bias[c] : 0.f;
        if (flags & FRCNN_CONV_STATS)
            for (int c = tid; c < 2 * p.Cout; c += T) stat_s[c] = 0.f;
        __syncthreads();
    }

    // ------------------------------------------------------------------ loader (LDS-DMA producer) state
    // buffer_load_dwordx4 ... lds: address = base + soffset (scalar: filter tap / K offset, advanced per slice) +
    // voffset (per lane: pixel row + swizzled chunk, fixed per item).  Lanes of rows outside the input / tile use an
    // out-of-range voffset: the buffer range check writes zeros, so halo, M tail and N tail need no branches.
    // The x descriptor starts pad rows/pixels BEFORE the tensor so that voffset (pixel oy*s, ox*s) and soffset
    // (tap kh, kw) are both non-negative; taps that fall outside the image are masked per lane.
    const long long halo = (long long)p.pad_h * p.in_row_stride + (long long)p.pad_w * p.in_pix_stride;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - halo), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.y_bytes, 0x00020000);
    ItemWalk ld_items;
    ld_items.init(p.items, p.tiles_n, p.split, blockIdx.x, gridDim.x);
    ItemWalk cp_items = ld_items;
    int ld_left = 0;                             // K-slices left in the loader's current item
    int ld_c0 = 0, ld_kh = 0, ld_kw = 0;         // filter tap / channel offset of the next slice
    unsigned ld_soff_a = 0, ld_soff_b = 0;       // scalar byte offsets of the next slice
    unsigned a_voff[A_IT], b_voff[B_IT];
    int a_iy0[A_IT] = {}, a_ix0[A_IT] = {};
    unsigned a_mask[A_IT] = {};                  // bit t: filter tap t of this row lies inside the image (taps <= 32)
    int ld_tap = 0;                              // kh * KW + kw of the next slice
    int ld_seq = 0, cp_seq = 0;                  // item sequence numbers of loader / consumer
    const int lrow = lane / CPR, lslot = lane % CPR;
    const int hw = p.Ho * p.Wo;
    const int Lw = [&]() {                       // DMA instructions THIS wave issues per slice (vmcnt units)
        int l = 0;
        for (int i = 0; i < A_IT; ++i) l += (wave + NW * i < A_INSTR) ? 1 : 0;
        for (int i = 0; i < B_IT; ++i) l += (wave + NW * i < B_INSTR) ? 1 : 0;
        return l;
    }();

    auto loader_next_item = [&]() {
        const int m0 = ld_items.tm * BM, n0 = ld_items.tn * BN;
        const int kt = ld_items.ks * p.k_tiles_per_split;
        ld_left = min(p.k_tiles, kt + p.k_tiles_per_split) - kt;
        ++ld_seq;
        if (p.taps == 1) {
            ld_kh = ld_kw = ld_tap = 0;
            ld_c0 = kt * BK;
        } else {
            const int k0 = kt * BK;
            const int tap = k0 / p.Cin;
            ld_c0 = k0 - tap * p.Cin;
            ld_kh = tap / p.KW;
            ld_kw = tap - ld_kh * p.KW;
            ld_tap = tap;
        }
        ld_soff_a = (unsigned)((ld_kh * p.in_row_stride32 + ld_kw * p.in_pix_stride + ld_c0) * 2);
        ld_soff_b = (unsigned)(kt * BK * 2);
        // straight-line selects with compile-time indices only: a branchy per-row form makes hipcc keep these small
        // arrays in scratch memory, whose reloads drain the DMA ring (s_waitcnt vmcnt(0)) at every slice
        if (p.linear_a) {                                    // 1x1 stride-1: output pixel m reads input pixel m
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int r = (wave + NW * i) * RPI + lrow;
                const int m = m0 + r;
                const unsigned chunk_b = (unsigned)swz<BK>(lslot, r) * 16u; // LDS slot lslot of row r must hold this chunk
                a_voff[i] = (m < p.M && r < BM) ? (unsigned)m * (unsigned)(p.in_pix_stride * 2) + chunk_b : kOob;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int r = (wave + NW * i) * RPI + lrow;
                const int m = m0 + r;
                const unsigned chunk_b = (unsigned)swz<BK>(lslot, r) * 16u;
                const int n = m / hw;                        // rows beyond M compute harmless garbage, masked below
                const int rem = m - n * hw;
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                a_iy0[i] = oy * p.stride - p.pad_h;
                a_ix0[i] = ox * p.stride - p.pad_w;
                const unsigned vo = (unsigned)(((n * p.Hi + oy * p.stride) * p.Wi + ox * p.stride) * p.in_pix_stride * 2) + chunk_b;
                a_voff[i] = (m < p.M && r < BM) ? vo : kOob;
            }
            if (p.tap_mask) {                                // per-tap validity as one bit each: 3 VALU per DMA piece and slice
#pragma unroll
                for (int i = 0; i < A_IT; ++i) a_mask[i] = 0u;
                int t = 0;
                for (int kh = 0; kh * p.KW < p.taps; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw, ++t) {
#pragma unroll
                        for (int i = 0; i < A_IT; ++i)       // innermost and fully unrolled: the arrays stay in registers
                            a_mask[i] |= ((unsigned)(a_iy0[i] + kh) < (unsigned)p.Hi && (unsigned)(a_ix0[i] + kw) < (unsigned)p.Wi) ? (1u << t) : 0u;
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = (wave + NW * i) * RPI + lrow;
            const int n = n0 + r;
            b_voff[i] = (n < p.Cout && r < BN) ? (unsigned)n * (unsigned)(p.Ktot * 2) + (unsigned)swz<BK>(lslot, r) * 16u : kOob;
        }
        ld_items.advance(p.tiles_n, p.split);
    };

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto issue_slice = [&](const int slot) {     // DMA one K-slice of the loader's item into ring slot
        unsigned char* sa = ring + slot * A_BYTES;
        unsigned char* sb = ring + S * A_BYTES + slot * B_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if (wave + NW * i < A_INSTR) {       // wave-uniform
                unsigned vo = a_voff[i];
                if (!p.linear_a) {
                    if (p.tap_mask) {
                        vo = ((a_mask[i] >> ld_tap) & 1u) ? vo : kOob;
                    } else {
                        const int iy = a_iy0[i] + ld_kh, ix = a_ix0[i] + ld_kw;
                        vo = ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) ? vo : kOob;
                    }
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sa + (wave + NW * i) * 1024), 16, vo, ld_soff_a, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            if (wave + NW * i < B_INSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave + NW * i) * 1024), 16, b_voff[i], ld_soff_b, 0, 0);
        }
        --ld_left;
        ld_soff_b += BK * 2;
        ld_soff_a += BK * 2;
        ld_c0 += BK;
        if (ld_c0 == p.Cin) {                    // next filter tap
            ld_c0 = 0;
            ++ld_tap;
            if (++ld_kw == p.KW) { ld_kw = 0; ++ld_kh; }
            ld_soff_a = (unsigned)((ld_kh * p.in_row_stride32 + ld_kw * p.in_pix_stride) * 2);
        }
    };
    // total slices this block will stream (every item has >= 1 slice; only the last K-split of a tile can be shorter)
    int total_slices = 0;
    {
        ItemWalk it = ld_items;
        while (it.left > 0) {
            const int kb = it.ks * p.k_tiles_per_split;
            total_slices += min(p.k_tiles, kb + p.k_tiles_per_split) - kb;
            it.advance(p.tiles_n, p.split);
        }
    }

    // ------------------------------------------------------------------ consumer (MFMA) state
    int cp_left = 0;                             // slices left in the consumer's current item
    int cp_m0 = 0, cp_n0 = 0, cp_ks = 0;
    f32x4 acc[MI][NI];
    const int frow = lane & 15, fchunk = lane >> 4;
    // BN statistics stay in registers across all tiles of this block that share an n-tile (the usual case: the item
    // stride of a block is a multiple of tiles_n); they are reduced and flushed only when n0 changes / at the end
    float ssum[NI][4], ssq[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[j][e] = ssq[j][e] = 0.f;
    int stats_n0 = -1;
    auto flush_stats = [&]() {                   // all lanes active here
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = ssum[j][e], b = ssq[j][e];
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) {
                    a += __shfl_xor(a, sh);
                    b += __shfl_xor(b, sh);
                }
                const int c = stats_n0 + wn * WTN + j * 16 + fchunk * 4 + e;
                if (frow == 0 && c < p.Cout) {
                    lds_add_f32(stat_a + c * 4, a);
                    lds_add_f32(stat_a + (p.Cout + c) * 4, b);
                }
                ssum[j][e] = ssq[j][e] = 0.f;
            }
    };

    // fragment read offsets inside a tile, per 32-wide K step: the swizzle term only depends on the lane (every
    // fragment starts on a multiple of 16 rows), so slot / fragment-row offsets are immediates of the ds_read
    constexpr int KK = BK / 32;
    unsigned a_foff[KK], b_foff[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        a_foff[kk] = (unsigned)((wm * WTM + frow) * (BK * 2) + swz<BK>(kk * 4 + fchunk, frow) * 16);
        b_foff[kk] = (unsigned)((wn * WTN + frow) * (BK * 2) + swz<BK>(kk * 4 + fchunk, frow) * 16);
    }
    auto mfma_slice = [&](const int slot) {      // acc += A(slot) * B(slot)^T for this wave's WTM x WTN sub-tile
        const unsigned char* cA = ring + slot * A_BYTES;
        const unsigned char* cB = ring + S * A_BYTES + slot * B_BYTES;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(cA + i * 16 * (BK * 2) + a_foff[kk]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[kk]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    int issued = 0, consumed = 0, ld_slot = 0, cp_slot = 0;
    int ep_sum = 0, ep_hist[3] = {0, 0, 0};      // epilogue vector-memory ops of this wave in the last 3 iterations
    // prologue: fill S-1 ring slots
    for (int s = 0; s < S - 1; ++s) {
        if (issued < total_slices) {
            if (ld_left == 0) loader_next_item();
            issue_slice(ld_slot);
            ld_slot = ld_slot + 1 == S ? 0 : ld_slot + 1;
            ++issued;
        }
    }

    while (consumed < total_slices) {
        if (cp_left == 0) {                      // start the next item
            cp_ks = cp_items.ks;
            cp_m0 = cp_items.tm * BM;
            cp_n0 = cp_items.tn * BN;
            const int kb = cp_ks * p.k_tiles_per_split;
            cp_left = min(p.k_tiles, kb + p.k_tiles_per_split) - kb;
            cp_items.advance(p.tiles_n, p.split);
            ++cp_seq;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((flags & FRCNN_CONV_STATS) && bf16_path && cp_n0 != stats_n0) {
                if (stats_n0 >= 0) flush_stats();
                stats_n0 = cp_n0;
            }
        }
#define FRCNN_WAIT_IMM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
        // ---- steady state of a long-K item: the loader is still inside the consumer's item with >= S slices to go, the
        // ring is full and no epilogue store is in flight.  S slices per trip with compile-time ring slots: per slice one
        // counted wait, one barrier, the DMA of slice +S-1 and the MFMA block -- no item / epilogue bookkeeping.
        if (UNIFORM_L && cp_slot == 0 && ep_sum == 0 && ld_seq == cp_seq && ld_left >= S && issued - consumed == S - 1) {
            int trips = ld_left / S;
            issued += trips * S;
            consumed += trips * S;
            cp_left -= trips * S;
            do {
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    FRCNN_WAIT_IMM((S - 2) * LC);
                    __builtin_amdgcn_s_barrier();
                    issue_slice((c + S - 1) % S);
                    mfma_slice(c);
                }
            } while (--trips != 0);
            continue;                            // cp_left >= S - 1 slices of this item remain for the general path
        }
        // slice `consumed` must have landed.  Ops this wave issued AFTER that slice's DMA: the DMA of the younger
        // slices and the epilogue stores of the last S-1 iterations (ep_sum; under-estimates are safe).
        const int younger = issued - consumed - 1;
        // the immediate must be a compile-time constant: a short chain over the values that occur in steady state
        // (ring full, 0..S-1 epilogues in the window); everything else (pipeline drain) rounds down, i.e. waits longer
        if (UNIFORM_L) {
            if (younger == S - 2) {
                if (ep_sum == 0) FRCNN_WAIT_IMM((S - 2) * LC);
                else if (ep_sum == ST_IT) FRCNN_WAIT_IMM((S - 2) * LC + ST_IT);
                else if (ep_sum == 2 * ST_IT) FRCNN_WAIT_IMM((S - 2) * LC + 2 * ST_IT);
                else FRCNN_WAIT_IMM((S - 2) * LC + (S > 3 ? 3 : 2) * ST_IT);
            } else if (S > 3 && younger == S - 3) {
                FRCNN_WAIT_IMM((S > 3 ? S - 3 : 0) * LC);
            } else {
                FRCNN_WAIT_IMM(0);
            }
        } else {
            wait_vmcnt_at_most(younger * Lw);    // few values (Lw <= 2 on the non-uniform tiles)
        }
#undef FRCNN_WAIT_IMM
        __builtin_amdgcn_s_barrier();            // everyone's DMA of this slice landed; everyone finished reading the previous slot
        // age the epilogue history: the slice consumed next was issued one iteration later than this one
        if (ep_sum != 0) {
            if (S == 4) { ep_hist[2] = ep_hist[1]; ep_hist[1] = ep_hist[0]; }
            else if (S == 3) { ep_hist[1] = ep_hist[0]; }
            ep_hist[0] = 0;
            ep_sum = ep_hist[1] + ep_hist[2];
        }
        if (issued < total_slices) {             // refill the slot that was read in the previous iteration
            if (ld_left == 0) loader_next_item();
            issue_slice(ld_slot);
            ld_slot = ld_slot + 1 == S ? 0 : ld_slot + 1;
            ++issued;
        }
        mfma_slice(cp_slot);
        cp_slot = cp_slot + 1 == S ? 0 : cp_slot + 1;
        ++consumed;
        if (--cp_left != 0) continue;

        // ------------------------------------------------------------------ epilogue of the finished item
        // lane holds, for tile (i,j): pixel = wm*WTM + i*16 + (lane&15); couts = wn*WTN + j*16 + (lane>>4)*4 + 0..3
        const int m0 = cp_m0, n0 = cp_n0;
        if (!bf16_path) {
            float* y = reinterpret_cast<float*>(p.y);
            const bool add_bias = (flags & FRCNN_CONV_BIAS) && cp_ks == 0;
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
            continue;
        }

        // Specialised on (STATS, M-tail tile) through wave-uniform branches taken once per tile: the per-element work is
        // bias add, clamp, one packed bf16 convert per pair and three flops of statistics -- no LDS round trips, no branches.
        const bool tail = m0 + BM > p.M;
        float bv[NI][4];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
            if (flags & FRCNN_CONV_BIAS) b = *reinterpret_cast<const f32x4*>(bias_s + n0 + wn * WTN + j * 16 + fchunk * 4);   // zero padded to the tile grid
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][e] = b[e];
        }
        const float lo = (flags & FRCNN_CONV_RELU) ? 0.f : -__builtin_inff();
        auto convert_tile = [&](auto stats_c, auto tail_c) {
            constexpr bool ST = decltype(stats_c)::value, TL = decltype(tail_c)::value;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * WTM + i * 16 + frow;
                const bool row_ok = !TL || m0 + r < p.M;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int cl = wn * WTN + j * 16 + fchunk * 4;
                    u32x2 pk;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 v;
                        v[0] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h] + bv[j][2 * h], lo, __builtin_inff());
                        v[1] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h + 1] + bv[j][2 * h + 1], lo, __builtin_inff());
                        const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32 (RNE)
                        pk[h] = bits;
                        if (ST) {
                            float q0 = __uint_as_float(bits << 16), q1 = __uint_as_float(bits & 0xFFFF0000u);
                            if (TL) { q0 = row_ok ? q0 : 0.f; q1 = row_ok ? q1 : 0.f; }
                            ssum[j][2 * h] += q0;
                            ssq[j][2 * h] += q0 * q0;
                            ssum[j][2 * h + 1] += q1;
                            ssq[j][2 * h + 1] += q1 * q1;
                        }
                    }
                    lds_write_b64(stage_a + r * ROWB + cl * 2, pk);
                }
            }
        };
        if (flags & FRCNN_CONV_STATS) {
            if (tail) convert_tile(std::true_type{}, std::true_type{});
            else convert_tile(std::true_type{}, std::false_type{});
        } else {
            convert_tile(std::false_type{}, std::false_type{});
        }
        // staging tile complete: LDS writes are tracked by lgkmcnt; a raw barrier does NOT drain the DMA ring
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        if (p.direct_out) {
            // output row == GEMM row: buffer stores with a per-lane offset that is fixed for the whole kernel (row-in-pass,
            // 16-byte column chunk) and a scalar offset per pass; lanes outside the tensor carry an out-of-range offset
            const int lrow_o = tid / C8, lc8 = tid - lrow_o * C8;
            const bool col_ok = n0 + lc8 * 8 < p.Cout;
            const unsigned vo_lane = (unsigned)(lrow_o * p.Cout * 2 + lc8 * 16);
            unsigned soff = (unsigned)((m0 * p.Cout + n0) * 2);
            const unsigned pass_pitch = (unsigned)((T / C8) * p.Cout * 2);
            ep_hist[0] = ST_IT;
            ep_sum = ep_hist[0] + ep_hist[1] + ep_hist[2];
#pragma unroll
            for (int it = 0; it < ST_IT; ++it) {
                const int r = lrow_o + it * (T / C8);
                // the pass offset travels in the VGPR offset, not in soffset: with an SGPR soffset hipcc omits the wait state
                // between a 16-byte buffer store and a VALU overwrite of its data registers, and gfx950 does need it
                const unsigned vo = (col_ok && (!tail || m0 + r < p.M)) ? vo_lane + soff : kOob;
                u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * ROWB + lc8 * 16);
                if (flags & FRCNN_CONV_ADD_RES) {
                    const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, vo, 0, 0);
                    float a[8], b[8];
                    unpack8(v, a);
                    unpack8(rv, b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += b[e];
                    v = pack8(a);
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, vo, 0, 0);
                soff += pass_pitch;
            }
        } else {
            // strided scatter (data gradient of a stride-2 1x1 convolution): per-row address computation
            bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
            ep_hist[0] = 0;                      // not counted: the next waits only become more conservative
            ep_sum = ep_hist[1] + ep_hist[2];
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
        }
        // the next item's epilogue writes `stage` again only after >= 1 slice barrier of the main loop
    }
    if (bf16_path && (flags & FRCNN_CONV_STATS)) {
        // flush the block's BN partial sums: one float atomic per channel and statistic, spread over 64 pre-zeroed slots
        if (stats_n0 >= 0) flush_stats();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        double* slot = p.stats + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2) * p.Cout;
        for (int c = tid; c < 2 * p.Cout; c += T) {
            const float v = stat_s[c];
            if (v != 0.f) atomicAdd(slot + c, (double)v);
        }
    }
#endif
}

struct TileCfg { int bm, bn, bk, stages, waves; };

TileCfg pick_tile(const frcnn_conv_desc* d) {
    TileCfg t;
    t.bk = (d->cin % 64 == 0) ? 64 : 32;
    t.bn = d->cout >= 128 ? 128 : 64;
    const long long M = (long long)d->n * d->ho * d->wo;
    const int split = d->split_k > 1 ? d->split_k : 1;
    const long long items128 = ((M + 127) / 128) * ((d->cout + t.bn - 1) / t.bn) * split;
    t.bm = items128 >= 256 ? 128 : 64;
    // ring depth: as deep as 160 KiB allows next to the epilogue staging tile
    t.stages = (t.bm == 128 && t.bn == 128 && t.bk == 64) ? 3 : 4;
    t.waves = 8;
    if (const char* e = getenv("FRCNN_IGEMM_TILE")) {          // kernel development aid: "bm,bn,stages"
        int bm = 0, bn = 0, st = 0, bk = t.bk, nw = t.waves;
        if (sscanf(e, "%d,%d,%d,%d,%d", &bm, &bn, &st, &bk, &nw) >= 3) { t.bm = bm; t.bn = bn; t.stages = st; t.bk = bk; t.waves = nw; }
    }
    return t;
}

template <int BM, int BN, int BK, int S, int NW>
int launch(const ConvParams& p, hipStream_t s) {
    constexpr int base = S * (BM + BN) * BK * 2 + BM * (BN * 2 + 16);
    static_assert(base <= 163840 - 3 * 1024 * 4, "LDS budget (ring + staging + bias/stat arrays of <= 1024 channels)");
    int smem = base;
    if (!(p.flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC))) {
        if (p.flags & FRCNN_CONV_BIAS) smem += p.tiles_n * BN * 4;
        if (p.flags & FRCNN_CONV_STATS) smem += 2 * p.Cout * 4;
    }
    if (smem > 163840) {
        frcnn_set_error("frcnn_conv2d_fprop: cout=%d too large for the LDS bias/statistics arrays", p.Cout);
        return FRCNN_EINVAL;
    }
    if (frcnn_allow_big_lds(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, BK, S, NW>), smem) != 0) {
        frcnn_set_error("frcnn_conv2d_fprop: cannot reserve %d B of LDS", smem);
        return FRCNN_EINVAL;
    }
    const int per_cu = 163840 / smem >= 2 ? 2 : 1;
    int grid = num_cus() * per_cu;
    if (grid > p.items) grid = p.items;
    hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, S, NW>), dim3(grid), dim3(NW * 64), smem, s, p);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_fprop");
    return FRCNN_OK;
}

}  // namespace

int frcnn_conv_tile_dispatch(const void* params, const frcnn_conv_desc* d, hipStream_t s);     // conv_tile.hip

extern "C" int frcnn_conv2d_stat_tiles(const frcnn_conv_desc* d) {
    if (!d) return FRCNN_EINVAL;
    return FRCNN_STAT_SLOTS;
}

static int conv2d_fprop_impl(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* w, const float* bias,
                             const frcnn_bf16* res, const uint8_t* res_mask, void* y, double* stats_partial, const frcnn_bn_reduce* red,
                             frcnn_stream_t stream);

extern "C" int frcnn_conv2d_fprop(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* w, const float* bias,
                                  const frcnn_bf16* res, void* y, double* stats_partial, frcnn_stream_t stream) {
    return conv2d_fprop_impl(d, x, w, bias, res, nullptr, y, stats_partial, nullptr, stream);
}

extern "C" int frcnn_conv2d_dgrad_bnreduce(const frcnn_conv_desc* d, const frcnn_bf16* dz, const frcnn_bf16* w_t, const frcnn_bf16* res,
                                           const uint8_t* res_mask, frcnn_bf16* gx, const frcnn_bn_reduce* red, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(!res_mask || (res && d && (d->flags & FRCNN_CONV_ADD_RES)), "conv2d_dgrad_bnreduce: res_mask without ADD_RES residual");
    FRCNN_CHECK_ARG(red && red->z && red->mean && red->invstd && red->partial, "conv2d_dgrad_bnreduce: incomplete reduce arguments");
    FRCNN_CHECK_ARG(d && !(d->flags & (FRCNN_CONV_STATS | FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC | FRCNN_CONV_BIAS | FRCNN_CONV_RELU)),
                    "conv2d_dgrad_bnreduce: only ADD_RES may be set");
    return conv2d_fprop_impl(d, dz, w_t, nullptr, res, res_mask, gx, nullptr, red, stream);
}

static int conv2d_fprop_impl(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* w, const float* bias,
                             const frcnn_bf16* res, const uint8_t* res_mask, void* y, double* stats_partial, const frcnn_bn_reduce* red,
                             frcnn_stream_t stream) {
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
    p.red_z = red ? reinterpret_cast<const bf16_t*>(red->z) : nullptr;
    p.red_mask = red ? red->relu_mask : nullptr;
    p.red_mean = red ? red->mean : nullptr;
    p.red_invstd = red ? red->invstd : nullptr;
    p.red_part = red ? red->partial : nullptr;
    p.res_mask = res_mask;
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
    p.split = (p.k_tiles + p.k_tiles_per_split - 1) / p.k_tiles_per_split;     // no empty K-splits
    p.tiles_m = (int)((M + t.bm - 1) / t.bm);
    p.tiles_n = (d->cout + t.bn - 1) / t.bn;
    const long long items = (long long)p.tiles_m * p.tiles_n * p.split;
    FRCNN_CHECK_ARG(items < (1ll << 30), "conv2d_fprop: too many tiles");
    p.items = (int)items;
    p.taps = d->kh * d->kw;
    p.tap_mask = p.taps <= 32 ? 1 : 0;
    p.linear_a = (d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->ho == d->hi && d->wo == d->wi) ? 1 : 0;
    p.in_row_stride = (long long)d->wi * d->in_pix_stride;
    p.in_img_stride = (long long)d->hi * p.in_row_stride;
    {
        // 32-bit buffer addressing: voffset (pixel) + soffset (tap) must stay below 2^32 - 16
        const long long halo = (long long)d->pad_h * p.in_row_stride + (long long)d->pad_w * d->in_pix_stride;
        const long long x_elems = (long long)d->n * p.in_img_stride + (long long)d->kw * d->in_pix_stride + 64;   // slack: stem tap reads
        const long long xb = (x_elems + halo) * 2, wb = (long long)d->cout * p.Ktot * 2;
        FRCNN_CHECK_ARG(xb < 0xFFFF0000ll && wb < 0xFFFF0000ll, "conv2d_fprop: operand larger than 4 GiB (32-bit buffer offsets)");
        p.x_bytes = (unsigned)xb;
        p.w_bytes = (unsigned)wb;
        p.in_row_stride32 = (int)p.in_row_stride;
        const long long yb = M * d->cout * 2;
        p.direct_out = (d->out_scatter == 1 && d->out_h == d->ho && d->out_w == d->wo && yb < 0xFFFF0000ll) ? 1 : 0;
        p.y_bytes = p.direct_out ? (unsigned)yb : 0u;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    {
        // one tile (or a run of tiles) per workgroup (conv_tile.hip) unless no instantiation fits
        const char* sel = getenv("FRCNN_TILE_KERNEL");              // "0": force the general kernel (A/B testing aid)
        const bool use_tile = !(sel && sel[0] == '0');
        if (use_tile) {
            const int rc = frcnn_conv_tile_dispatch(&p, d, s);
            if (rc != FRCNN_ENOTSUP) return rc;
        }
    }
    FRCNN_CHECK_ARG(!red, "conv2d_dgrad_bnreduce: shape not supported by the tile kernel (filters with > 32 taps)");

#define FRCNN_DISPATCH(BM_, BN_, BK_, S_) FRCNN_DISPATCH_W(BM_, BN_, BK_, S_, 8)
#define FRCNN_DISPATCH_W(BM_, BN_, BK_, S_, W_) \
    if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_ && t.stages == S_ && t.waves == W_) return launch<BM_, BN_, BK_, S_, W_>(p, s);
    FRCNN_DISPATCH(128, 128, 64, 3)
    FRCNN_DISPATCH(128, 64, 64, 4)
    FRCNN_DISPATCH(64, 128, 64, 4)
    FRCNN_DISPATCH(64, 64, 64, 4)
    FRCNN_DISPATCH(64, 64, 64, 3)
    FRCNN_DISPATCH(64, 64, 64, 2)
    FRCNN_DISPATCH(128, 64, 64, 3)
    FRCNN_DISPATCH(128, 64, 64, 2)
    FRCNN_DISPATCH_W(128, 128, 64, 3, 4)
    FRCNN_DISPATCH_W(128, 64, 128, 2, 4)
    FRCNN_DISPATCH_W(128, 64, 128, 2, 8)
    FRCNN_DISPATCH_W(64, 128, 128, 2, 4)
    FRCNN_DISPATCH_W(128, 64, 64, 4, 4)
    FRCNN_DISPATCH_W(64, 64, 128, 3, 4)
    FRCNN_DISPATCH(128, 128, 32, 4)
    FRCNN_DISPATCH(128, 64, 32, 4)
    FRCNN_DISPATCH(64, 128, 32, 4)
    FRCNN_DISPATCH(64, 64, 32, 4)
#undef FRCNN_DISPATCH
#undef FRCNN_DISPATCH_W
    frcnn_set_error("conv2d_fprop: no tile configuration for bm=%d bn=%d bk=%d", t.bm, t.bn, t.bk);
    return FRCNN_EINVAL;
}
