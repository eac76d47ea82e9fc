// Implicit-GEMM convolution, bf16 output: ONE output tile per workgroup, several workgroups per CU.
//
// GEMM view: C[M = N*Ho*Wo pixels][Cout] = A[M][K = KH*KW*Cin] * W[Cout][K]^T, A gathered on the fly from the NHWC input (im2col is never
// materialised); the MFMA is issued with swapped operands so that a lane ends with 4 consecutive output channels of one pixel;
// K walked in BK-wide slices that never straddle a filter tap, operands DMA'd global -> LDS with the XOR swizzle
// applied on the source side).  What differs is the schedule: a workgroup owns one BM x BN tile, streams its K slices
// through a short ring (S = 2 or 3 slots), runs the epilogue and exits.  Latency is hidden by OCCUPANCY instead of
// cross-tile prefetch: the epilogue staging tile aliases the ring (no DMA is in flight any more), so the LDS footprint
// is max(ring, staging) and two or three workgroups share a CU -- one tile's prologue / epilogue overlaps its
// neighbours' MFMA loops.  Everything the inner loop could branch on is a template parameter (LIN: 1x1 stride-1
// addressing, STATS: BatchNorm partial sums) or folded into per-lane registers computed once (tap validity bit masks,
// fragment read offsets), which keeps the per-slice instruction stream short: the loop is issue-bound long before it is
// MFMA- or LDS-bound (measured: 83 SALU + 60 VALU per slice and wave in the general persistent kernel).
#include "conv_common.h"
#include <stdlib.h>

namespace {

// BNIN (frcnn_conv2d_fprop_bnin on a 1x1 / stride-1 layer: the third convolution of a bottleneck block): the A operand is the RAW output z of the
// previous convolution and this kernel applies that layer's training-mode BatchNorm + ReLU itself.  Scale / shift of the Cin input channels
// come from the f64 statistics slots (one channel per thread, bn_train_apply_kernel's additions in its order: same bits) into LDS behind
// everything else; every landed A slice is transformed IN PLACE by all 512 threads (two 16-byte slots each) between the slice barrier and
// one more barrier -- an element is transformed once per workgroup, not once per wave column as a fragment-register form would -- and the
// activation + its ReLU bit mask, which the backward pass reads, leave from there: rows are dealt to the channel-part workgroups of a
// pixel tile by (row / 8 + slice) mod parts, a wave-uniform rule, with buffer stores whose masked lanes go out of range, so every wave
// knows how many stores it has in flight behind the next slice's DMA and the counted vmcnt waits stay exact.
template <int BM, int BN, int BK, int S, bool LIN, int SMODE, int OCC, bool MULTI, bool F32, bool KWS, bool FIX, int F8, bool BNIN = false>
__global__ __launch_bounds__(512, 2 * OCC) void conv_tile_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!BNIN || (LIN && S == 2 && BM == 128 && BK == 64 && SMODE != 2 && !F32 && !KWS && !FIX && !F8), "BatchNorm on the input side: 1x1 stride-1 forward layers, two-slot ring");
    // SMODE 0: plain, 1: BatchNorm statistics of the output (forward), 2: BatchNorm-backward reduce of the consumer layer
    constexpr bool STATS = SMODE == 1, RED = SMODE == 2;
    // 8 waves as a 2 x 4 grid (a 4-wave variant with four times the MFMA work per wave and a software-pipelined fragment-read
    // loop were built and measured within +-5 % on every layer shape: removed)
    constexpr int NW = 8, T = NW * 64, WM = 2, WN = NW / 2;
    constexpr int CPR = BK / 8;                  // 16-byte chunks per tile row
    constexpr int RPI = 64 / CPR;                // rows written by one DMA wave instruction (1 KiB)
    constexpr int A_INSTR = BM / RPI, B_INSTR = BN / RPI;
    constexpr int A_IT = (A_INSTR + NW - 1) / NW, B_IT = (B_INSTR + NW - 1) / NW;
    constexpr bool A_UNI = A_INSTR % NW == 0, B_UNI = B_INSTR % NW == 0;
    constexpr int LC = A_INSTR / NW + B_INSTR / NW;
    static_assert(S == 2 || (S >= 3 && S <= 6 && A_UNI && B_UNI), "counted waits need a uniform DMA split");
    static_assert(!MULTI || (S == 2 && A_UNI && B_UNI), "tile runs use the two-slot ring with a uniform DMA split");
    static_assert(!F32 || (!MULTI && SMODE == 0), "fp32 / split-K output: one tile per workgroup, no statistics");
    constexpr int STG32 = BM * (BN * 4 + 16);   // fp32 staging tile (F32)
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16, KK = BK / 32;
    static_assert(MI >= 1 && NI >= 1, "tile too small for the wave grid");
    static_assert(2 * BN <= T, "statistics flush: one thread per (statistic, channel)");
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int ROWB = BN * 2 + 16;            // staging row pitch (bytes)
    constexpr int C8 = BN / 8, ST_IT = (BM * C8) / T;
    static_assert((BM * C8) % T == 0, "store loop covers the tile in whole passes");
    // KWS (3x3, stride 1, pad 1): the three kw taps of one (kh, channel chunk) group share ONE staged A image -- the BM + 2
    // pixel rows around the tile, fetched once instead of three times; tap kw reads it shifted by kw rows.  Pixels whose
    // left / right neighbour lies in another image row (ox == 0 for kw = 0, ox == Wo-1 for kw = 2) are zeroed in the fragment
    // registers; vertical validity is a property of the staged source row (per-kh bit mask, zero-filled by the range check).
    static_assert(!KWS || (!LIN && !MULTI && !F32 && BM == 128 && BK == 64 && S == 3), "kw-sharing mode: 128-row tiles, 3-slot B ring");
    constexpr int AK_IT = 3, AK_BYTES = AK_IT * NW * 1024;          // shared A image: 192 rows x 128 B (130 used), two of them
    // FIX (layers with fewer tiles than CUs and a very long K: the RPN's 3x3 1024->256 at M = 7,488): the K range of a tile is split
    // over TWO workgroups placed on one XCD, so every CU holds two workgroups.  Each writes its fp32 partial tile to the
    // caller's workspace; the SECOND to arrive (one relaxed atomic per tile, no waiting) adds its partner's partial and runs the
    // ordinary epilogue -- bias / residual / statistics / fused reduce see the complete sums.  a + b == b + a: the result does not
    // depend on who arrives last.
    static_assert(!FIX || (!MULTI && !F32 && !KWS), "split-K fix-up: one-tile kernel only");
    // F8 (1: x in e4m3 -- forward activations; 2: x in e5m2 -- the gradients of a data-gradient launch; weights always e4m3): both
    // operands are fp8 bytes.  The host passes every element count HALVED (Cin, pixel stride, K: two fp8 values take the
    // place of one bf16), so tiles, DMA, swizzle and fragment reads are byte for byte those of the bf16 kernel: a 128-byte staged row
    // is 128 K values instead of 64.  The two 16-byte fragment reads of a slice (kk = 0, 1) are the 32 bytes ONE
    // v_mfma_scale_f32_16x16x128_f8f6f4 takes per lane (lane group g contracts bytes [16g, 16g+16) and [64+16g, 64+16g+16) of the row
    // for BOTH operands -- a consistent permutation of K, which a dot product does not see), issued with unit block scales: the same
    // matrix-pipe cycles per slice as the two bf16 MFMAs it replaces, at twice the K.  Dequantisation (activation scale x weight
    // scale of the output channel) is one multiply in the epilogue.
    static_assert(!F8 || (BK == 64 && !F32), "fp8 operands: 128-byte slices, bf16 output");
    constexpr int B_BASE = KWS ? 2 * AK_BYTES : S * A_BYTES;
    constexpr int RING = KWS ? 2 * AK_BYTES + S * B_BYTES : S * (A_BYTES + B_BYTES), STG = BM * ROWB;
    // one tile per workgroup: the staging tile aliases the drained ring.  Tile runs (MULTI): the ring keeps prefetching the
    // next tile while the epilogue runs, so the staging tile has its own LDS.
    constexpr int STAGE_OFF = MULTI ? RING : 0;
    constexpr int BIG = F32 ? (RING > STG32 ? RING : STG32) : MULTI ? RING + STG : (RING > STG ? RING : STG);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                  // [S A tiles][S B tiles]
    unsigned char* stage = smem + STAGE_OFF;     // [BM][ROWB]
    float* s_scale = reinterpret_cast<float*>(smem + BIG + 2 * BN * 4);      // BNIN: [Cin] scale, [Cin] shift

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef FRCNN_STAMPS
    // kernel-development build (tools/conv_stamps.py): wave 0 records shader-clock stamps of the workgroup's phases
    // dbg[blockIdx.x][24] = {entry, K loop start, K loop end (last tile), epilogue stores issued (last tile), exit, HW_ID, realtime, XCC,
    //   slices 4 and 5 of the K loop, 5 stamps each: top of the step, after the vmcnt wait, after the barrier, after the DMA issue, after
    //   the last MFMA was issued -- kept in registers until the kernel's end (a store inside the loop would count in vmcnt)}
    unsigned long long ks[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FRCNN_KSTAMP(slice, j) do { if (p.dbg && ((slice) == 4 || (slice) == 5)) ks[((slice) - 4) * 5 + (j)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FRCNN_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 24 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    if (p.dbg && tid == 0) {
        p.dbg[(size_t)blockIdx.x * 24 + 5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, 32 bits
        p.dbg[(size_t)blockIdx.x * 24 + 6] = __builtin_amdgcn_s_memrealtime();
        p.dbg[(size_t)blockIdx.x * 24 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
#else
#define FRCNN_STAMP(i) do { } while (0)
#define FRCNN_KSTAMP(slice, j) do { } while (0)
#endif
    FRCNN_STAMP(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int flags = p.flags;

    // workgroup -> run of tiles_per_block consecutive m-tiles of one n-tile (one tile unless MULTI).  Workgroups b, b+8, ...
    // run on one XCD: each XCD takes a contiguous chunk of the unit list (n fastest), so the workgroups that share an L2
    // read neighbouring pixel rows and the same weight panels.
    int tm_begin, tn, tile_count;
    int fix_pair = 0, fix_split = 0;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3;
        const int q = nb >> 3, r = nb & 7;
        int unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        if (FIX) {
            // gridDim.x is a multiple of 16: every XCD's chunk starts at an even unit and has an even length, so the two halves
            // (units 2t, 2t+1) of tile t run on the same XCD, eight workgroup ids apart
            fix_pair = unit >> 1;
            fix_split = unit & 1;
            if (fix_pair >= p.items) return;
            unit = fix_pair;
        }
        const int run = unit / p.tiles_n;
        tn = unit - run * p.tiles_n;
        tm_begin = MULTI ? run * p.tiles_per_block : run;
        tile_count = MULTI ? min(p.tiles_per_block, p.tiles_m - tm_begin) : 1;
    }
    const int n0 = tn * BN;

    // bias of this tile's channels (lane: 4 consecutive channels per 16-wide fragment column), issued before the K loop
    float bv[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int c = n0 + wn * WTN + j * 16 + fchunk * 4;
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!F8 && (flags & FRCNN_CONV_BIAS) && c < p.Cout) b = *reinterpret_cast<const f32x4*>(p.bias + c);   // Cout % 8 == 0
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = b[e];
    }

    // F8: x_scale * w_scale[cout] and the bias of the tile's BN channels wait in LDS for the epilogue (held in registers through the K
    // loop they were the 8 VGPRs that made the fp8 forms spill: a scratch reload in front of every slice's DMA issue)
    float* s_dq = reinterpret_cast<float*>(smem + BIG + 2 * BN * 4);           // F8: [BN] dequantisation factors, [BN] bias
    if (F8) {
        const float xs = *p.f8_x_scale;
        if (tid < BN) {
            const int c = n0 + tid;
            s_dq[tid] = c < p.Cout ? p.f8_w_scale[c] * xs : 0.f;
            s_dq[BN + tid] = ((flags & FRCNN_CONV_BIAS) && c < p.Cout) ? p.bias[c] : 0.f;
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ loader (LDS-DMA) state
    const long long halo = (long long)p.pad_h * p.in_row_stride + (long long)p.pad_w * p.in_pix_stride;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - halo), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned a_voff[KWS ? AK_IT : A_IT], a_mask[KWS ? AK_IT : A_IT], b_voff[B_IT];
    unsigned edge_l[MI], edge_r[MI];            // KWS: all-ones, or zero where the lane's pixel has no left / right neighbour in its image row
    const int lrow = lane / CPR, lslot = lane % CPR;
    auto setup_tile = [&](const int m0) {        // per-lane source offsets (and tap validity masks) of the A rows of tile m0
        if (KWS) {
            const int hw = p.Ho * p.Wo;
#pragma unroll
            for (int i = 0; i < AK_IT; ++i) {
                const int q = (wave + NW * i) * RPI + lrow;          // staged row q holds the kw = 1 source of virtual pixel m0 + q - 1
                const int mv = m0 + q - 1;
                const bool ok = q < BM + 2 && mv >= 0 && mv < p.M;
                const int mc = ok ? mv : 0;
                const int n = mc / hw;
                const int rem = mc - n * hw;
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                a_voff[i] = ok ? (unsigned)(((n * p.Hi + oy) * p.Wi + ox) * p.in_pix_stride * 2) + (unsigned)swz<BK>(lslot, q) * 16u : kOob;
                a_mask[i] = ((unsigned)(oy - 1) < (unsigned)p.Hi ? 1u : 0u) | 2u | ((unsigned)(oy + 1) < (unsigned)p.Hi ? 4u : 0u);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * WTM + i * 16 + frow;
                const int ox = m % p.Wo;
                edge_l[i] = ox == 0 ? 0u : 0xFFFFFFFFu;
                edge_r[i] = ox == p.Wo - 1 ? 0u : 0xFFFFFFFFu;
            }
        } else if (LIN) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int r = (wave + NW * i) * RPI + lrow;
                const int m = m0 + r;
                a_voff[i] = (m < p.M && r < BM) ? (unsigned)m * (unsigned)(p.in_pix_stride * 2) + (unsigned)swz<BK>(lslot, r) * 16u : kOob;
                a_mask[i] = 1u;
            }
        } else {
            const int hw = p.Ho * p.Wo;
            int iy0[A_IT], ix0[A_IT];
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int r = (wave + NW * i) * RPI + lrow;
                const int m = m0 + r;
                const int n = m / hw;            // rows beyond M compute harmless garbage, masked below
                const int rem = m - n * hw;
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                iy0[i] = oy * p.stride - p.pad_h;
                ix0[i] = ox * p.stride - p.pad_w;
                const unsigned vo = (unsigned)(((n * p.Hi + oy * p.stride) * p.Wi + ox * p.stride) * p.in_pix_stride * 2) + (unsigned)swz<BK>(lslot, r) * 16u;
                a_voff[i] = (m < p.M && r < BM) ? vo : kOob;
                a_mask[i] = 0u;
            }
            int t = 0;                           // bit t of a_mask: filter tap t of this row lies inside the image
            for (int kh = 0; kh * p.KW < p.taps; ++kh)
                for (int kw = 0; kw < p.KW; ++kw, ++t) {
#pragma unroll
                    for (int i = 0; i < A_IT; ++i)
                        a_mask[i] |= ((unsigned)(iy0[i] + kh) < (unsigned)p.Hi && (unsigned)(ix0[i] + kw) < (unsigned)p.Wi) ? (1u << t) : 0u;
                }
        }
    };
    setup_tile(tm_begin * BM);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int r = (wave + NW * i) * RPI + lrow;
        const int n = n0 + r;
        b_voff[i] = (n < p.Cout && r < BN) ? (unsigned)n * (unsigned)(p.Ktot * 2) + (unsigned)swz<BK>(lslot, r) * 16u : kOob;
    }
    // F32: blockIdx.y selects a K range of k_tiles_per_split slices (split-K, 1x1 filters only)
    const int kt0 = F32 ? blockIdx.y * p.k_tiles_per_split : FIX ? fix_split * p.k_tiles_per_split : 0;
    const int nk = (F32 || FIX) ? min(p.k_tiles, kt0 + p.k_tiles_per_split) - kt0 : p.k_tiles;
    int ld_c0 = kt0 * BK, ld_tap = 0, ld_kh = 0, ld_kw = 0;
    int ld_k = 0, ld_m0 = tm_begin * BM;         // loader position: K slice inside its tile, first row of its tile
    unsigned ld_soff_a = (unsigned)(kt0 * BK * 2), ld_soff_b = (unsigned)(kt0 * BK * 2);
    if (FIX && !LIN) {                           // the second half starts inside the filter: slice kt0 = (tap, channel chunk)
        ld_tap = (kt0 * BK) / p.Cin;
        ld_c0 = kt0 * BK - ld_tap * p.Cin;
        ld_kh = ld_tap / p.KW;
        ld_kw = ld_tap - ld_kh * p.KW;
        ld_soff_a = (unsigned)((ld_kh * p.in_row_stride32 + ld_kw * p.in_pix_stride + ld_c0) * 2);
    }

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto issue_slice = [&](const int slot) {     // DMA the loader's next K slice into ring slot
        unsigned char* sa = ring + slot * A_BYTES;
        unsigned char* sb = ring + B_BASE + slot * B_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if (A_UNI || wave + NW * i < A_INSTR) {
                const unsigned vo = LIN ? a_voff[i] : (((a_mask[i] >> ld_tap) & 1u) ? a_voff[i] : kOob);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sa + (wave + NW * i) * 1024), 16, vo, ld_soff_a, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            if (B_UNI || wave + NW * i < B_INSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave + NW * i) * 1024), 16, b_voff[i], ld_soff_b, 0, 0);
        }
        ld_soff_b += BK * 2;
        ld_soff_a += BK * 2;
        if (!LIN) {
            ld_c0 += BK;
            if (ld_c0 == p.Cin) {                // next filter tap
                ld_c0 = 0;
                ++ld_tap;
                if (++ld_kw == p.KW) { ld_kw = 0; ++ld_kh; }
                ld_soff_a = (unsigned)((ld_kh * p.in_row_stride32 + ld_kw * p.in_pix_stride) * 2);
            }
        }
        if (MULTI && ++ld_k == nk) {             // the loader moves on to the next tile of the run
            ld_k = 0;
            ld_m0 += BM;
            ld_c0 = ld_tap = ld_kh = ld_kw = 0;
            ld_soff_a = ld_soff_b = 0;
            setup_tile(ld_m0);                   // (beyond the run's last tile nothing is issued any more)
        }
    };

    // KWS loader: slices in (channel chunk, kh, kw) order; the kw = 0 slice of a group also brings the group's A image
    int kl_c0 = 0, kl_kh = 0, kl_kw = 0, kl_abuf = 0;
    auto issue_slice_kws = [&](const int slot) {
        unsigned char* sb = ring + B_BASE + slot * B_BYTES;
        const unsigned soff_b = (unsigned)(((kl_kh * 3 + kl_kw) * p.Cin + kl_c0) * 2);
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave + NW * i) * 1024), 16, b_voff[i], soff_b, 0, 0);
        if (kl_kw == 0) {
            unsigned char* sa = ring + kl_abuf * AK_BYTES;
            const unsigned soff_a = (unsigned)((kl_kh * p.in_row_stride32 + p.in_pix_stride + kl_c0) * 2);
#pragma unroll
            for (int i = 0; i < AK_IT; ++i) {
                const unsigned vo = ((a_mask[i] >> kl_kh) & 1u) ? a_voff[i] : kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sa + (wave + NW * i) * 1024), 16, vo, soff_a, 0, 0);
            }
            kl_abuf ^= 1;
        }
        if (++kl_kw == 3) {
            kl_kw = 0;
            if (++kl_kh == 3) { kl_kh = 0; kl_c0 += BK; }
        }
    };

    // ------------------------------------------------------------------ consumer (MFMA) state
    f32x4 acc[MI][NI];
    // fragment read offsets per 32-wide K step: the swizzle term only depends on the lane (fragments start on multiples
    // of 16 rows), so slot and fragment-row offsets are immediates of the ds_read
    unsigned a_foff[KK], b_foff[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        a_foff[kk] = (unsigned)((wm * WTM + frow) * (BK * 2) + swz<BK>(kk * 4 + fchunk, frow) * 16);
        b_foff[kk] = (unsigned)((wn * WTN + frow) * (BK * 2) + swz<BK>(kk * 4 + fchunk, frow) * 16);
    }
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    auto cat8 = [](const bf16x8 lo, const bf16x8 hi) {           // the 32 fp8 bytes of a lane: its kk = 0 and kk = 1 fragments
        const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
        return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    };
    auto mfma_slice = [&](const int slot) {      // acc += A(slot) * B(slot)^T, fragments of step kk+1 fetched under the MFMAs of kk
        const unsigned char* cA = ring + slot * A_BYTES;
        const unsigned char* cB = ring + B_BASE + slot * B_BYTES;
        bf16x8 af[2][MI], bfr[2][NI];
        if (F8) {
            static_assert(!F8 || KK == 2, "fp8: one 128-deep MFMA per 128-byte slice");
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < NI; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[kk]);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[kk][i] = *reinterpret_cast<const bf16x8*>(cA + i * 16 * (BK * 2) + a_foff[kk]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const i32x8 a8 = cat8(af[0][i], af[1][i]);
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(bfr[0][j], bfr[1][j]), a8, acc[i][j], 0, F8 == 2 ? 1 : 0, 0, 0x7F7F7F7F, 0,
                                                                                  0x7F7F7F7F);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(cA + i * 16 * (BK * 2) + a_foff[0]);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[0]);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[nxt][i] = *reinterpret_cast<const bf16x8*>(cA + i * 16 * (BK * 2) + a_foff[kk + 1]);
#pragma unroll
                for (int j = 0; j < NI; ++j) bfr[nxt][j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[kk + 1]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
        }
    };

    // KWS: fragment offsets of the A image per kw (the swizzle term follows the shifted row)
    unsigned ak_foff[3][KK];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            ak_foff[kw][kk] = (unsigned)((wm * WTM + frow + kw) * (BK * 2) + swz<BK>(kk * 4 + fchunk, frow + kw) * 16);
    auto mfma_slice_kws = [&](const int slot, const int abuf, auto kw_c) {
        constexpr int kw = decltype(kw_c)::value;
        const unsigned char* cA = ring + abuf * AK_BYTES;
        const unsigned char* cB = ring + B_BASE + slot * B_BYTES;
        u32x4 af[2][MI];
        bf16x8 bfr[2][NI];
        if (F8) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < NI; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[kk]);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    af[kk][i] = *reinterpret_cast<const u32x4*>(cA + i * 16 * (BK * 2) + ak_foff[kw][kk]);
                    if (kw == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) af[kk][i][q] &= edge_l[i];
                    }
                    if (kw == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) af[kk][i][q] &= edge_r[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const i32x8 a8 = cat8(__builtin_bit_cast(bf16x8, af[0][i]), __builtin_bit_cast(bf16x8, af[1][i]));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(bfr[0][j], bfr[1][j]), a8, acc[i][j], 0, F8 == 2 ? 1 : 0, 0, 0x7F7F7F7F, 0,
                                                                                  0x7F7F7F7F);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const u32x4*>(cA + i * 16 * (BK * 2) + ak_foff[kw][0]);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[0]);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[nxt][i] = *reinterpret_cast<const u32x4*>(cA + i * 16 * (BK * 2) + ak_foff[kw][kk + 1]);
#pragma unroll
                for (int j = 0; j < NI; ++j) bfr[nxt][j] = *reinterpret_cast<const bf16x8*>(cB + j * 16 * (BK * 2) + b_foff[kk + 1]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                u32x4 a = af[cur][i];
                if (kw == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] &= edge_l[i];
                }
                if (kw == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] &= edge_r[i];
                }
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[cur][j], __builtin_bit_cast(bf16x8, a), acc[i][j], 0, 0, 0);
            }
        }
    };

    // epilogue state that does not depend on the tile
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.y_bytes, 0x00020000);
    const int lrow_o = tid / C8, lc8 = tid - lrow_o * C8;
    const bool col_ok = n0 + lc8 * 8 < p.Cout;
    const unsigned vo_lane = (unsigned)(lrow_o * p.Cout * 2 + lc8 * 16);
    const unsigned pass_pitch = (unsigned)((T / C8) * p.Cout * 2);
    const unsigned stage_a = lds_addr(stage);
    float ssum[NI][4], ssq[NI][4];               // BatchNorm partial sums of this workgroup's tiles (one n-tile)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[j][e] = ssq[j][e] = 0.f;
    const float lo = (flags & FRCNN_CONV_RELU) ? 0.f : -__builtin_inff();
    // RED: this kernel's output is the gradient g arriving at a BatchNorm layer; accumulate that layer's backward sums
    // sum(g*m) and sum(g*m*z) (m = its ReLU mask) for this thread's 8 channels over the rows it stores -- the separate reduce
    // pass (one more read of g, z and the mask, one more launch per layer) disappears.  mean / invstd enter at the flush:
    // sum(g*m*xhat) = invstd * (sum(g*m*z) - mean * sum(g*m)).
    const __amdgpu_buffer_rsrc_t rsrc_rz = __builtin_amdgcn_make_buffer_rsrc((void*)p.red_z, 0, p.y_bytes, 0x00020000);
    float rsg[8], rsgz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) rsg[e] = rsgz[e] = 0.f;

    // ------------------------------------------------------------------ K loops of the run's tiles
#define FRCNN_WAIT_IMM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    FRCNN_STAMP(1);
    const int total_slices = tile_count * nk;
    if (KWS) {
        issue_slice_kws(0);                      // (nk = 9 * Cin / 64 >= 9)
        issue_slice_kws(1);
    } else {
        const int pre = total_slices < S - 1 ? total_slices : S - 1;
        for (int s = 0; s < pre; ++s) issue_slice(s);
    }
    int to_issue = total_slices - (total_slices < S - 1 ? total_slices : S - 1);
    int slot = 0;
    int bn_st = 0;                               // BNIN: stores this wave issued after its most recent DMA issue
    // wait until the DMA issued `base + bn_st` vector-memory operations ago has landed (base: epilogue stores younger than it)
    auto wait_dma = [&](const int base) {
        switch (base + bn_st) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    const __amdgpu_buffer_rsrc_t rsrc_act = __builtin_amdgcn_make_buffer_rsrc(BNIN ? (void*)p.bnin_act : (void*)p.y, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_msk = __builtin_amdgcn_make_buffer_rsrc(BNIN ? (void*)p.bnin_mask : (void*)p.y, 0, p.x_bytes >> 4, 0x00020000);
    if (BNIN) {
        if (tid < p.Cin) {
            const int c = tid;
            double a[4][FRCNN_STAT_SLOTS / 4], b[4][FRCNN_STAT_SLOTS / 4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) {
                    a[sl][k] = p.bnin_part[((long long)(sl + 4 * k) * 2) * p.Cin + c];
                    b[sl][k] = p.bnin_part[((long long)(sl + 4 * k) * 2 + 1) * p.Cin + c];
                }
            double ps[4], pq[4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) { s0 += a[sl][k]; s1 += b[sl][k]; }
                ps[sl] = s0;
                pq[sl] = s1;
            }
            const double sum = ps[0] + ps[1] + ps[2] + ps[3];
            const double ssq_ = pq[0] + pq[1] + pq[2] + pq[3];
            const double mean = sum * p.bnin_inv_count;
            double var = ssq_ * p.bnin_inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)p.bnin_eps));
            const float sc = p.bnin_gamma[c] * invstd;
            const float sh = p.bnin_beta[c] - (float)mean * sc;
            // (asm forms: a DS write hipcc emits itself waits for ALL pending LDS-DMA)
            asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(s_scale + c)), "v"(sc) : "memory");
            asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(s_scale + p.Cin + c)), "v"(sh) : "memory");
            if (blockIdx.x == 0) {
                p.bnin_mean[c] = (float)mean;
                p.bnin_invstd[c] = invstd;
                p.bnin_mm[c] = p.bnin_mm[c] * p.bnin_momentum + (float)mean * (1.f - p.bnin_momentum);
                p.bnin_mv[c] = p.bnin_mv[c] * p.bnin_momentum + (float)(var * p.bnin_unbias) * (1.f - p.bnin_momentum);
            }
        }
        // (workgroup 0's four stores are YOUNGER than the slices issued so far: operations beyond a wait's count only make it conservative)
    }
    // BNIN: BatchNorm + ReLU of the landed A slice `ks` of the tile at row m0, in place (slot position sp of row r holds channel vector
    // sp ^ (r & 7) of the slice), and this workgroup's share of the activation / mask rows
    auto bnin_slice = [&](const int slot_, const int ks, const int m0_) {
        unsigned char* cA = ring + slot_ * A_BYTES;
        const unsigned abase = lds_addr(cA);
        const int sp = tid & 7, r0 = tid >> 3;
        const int c8 = sp ^ (r0 & 7);
        const float* csc = s_scale + ks * 64 + c8 * 8;
        const float* csh = csc + p.Cin;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(csc), sc1 = *reinterpret_cast<const f32x4*>(csc + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(csh), sh1 = *reinterpret_cast<const f32x4*>(csh + 4);
#pragma unroll
        for (int k = 0; k < BM / 64; ++k) {
            const int r = r0 + 64 * k;
#ifndef FRCNN_BNIN_NOMATH
            const u32x4 raw = *reinterpret_cast<const u32x4*>(cA + r * 128 + sp * 16);
            float x[8];
            unpack8(raw, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = fmaxf(x[e] * sc0[e] + sh0[e], 0.f);
                x[4 + e] = fmaxf(x[4 + e] * sc1[e] + sh1[e], 0.f);
            }
            const u32x4 pk = pack8(x);
            asm volatile("ds_write_b128 %0, %1" ::"v"(abase + (unsigned)(r * 128 + sp * 16)), "v"(pk) : "memory");
#else
            const u32x4 pk = u32x4{(unsigned)r, 0u, 0u, 0u};
#endif
#ifndef FRCNN_BNIN_NOSTORE
            if (((wave + 8 * k + ks) % p.tiles_n) == tn) {          // (r >> 3 == wave + 8 k: wave-uniform)
                const bool ok = m0_ + r < p.M;
                const unsigned row = (unsigned)(m0_ + r);
                __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc_act, ok ? (row * (unsigned)p.Cin + (unsigned)(ks * 64 + c8 * 8)) * 2u : kOob, 0, 0);
                bn_st += 1;
#ifndef FRCNN_BNIN_NOMASK
#ifndef FRCNN_BNIN_POOLMASK
                // one byte per lane.  (Pooling the eight bytes of a row with cross-lane shuffles into one 8-byte store -- what the patch /
                // weights-resident forms do, where no LDS-DMA is pending in the transforming waves -- measured SLOWER here: 64 -> 256 at
                // M = 116,936 36.0 against 32.2 us, 128 -> 512 28.9 against 27.6: the shuffles are DS operations behind the ring's DMA)
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)relu_bits8(pk), rsrc_msk, ok ? row * (unsigned)(p.Cin >> 3) + (unsigned)(ks * 8 + c8) : kOob, 0, 0);
#else
                const u32x2 mk = pool_mask8(relu_bits8(pk), c8);
                __builtin_amdgcn_raw_buffer_store_b64(mk, rsrc_msk, (ok && sp == 0) ? row * (unsigned)(p.Cin >> 3) + (unsigned)(ks * 8) : kOob, 0, 0);
#endif
                bn_st += 1;
#endif
            }
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int t = 0; t < tile_count; ++t) {
        const int m0 = (tm_begin + t) * BM;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int left = nk;                           // slices of this tile still to consume
        if (KWS) {
            // one (channel chunk, kh) group per trip, its three kw slices unrolled: slice s waits until only the DMA of
            // slice s+1 is outstanding (B piece, plus the next group's A image when s+1 opens a group)
            constexpr int LCB = B_INSTR / NW;
            static_assert(!KWS || B_UNI, "kw-sharing mode: uniform B split");
            int sl = 0, bslot = 0;
            auto kws_step = [&](auto kw_c, const int abuf) {
                constexpr int kw = decltype(kw_c)::value;
                FRCNN_KSTAMP(sl, 0);
                if (sl + 1 < nk) FRCNN_WAIT_IMM(kw == 2 ? LCB + AK_IT : LCB);
                else FRCNN_WAIT_IMM(0);
                FRCNN_KSTAMP(sl, 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                FRCNN_KSTAMP(sl, 2);
                if (sl + 2 < nk) issue_slice_kws(bslot == 0 ? 2 : bslot - 1);
                FRCNN_KSTAMP(sl, 3);
                mfma_slice_kws(bslot, abuf, kw_c);
                FRCNN_KSTAMP(sl, 4);
                bslot = bslot == 2 ? 0 : bslot + 1;
                ++sl;
            };
            for (int g = 0; 3 * g < nk; ++g) {
                kws_step(std::integral_constant<int, 0>{}, g & 1);
                kws_step(std::integral_constant<int, 1>{}, g & 1);
                kws_step(std::integral_constant<int, 2>{}, g & 1);
            }
            left = 0;
        }
        // a tile's first slice was issued BEFORE the previous tile's epilogue stores: those ST_IT stores may stay in flight
        bool after_epilogue = MULTI && t > 0 && p.direct_out;
        while (left > 0) {
            if (slot == 0 && !after_epilogue && left >= S && to_issue >= S) {
                // whole trips around the ring: compile-time slots, ring stays full
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    FRCNN_KSTAMP(nk - left + c, 0);
                    if (BNIN) wait_dma(0);
                    else FRCNN_WAIT_IMM((S - 2) * LC);
                    FRCNN_KSTAMP(nk - left + c, 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's fragment reads of the slot refilled next have completed
                    __builtin_amdgcn_s_barrier();        // slice landed for everyone; everyone finished reading the slot refilled next
                    FRCNN_KSTAMP(nk - left + c, 2);
                    issue_slice((c + S - 1) % S);
                    if (BNIN) { bn_st = 0; bnin_slice(c, nk - left + c, m0); }
                    FRCNN_KSTAMP(nk - left + c, 3);
                    mfma_slice(c);
                    FRCNN_KSTAMP(nk - left + c, 4);
                }
                to_issue -= S;
                left -= S;
                continue;
            }
            if (BNIN) {
                if (after_epilogue) wait_dma(ST_IT);                       // (the previous tile's epilogue stores are younger than this slice's DMA)
                else if (to_issue > 0) wait_dma(0);
                else FRCNN_WAIT_IMM(0);
            } else if (after_epilogue) FRCNN_WAIT_IMM(ST_IT);              // (MULTI: S == 2)
            else if (to_issue > 0) FRCNN_WAIT_IMM((S - 2) * LC);
            else FRCNN_WAIT_IMM(0);
            after_epilogue = false;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (to_issue > 0) {
                issue_slice(slot == 0 ? S - 1 : slot - 1);
                --to_issue;
                bn_st = 0;
            }
            if (BNIN) bnin_slice(slot, nk - left, m0);
            mfma_slice(slot);
            slot = slot + 1 == S ? 0 : slot + 1;
            --left;
        }
        if (!MULTI) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // every wave is done with the ring: the staging tile may overwrite it
        }

        FRCNN_STAMP(2);
        if (FIX) {
            // fragment-major partial tile: store (i, j, e) of all 512 threads is one run of 512 floats.  Agent-scope relaxed
            // atomics = write-through (sc1) stores / L1-bypassing (sc1) loads: no L2 write-back or invalidate, only the wait for
            // this wave's stores before the workgroup announces itself.  This is the hand-off form MI355X_MICROARCH.md lists as
            // measured-valid on gfx950 (every payload store sc1 and drained by its wave's vmcnt(0), a workgroup barrier, ONE lane's
            // agent-scope counter add; the last arriver -- told by the value its add returned -- loads every payload byte sc1 behind
            // a workgroup barrier): it relies on that lowering, not on the HIP memory model's release / acquire, which would cost a
            // write-back + invalidate (~3.5 us) per tile against the 7 us the split saves.  The counters are zeroed by the caller's
            // per-step fill (frcnn_conv2d_workspace_counter_bytes), so an aborted launch cannot poison the next one.
            __shared__ unsigned fix_order;
            float* mine = p.fix_partial + ((size_t)fix_pair * 2 + fix_split) * (BM * BN);
            const float* other = p.fix_partial + ((size_t)fix_pair * 2 + (fix_split ^ 1)) * (BM * BN);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __hip_atomic_store(mine + ((i * NI + j) * 4 + e) * T + tid, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) fix_order = __hip_atomic_fetch_add(p.fix_counter + fix_pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (fix_order == 0u) return;             // first to arrive: the partner completes the tile
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j][e] += __hip_atomic_load(other + ((i * NI + j) * 4 + e) * T + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) __hip_atomic_store(p.fix_counter + fix_pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        }
        // -------------------------------------------------------------- epilogue of tile t
        // lane holds, for fragment (i,j): pixel = wm*WTM + i*16 + (lane&15); couts = wn*WTN + j*16 + (lane>>4)*4 + 0..3
        if (F32) {
            // fp32 output (RPN heads) / split-K partial sums (Dense heads): the tile is staged in the drained ring and leaves
            // with whole rows -- 64 lanes x 4 B = 256 contiguous bytes per store / float-atomic wave instruction
            constexpr int ROW32 = BN * 4 + 16;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * WTM + i * 16 + frow;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int cl = wn * WTN + j * 16 + fchunk * 4;
                    f32x4 v = acc[i][j];
                    if (blockIdx.y == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[j][e];
                    }
                    if (flags & FRCNN_CONV_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *reinterpret_cast<f32x4*>(smem + r * ROW32 + cl * 4) = v;
                }
            }
            __syncthreads();
            float* y = reinterpret_cast<float*>(p.y);
            for (int idx = tid; idx < BM * BN; idx += T) {
                const int r = idx / BN, c = idx - r * BN;
                const int m = m0 + r;
                if (m < p.M && n0 + c < p.Cout) {
                    const float v = *reinterpret_cast<const float*>(smem + r * ROW32 + c * 4);
                    if (flags & FRCNN_CONV_SPLITK_ATOMIC) atomicAdd(y + (long long)m * p.Cout + n0 + c, v);
                    else y[(long long)m * p.Cout + n0 + c] = v;
                }
            }
            continue;
        }
        const bool tail = m0 + BM > p.M;
        const unsigned tile_off = (unsigned)((m0 * p.Cout + n0) * 2);
        u32x4 resv[ST_IT];
        if (p.direct_out && (flags & FRCNN_CONV_ADD_RES)) {  // residual rows of this tile: in flight under the convert phase
#pragma unroll
            for (int it = 0; it < ST_IT; ++it) {
                const int r = lrow_o + it * (T / C8);
                const unsigned vo = (col_ok && (!tail || m0 + r < p.M)) ? vo_lane + tile_off + it * pass_pitch : kOob;
                resv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, vo, 0, 0);
                if (p.res_mask) {                // masked residual: zero the elements whose mask bit is clear (no gpre tensor)
                    const unsigned mk = vo != kOob ? p.res_mask[vo >> 4] : 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        resv[it][q] &= (((mk >> (2 * q)) & 1u) ? 0x0000FFFFu : 0u) | (((mk >> (2 * q + 1)) & 1u) ? 0xFFFF0000u : 0u);
                }
            }
        }
        u32x4 redz[ST_IT];
        unsigned redm[ST_IT];
        if (RED && p.direct_out) {               // consumer layer's z rows and mask bytes: also in flight under the convert phase
#pragma unroll
            for (int it = 0; it < ST_IT; ++it) {
                const int r = lrow_o + it * (T / C8);
                const bool ok = col_ok && (!tail || m0 + r < p.M);
                const unsigned vo = ok ? vo_lane + tile_off + it * pass_pitch : kOob;
                redz[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rz, vo, 0, 0);
                redm[it] = (p.red_mask && ok) ? p.red_mask[vo >> 4] : 0xFFu;      // byte index = (row * Cout + c) / 8
            }
        }
        auto convert_tile = [&](auto tail_c) {
            constexpr bool TL = decltype(tail_c)::value;
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
                        if (F8) {
                            const f32x2 dqv = *reinterpret_cast<const f32x2*>(s_dq + cl + 2 * h), bvv = *reinterpret_cast<const f32x2*>(s_dq + BN + cl + 2 * h);
                            v[0] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h] * dqv[0] + bvv[0], lo, __builtin_inff());
                            v[1] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h + 1] * dqv[1] + bvv[1], lo, __builtin_inff());
                        } else {
                            v[0] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h] + bv[j][2 * h], lo, __builtin_inff());
                            v[1] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h + 1] + bv[j][2 * h + 1], lo, __builtin_inff());
                        }
                        const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32 (RNE)
                        pk[h] = bits;
                        if (STATS) {             // sums of the ROUNDED outputs (what the next layer reads)
                            float q0 = __uint_as_float(bits << 16), q1 = __uint_as_float(bits & 0xFFFF0000u);
                            if (TL) { q0 = row_ok ? q0 : 0.f; q1 = row_ok ? q1 : 0.f; }
                            ssum[j][2 * h] += q0;
                            ssq[j][2 * h] += q0 * q0;
                            ssum[j][2 * h + 1] += q1;
                            ssq[j][2 * h + 1] += q1 * q1;
                        }
                    }
                    // MULTI: the next tile's DMA is in flight and hipcc would order a DS write it emits itself behind ALL
                    // pending LDS-DMA (vmcnt(0)); the asm form is invisible to that pass
                    if (MULTI) lds_write_b64(stage_a + r * ROWB + cl * 2, pk);
                    else *reinterpret_cast<u32x2*>(stage + r * ROWB + cl * 2) = pk;
                }
            }
        };
        if (STATS && tail) convert_tile(std::true_type{});
        else convert_tile(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // staging tile complete (a raw barrier does not drain the DMA ring)

        if (p.direct_out) {
#pragma unroll
            for (int it = 0; it < ST_IT; ++it) {
                const int r = lrow_o + it * (T / C8);
                // the offset travels in the VGPR, not in soffset: with an SGPR soffset hipcc omits the wait state between a
                // 16-byte buffer store and a VALU overwrite of its data registers, and gfx950 does need it
                const unsigned vo = (col_ok && (!tail || m0 + r < p.M)) ? vo_lane + tile_off + it * pass_pitch : kOob;
                u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * ROWB + lc8 * 16);
                if (flags & FRCNN_CONV_ADD_RES) {
                    float a[8], b[8];
                    unpack8(v, a);
                    unpack8(resv[it], b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += b[e];
                    v = pack8(a);
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, vo, 0, 0);
                if (RED) {                       // rows / columns outside the tensor: z loads returned zeros -> g*z = 0; mask g too
                    float g[8], zz[8];
                    unpack8(v, g);
                    unpack8(redz[it], zz);
                    const bool ok = vo != kOob;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gm = (ok && ((redm[it] >> e) & 1u)) ? g[e] : 0.f;
                        rsg[e] += gm;
                        rsgz[e] += gm * zz[e];
                    }
                }
            }
        } else {
            // strided scatter (data gradient of a stride-2 1x1 convolution): per-row address computation
            bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
            for (int idx = tid; idx < BM * C8; idx += T) {
                const int r = idx / C8, c8 = idx - r * C8;
                const int m = m0 + r, c = n0 + c8 * 8;
                if (m >= p.M || c >= p.Cout) continue;
                u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * ROWB + c8 * 16);
                const long long off = out_row_of(p, m) * p.Cout + c;
                if (flags & FRCNN_CONV_ADD_RES) {
                    u32x4 rv = *reinterpret_cast<const u32x4*>(p.res + off);
                    if (p.res_mask) {
                        const unsigned mk = p.res_mask[off >> 3];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            rv[q] &= (((mk >> (2 * q)) & 1u) ? 0x0000FFFFu : 0u) | (((mk >> (2 * q + 1)) & 1u) ? 0xFFFF0000u : 0u);
                    }
                    float a[8], b[8];
                    unpack8(v, a);
                    unpack8(rv, b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += b[e];
                    v = pack8(a);
                }
                *reinterpret_cast<u32x4*>(y + off) = v;
                if (RED) {
                    float g[8], zz[8];
                    unpack8(v, g);
                    unpack8(*reinterpret_cast<const u32x4*>(p.red_z + off), zz);
                    const unsigned mk = p.red_mask ? p.red_mask[off >> 3] : 0xFFu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gm = ((mk >> e) & 1u) ? g[e] : 0.f;
                        rsg[e] += gm;
                        rsgz[e] += gm * zz[e];
                    }
                }
            }
        }
        FRCNN_STAMP(3);
        // (MULTI) the next tile's convert phase writes the staging tile only after >= 1 barrier of its K loop
    }
#undef FRCNN_WAIT_IMM

    // ---- flushes of the per-workgroup sums.  Nothing here may wait for the epilogue's global stores (a __syncthreads() or a
    // DS operation the compiler orders behind vmcnt(0) costs the store round trip, measured 1.3 - 3 us of a 9 - 12 us workgroup
    // lifetime with only two workgroups per CU): cross-lane shuffles, asm LDS writes, an lgkmcnt-only wait, a raw barrier.
    // The scratch lies in the drained ring, beside the staging tile (the last tile's store loop may still be reading it).
    constexpr int FLUSH_OFF = MULTI ? 0 : STG;
    static_assert(FLUSH_OFF + NW * 2 * BN * 4 <= BIG + 2 * BN * 4, "flush scratch must fit the idle ring");
    float* fl = reinterpret_cast<float*>(smem + FLUSH_OFF);      // [NW waves][2 sums][BN channels]
    const unsigned fl_a = lds_addr(fl);
    if (RED) {
        // the T / C8 row lanes that share a channel vector: first inside the wave (lanes with equal lane % C8), then the eight
        // waves through LDS; 2 x BN coalesced float atomics into the consumer layer's slot partial sums
#pragma unroll
        for (int sh = C8; sh < 64; sh <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                rsg[e] += __shfl_xor(rsg[e], sh);
                rsgz[e] += __shfl_xor(rsgz[e], sh);
            }
        }
        if (lane < C8) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                lds_write_b64(fl_a + ((wave * 2 + 0) * BN + lane * 8 + e) * 4, u32x2{__float_as_uint(rsg[e]), __float_as_uint(rsg[e + 1])});
                lds_write_b64(fl_a + ((wave * 2 + 1) * BN + lane * 8 + e) * 4, u32x2{__float_as_uint(rsgz[e]), __float_as_uint(rsgz[e + 1])});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid - st * BN;
            const int c = n0 + cl;
            float sg = 0.f, sgz = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {                       // fixed order: the workgroup's partial is reproducible
                sg += fl[(w * 2 + 0) * BN + cl];
                sgz += fl[(w * 2 + 1) * BN + cl];
            }
            if (c < p.Cout) {
                const float v = st == 0 ? sg : p.red_invstd[c] * (sgz - p.red_mean[c] * sg);
                atomicAdd(p.red_part + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * p.Cout + c, v);
            }
        }
    }
    if (STATS) {
        // the run's tiles share one n-tile: 16-lane butterflies, the two row halves (wm) meet in LDS, then one f64 atomic per
        // channel and statistic into one of FRCNN_STAT_SLOTS pre-zeroed slots (consecutive lanes: consecutive channels).
        // The workgroup's own partial is an fp32 sum in a fixed order; the cross-workgroup sum is accumulated in f64, whose
        // rounding (1e-16) makes the arrival order of the atomics invisible in the fp32 statistics derived from it
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
                ssum[j][e] = a;
                ssq[j][e] = b;
            }
        if (frow == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * WTN + j * 16 + fchunk * 4;
                lds_write_b64(fl_a + ((wm * 2 + 0) * BN + cl) * 4, u32x2{__float_as_uint(ssum[j][0]), __float_as_uint(ssum[j][1])});
                lds_write_b64(fl_a + ((wm * 2 + 0) * BN + cl + 2) * 4, u32x2{__float_as_uint(ssum[j][2]), __float_as_uint(ssum[j][3])});
                lds_write_b64(fl_a + ((wm * 2 + 1) * BN + cl) * 4, u32x2{__float_as_uint(ssq[j][0]), __float_as_uint(ssq[j][1])});
                lds_write_b64(fl_a + ((wm * 2 + 1) * BN + cl + 2) * 4, u32x2{__float_as_uint(ssq[j][2]), __float_as_uint(ssq[j][3])});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid - st * BN;
            const float v = fl[(0 * 2 + st) * BN + cl] + fl[(1 * 2 + st) * BN + cl];          // wm = 0, 1
            if (n0 + cl < p.Cout)
                atomicAdd(p.stats + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * p.Cout + n0 + cl, (double)v);
        }
    }
    FRCNN_STAMP(4);
#ifdef FRCNN_STAMPS
    if (p.dbg && tid == 0) {
#pragma unroll
        for (int j = 0; j < 10; ++j) p.dbg[(size_t)blockIdx.x * 24 + 8 + j] = ks[j];
    }
#endif
#undef FRCNN_STAMP
#undef FRCNN_KSTAMP
#endif
}

// ---------------------------------------------------------------------------------------------------- the ResNet stem, on its own
// conv1 of the backbone (7x7 / 2 on the 4-channel padded image, K = 7 rows x (8 pixels x 4 channels), 64 output channels;
// models/feature_extractor.py:8) through the general kernel is INGEST-bound: every 128-pixel tile re-fetches its 7 x 8 KB of
// overlapping A rows (adjacent output pixels share 6 of their 8 input pixels) and the whole 28 KB filter bank -- 100 KB through the CU's
// load path per 16 KB of output, 57 us for a layer whose HBM roofline is 12.  Here a WAVE owns 16 consecutive output pixels x all 64
// channels and needs no LDS for its operands at all:
//   * the filter bank lives in REGISTERS for the whole launch (7 x 4 fragments of 16 bytes per lane = 112 VGPRs: lane (c, q) holds
//     K chunk q of output channel 16 j + c of every tap row kh);
//   * the pixel operand of tap row kh is ONE 16-byte load per lane straight from the image (lane (p, q): pixels 2p + 2q, 2p + 2q + 1 of
//     input row 2 oy + kh -- 16-byte aligned because the stride is 2 pixels of 4 bf16 channels); neighbouring lanes overlap, which
//     the vector L1 absorbs; the next unit's 7 loads are in flight under this unit's 28 MFMAs;
//   * the epilogue (bias, bf16 rounding, the BatchNorm statistics of the rounded values) stages the wave's 16 x 64 outputs in 2 KB of
//     wave-private LDS and stores them as one contiguous 2 KB run.
// No barrier until the statistics flush at the very end.
template <bool STATS>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(const ConvParams p, const int units_per_row, const int total_units) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PITCH = 144;                   // staging row pitch (128 B of channels + 16: the 16 pixel rows spread over the banks)
    __shared__ __attribute__((aligned(16))) unsigned char stage_all[4][16 * PITCH];
    __shared__ float wsum[4][2][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pl = lane & 15, q = lane >> 4;
    unsigned char* stage = stage_all[wave];
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    bf16x8 wf[7][4];                             // the filter bank: [tap row][16-channel block], K chunk q of channel 16 j + pl
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[kh][j] = *reinterpret_cast<const bf16x8*>(p.w + ((16 * j + pl) * 7 + kh) * 32 + q * 8);
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.flags & FRCNN_CONV_BIAS) b = *reinterpret_cast<const f32x4*>(p.bias + 16 * j + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = b[e];
    }
    float ssum[4][4], ssq[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[j][e] = ssq[j][e] = 0.f;

    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const unsigned row_pitch = (unsigned)p.Wi * 8u;          // bytes per input row (4 bf16 channels per pixel)
    auto issue = [&](const int u, u32x4 (&a)[7]) {
        const int row = u / units_per_row;                    // n * Ho + oy
        const int ox0 = (u - row * units_per_row) * 16;
        const int n = row / p.Ho, oy = row - n * p.Ho;
        const unsigned base = (unsigned)(((n * p.Hi + 2 * oy) * p.Wi + 2 * (ox0 + pl) + 2 * q) * 8);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) a[kh] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, base + (unsigned)kh * row_pitch, 0, 0);
    };
    u32x4 cur[7], nxt[7];
    int u = gw;
    if (u < total_units) issue(u, cur);
    for (; u < total_units; u += nw) {
        if (u + nw < total_units) issue(u + nw, nxt);
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], __builtin_bit_cast(bf16x8, cur[kh]), acc[j], 0, 0, 0);
        // ---- epilogue of the unit: lane (pl, q) holds channels 16 j + 4 q + {0..3} of pixel ox0 + pl
        const int row = u / units_per_row;
        const int ox0 = (u - row * units_per_row) * 16;
        const bool valid = ox0 + pl < p.Wo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x2 pk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 v;
                v[0] = acc[j][2 * h] + bv[j][2 * h];
                v[1] = acc[j][2 * h + 1] + bv[j][2 * h + 1];
                const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32 (RNE)
                pk[h] = bits;
                if (STATS) {                     // sums of the ROUNDED outputs (what the BatchNorm kernel reads)
                    const float q0 = valid ? __uint_as_float(bits << 16) : 0.f, q1 = valid ? __uint_as_float(bits & 0xFFFF0000u) : 0.f;
                    ssum[j][2 * h] += q0;
                    ssq[j][2 * h] += q0 * q0;
                    ssum[j][2 * h + 1] += q1;
                    ssq[j][2 * h + 1] += q1 * q1;
                }
            }
            *reinterpret_cast<u32x2*>(stage + pl * PITCH + (16 * j + 4 * q) * 2) = pk;
        }
        // (same wave writes and reads: LDS operations of a wave complete in order, no barrier)
        bf16_t* out = reinterpret_cast<bf16_t*>(p.y) + ((long long)row * p.Wo + ox0) * 64;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int L = lane + 64 * half, pix = L >> 3, c = L & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stage + pix * PITCH + c * 16);
            if (ox0 + pix < p.Wo) *reinterpret_cast<u32x4*>(out + pix * 64 + c * 8) = v;
        }
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) cur[kh] = nxt[kh];
    }
    if (STATS) {
        // the 16 pixel lanes of a wave meet by butterflies, the four waves in LDS, then one f64 atomic per (statistic, channel) and
        // workgroup into one of the pre-zeroed slots (as conv_tile_kernel's statistics flush)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = ssum[j][e], b = ssq[j][e];
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) {
                    a += __shfl_xor(a, sh);
                    b += __shfl_xor(b, sh);
                }
                if (pl == 0) {
                    wsum[wave][0][16 * j + 4 * q + e] = a;
                    wsum[wave][1][16 * j + 4 * q + e] = b;
                }
            }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int st = threadIdx.x >> 6, c = threadIdx.x & 63;
            const float v = (wsum[0][st][c] + wsum[1][st][c]) + (wsum[2][st][c] + wsum[3][st][c]);
            atomicAdd(p.stats + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * 64 + c, (double)v);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------- short-K 1x1 layers, streamed
// The expansion layers of conv2 / conv3 (1x1, stride 1, 64 -> 256 and 128 -> 512; also conv2's 64 -> 64) have ONE or TWO K slices: in
// the tile kernel they are all prologue and epilogue (19-22 us against an HBM roofline of 9.4 at 64 -> 256: input once, output once).
// The stem kernel's scheme fits them as well: a wave owns 16 consecutive pixels x NC output channels, its slice of the filter bank
// (NC x K bf16 = 16 KB) stays in REGISTERS for the whole launch, the pixel operand is K / 32 16-byte loads per lane straight from the
// activation tensor (a pixel row is K * 2 contiguous bytes: lanes (p, q) and K step kk read bytes [64 kk + 16 q, +16) of row p --
// 16 rows x 128 / 256 contiguous bytes per unit), the next unit's loads fly under this unit's MFMAs, and the epilogue (bias, bf16
// rounding, BatchNorm statistics of the rounded values) goes through wave-private LDS to contiguous 16-byte stores.  The Cout / NC
// channel parts of a pixel group are taken by neighbouring waves (same A rows: vector-L1 hits).
template <int K, int NC, bool STATS>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const ConvParams p, const int parts, const int total_units) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KK = K / 32, NJ = NC / 16;
    constexpr int PITCH = NC * 2 + 16;           // staging row pitch
    __shared__ __attribute__((aligned(16))) unsigned char stage_all[4][16 * PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pl = lane & 15, q = lane >> 4;
    unsigned char* stage = stage_all[wave];
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    // a WORKGROUP keeps one channel part for the whole launch (its four waves meet in LDS for the statistics: one atomic per channel
    // and workgroup); workgroup b takes part b % parts, its waves the pixel groups (b / parts) * 4 + wave, + stride
    const int part = blockIdx.x % parts;         // (the host makes gridDim.x a multiple of parts)
    const int gw = (blockIdx.x / parts) * 4 + wave, nw = (gridDim.x / parts) * 4;
    const int n0 = part * NC;
    bf16x8 wf[KK][NJ];                           // K chunk q of channel n0 + 16 j + pl, K step kk
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[kk][j] = *reinterpret_cast<const bf16x8*>(p.w + (long long)(n0 + 16 * j + pl) * K + kk * 32 + q * 8);
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.flags & FRCNN_CONV_BIAS) b = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * j + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = b[e];
    }
    float ssum[STATS ? NJ : 1][4], ssq[STATS ? NJ : 1][4];
#pragma unroll
    for (int j = 0; j < (STATS ? NJ : 1); ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[j][e] = ssq[j][e] = 0.f;
    const float lo = (p.flags & FRCNN_CONV_RELU) ? 0.f : -__builtin_inff();

    auto issue = [&](const int grp, u32x4 (&a)[KK]) {
        const unsigned base = (unsigned)(grp * 16 + pl) * (unsigned)(p.in_pix_stride * 2) + (unsigned)q * 16u;      // (rows beyond M: beyond the descriptor, zeros)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (grp * 16 + pl) < p.M ? base + kk * 64u : kOob, 0, 0);
    };
    const int groups = total_units;              // 16-pixel groups
    const int gstride = nw;
    u32x4 cur[KK], nxt[KK];
    int grp = gw;
    if (grp < groups) issue(grp, cur);
    for (; grp < groups; grp += gstride) {
        if (grp + gstride < groups) issue(grp + gstride, nxt);
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], __builtin_bit_cast(bf16x8, cur[kk]), acc[j], 0, 0, 0);
        const int m0 = grp * 16;
        const bool valid = m0 + pl < p.M;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u32x2 pk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 v;
                v[0] = __builtin_amdgcn_fmed3f(acc[j][2 * h] + bv[j][2 * h], lo, __builtin_inff());
                v[1] = __builtin_amdgcn_fmed3f(acc[j][2 * h + 1] + bv[j][2 * h + 1], lo, __builtin_inff());
                const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                pk[h] = bits;
                if (STATS) {
                    const float q0 = valid ? __uint_as_float(bits << 16) : 0.f, q1 = valid ? __uint_as_float(bits & 0xFFFF0000u) : 0.f;
                    ssum[STATS ? j : 0][2 * h] += q0;
                    ssq[STATS ? j : 0][2 * h] += q0 * q0;
                    ssum[STATS ? j : 0][2 * h + 1] += q1;
                    ssq[STATS ? j : 0][2 * h + 1] += q1 * q1;
                }
            }
            *reinterpret_cast<u32x2*>(stage + pl * PITCH + (16 * j + 4 * q) * 2) = pk;
        }
        // 16 rows x NC * 2 bytes staged: NC / 8 16-byte vectors per row, NC * 2 / 64 per lane
        bf16_t* out = reinterpret_cast<bf16_t*>(p.y) + (long long)m0 * p.Cout + n0;
        constexpr int VPR = NC / 8;              // vectors per row
#pragma unroll
        for (int it = 0; it < (16 * VPR) / 64; ++it) {
            const int L = lane + 64 * it, pix = L / VPR, c = L - pix * VPR;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stage + pix * PITCH + c * 16);
            if (m0 + pix < p.M) *reinterpret_cast<u32x4*>(out + (long long)pix * p.Cout + c * 8) = v;
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) cur[kk] = nxt[kk];
    }
    if (STATS) {
        // the 16 pixel lanes meet by butterflies, the four waves in LDS (the staging area is free now), then one f64 atomic per
        // (statistic, channel) and workgroup
        __syncthreads();
        float* wsum = reinterpret_cast<float*>(&stage_all[0][0]);      // [4 waves][2][NC]
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = ssum[STATS ? j : 0][e], b = ssq[STATS ? j : 0][e];
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) {
                    a += __shfl_xor(a, sh);
                    b += __shfl_xor(b, sh);
                }
                if (pl == 0) {
                    wsum[(wave * 2 + 0) * NC + 16 * j + 4 * q + e] = a;
                    wsum[(wave * 2 + 1) * NC + 16 * j + 4 * q + e] = b;
                }
            }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * NC; t += 256) {
            const int st = t / NC, c = t - st * NC;
            const float v = (wsum[(0 * 2 + st) * NC + c] + wsum[(1 * 2 + st) * NC + c]) + (wsum[(2 * 2 + st) * NC + c] + wsum[(3 * 2 + st) * NC + c]);
            atomicAdd(p.stats + ((long long)((blockIdx.x / parts) & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * p.Cout + n0 + c, (double)v);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------- 3x3 / stride 1 / pad 1, patch-resident
// The tile kernel above walks a 3x3 filter as nine GEMM slices per 64-channel chunk and fetches the A rows of every tap again: a 128-pixel
// tile takes in 9 x 16 KB of activations per chunk, although the nine taps read the SAME pixels shifted (kw sharing brings that to 3 x).
// On the M = 7,488 layers (conv4, the RPN) that is most of what a CU's load path carries (~37 GB/s per CU whatever
// the schedule), and the load path, not the matrix pipe, is what bounds them.  Here the workgroup owns a SPATIAL tile of 8 x 16 output
// pixels x 64 output channels and keeps the 10 x 18 input patch of one 64-channel chunk in LDS (180 pixel rows of 128 B, double
// buffered): all nine taps read it at shifted rows -- tap (kh, kw) of output pixel (ty, tx) is patch row (ty + kh) * 18 + tx + kw -- and
// pixels outside the image are zero-filled once by the buffer range check, so the K loop carries no validity masks.  Per chunk the CU
// takes in 23 KB of activations instead of 144 (kw sharing: 50), and the filter slab (9 x 8 KB) becomes most of the traffic.
//   * K order: (64-channel chunk, kh, kw).  One STEP = one (chunk, kh) = three taps = 192 K values = 24 KB of weights in one ring slot;
//     a barrier per step (a third of the tile kernel's), 24 MFMAs per wave between barriers;
//   * DMA per wave and step: 3 weight pieces of step k + SB - 1, and at kh == 0 the 3 patch pieces of the NEXT chunk (they are older than
//     the weights of the step that first reads them, so one counted vmcnt wait covers both);
//   * 8 waves as 4 (tile row pairs) x 2 (32 output channels): 2 + 2 fragment reads per 4 MFMAs;
//   * epilogue as the tile kernel's (bias / ReLU, bf16 rounding, BatchNorm statistics of the rounded values or the fused
//     BatchNorm-backward reduce of the consumer layer), rows of the staging tile mapped back to (oy, ox).
//   * LW = 4: four more waves per workgroup that do nothing but issue the DMA (each the pieces of two of the eight "virtual" loader waves)
//     and wait for it; the eight MFMA waves never execute an LDS-DMA instruction (it stalls its wave while the CU's load path is backed up)
//   * BNT = 128: 128 output channels per workgroup (wave tiles 32 pixels x 64 channels: 2 + 4 fragment reads per 8 MFMAs), 48 KB of weights
//     per step in a 2-slot ring -- for layers with 128 output channels and few chunks (conv3), where it halves the workgroups to one round
//   * BNIN (frcnn_conv2d_fprop_bnin, loader-wave forms only): the input is the RAW output z of the previous convolution and this kernel
//     applies that layer's training-mode BatchNorm + ReLU itself, as conv3x3_wres_kernel<., BNIN> does for the one-chunk layers: every MFMA
//     thread below Cin derives scale / shift of one input channel from the f64 statistics slots (the additions of bn_train_apply_kernel in
//     its order: same bits) into LDS behind the weight ring; when a chunk's patch has landed (the step with kh == 0) the eight MFMA waves
//     transform it in place -- pixels outside the image stay zero: the padding applies AFTER the BatchNorm -- while the loader waves keep
//     issuing, one more barrier per chunk; the workgroup whose channel part equals chunk mod parts writes the chunk's activation and ReLU
//     bit mask for the backward pass from its patch interior.  One launch and one read of z less than bn_train_apply + this kernel.
template <int SB, int SMODE, int OCCW, int LW, int BNT, bool BNIN = false>
__global__ __launch_bounds__(512 + 64 * LW, OCCW) void conv3x3_patch_kernel(const ConvParams p, const int tiles_x, const int tiles_y) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool STATS = SMODE == 1, RED = SMODE == 2;
    static_assert(!BNIN || (LW == 4 && SMODE != 2), "BatchNorm on the input side: forward convolutions on the loader-wave forms");
    constexpr bool INTERLEAVE = false;          // a step's DMA pieces spread between its MFMA groups: measured slower (c4 3x3 17.4 -> 18.7 us)
    constexpr int NW = 8, T = 512, BM = 128, BN = BNT, TW = 16, TH = 8, PW = TW + 2;
    constexpr int MI = 2, NI = BN / 32, BPV = BN / 64;            // BPV: weight pieces per (virtual) wave and tap
    constexpr int TAPB = BN * 128;                               // bytes of one tap's weights in a ring slot
    constexpr int A_BUF = 24 * 1024, B_STEP = 3 * TAPB, B_BASE = 2 * A_BUF, PB = SB - 1;
    static_assert(BN == 64 || BN == 128, "64 or 128 output channels per workgroup");
    constexpr int ROWB = BN * 2 + 16, C8 = BN / 8, ST_IT = (BM * C8) / T, STG = BM * ROWB;
    static_assert(SB >= 2 && SB <= 4, "weight ring: 2..4 steps");
    static_assert(STG + NW * 2 * BN * 4 <= B_BASE + SB * B_STEP, "staging tile + flush scratch alias the drained buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int flags = p.flags;

    // workgroup -> (spatial tile, 64-channel part), channel part fastest; every XCD takes a contiguous chunk of that list, so the
    // workgroups that share an L2 read the same patches and neighbouring ones
    int unit;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3;
        const int q = nb >> 3, r = nb & 7;
        unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int tmi = unit / p.tiles_n, tn = unit - tmi * p.tiles_n;
    const int per_img = tiles_x * tiles_y;
    const int img = tmi / per_img;
    const int trem = tmi - img * per_img;
    const int tyt = trem / tiles_x;
    const int oy0 = tyt * TH, ox0 = (trem - tyt * tiles_x) * TW, n0 = tn * BN;

    float bv[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (flags & FRCNN_CONV_BIAS) b = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * (BN / 2) + j * 16 + fchunk * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = b[e];
    }

    // BNIN: scale / shift of the Cin input channels ([Cin] + [Cin] floats behind the weight ring), one channel per MFMA thread: the four slot
    // slices of bn_train_apply_kernel's prologue (slot_sums) and their sum, in that order; workgroup 0 publishes mean / invstd and updates
    // the moving statistics.  Published to the other waves by the first step's barrier.
    float* s_scale = reinterpret_cast<float*>(smem + B_BASE + SB * B_STEP);
    float* s_shift = s_scale + p.Cin;
    if (BNIN && tid < p.Cin) {
        const int c = tid;
        double a[4][FRCNN_STAT_SLOTS / 4], b[4][FRCNN_STAT_SLOTS / 4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) {
                a[sl][k] = p.bnin_part[((long long)(sl + 4 * k) * 2) * p.Cin + c];
                b[sl][k] = p.bnin_part[((long long)(sl + 4 * k) * 2 + 1) * p.Cin + c];
            }
        double ps[4], pq[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) { s0 += a[sl][k]; s1 += b[sl][k]; }
            ps[sl] = s0;
            pq[sl] = s1;
        }
        const double sum = ps[0] + ps[1] + ps[2] + ps[3];
        const double ssq_ = pq[0] + pq[1] + pq[2] + pq[3];
        const double mean = sum * p.bnin_inv_count;
        double var = ssq_ * p.bnin_inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)p.bnin_eps));
        const float sc = p.bnin_gamma[c] * invstd;
        s_scale[c] = sc;
        s_shift[c] = p.bnin_beta[c] - (float)mean * sc;
        if (blockIdx.x == 0) {
            p.bnin_mean[c] = (float)mean;
            p.bnin_invstd[c] = invstd;
            p.bnin_mm[c] = p.bnin_mm[c] * p.bnin_momentum + (float)mean * (1.f - p.bnin_momentum);
            p.bnin_mv[c] = p.bnin_mv[c] * p.bnin_momentum + (float)(var * p.bnin_unbias) * (1.f - p.bnin_momentum);
        }
    }
    // BatchNorm + ReLU of the landed patch of chunk s, in place: 192 rows x 8 sixteen-byte slots, three per MFMA thread (slot sl of row q
    // holds channels 8 (sl ^ (q & 7)) .. of the chunk).  Rows outside the image (and beyond the patch) become zeros.
    auto bnin_transform = [&](const int s) {
        unsigned char* buf = smem + (s & 1) * A_BUF;
        const unsigned pbase = lds_addr(buf);
        const bool mine = (s % p.tiles_n) == tn;                     // this workgroup writes chunk s of its tile's activation
        const float* csc = s_scale + s * 64;
        const float* csh = s_shift + s * 64;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + T * k;
            const int q = idx >> 3, slot = idx & 7, c8 = slot ^ (q & 7);
            const int py = q / PW, px = q - py * PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool valid = q < (TH + 2) * PW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(buf + q * 128 + slot * 16);
            const f32x4 sc0 = *reinterpret_cast<const f32x4*>(csc + c8 * 8), sc1 = *reinterpret_cast<const f32x4*>(csc + c8 * 8 + 4);
            const f32x4 sh0 = *reinterpret_cast<const f32x4*>(csh + c8 * 8), sh1 = *reinterpret_cast<const f32x4*>(csh + c8 * 8 + 4);
            float x[8];
            unpack8(raw, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = fmaxf(x[e] * sc0[e] + sh0[e], 0.f);
                x[4 + e] = fmaxf(x[4 + e] * sc1[e] + sh1[e], 0.f);
            }
            u32x4 pk = pack8(x);
            if (!valid) pk = u32x4{0u, 0u, 0u, 0u};
            asm volatile("ds_write_b128 %0, %1" ::"v"(pbase + (unsigned)(q * 128 + slot * 16)), "v"(pk) : "memory");
            if (mine) {                          // (workgroup-uniform) this tile's own pixels: the activation and its mask
                const u32x2 mk = pool_mask8(relu_bits8(pk), c8);      // the row's eight mask bytes of this chunk: one 8-byte store by its first lane
                if (valid && py >= 1 && py <= TH && px >= 1 && px <= TW) {
                    const long long pix = ((long long)img * p.Hi + iy) * p.Wi + ix;
                    // (the activation and its ReLU mask are the BACKWARD pass's operands -- this kernel has consumed them already: non-temporal
                    // stores; same-box A/B 3.798 -> 3.781, 3.802 -> 3.799)
                    __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(p.bnin_act + pix * p.Cin + s * 64 + c8 * 8));
                    if (slot == 0) __builtin_nontemporal_store(mk, reinterpret_cast<u32x2*>(p.bnin_mask + pix * (p.Cin >> 3) + s * 8));
                }
            }
        }
    };

    // ------------------------------------------------------------------ loader state
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const unsigned dma_swz = (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);      // piece rows start at multiples of 8: row & 7 == lane >> 3
    // a loader wave (LW: waves 8 .. 11) issues the pieces of the virtual waves vw[0] = wave - 8 and vw[1] = wave - 4; else vw[0] = wave
    constexpr int NV = LW ? 2 : 1;
    const bool is_loader = LW && wave >= NW;
    const int vw0 = is_loader ? wave - NW : (wave & 7);
    unsigned a_voff[NV][3], b_voff[NV][BPV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int vw = vw0 + 4 * v;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int q = (vw + NW * t) * 8 + (lane >> 3);                       // patch row = patch pixel
            const int py = q / PW, px = q - py * PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = q < (TH + 2) * PW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            a_voff[v][t] = ok ? (unsigned)((img * p.Hi + iy) * p.Wi + ix) * (unsigned)(p.in_pix_stride * 2) + dma_swz : kOob;
        }
#pragma unroll
        for (int u = 0; u < BPV; ++u) b_voff[v][u] = (unsigned)(n0 + (vw + NW * u) * 8 + (lane >> 3)) * (unsigned)(p.Ktot * 2) + dma_swz;
    }
    const int nsub = p.Cin >> 6, K = nsub * 3;
    int ld_k = 0, ld_s = 0, ld_kh = 0, ld_slot = 0;                               // next weight step to issue
    // one weight piece (tap kw of the loader's step) / one patch piece (t) per call: the K loop spreads a step's pieces between its MFMA
    // groups -- an LDS-DMA instruction issued into a backed-up load path stalls its wave (75 - 100 ns per piece when the eight waves issue
    // theirs back to back), one per ~130 cycles of MFMA does not
    auto issue_b_piece = [&](const int kw) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int u = 0; u < BPV; ++u) {
                unsigned char* dst = smem + B_BASE + ld_slot * B_STEP + (vw0 + 4 * v + NW * u) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(dst + kw * TAPB), 16, b_voff[v][u], (unsigned)((((ld_kh * 3 + kw) * p.Cin) + ld_s * 64) * 2), 0, 0);
            }
    };
    auto advance_b = [&]() {
        ++ld_k;
        ld_slot = ld_slot + 1 == SB ? 0 : ld_slot + 1;
        if (++ld_kh == 3) { ld_kh = 0; ++ld_s; }
    };
    auto issue_b = [&]() {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) issue_b_piece(kw);
        advance_b();
    };
    auto issue_a_piece = [&](const int s, const int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned char* dst = smem + (s & 1) * A_BUF + (vw0 + 4 * v) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(dst + t * NW * 1024), 16, a_voff[v][t], (unsigned)(s * 128), 0, 0);
        }
    };
    auto issue_a = [&](const int s) {
#pragma unroll
        for (int t = 0; t < 3; ++t) issue_a_piece(s, t);
    };
    // pieces a wave issues in consumer step j: the next chunk's patch at kh == 0, the weights of step j + PB
    auto issued_in = [&](const int j) { return ((j % 3 == 0 && j / 3 + 1 < nsub) ? 3 : 0) + (j + PB < K ? 3 * BPV : 0); };

    // ------------------------------------------------------------------ consumer state
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned a_q0[MI], b_foff[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) a_q0[i] = (unsigned)((2 * wm + i) * PW + frow);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_foff[kk] = (unsigned)((wn * (BN / 2) + frow) * 128 + (((kk * 4 + fchunk) ^ (frow & 7)) << 4));
    auto mfma_step = [&](const int abuf, const int slot, const int kh, const bool do_a, const bool do_b, const int s_next) {
        const unsigned char* cA = smem + abuf * A_BUF;
        const unsigned char* cB = smem + B_BASE + slot * B_STEP;
        unsigned qrow[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) qrow[i] = a_q0[i] + (unsigned)(kh * PW);
        bf16x8 af[2][MI], bfr[2][NI];
        auto fetch = [&](const int st, const int buf) {      // fragments of K step st = kw * 2 + kk of this step
            const int kw = st >> 1, kk = st & 1;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const unsigned q = qrow[i] + (unsigned)kw;
                af[buf][i] = *reinterpret_cast<const bf16x8*>(cA + (q << 7) + ((((unsigned)(kk * 4 + fchunk)) ^ (q & 7u)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[buf][j] = *reinterpret_cast<const bf16x8*>(cB + kw * TAPB + j * 2048 + b_foff[kk]);
        };
        fetch(0, 0);
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int cur = st & 1;
            if (st + 1 < 6) fetch(st + 1, cur ^ 1);
            // this step's DMA, one piece per MFMA group; a patch piece always BEFORE the weight piece that follows it: the step's last piece is
            // a weight piece, so "weight step k has landed" implies "every older patch piece has landed" (the counted wait below)
            if (INTERLEAVE) {
                if ((st & 1) == 0) { if (do_a) issue_a_piece(s_next, st >> 1); }
                else if (do_b) issue_b_piece(st >> 1);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
        }
        if (INTERLEAVE && do_b) advance_b();
    };

    // wait until at most n of this wave's DMA pieces are outstanding (n: a multiple of 3 up to 60; the immediate must be a constant)
    auto wait_pending = [&](const int n) {
        switch (n) {
#define FRCNN_WP(c) case c: asm volatile("s_waitcnt vmcnt(" #c ")" ::: "memory"); break;
            FRCNN_WP(0) FRCNN_WP(3) FRCNN_WP(6) FRCNN_WP(9) FRCNN_WP(12) FRCNN_WP(15) FRCNN_WP(18) FRCNN_WP(21) FRCNN_WP(24) FRCNN_WP(27) FRCNN_WP(30)
            FRCNN_WP(36) FRCNN_WP(42) FRCNN_WP(48) FRCNN_WP(54) FRCNN_WP(60)
#undef FRCNN_WP
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    // ------------------------------------------------------------------ K loop
    int pend = 3 * BPV * ((PB < K ? PB : K) - 1);      // pieces (per virtual wave) issued after the last piece of weight step 0
    int s = 0, kh = 0, slot = 0;
    if (is_loader) {
        // loader waves: prologue, then per step: wait for the step's pieces (twice the per-virtual-wave count), meet the MFMA waves at the
        // step's barrier, issue the pieces of step k + PB / the next chunk's patch.  Same barrier count as the MFMA waves, epilogue included.
        issue_a(0);
        for (int k = 0; k < PB && k < K; ++k) issue_b();
        for (int k = 0; k < K; ++k) {
            wait_pending(2 * pend);
            __builtin_amdgcn_s_barrier();
            if (kh == 0 && s + 1 < nsub) issue_a(s + 1);
            if (ld_k < K) issue_b();
            if (BNIN && kh == 0) __builtin_amdgcn_s_barrier();      // (the MFMA waves have transformed this chunk's patch)
            pend += issued_in(k) - (k + 1 < PB ? 3 * BPV : issued_in(k + 1 - PB));
            if (++kh == 3) { kh = 0; ++s; }
        }
        __builtin_amdgcn_s_barrier();            // (the MFMA waves' barriers: ring drained, staging tile complete, flushes)
        __builtin_amdgcn_s_barrier();
        if (RED) __builtin_amdgcn_s_barrier();
        if (STATS) __builtin_amdgcn_s_barrier();
        return;
    }
    if (!LW) {
        issue_a(0);
        for (int k = 0; k < PB && k < K; ++k) issue_b();
    }
    for (int k = 0; k < K; ++k) {
        // wait until only the `pend` youngest pieces are outstanding: weight step k (and, older than it, this chunk's patch) have landed
        if (!LW) {
            wait_pending(pend);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads of the buffers refilled next have completed
        __builtin_amdgcn_s_barrier();
        if (BNIN && kh == 0) {                                 // this chunk's patch has landed: BatchNorm + ReLU in place, then everyone reads it
            bnin_transform(s);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const bool do_a = !LW && kh == 0 && s + 1 < nsub, do_b = !LW && ld_k < K;
        if (!INTERLEAVE) {
            if (do_a) issue_a(s + 1);
            if (do_b) issue_b();
        }
        mfma_step(s & 1, slot, kh, do_a, do_b, s + 1);
        // pieces younger than weight step k + 1: what was younger than step k, plus this step's, minus the group that ends with step k + 1
        pend += issued_in(k) - (k + 1 < PB ? 3 * BPV : issued_in(k + 1 - PB));
        slot = slot + 1 == SB ? 0 : slot + 1;
        if (++kh == 3) { kh = 0; ++s; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // every wave is done with the patch and the ring: the staging tile may overwrite them

    // ------------------------------------------------------------------ epilogue
    // lane holds, for fragment (i, j): pixel (ty = 2 wm + i, tx = lane & 15); couts n0 + 32 wn + 16 j + 4 (lane >> 4) + 0..3
    unsigned char* stage = smem;
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_rz = __builtin_amdgcn_make_buffer_rsrc((void*)p.red_z, 0, p.y_bytes, 0x00020000);
    const int lrow_o = tid / C8, lc8 = tid - lrow_o * C8;
    unsigned vo_out[ST_IT];
#pragma unroll
    for (int it = 0; it < ST_IT; ++it) {
        const int r = lrow_o + it * (T / C8);
        const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
        vo_out[it] = (oy < p.Ho && ox < p.Wo) ? (unsigned)((img * p.Ho + oy) * p.Wo + ox) * (unsigned)(p.Cout * 2) + (unsigned)((n0 + lc8 * 8) * 2) : kOob;
    }
    u32x4 redz[ST_IT];
    unsigned redm[ST_IT];
    if (RED) {                                   // consumer layer's z rows and mask bytes: in flight under the convert phase
#pragma unroll
        for (int it = 0; it < ST_IT; ++it) {
            redz[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rz, vo_out[it], 0, 0);
            redm[it] = (p.red_mask && vo_out[it] != kOob) ? p.red_mask[vo_out[it] >> 4] : 0xFFu;
        }
    }
    const float lo = (flags & FRCNN_CONV_RELU) ? 0.f : -__builtin_inff();
    float ssum[NI][4], ssq[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[j][e] = ssq[j][e] = 0.f;
    const bool col_in = ox0 + frow < p.Wo;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = (2 * wm + i) * 16 + frow;
        const bool row_ok = col_in && oy0 + 2 * wm + i < p.Ho;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int cl = wn * (BN / 2) + j * 16 + fchunk * 4;
            u32x2 pk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 v;
                v[0] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h] + bv[j][2 * h], lo, __builtin_inff());
                v[1] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h + 1] + bv[j][2 * h + 1], lo, __builtin_inff());
                const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32 (RNE)
                pk[h] = bits;
                if (STATS) {                     // sums of the ROUNDED outputs (what the next layer reads), pixels inside the image only
                    const float q0 = row_ok ? __uint_as_float(bits << 16) : 0.f, q1 = row_ok ? __uint_as_float(bits & 0xFFFF0000u) : 0.f;
                    ssum[j][2 * h] += q0;
                    ssq[j][2 * h] += q0 * q0;
                    ssum[j][2 * h + 1] += q1;
                    ssq[j][2 * h + 1] += q1 * q1;
                }
            }
            *reinterpret_cast<u32x2*>(stage + r * ROWB + cl * 2) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float rsg[8], rsgz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) rsg[e] = rsgz[e] = 0.f;
#pragma unroll
    for (int it = 0; it < ST_IT; ++it) {
        const int r = lrow_o + it * (T / C8);
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * ROWB + lc8 * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, vo_out[it], 0, 0);      // (the offset travels in the VGPR: see conv_tile_kernel)
        if (RED) {
            float g[8], zz[8];
            unpack8(v, g);
            unpack8(redz[it], zz);
            const bool ok = vo_out[it] != kOob;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gm = (ok && ((redm[it] >> e) & 1u)) ? g[e] : 0.f;
                rsg[e] += gm;
                rsgz[e] += gm * zz[e];
            }
        }
    }
    // ---- flushes (as conv_tile_kernel's: nothing here waits for the global stores above)
    float* fl = reinterpret_cast<float*>(smem + STG);            // [NW][2][BN]
    const unsigned fl_a = lds_addr(fl);
    if (RED) {
#pragma unroll
        for (int sh = C8; sh < 64; sh <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                rsg[e] += __shfl_xor(rsg[e], sh);
                rsgz[e] += __shfl_xor(rsgz[e], sh);
            }
        }
        if (lane < C8) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                lds_write_b64(fl_a + ((wave * 2 + 0) * BN + lane * 8 + e) * 4, u32x2{__float_as_uint(rsg[e]), __float_as_uint(rsg[e + 1])});
                lds_write_b64(fl_a + ((wave * 2 + 1) * BN + lane * 8 + e) * 4, u32x2{__float_as_uint(rsgz[e]), __float_as_uint(rsgz[e + 1])});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid - st * BN;
            const int c = n0 + cl;
            float sg = 0.f, sgz = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                sg += fl[(w * 2 + 0) * BN + cl];
                sgz += fl[(w * 2 + 1) * BN + cl];
            }
            const float v = st == 0 ? sg : p.red_invstd[c] * (sgz - p.red_mean[c] * sg);
            atomicAdd(p.red_part + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * p.Cout + c, v);
        }
    }
    if (STATS) {
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
                ssum[j][e] = a;
                ssq[j][e] = b;
            }
        if (frow == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * (BN / 2) + j * 16 + fchunk * 4;
                lds_write_b64(fl_a + ((wm * 2 + 0) * BN + cl) * 4, u32x2{__float_as_uint(ssum[j][0]), __float_as_uint(ssum[j][1])});
                lds_write_b64(fl_a + ((wm * 2 + 0) * BN + cl + 2) * 4, u32x2{__float_as_uint(ssum[j][2]), __float_as_uint(ssum[j][3])});
                lds_write_b64(fl_a + ((wm * 2 + 1) * BN + cl) * 4, u32x2{__float_as_uint(ssq[j][0]), __float_as_uint(ssq[j][1])});
                lds_write_b64(fl_a + ((wm * 2 + 1) * BN + cl + 2) * 4, u32x2{__float_as_uint(ssq[j][2]), __float_as_uint(ssq[j][3])});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid - st * BN;
            const float v = (fl[(0 * 2 + st) * BN + cl] + fl[(1 * 2 + st) * BN + cl]) + (fl[(2 * 2 + st) * BN + cl] + fl[(3 * 2 + st) * BN + cl]);
            atomicAdd(p.stats + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * p.Cout + n0 + cl, (double)v);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------- 3x3 / stride 1 / pad 1, 64 input channels: weights resident
// conv2's 3x3 layers (64 -> 64 at M = 116,936) have ONE channel chunk: the whole filter slab of a 64-channel output part is 9 x 8 KB and
// fits LDS beside two input patches.  A persistent workgroup loads it ONCE and walks its share of the 8 x 16 pixel tiles: per tile it
// takes in one 23 KB patch (DMA'd a tile ahead) and nothing else, runs 18 K steps of MFMA between two barriers and stores 16 KB.  The
// tile kernel's kw-sharing form fetches 74 KB of weights and 50 KB of activations per 128-pixel tile and pays a barrier per tap
// (23 - 26 us per launch against an HBM roofline of 3.8); here the CU's load path carries a third of that and the statistics /
// reduce flushes (and their atomics) happen once per workgroup instead of once per tile.
//   * BNIN: the kernel's input is the RAW output z of the previous convolution and the kernel applies that layer's training-mode BatchNorm +
//     ReLU itself (frcnn_conv2d_fprop_bnin): every workgroup derives scale / shift of the 64 input channels from the statistics slots (the
//     arithmetic and summation order of bn_train_apply_kernel: same bits), transforms each landed patch in LDS (pixels outside the image
//     stay zero: the padding is applied AFTER the BatchNorm), and the channel-part-0 workgroups write the activation and its ReLU bit mask
//     for the backward pass from their patches' interiors.  One launch and one read of z less per layer than bn_train_apply + this kernel.
template <int SMODE, bool BNIN>
__global__ __launch_bounds__(512, 2) void conv3x3_wres_kernel(const ConvParams p, const int tiles_x, const int tiles_y, const int tiles_total) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool STATS = SMODE == 1, RED = SMODE == 2;
    constexpr int BNIN_OFF = 9 * 8 * 1024 + 2 * 24 * 1024 + 128 * (64 * 2 + 16) + 8 * 2 * 64 * 4;      // scale / shift / reduction scratch behind the flush scratch
    constexpr int NW = 8, T = 512, BM = 128, BN = 64, TW = 16, TH = 8, PW = TW + 2;
    constexpr int MI = 2, NI = 2;
    constexpr int W_BYTES = 9 * 8 * 1024, A_BUF = 24 * 1024, A_BASE = W_BYTES, STG_BASE = A_BASE + 2 * A_BUF;
    constexpr int ROWB = BN * 2 + 16, C8 = BN / 8, ST_IT = (BM * C8) / T, STG = BM * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int flags = p.flags;
    // workgroup b: channel part b % tiles_n, spatial tiles b / tiles_n, + gridDim.x / tiles_n, ... (the host makes the grid a multiple of tiles_n)
    const int tn = blockIdx.x % p.tiles_n, n0 = tn * BN;
    const int t_first = blockIdx.x / p.tiles_n, t_stride = gridDim.x / p.tiles_n;
    const int per_img = tiles_x * tiles_y;

    float bv[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (flags & FRCNN_CONV_BIAS) b = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 32 + j * 16 + fchunk * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = b[e];
    }
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_rz = __builtin_amdgcn_make_buffer_rsrc((void*)p.red_z, 0, p.y_bytes, 0x00020000);
    const unsigned dma_swz = (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
    // the filter slab of this channel part: [tap][64 output channels][128 B], 9 pieces per wave, once
    {
        const unsigned b_voff = (unsigned)(n0 + wave * 8 + (lane >> 3)) * (unsigned)(p.Ktot * 2) + dma_swz;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(smem + tap * 8192 + wave * 1024), 16, b_voff, (unsigned)(tap * 128), 0, 0);
    }
    float* s_scale = reinterpret_cast<float*>(smem + BNIN_OFF);
    float* s_shift = s_scale + 64;
    if (BNIN) {
        // scale / shift of the input layer's BatchNorm from its f64 statistics slots, as bn_train_apply_kernel's prologue (four slot
        // slices per channel summed in the same order, the same f64 -> f32 conversions); workgroup 0 publishes mean / invstd and
        // updates the moving statistics
        double* red = reinterpret_cast<double*>(smem + BNIN_OFF + 512);          // [2][4][64]
        if (tid < 256) {
            const int cl = tid & 63, sl = tid >> 6;
            double a[FRCNN_STAT_SLOTS / 4], b[FRCNN_STAT_SLOTS / 4], s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) {
                a[k] = p.bnin_part[((long long)(sl + 4 * k) * 2) * 64 + cl];
                b[k] = p.bnin_part[((long long)(sl + 4 * k) * 2 + 1) * 64 + cl];
            }
#pragma unroll
            for (int k = 0; k < FRCNN_STAT_SLOTS / 4; ++k) { s0 += a[k]; s1 += b[k]; }
            red[(0 * 4 + sl) * 64 + cl] = s0;
            red[(1 * 4 + sl) * 64 + cl] = s1;
        }
        __syncthreads();
        if (tid < 64) {
            const int cl = tid;
            const double sum = red[(0 * 4 + 0) * 64 + cl] + red[(0 * 4 + 1) * 64 + cl] + red[(0 * 4 + 2) * 64 + cl] + red[(0 * 4 + 3) * 64 + cl];
            const double ssq = red[(1 * 4 + 0) * 64 + cl] + red[(1 * 4 + 1) * 64 + cl] + red[(1 * 4 + 2) * 64 + cl] + red[(1 * 4 + 3) * 64 + cl];
            const double mean = sum * p.bnin_inv_count;
            double var = ssq * p.bnin_inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)p.bnin_eps));
            const float sc = p.bnin_gamma[cl] * invstd;
            s_scale[cl] = sc;
            s_shift[cl] = p.bnin_beta[cl] - (float)mean * sc;
            if (blockIdx.x == 0) {
                p.bnin_mean[cl] = (float)mean;
                p.bnin_invstd[cl] = invstd;
                p.bnin_mm[cl] = p.bnin_mm[cl] * p.bnin_momentum + (float)mean * (1.f - p.bnin_momentum);
                p.bnin_mv[cl] = p.bnin_mv[cl] * p.bnin_momentum + (float)(var * p.bnin_unbias) * (1.f - p.bnin_momentum);
            }
        }
        __syncthreads();
    }
    // patch rows of this lane's three DMA pieces (tile-independent part)
    int a_py[3], a_px[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = (wave + NW * t) * 8 + (lane >> 3);
        a_py[t] = q < (TH + 2) * PW ? q / PW : -100000;          // rows beyond the patch: never inside an image
        a_px[t] = q - (q / PW) * PW;
    }
    auto issue_patch = [&](const int tile, const int buf) {
        const int img = tile / per_img, trem = tile - img * per_img;
        const int tyt = trem / tiles_x;
        const int oy0 = tyt * TH, ox0 = (trem - tyt * tiles_x) * TW;
        unsigned char* dst = smem + A_BASE + buf * A_BUF + wave * 1024;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int iy = oy0 - 1 + a_py[t], ix = ox0 - 1 + a_px[t];
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const unsigned vo = ok ? (unsigned)((img * p.Hi + iy) * p.Wi + ix) * (unsigned)(p.in_pix_stride * 2) + dma_swz : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(dst + t * NW * 1024), 16, vo, 0, 0, 0);
        }
    };
    unsigned a_q0[MI], b_foff[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) a_q0[i] = (unsigned)((2 * wm + i) * PW + frow);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_foff[kk] = (unsigned)((wn * 32 + frow) * 128 + (((kk * 4 + fchunk) ^ (frow & 7)) << 4));

    const int lrow_o = tid / C8, lc8 = tid - lrow_o * C8;
    const float lo = (flags & FRCNN_CONV_RELU) ? 0.f : -__builtin_inff();
    float ssum[NI][4], ssq[NI][4];               // BatchNorm partial sums over ALL tiles of this workgroup
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[j][e] = ssq[j][e] = 0.f;
    float rsg[8], rsgz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) rsg[e] = rsgz[e] = 0.f;
    unsigned char* stage = smem + STG_BASE;

    int tile = t_first, buf = 0;
    if (tile < tiles_total) issue_patch(tile, 0);
    bool first = true;
    for (; tile < tiles_total; tile += t_stride, buf ^= 1) {
        // this tile's patch (and, the first time, the filter slab) has landed; younger than its pieces are only the previous tile's ST_IT stores
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST_IT) : "memory");
        first = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave is done with the other patch buffer and the staging tile
        __builtin_amdgcn_s_barrier();
        if (BNIN) {
            // BatchNorm + ReLU of the landed patch, in place: 192 rows x 8 sixteen-byte slots, three per thread (slot s of row q holds
            // channels 8 (s ^ (q & 7)) ..).  Rows outside the image (and beyond the patch) become zeros -- the convolution's padding.
            const int img_t = tile / per_img, trem_t = tile - img_t * per_img;
            const int tyt_t = trem_t / tiles_x;
            const int oy0_t = tyt_t * TH, ox0_t = (trem_t - tyt_t * tiles_x) * TW;
            const unsigned pbase = lds_addr(smem + A_BASE + buf * A_BUF);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = tid + T * k;
                const int q = idx >> 3, slot = idx & 7, c8 = slot ^ (q & 7);
                const int py = q / PW, px = q - py * PW;
                const int iy = oy0_t - 1 + py, ix = ox0_t - 1 + px;
                const bool valid = q < (TH + 2) * PW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(smem + A_BASE + buf * A_BUF + q * 128 + slot * 16);
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s_scale + c8 * 8), sc1 = *reinterpret_cast<const f32x4*>(s_scale + c8 * 8 + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s_shift + c8 * 8), sh1 = *reinterpret_cast<const f32x4*>(s_shift + c8 * 8 + 4);
                float x[8];
                unpack8(raw, x);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = fmaxf(x[e] * sc0[e] + sh0[e], 0.f);
                    x[4 + e] = fmaxf(x[4 + e] * sc1[e] + sh1[e], 0.f);
                }
                u32x4 pk = pack8(x);
                if (!valid) pk = u32x4{0u, 0u, 0u, 0u};
                asm volatile("ds_write_b128 %0, %1" ::"v"(pbase + (unsigned)(q * 128 + slot * 16)), "v"(pk) : "memory");
                if (tn == 0) {                   // (workgroup-uniform) this tile's own pixels: the activation and its mask
                    const u32x2 mk = pool_mask8(relu_bits8(pk), c8);      // the row's eight mask bytes: one 8-byte store by its first lane
                    if (valid && py >= 1 && py <= TH && px >= 1 && px <= TW) {
                        const long long pix = ((long long)img_t * p.Hi + iy) * p.Wi + ix;
                        __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(p.bnin_act + pix * 64 + c8 * 8));      // (as in the patch kernel)
                        if (slot == 0) __builtin_nontemporal_store(mk, reinterpret_cast<u32x2*>(p.bnin_mask + pix * 8));
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (tile + t_stride < tiles_total) issue_patch(tile + t_stride, buf ^ 1);

        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const unsigned char* cA = smem + A_BASE + buf * A_BUF;
            bf16x8 af[2][MI], bfr[2][NI];
            auto fetch = [&](const int st, const int b) {       // K step st = (kh * 3 + kw) * 2 + kk
                const int tap = st >> 1, kk = st & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const unsigned q = a_q0[i] + (unsigned)(kh * PW + kw);
                    af[b][i] = *reinterpret_cast<const bf16x8*>(cA + (q << 7) + ((((unsigned)(kk * 4 + fchunk)) ^ (q & 7u)) << 4));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) bfr[b][j] = *reinterpret_cast<const bf16x8*>(smem + tap * 8192 + j * 2048 + b_foff[kk]);
            };
            fetch(0, 0);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                const int cur = st & 1;
                if (st + 1 < 18) fetch(st + 1, cur ^ 1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
            }
        }
        // ---- epilogue of this tile (as conv3x3_patch_kernel's)
        const int img = tile / per_img, trem = tile - img * per_img;
        const int tyt = trem / tiles_x;
        const int oy0 = tyt * TH, ox0 = (trem - tyt * tiles_x) * TW;
        unsigned vo_out[ST_IT];
#pragma unroll
        for (int it = 0; it < ST_IT; ++it) {
            const int r = lrow_o + it * (T / C8);
            const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
            vo_out[it] = (oy < p.Ho && ox < p.Wo) ? (unsigned)((img * p.Ho + oy) * p.Wo + ox) * (unsigned)(p.Cout * 2) + (unsigned)((n0 + lc8 * 8) * 2) : kOob;
        }
        u32x4 redz[ST_IT];
        unsigned redm[ST_IT];
        if (RED) {
#pragma unroll
            for (int it = 0; it < ST_IT; ++it) {
                redz[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rz, vo_out[it], 0, 0);
                redm[it] = (p.red_mask && vo_out[it] != kOob) ? p.red_mask[vo_out[it] >> 4] : 0xFFu;
            }
        }
        const bool col_in = ox0 + frow < p.Wo;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = (2 * wm + i) * 16 + frow;
            const bool row_ok = col_in && oy0 + 2 * wm + i < p.Ho;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * 32 + j * 16 + fchunk * 4;
                u32x2 pk;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 v;
                    v[0] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h] + bv[j][2 * h], lo, __builtin_inff());
                    v[1] = __builtin_amdgcn_fmed3f(acc[i][j][2 * h + 1] + bv[j][2 * h + 1], lo, __builtin_inff());
                    const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
                    pk[h] = bits;
                    if (STATS) {
                        const float q0 = row_ok ? __uint_as_float(bits << 16) : 0.f, q1 = row_ok ? __uint_as_float(bits & 0xFFFF0000u) : 0.f;
                        ssum[j][2 * h] += q0;
                        ssq[j][2 * h] += q0 * q0;
                        ssum[j][2 * h + 1] += q1;
                        ssq[j][2 * h + 1] += q1 * q1;
                    }
                }
                // (asm form: hipcc would order a DS write it emits itself behind the next patch's pending LDS-DMA -- vmcnt(0))
                lds_write_b64(lds_addr(stage) + r * ROWB + cl * 2, pk);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int it = 0; it < ST_IT; ++it) {
            const int r = lrow_o + it * (T / C8);
            const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * ROWB + lc8 * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, vo_out[it], 0, 0);
            if (RED) {
                float g[8], zz[8];
                unpack8(v, g);
                unpack8(redz[it], zz);
                const bool ok = vo_out[it] != kOob;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gm = (ok && ((redm[it] >> e) & 1u)) ? g[e] : 0.f;
                    rsg[e] += gm;
                    rsgz[e] += gm * zz[e];
                }
            }
        }
    }
    // ---- flushes, once per workgroup (scratch behind the staging tile)
    float* fl = reinterpret_cast<float*>(smem + STG_BASE + STG);            // [NW][2][BN]
    const unsigned fl_a = lds_addr(fl);
    if (RED) {
#pragma unroll
        for (int sh = C8; sh < 64; sh <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                rsg[e] += __shfl_xor(rsg[e], sh);
                rsgz[e] += __shfl_xor(rsgz[e], sh);
            }
        }
        if (lane < C8) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                lds_write_b64(fl_a + ((wave * 2 + 0) * BN + lane * 8 + e) * 4, u32x2{__float_as_uint(rsg[e]), __float_as_uint(rsg[e + 1])});
                lds_write_b64(fl_a + ((wave * 2 + 1) * BN + lane * 8 + e) * 4, u32x2{__float_as_uint(rsgz[e]), __float_as_uint(rsgz[e + 1])});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid - st * BN;
            const int c = n0 + cl;
            float sg = 0.f, sgz = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                sg += fl[(w * 2 + 0) * BN + cl];
                sgz += fl[(w * 2 + 1) * BN + cl];
            }
            const float v = st == 0 ? sg : p.red_invstd[c] * (sgz - p.red_mean[c] * sg);
            atomicAdd(p.red_part + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * p.Cout + c, v);
        }
    }
    if (STATS) {
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
                ssum[j][e] = a;
                ssq[j][e] = b;
            }
        if (frow == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * 32 + j * 16 + fchunk * 4;
                lds_write_b64(fl_a + ((wm * 2 + 0) * BN + cl) * 4, u32x2{__float_as_uint(ssum[j][0]), __float_as_uint(ssum[j][1])});
                lds_write_b64(fl_a + ((wm * 2 + 0) * BN + cl + 2) * 4, u32x2{__float_as_uint(ssum[j][2]), __float_as_uint(ssum[j][3])});
                lds_write_b64(fl_a + ((wm * 2 + 1) * BN + cl) * 4, u32x2{__float_as_uint(ssq[j][0]), __float_as_uint(ssq[j][1])});
                lds_write_b64(fl_a + ((wm * 2 + 1) * BN + cl + 2) * 4, u32x2{__float_as_uint(ssq[j][2]), __float_as_uint(ssq[j][3])});
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * BN) {
            const int st = tid / BN, cl = tid - st * BN;
            const float v = (fl[(0 * 2 + st) * BN + cl] + fl[(1 * 2 + st) * BN + cl]) + (fl[(2 * 2 + st) * BN + cl] + fl[(3 * 2 + st) * BN + cl]);
            atomicAdd(p.stats + ((long long)(blockIdx.x & (FRCNN_STAT_SLOTS - 1)) * 2 + st) * p.Cout + n0 + cl, (double)v);
        }
    }
#endif
}

// Name of the instantiation the calling thread launched last (frcnn_last_conv_instantiation): lets the parity tests assert
// WHICH kernel a shape dispatched to, so that their coverage cannot rot silently when the heuristics below move.
thread_local char g_last_inst[512] = "";

template <int K, int NC>
int launch_stream_1x1(const ConvParams& p, hipStream_t s) {
    const int parts = p.Cout / NC;
    const int groups = (p.M + 15) / 16;
    // two workgroups of four waves per CU; the number of workgroups a multiple of the channel parts (a workgroup keeps its part)
    long long wgs = 2ll * num_cus();
    if (wgs * 4 > (long long)groups * parts) wgs = ((long long)groups * parts + 3) / 4;
    wgs = (wgs + parts - 1) / parts * parts;
    const int grid = (int)wgs;
    const bool stats = (p.flags & FRCNN_CONV_STATS) != 0;
    snprintf(g_last_inst, sizeof(g_last_inst), "conv1x1_stream<K=%d,NC=%d,STATS=%d> grid=%dx1 tpb=1", K, NC, stats ? 1 : 0, grid);
    if (p.dry_run) return FRCNN_OK;
    if (stats) hipLaunchKernelGGL((conv1x1_stream_kernel<K, NC, true>), dim3(grid), dim3(256), 0, s, p, parts, groups);
    else hipLaunchKernelGGL((conv1x1_stream_kernel<K, NC, false>), dim3(grid), dim3(256), 0, s, p, parts, groups);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_fprop(stream 1x1)");
    return FRCNN_OK;
}

template <int SB, int SMODE, int LW = 0, int BNT = 64, bool BNIN = false>
int launch_patch_sm(const ConvParams& p, hipStream_t s, const int tiles_x, const int tiles_y, const int grid) {
    constexpr int smem = 2 * 24 * 1024 + SB * 3 * BNT * 128 + (BNIN ? 2 * 512 * 4 : 0);      // BNIN: + scale / shift of up to 512 input channels
    static_assert(smem <= 163840, "LDS budget");
    constexpr int occw = LW ? 3 : 2;             // waves per SIMD of the one workgroup a CU holds
    if (!p.dry_run && frcnn_allow_big_lds(reinterpret_cast<const void*>(&conv3x3_patch_kernel<SB, SMODE, occw, LW, BNT, BNIN>), smem) != 0) {
        frcnn_set_error("frcnn_conv2d_fprop(patch 3x3): cannot reserve %d B of LDS", smem);
        return FRCNN_EINVAL;
    }
    snprintf(g_last_inst, sizeof(g_last_inst), "conv3x3_patch<SB=%d,SMODE=%d%s%s%s> grid=%dx1 tpb=1", SB, SMODE, LW ? ",LW=4" : "", BNT == 128 ? ",BN=128" : "",
             BNIN ? ",BNIN=1" : "", grid);
    if (p.dry_run) return FRCNN_OK;
    hipLaunchKernelGGL((conv3x3_patch_kernel<SB, SMODE, occw, LW, BNT, BNIN>), dim3(grid), dim3(512 + 64 * LW), smem, s, p, tiles_x, tiles_y);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_fprop(patch 3x3)");
    return FRCNN_OK;
}

// 3x3 / stride 1 / pad 1 on the patch-resident kernel: 8 x 16 pixel tiles x 64 output channels, one workgroup per CU
int launch_patch(ConvParams p, hipStream_t s, const int n_img, int sb, const int bn = 64) {
    const int tiles_x = (p.Wo + 15) / 16, tiles_y = (p.Ho + 7) / 8;
    p.tiles_m = n_img * tiles_x * tiles_y;
    p.tiles_n = p.Cout / bn;
    p.items = p.tiles_m * p.tiles_n;
    const int smode = (p.flags & FRCNN_CONV_STATS) ? 1 : (p.red_part ? 2 : 0);
    if (p.bnin_part) {                           // the input layer's BatchNorm + ReLU applied by this launch (loader-wave forms, forward only)
        if (smode == 2 || p.Cin > 512 || !(bn == 128 || sb == 4)) {
            frcnn_set_error("frcnn_conv2d_fprop_bnin(patch 3x3): forward convolutions with at most 512 input channels on the loader-wave forms");
            return FRCNN_EINVAL;
        }
        if (bn == 128) return smode == 1 ? launch_patch_sm<2, 1, 4, 128, true>(p, s, tiles_x, tiles_y, p.items) : launch_patch_sm<2, 0, 4, 128, true>(p, s, tiles_x, tiles_y, p.items);
        return smode == 1 ? launch_patch_sm<4, 1, 4, 64, true>(p, s, tiles_x, tiles_y, p.items) : launch_patch_sm<4, 0, 4, 64, true>(p, s, tiles_x, tiles_y, p.items);
    }
    if (bn == 128) {                             // 128 output channels per workgroup: 2-slot weight ring (144 KB of LDS), loader waves
        if (smode == 1) return launch_patch_sm<2, 1, 4, 128>(p, s, tiles_x, tiles_y, p.items);
        if (smode == 2) return launch_patch_sm<2, 2, 4, 128>(p, s, tiles_x, tiles_y, p.items);
        return launch_patch_sm<2, 0, 4, 128>(p, s, tiles_x, tiles_y, p.items);
    }
#define FRCNN_PATCH_CASE(SB_)                                                                           \
    if (sb == SB_) {                                                                                    \
        if (smode == 1) return launch_patch_sm<SB_, 1>(p, s, tiles_x, tiles_y, p.items);              \
        if (smode == 2) return launch_patch_sm<SB_, 2>(p, s, tiles_x, tiles_y, p.items);              \
        return launch_patch_sm<SB_, 0>(p, s, tiles_x, tiles_y, p.items);                              \
    }
    // four dedicated loader waves (measured, tools/patch_bench.py, us, without -> with: conv4 3x3 17.3 -> 15.4, RPN 3x3 48.2 -> 38.0,
    // conv4 at batch 2 15.7 -> 13.6)
    int lw = 4;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_PATCH_LW")) lw = atoi(e);
#endif
    if (lw == 4 && sb == 4) {
        if (smode == 1) return launch_patch_sm<4, 1, 4>(p, s, tiles_x, tiles_y, p.items);
        if (smode == 2) return launch_patch_sm<4, 2, 4>(p, s, tiles_x, tiles_y, p.items);
        return launch_patch_sm<4, 0, 4>(p, s, tiles_x, tiles_y, p.items);
    }
#ifdef FRCNN_SWEEP
    FRCNN_PATCH_CASE(4)
    FRCNN_PATCH_CASE(3)
    FRCNN_PATCH_CASE(2)
#endif
#undef FRCNN_PATCH_CASE
    frcnn_set_error("conv2d_fprop(patch 3x3): no instantiation with a %d-step weight ring", sb);
    return FRCNN_EINVAL;
}

// 3x3 / stride 1 / pad 1 with 64 input channels on the weights-resident kernel: persistent workgroups, equal shares of the 8 x 16 pixel tiles
int launch_wres(ConvParams p, hipStream_t s, const int n_img) {
    const int tiles_x = (p.Wo + 15) / 16, tiles_y = (p.Ho + 7) / 8;
    const int tiles_total = n_img * tiles_x * tiles_y;
    p.tiles_m = tiles_total;
    p.tiles_n = p.Cout / 64;
    p.items = p.tiles_m * p.tiles_n;
    int wg_sp = num_cus() / p.tiles_n;                           // workgroups per channel part: one workgroup per CU in all
    if (wg_sp < 1) wg_sp = 1;
    const int per = (tiles_total + wg_sp - 1) / wg_sp;           // tiles per workgroup, the same for all but the last few
    const int grid = ((tiles_total + per - 1) / per) * p.tiles_n;
    constexpr int smem = 9 * 8 * 1024 + 2 * 24 * 1024 + 128 * (64 * 2 + 16) + 8 * 2 * 64 * 4 + 512 + 2 * 4 * 64 * 8;
    static_assert(smem <= 163840, "LDS budget");
    const int smode = (p.flags & FRCNN_CONV_STATS) ? 1 : (p.red_part ? 2 : 0);
    const bool bnin = p.bnin_part != nullptr;
    if (bnin && smode == 2) {
        frcnn_set_error("frcnn_conv2d_fprop_bnin: forward convolutions only");
        return FRCNN_EINVAL;
    }
    const void* fn = bnin ? (smode == 1 ? reinterpret_cast<const void*>(&conv3x3_wres_kernel<1, true>) : reinterpret_cast<const void*>(&conv3x3_wres_kernel<0, true>))
                   : smode == 1 ? reinterpret_cast<const void*>(&conv3x3_wres_kernel<1, false>)
                   : smode == 2 ? reinterpret_cast<const void*>(&conv3x3_wres_kernel<2, false>) : reinterpret_cast<const void*>(&conv3x3_wres_kernel<0, false>);
    if (!p.dry_run && frcnn_allow_big_lds(fn, smem) != 0) {
        frcnn_set_error("frcnn_conv2d_fprop(3x3, weights resident): cannot reserve %d B of LDS", smem);
        return FRCNN_EINVAL;
    }
    snprintf(g_last_inst, sizeof(g_last_inst), "conv3x3_wres<SMODE=%d%s> grid=%dx1 tpb=%d", smode, bnin ? ",BNIN=1" : "", grid, per);
    if (p.dry_run) return FRCNN_OK;
    if (bnin && smode == 1) hipLaunchKernelGGL((conv3x3_wres_kernel<1, true>), dim3(grid), dim3(512), smem, s, p, tiles_x, tiles_y, tiles_total);
    else if (bnin) hipLaunchKernelGGL((conv3x3_wres_kernel<0, true>), dim3(grid), dim3(512), smem, s, p, tiles_x, tiles_y, tiles_total);
    else if (smode == 1) hipLaunchKernelGGL((conv3x3_wres_kernel<1, false>), dim3(grid), dim3(512), smem, s, p, tiles_x, tiles_y, tiles_total);
    else if (smode == 2) hipLaunchKernelGGL((conv3x3_wres_kernel<2, false>), dim3(grid), dim3(512), smem, s, p, tiles_x, tiles_y, tiles_total);
    else hipLaunchKernelGGL((conv3x3_wres_kernel<0, false>), dim3(grid), dim3(512), smem, s, p, tiles_x, tiles_y, tiles_total);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_fprop(3x3, weights resident)");
    return FRCNN_OK;
}

unsigned long long* g_stamp_buffer = nullptr;    // FRCNN_STAMPS builds: set through frcnn_debug_set_stamp_buffer (tools/conv_stamps.py)

template <int BM, int BN, int BK, int S, bool LIN, int SMODE, int OCC, bool MULTI, bool F32 = false, bool KWS = false, bool FIX = false, int F8 = 0, bool BNIN = false>
int launch_tile(const ConvParams& p, hipStream_t s) {
    constexpr int ring = KWS ? 2 * 3 * 8 * 1024 + S * BN * BK * 2 : S * (BM + BN) * BK * 2, stg = BM * (BN * 2 + 16), stg32 = BM * (BN * 4 + 16);
    constexpr int smem = (F32 ? (ring > stg32 ? ring : stg32) : MULTI ? ring + stg : (ring > stg ? ring : stg)) + 2 * BN * 4 + (BNIN ? 2 * 512 * 4 : 0) + (F8 ? 2 * BN * 4 : 0);   // BNIN: + scale / shift of <= 512 input channels; F8: + dequantisation factors / bias
    static_assert(smem <= 163840, "LDS budget");
    static_assert(smem * OCC <= 163840, "occupancy target does not fit the LDS");
    if (!p.dry_run && frcnn_allow_big_lds(reinterpret_cast<const void*>(&conv_tile_kernel<BM, BN, BK, S, LIN, SMODE, OCC, MULTI, F32, KWS, FIX, F8, BNIN>), smem) != 0) {
        frcnn_set_error("frcnn_conv2d_fprop: cannot reserve %d B of LDS", smem);
        return FRCNN_EINVAL;
    }
    const int grid_x = FIX ? ((2 * p.items + 15) / 16) * 16 : p.items;      // FIX: two halves per tile, whole pairs per XCD
    snprintf(g_last_inst, sizeof(g_last_inst), "conv_tile<BM=%d,BN=%d,BK=%d,S=%d,LIN=%d,SMODE=%d,OCC=%d,MULTI=%d,F32=%d,KWS=%d%s%s%s> grid=%dx%d tpb=%d",
             BM, BN, BK, S, (int)LIN, SMODE, OCC, (int)MULTI, (int)F32, (int)KWS, FIX ? ",FIX=1" : "", F8 == 1 ? ",F8=1" : F8 == 2 ? ",F8=2" : "", BNIN ? ",BNIN=1" : "",
             grid_x, F32 ? p.split : 1, p.tiles_per_block);
    if (p.dry_run) return FRCNN_OK;              // frcnn_conv2d_describe: the dispatch decision only
    hipLaunchKernelGGL((conv_tile_kernel<BM, BN, BK, S, LIN, SMODE, OCC, MULTI, F32, KWS, FIX, F8, BNIN>), dim3(grid_x, F32 ? p.split : 1), dim3(512), smem, s, p);
    FRCNN_CHECK_LAUNCH("frcnn_conv2d_fprop");
    return FRCNN_OK;
}

template <int BM, int BN, int BK, int S, int OCC, bool MULTI, bool FIX = false>
int launch_tile_flags(const ConvParams& p, hipStream_t s) {
    const int smode = (p.flags & FRCNN_CONV_STATS) ? 1 : (p.red_part ? 2 : 0);
    if (p.bnin_part) {
        // the input layer's BatchNorm + ReLU applied to the landed A slices (conv_tile_kernel, BNIN): 1x1 / stride-1 forward layers on the
        // two-slot 128-row forms
        if constexpr (S == 2 && BM == 128 && BK == 64 && !FIX) {
            if (!p.linear_a || smode == 2 || p.f8_x_scale || p.Cin > 512 || p.in_pix_stride != p.Cin) {
                frcnn_set_error("conv2d_fprop_bnin(1x1): dense 1x1 / stride-1 forward layers with at most 512 input channels");
                return FRCNN_EINVAL;
            }
            constexpr int O = OCC > 2 ? 2 : OCC;    // (the transform pass does not fit the 80 VGPRs of three workgroups per CU: 10 would spill)
            if (smode == 1) return launch_tile<BM, BN, BK, S, true, 1, O, MULTI, false, false, false, 0, true>(p, s);
            return launch_tile<BM, BN, BK, S, true, 0, O, MULTI, false, false, false, 0, true>(p, s);
        } else {
            frcnn_set_error("conv2d_fprop_bnin(1x1): no BatchNorm-in form of the %d x %d x %d tile with %d ring slots", BM, BN, BK, S);
            return FRCNN_EINVAL;
        }
    }
    if (p.f8_x_scale) {
        // fp8 operands: forward convolutions (x e4m3; with or without statistics) and data gradients (x e5m2; plain or with the fused
        // BatchNorm-backward reduce)
        if constexpr (BK == 64) {
            if (p.f8_fmt == 1) {
                if (smode == 2) {
                    frcnn_set_error("conv2d fp8: the fused BatchNorm-backward reduce belongs to the data-gradient entry point");
                    return FRCNN_EINVAL;
                }
                if (p.linear_a) {
                    if (smode == 1) return launch_tile<BM, BN, BK, S, true, 1, OCC, MULTI, false, false, FIX, 1>(p, s);
                    return launch_tile<BM, BN, BK, S, true, 0, OCC, MULTI, false, false, FIX, 1>(p, s);
                }
                if (smode == 1) return launch_tile<BM, BN, BK, S, false, 1, OCC, MULTI, false, false, FIX, 1>(p, s);
                return launch_tile<BM, BN, BK, S, false, 0, OCC, MULTI, false, false, FIX, 1>(p, s);
            }
            if (smode == 1) {
                frcnn_set_error("conv2d fp8 data gradient: no forward statistics");
                return FRCNN_EINVAL;
            }
            if (p.linear_a) {
                if (smode == 2) return launch_tile<BM, BN, BK, S, true, 2, OCC, MULTI, false, false, FIX, 2>(p, s);
                return launch_tile<BM, BN, BK, S, true, 0, OCC, MULTI, false, false, FIX, 2>(p, s);
            }
            if (smode == 2) return launch_tile<BM, BN, BK, S, false, 2, OCC, MULTI, false, false, FIX, 2>(p, s);
            return launch_tile<BM, BN, BK, S, false, 0, OCC, MULTI, false, false, FIX, 2>(p, s);
        } else {
            frcnn_set_error("conv2d fp8: needs 128-byte K slices (cin %% 128 == 0)");
            return FRCNN_EINVAL;
        }
    }
    if (smode == 2)
        return p.linear_a ? launch_tile<BM, BN, BK, S, true, 2, OCC, MULTI, false, false, FIX>(p, s)
                          : launch_tile<BM, BN, BK, S, false, 2, OCC, MULTI, false, false, FIX>(p, s);
    if (p.linear_a) {
        if (smode == 1) return launch_tile<BM, BN, BK, S, true, 1, OCC, MULTI, false, false, FIX>(p, s);
        return launch_tile<BM, BN, BK, S, true, 0, OCC, MULTI, false, false, FIX>(p, s);
    }
    if (smode == 1) return launch_tile<BM, BN, BK, S, false, 1, OCC, MULTI, false, false, FIX>(p, s);
    return launch_tile<BM, BN, BK, S, false, 0, OCC, MULTI, false, false, FIX>(p, s);
}

thread_local bool g_ws_query = false;             // inside frcnn_conv2d_workspace_bytes: answer for the tile kernel's forms (see there)
thread_local size_t g_last_ws_bytes = 0;          // workspace the last dispatch decision would use (frcnn_conv2d_workspace_bytes)
thread_local size_t g_last_ws_counter_bytes = 0;  // ... and the size of its arrival-counter tail

#ifdef FRCNN_SWEEP
// kernel-development builds only (FRCNN_SWEEP=1 python .../build.py --force; tools/tile_sweep.py): tile shape / kw-sharing
// overrides, re-read per launch (the sweep tool changes them between launches; production builds contain no getenv)
struct SweepEnv {
    int bm = 0, bn = 0, bk = 0, stages = 0, tpb = 1, kws = -1;
    SweepEnv() {
        if (const char* e = getenv("FRCNN_TILE")) {              // "bm,bn,bk,stages[,tiles_per_block]"
            int a = 0, b = 0, c = 0, st = 0, tp = 1;
            if (sscanf(e, "%d,%d,%d,%d,%d", &a, &b, &c, &st, &tp) >= 4) { bm = a; bn = b; bk = c; stages = st; tpb = tp; }
        }
        if (const char* e = getenv("FRCNN_KWS")) kws = e[0] == '1' ? 1 : 0;
    }
};
SweepEnv sweep_env() { return SweepEnv(); }
#endif

// Tile choice + launch.  p arrives with the geometry fields filled in; tiles_m / tiles_n / k_tiles are set here.
int conv_tile_dispatch(ConvParams p, const frcnn_conv_desc* d, hipStream_t s) {
    if (p.taps > 32) {
        frcnn_set_error("conv2d_fprop: filters with more than 32 taps are not supported (per-row tap validity masks are 32 bits)");
        return FRCNN_EINVAL;
    }
    FRCNN_CHECK_ARG(!(p.f8_x_scale && (p.flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC))), "conv2d fp8: bf16 output only");
    if (p.flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC)) {
        // fp32 output / split-K partial sums: 1x1 filters whose output rows are the GEMM rows, K a multiple of 64
        FRCNN_CHECK_ARG(p.taps == 1 && p.linear_a && d->cin % 64 == 0 && !(p.flags & (FRCNN_CONV_STATS | FRCNN_CONV_ADD_RES)),
                        "conv2d_fprop: fp32 / split-K output needs a 1x1 stride-1 filter, cin %% 64 == 0, no STATS / ADD_RES");
        p.k_tiles = p.Ktot / 64;
        int split = d->split_k > 1 ? d->split_k : 1;
        p.k_tiles_per_split = (p.k_tiles + split - 1) / split;
        p.split = (p.k_tiles + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
        p.tiles_m = (p.M + 127) / 128;
        p.tiles_n = (d->cout + 63) / 64;
        p.tiles_per_block = 1;
        p.items = p.tiles_m * p.tiles_n;
        return launch_tile<128, 64, 64, 3, true, 0, 2, false, true>(p, s);
    }
    FRCNN_CHECK_ARG(d->split_k <= 1, "conv2d_fprop: split_k needs SPLITK_ATOMIC");
    // measured on the R50-C4 layer shapes (tools/tile_sweep.py): 128-row tiles and BK = 64 win almost everywhere (two
    // workgroups per CU); 128 output channels per tile once that still leaves >= ~400 tiles, else 64; a third ring slot
    // only pays on very long K with the narrow tile
    int bk = d->cin % 64 == 0 ? 64 : 32;
    const long long M = p.M;
    int bm = 128;
    const long long tiles_m128 = (M + 127) / 128;
    // (same-box A/B of this threshold in the step, ms: batch 4: 200 4.29-4.31, 300 / 400 4.26-4.28, 500 4.29-4.31; R101 batch 2:
    // 6.09 -> 6.03; fp8 batch 8: 6.86 -> 6.77: a wide tile wants ~1.5 workgroups per CU, else two narrow ones per CU fill better)
    int wide_min = 400;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_WIDE_MIN")) wide_min = atoi(e);
#endif
    int bn = (bk == 64 && d->cout >= 128 && tiles_m128 * ((d->cout + 127) / 128) >= wide_min) ? 128 : 64;
    int s3_min = 8;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_S3_MIN")) s3_min = atoi(e);
#endif
    int stages = (bn == 64 && bk == 64 && p.Ktot / bk >= s3_min) ? 3 : 2;     // (cold-cache sweep: the third slot pays from 8 slices on)
    // short K (<= 4 slices): runs of consecutive m-tiles per workgroup -- the ring prefetches the next tile under the
    // epilogue, bias / statistics / addressing are set up once per run; narrow tiles keep two workgroups per CU
    int tpb = 1;                                                // tiles per workgroup (1: one-tile kernel)
    if (bk == 64) {
        const int kt = p.Ktot / 64;
        // (four slices and 512+ output channels: the 128 x 128 one-tile kernel is as fast or faster -- tools/tile_sweep.py, us:
        // 256 -> 512 stride 2 at M = 29,328 24.2 against 30.4, 256 -> 1024 at M = 7,488 13.5 against 13.7)
        tpb = kt == 1 ? 8 : kt == 2 ? 4 : (kt <= 4 && d->cout < 512) ? 2 : 1;
#ifdef FRCNN_SWEEP
        if (const char* e = tpb > 1 ? getenv("FRCNN_TPB_SCALE") : nullptr) { const int sc = atoi(e); tpb = sc > 0 ? tpb * sc : tpb > 1 ? tpb / (-sc) : 1; if (tpb < 1) tpb = 1; if (tpb > 16) tpb = 16; }
#endif
        const int tn64 = (d->cout + 63) / 64;
        // a run is shortened until the launch has at least ~300 workgroups: in the step (same-box A/B, batch 4, ms) 512: 4.280 / 4.287 /
        // 4.294, 400: 4.247 / 4.259, 300: 4.242 / 4.220 / 4.250, 256: 4.265 / 4.249, 200: 4.244 / 4.255, 128: 4.248 / 4.260 -- one
        // workgroup per CU with long runs beats two with short ones (64 -> 256 at M = 116,936: runs of 8 20.8 us, of 4 24.7)
        int min_groups = 300;
#ifdef FRCNN_SWEEP
        if (const char* e = getenv("FRCNN_TPB_MIN")) min_groups = atoi(e);
#endif
        while (tpb > 1 && ((tiles_m128 + tpb - 1) / tpb) * tn64 < min_groups) tpb >>= 1;
        if (tpb > 1) { bn = 64; stages = 2; }
    }
    // Long K on a small grid (the 1024 -> 256 layers of conv4 at M = 3,744: ResNet-101 at batch 2, BASELINE configs[3]): 128-row tiles give
    // 120 workgroups on 256 CUs; 64-row tiles put a workgroup on (almost) every CU.  tools/tile_sweep.py --batch=2 (round 5, us): 1x1
    // 1024 -> 256 10.1 -> 8.7; no gain on the short-K layers of that grid (256 -> 1024: 8.4 either way) or with more tiles than 0.6 x CUs.
    if (bk == 64 && tpb == 1 && bn == 64 && stages == 3 && p.taps == 1 && !p.f8_x_scale && p.Ktot / 64 < 64 && 5 * tiles_m128 * ((d->cout + 63) / 64) <= 3 * num_cus()) bm = 64;   // (64+ slices: the split-K pair form below)
#ifdef FRCNN_WIDE_N
    // 256 output channels per tile for the short-K 1x1 layers that write 4x the channels they read (64 -> 256, 128 -> 512,
    // 256 -> 1024 and the data gradients of their mirror images): a CU takes in ~37 GB/s through its load path whatever the
    // tile, so time ~ bytes ingested + stored per CU; with all 256 columns in one workgroup the A rows are fetched once
    // instead of 2-4 times.  One workgroup per CU (96 KB of LDS, ~150 VGPRs).
    if (bk == 64 && p.taps == 1 && p.Ktot <= FRCNN_WIDE_N && d->cout % 256 == 0 && tiles_m128 * (d->cout / 256) >= 200) {
        bn = 256; stages = 2; tpb = 1;
    }
#endif
    int force_kws = -1;
#ifdef FRCNN_SWEEP
    if (sweep_env().bm) { bm = sweep_env().bm; bn = sweep_env().bn; bk = sweep_env().bk; stages = sweep_env().stages; tpb = sweep_env().tpb; force_kws = 0; }
    if (sweep_env().kws >= 0) force_kws = sweep_env().kws;
#endif
    FRCNN_CHECK_ARG(d->cin % bk == 0, "conv2d_fprop: cin=%d is not a multiple of the K slice %d", d->cin, bk);
    p.k_tiles = p.Ktot / bk;
    p.k_tiles_per_split = p.k_tiles;
    p.split = 1;
    p.tiles_m = (int)((M + bm - 1) / bm);
    p.tiles_n = (d->cout + bn - 1) / bn;
    p.tiles_per_block = tpb;
    p.items = ((p.tiles_m + tpb - 1) / tpb) * p.tiles_n;
    {
        // 3x3 / stride 1 / pad 1: the kw taps share one staged A image (conv_tile_kernel, KWS)
        const bool kws_ok = p.taps == 9 && p.KW == 3 && p.stride == 1 && p.pad_h == 1 && p.pad_w == 1 && p.Hi == p.Ho && p.Wi == p.Wo &&
                            d->cin % 64 == 0 && p.direct_out;
        // measured (per-layer table of the train step): pays where two workgroups share a CU (conv2 / conv3: -8 % / -14 %);
        // with one 128 x 64 tile per CU (M = 7488) the slice time is a latency chain that the smaller fill does not shorten,
        // and the wide 128 x 128 tiles of the RPN data gradient are faster there
        // ... and not where the layer has 256+ output channels: the filter slab, not the A image, is then most of what a 128 x 64
        // tile takes in, and the plain 128 x 128 tile halves it (tools/tile_sweep.py --fpn, M = 233,872 / 58,656, 3x3 256 -> 256,
        // us: bf16 kw-shared 393 / 98 against 322 / 79; fp8 220 / 52 against 165 / 42)
        int kws_min = 160;
#ifdef FRCNN_SWEEP
        if (const char* e = getenv("FRCNN_KWS_MIN")) kws_min = atoi(e);
#endif
        const bool kws_pays = tiles_m128 >= kws_min && !(d->cout >= 256 && d->cout % 128 == 0);
        if (kws_ok && (force_kws >= 0 ? force_kws == 1 : kws_pays)) {
            p.k_tiles = p.Ktot / 64;
            p.k_tiles_per_split = p.k_tiles;
            p.tiles_m = (int)((M + 127) / 128);
            p.tiles_n = (d->cout + 63) / 64;
            p.tiles_per_block = 1;
            p.items = p.tiles_m * p.tiles_n;
            const int smode = (p.flags & FRCNN_CONV_STATS) ? 1 : (p.red_part ? 2 : 0);
            if (p.f8_x_scale) {
                if (p.f8_fmt == 1) {
                    FRCNN_CHECK_ARG(smode != 2, "conv2d fp8: the fused BatchNorm-backward reduce belongs to the data-gradient entry point");
                    if (smode == 1) return launch_tile<128, 64, 64, 3, false, 1, 2, false, false, true, false, 1>(p, s);
                    return launch_tile<128, 64, 64, 3, false, 0, 2, false, false, true, false, 1>(p, s);
                }
                FRCNN_CHECK_ARG(smode != 1, "conv2d fp8 data gradient: no forward statistics");
                if (smode == 2) return launch_tile<128, 64, 64, 3, false, 2, 2, false, false, true, false, 2>(p, s);
                return launch_tile<128, 64, 64, 3, false, 0, 2, false, false, true, false, 2>(p, s);
            }
            if (smode == 2) return launch_tile<128, 64, 64, 3, false, 2, 2, false, false, true>(p, s);
            if (smode == 1) return launch_tile<128, 64, 64, 3, false, 1, 2, false, false, true>(p, s);
            return launch_tile<128, 64, 64, 3, false, 0, 2, false, false, true>(p, s);
        }
    }
    g_last_ws_bytes = g_last_ws_counter_bytes = 0;
    {
        // fewer tiles than CUs and a very long K: split-K fix-up form (conv_tile_kernel, FIX) when the caller provides the workspace.
        // Measured per layer at M = 7,488 (tools/fix_bench.py, warm, us): 3x3 1024->256 (144 slices) 59.7 -> 52.8; 3x3 256->256
        // (36 slices) 23.4 -> 26.9; 1x1 1024->256 (16 slices) 14.6 -> 16.7 -- the exchange (partial tile written through, one atomic
        // round trip, partner's tile read back: 4-6 us in which the pair computes nothing) is paid back only from ~64 slices on,
        // because two workgroups on a CU do NOT halve the K loop: the CU's LDS-DMA path delivers ~37 GB/s whether one workgroup
        // or two feed it.
        const bool want = tpb == 1 && bm == 128 && bn == 64 && bk == 64 && stages == 3 && p.items <= num_cus() && p.k_tiles >= 64;
        const size_t ctr_bytes = (((size_t)p.items * sizeof(unsigned)) + 15) & ~(size_t)15;      // counter tail: whole 16-byte units
        const size_t need = (size_t)p.items * 2 * 128 * 64 * sizeof(float) + ctr_bytes;
        bool fix = want && d->workspace != nullptr;
#ifdef FRCNN_SWEEP
        if (const char* e = getenv("FRCNN_FIX")) fix = fix && e[0] != '0';
#endif
        if (want) { g_last_ws_bytes = need; g_last_ws_counter_bytes = ctr_bytes; }
        if (fix) {
            FRCNN_CHECK_ARG(d->workspace_bytes >= need, "conv2d_fprop: workspace of %zu bytes given, %zu needed (frcnn_conv2d_workspace_bytes)",
                            (size_t)d->workspace_bytes, need);
            FRCNN_CHECK_ARG((reinterpret_cast<size_t>(d->workspace) & 15) == 0, "conv2d_fprop: workspace must be 16-byte aligned");
            p.fix_partial = reinterpret_cast<float*>(d->workspace);
            p.fix_counter = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(d->workspace) + (size_t)p.items * 2 * 128 * 64 * sizeof(float));
            p.k_tiles_per_split = (p.k_tiles + 1) / 2;
            return launch_tile_flags<128, 64, 64, 3, 2, false, true>(p, s);
        }
    }
    int rc = FRCNN_ENOTSUP;
#define FRCNN_TILE(BM_, BN_, BK_, S_, OCC_) \
    if (rc == FRCNN_ENOTSUP && tpb == 1 && bm == BM_ && bn == BN_ && bk == BK_ && stages == S_) rc = launch_tile_flags<BM_, BN_, BK_, S_, OCC_, false>(p, s);
#define FRCNN_RUN(BM_, BN_, BK_, OCC_) \
    if (rc == FRCNN_ENOTSUP && tpb > 1 && bm == BM_ && bn == BN_ && bk == BK_ && stages == 2) rc = launch_tile_flags<BM_, BN_, BK_, 2, OCC_, true>(p, s);
    // the instantiations the heuristics above can select
    FRCNN_RUN(128, 64, 64, 2)
    FRCNN_TILE(128, 128, 64, 2, 2)
#ifdef FRCNN_WIDE_N
    FRCNN_TILE(128, 256, 64, 2, 1)
#endif
    FRCNN_TILE(128, 64, 64, 2, 3)          // (48 KB of LDS, <= 80 VGPRs: three workgroups per CU)
    FRCNN_TILE(128, 64, 64, 3, 2)
    FRCNN_TILE(64, 64, 64, 3, 2)           // (long K on a small grid, see above)
    FRCNN_TILE(128, 64, 32, 2, 2)
#ifdef FRCNN_SWEEP
    FRCNN_RUN(128, 128, 64, 1)
    FRCNN_RUN(64, 64, 64, 3)
    FRCNN_TILE(64, 128, 64, 2, 2)
    FRCNN_TILE(64, 64, 64, 2, 3)
    FRCNN_TILE(128, 128, 128, 2, 1)
    FRCNN_TILE(128, 64, 128, 2, 1)
    FRCNN_TILE(64, 128, 128, 2, 1)
    FRCNN_TILE(64, 64, 128, 2, 2)
    FRCNN_TILE(128, 128, 64, 3, 1)
    FRCNN_TILE(64, 128, 64, 3, 2)
    FRCNN_TILE(128, 64, 64, 4, 1)
    FRCNN_TILE(128, 64, 64, 6, 1)
    FRCNN_TILE(128, 128, 64, 4, 1)
    FRCNN_TILE(64, 64, 64, 6, 1)
    FRCNN_TILE(256, 128, 64, 2, 1)
    FRCNN_TILE(256, 64, 64, 2, 1)
#endif
#undef FRCNN_TILE
#undef FRCNN_RUN
    if (rc == FRCNN_ENOTSUP) {
        frcnn_set_error("conv2d_fprop: no kernel for tile %dx%dx%d, %d slots, %d tiles per workgroup", bm, bn, bk, stages, tpb);
        rc = FRCNN_EINVAL;
    }
    return rc;
}

int conv2d_fprop_impl(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* w, const float* bias, const frcnn_bf16* res,
                      const uint8_t* res_mask, void* y, double* stats_partial, const frcnn_bn_reduce* red, frcnn_stream_t stream,
                      const bool dry_run = false, const float* f8_x_scale = nullptr, const float* f8_w_scale = nullptr, const int f8_fmt = 1,
                      const frcnn_bn_in* bn_in = nullptr) {
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
    p.dry_run = dry_run ? 1 : 0;
    p.bnin_part = nullptr;
    if (bn_in) {
        FRCNN_CHECK_ARG(dry_run || (bn_in->stats_partial && bn_in->gamma && bn_in->beta && bn_in->moving_mean && bn_in->moving_var && bn_in->act &&
                                    bn_in->relu_mask && bn_in->mean && bn_in->invstd), "conv2d_fprop_bnin: null pointer in frcnn_bn_in");
        p.bnin_part = dry_run ? reinterpret_cast<const double*>(x) : bn_in->stats_partial;
        p.bnin_gamma = bn_in->gamma; p.bnin_beta = bn_in->beta;
        p.bnin_mm = bn_in->moving_mean; p.bnin_mv = bn_in->moving_var;
        p.bnin_mean = bn_in->mean; p.bnin_invstd = bn_in->invstd;
        p.bnin_act = reinterpret_cast<bf16_t*>(bn_in->act);
        p.bnin_mask = bn_in->relu_mask;
        FRCNN_CHECK_ARG(dry_run || bn_in->count > 0, "conv2d_fprop_bnin: count must be positive");
        p.bnin_momentum = bn_in->momentum; p.bnin_eps = bn_in->eps;
        p.bnin_inv_count = bn_in->count > 0 ? (float)(1.0 / (double)bn_in->count) : 0.f;                       // (as frcnn_bn_train_apply)
        p.bnin_unbias = bn_in->count > 1 ? (float)((double)bn_in->count / (double)(bn_in->count - 1)) : 1.f;
    }
    p.f8_x_scale = f8_x_scale;
    p.f8_w_scale = f8_w_scale;
    p.f8_fmt = f8_fmt;
    p.dbg = g_stamp_buffer;
    p.fix_partial = nullptr;
    p.fix_counter = nullptr;
    p.Hi = d->hi; p.Wi = d->wi; p.in_pix_stride = d->in_pix_stride; p.Cin = d->cin; p.KW = d->kw;
    p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.Ho = d->ho; p.Wo = d->wo; p.Cout = d->cout; p.out_h = d->out_h; p.out_w = d->out_w; p.out_scatter = d->out_scatter;
    p.flags = flags;
    const long long M = (long long)d->n * d->ho * d->wo;
    FRCNN_CHECK_ARG(M < (1ll << 31) - 256, "conv2d_fprop: M too large");
    p.M = (int)M;
    p.Ktot = d->kh * d->kw * d->cin;
    p.k_tiles = p.k_tiles_per_split = 0;         // set by the dispatcher together with the tile shape
    p.split = 1;
    p.tiles_m = p.tiles_n = p.items = 0;
    p.tiles_per_block = 1;
    p.taps = d->kh * d->kw;
    p.tap_mask = 1;
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
    // the ResNet stem's packed descriptor (7 tap rows of 8 pixels x 4 channels, stride 2, 64 output channels, plain bf16 output):
    // its own kernel (conv_stem_kernel)
    bool stem = d->kh == 7 && d->kw == 1 && d->cin == 32 && d->in_pix_stride == 4 && d->stride == 2 && d->pad_h == 0 && d->pad_w == 0 && d->cout == 64 &&
                p.direct_out && !(flags & ~(FRCNN_CONV_BIAS | FRCNN_CONV_STATS | FRCNN_CONV_WGRAD_ACCUMULATE | FRCNN_CONV_WGRAD_STEM_UNPACK)) && !red &&
                !f8_x_scale && !d->workspace && d->wi % 2 == 0;       // (no ADD_RES among the flags: `res` is unused, whatever it points to)
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_STEM_OLD")) { if (atoi(e)) stem = false; }
#endif
    if (stem) {
        const int units_per_row = (d->wo + 15) / 16;
        const long long total = (long long)d->n * d->ho * units_per_row;
        FRCNN_CHECK_ARG(total < (1ll << 30), "conv2d_fprop: stem grid too large");
        // two workgroups of four waves per CU (the filter bank in registers: ~190 VGPRs per lane), every wave walks its share of units
        int grid = 2 * num_cus();
        if ((long long)grid * 4 > total) grid = (int)((total + 3) / 4);
        snprintf(g_last_inst, sizeof(g_last_inst), "conv_stem<STATS=%d> grid=%dx1 tpb=1", (flags & FRCNN_CONV_STATS) ? 1 : 0, grid);
        if (dry_run) return FRCNN_OK;
        FRCNN_CHECK_ARG((reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(w) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0,
                        "conv2d_fprop(stem): operands must be 16-byte aligned");
        hipStream_t s_ = reinterpret_cast<hipStream_t>(stream);
        if (flags & FRCNN_CONV_STATS) hipLaunchKernelGGL(conv_stem_kernel<true>, dim3(grid), dim3(256), 0, s_, p, units_per_row, (int)total);
        else hipLaunchKernelGGL(conv_stem_kernel<false>, dim3(grid), dim3(256), 0, s_, p, units_per_row, (int)total);
        FRCNN_CHECK_LAUNCH("frcnn_conv2d_fprop(stem)");
        return FRCNN_OK;
    }
    // short-K 1x1 / stride-1 layers with a plain bf16 output and NO statistics (inference-mode convolutions, the RPN heads' data
    // gradient): the streaming kernel.  Measured per layer at 375x1242, batch 4 (round 4, gpurun_out/r4_layer_table3.txt, tile kernel in
    // brackets): without statistics 64 -> 64 6.3 us (7.3), 128 -> 256 at M = 7,488 4.6 (5.2); WITH the BatchNorm statistics epilogue it
    // loses -- 64 -> 256 24.2 (21.3), 128 -> 512 19.5 (16.4), 64 -> 64 10.8 (9.9): its eight waves per CU finish together and their
    // per-workgroup f64 atomics queue up on the 16 slots, and 2 waves per SIMD do not hide the per-element epilogue arithmetic; in the step
    // 4.053 -> 4.105 ms.  The statistics form stays compiled (FRCNN_SWEEP builds: FRCNN_STREAM_1X1_STATS=1) for the record.
    bool use_stream = p.linear_a && p.direct_out && (d->cin == 64 || d->cin == 128) && d->in_pix_stride == d->cin && !(flags & FRCNN_CONV_STATS) &&
                  !(flags & ~(FRCNN_CONV_BIAS | FRCNN_CONV_RELU | FRCNN_CONV_STATS | FRCNN_CONV_WGRAD_ACCUMULATE)) && !red && !f8_x_scale &&
                  !d->workspace && M >= 4096 && !bn_in;
    if (use_stream) use_stream = d->cin == 64 ? d->cout % 128 == 0 || d->cout == 64 : d->cout % 64 == 0;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_STREAM_1X1_OLD")) { if (atoi(e)) use_stream = false; }
    if (const char* e = getenv("FRCNN_STREAM_1X1_STATS")) {
        if (atoi(e) && (flags & FRCNN_CONV_STATS))
            use_stream = p.linear_a && p.direct_out && d->in_pix_stride == d->cin && !(flags & FRCNN_CONV_ADD_RES) && !red && !f8_x_scale && !d->workspace &&
                         M >= 4096 && (d->cin == 64 ? d->cout % 128 == 0 || d->cout == 64 : d->cin == 128 && d->cout % 64 == 0);
    }
#endif
    if (use_stream) {
        hipStream_t s_ = reinterpret_cast<hipStream_t>(stream);
        FRCNN_CHECK_ARG(dry_run || ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(y)) & 15) == 0,
                        "conv2d_fprop(stream 1x1): operands must be 16-byte aligned");
        p.dry_run = dry_run ? 1 : 0;
        if (d->cin == 64) return d->cout == 64 ? launch_stream_1x1<64, 64>(p, s_) : launch_stream_1x1<64, 128>(p, s_);
        return launch_stream_1x1<128, 64>(p, s_);
    }
    // 3x3 / stride 1 / pad 1 with the input patch of a spatial tile resident in LDS (conv3x3_patch_kernel): from two 64-channel chunks on
    // (a one-chunk layer -- conv2's 64 -> 64 -- is three steps long: all prologue and epilogue, the kw-sharing tile kernel keeps it)
    bool patch = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->hi == d->ho && d->wi == d->wo && d->cin % 64 == 0 &&
                 d->cout % 64 == 0 && d->in_pix_stride % 8 == 0 && p.direct_out && !f8_x_scale && d->wo >= 16 && d->ho >= 4 &&
                 !(flags & ~(FRCNN_CONV_BIAS | FRCNN_CONV_RELU | FRCNN_CONV_STATS | FRCNN_CONV_WGRAD_ACCUMULATE | FRCNN_CONV_WGRAD_STEM_UNPACK)) &&
                 !((flags & FRCNN_CONV_STATS) && red) && (long long)d->n * d->hi * d->wi * d->in_pix_stride * 2 < 0xFFFF0000ll;
    // measured per layer (tools/patch_bench.py, graph replays, us, tile kernel -> this one with loader waves): 64 output channels per
    // workgroup where that gives every workgroup a CU in ONE round -- conv4 3x3 256 -> 256 at M = 7,488 20.7 -> 15.4, its data gradient with
    // the fused reduce 21.5 -> 15.2, at M = 3,744 (ResNet-101, batch 2) 18.3 -> 13.6, conv3 128 -> 128 at batch 2 13.9 -> 11.7, the RPN's
    // 1024 -> 256 (against the split-K pair form) 55.9 -> 38.0; else 128 output channels per workgroup where that fits two rounds -- conv3
    // 128 -> 128 at batch 4 19.6 -> 16.4 (data gradient 20.6 -> 14.8), at batch 8 35.5 -> 30.0 / 36.6 -> 28.5, 256 -> 256 at M = 14,976
    // 28.1 -> 22.3, the RPN's data gradient 256 -> 1024 41.8 -> 38.0.  It LOSES with more workgroups than that (64-wide parts in two rounds:
    // conv3 at batch 4 18.8 -> 20.2) and with one chunk (conv2 64 -> 64 22.7 -> 26.5: the weights-resident form below).
    const long long patch_tiles = (long long)d->n * ((d->wo + 15) / 16) * ((d->ho + 7) / 8);
    const long long patch_wgs = patch_tiles * (d->cout / 64);
    // (two-chunk layers only from 64 workgroups on: below that there is nothing to gain, and the small-geometry model tests keep the
    // rounding of the tile kernel they were tuned with)
    bool patch_on = patch && patch_wgs <= num_cus() && !g_ws_query && (d->cin >= 256 || (d->cin >= 128 && patch_wgs >= 64));
    int patch_sb = 4, patch_bn = 64;
    if (patch && !patch_on && d->cin >= 128 && d->cout % 128 == 0 && patch_tiles * (d->cout / 128) <= 2 * num_cus() && patch_tiles * (d->cout / 128) >= 64 &&
        !g_ws_query) {
        patch_on = true;
        patch_bn = 128;
    }
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_PATCH")) { patch_on = patch && atoi(e) != 0; if (atoi(e) == 2) patch_on = patch_on && (d->cin >= 256 && patch_wgs <= num_cus()); }
    if (const char* e = getenv("FRCNN_PATCH_BN")) { if (d->cout % 128 == 0 && atoi(e)) patch_bn = atoi(e) == 128 ? 128 : 64; }
    if (const char* e = getenv("FRCNN_PATCH_SB")) patch_sb = atoi(e);
#endif
    // ... and the one-chunk layers (64 input channels: conv2) on the weights-resident form of the same tiling (conv3x3_wres_kernel)
    bool wres_on = patch && d->cin == 64 && d->in_pix_stride % 8 == 0 && (long long)d->n * ((d->wo + 15) / 16) * ((d->ho + 7) / 8) >= 2ll * num_cus() / (d->cout / 64);
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_WRES")) wres_on = patch && d->cin == 64 && atoi(e) != 0;
#endif
    bool patch_bnin = false;
    // ... a 1x1 / stride-1 layer (the block's third convolution) takes the BatchNorm of its input on the tile kernel's two-slot forms
    const bool tile_bnin = bn_in && !patch && p.linear_a && d->kh == 1 && d->in_pix_stride == d->cin && d->cin % 64 == 0 && d->cin <= 512 && !red && !f8_x_scale &&
                           !d->workspace && !(flags & ~(FRCNN_CONV_BIAS | FRCNN_CONV_RELU | FRCNN_CONV_STATS));
    if (bn_in && !wres_on && !tile_bnin) {
        // ... or on the patch-resident kernel's loader-wave forms (conv3 / conv4 at the benchmark's sizes)
        patch_bnin = patch_on && d->in_pix_stride == d->cin && d->cin <= 512 && !red && (patch_bn == 128 || patch_sb == 4);
#ifdef FRCNN_SWEEP
        if (const char* e = getenv("FRCNN_PATCH_LW")) { if (atoi(e) != 4 && patch_bn != 128) patch_bnin = false; }
#endif
        if (!patch_bnin) {
            frcnn_set_error("conv2d_fprop_bnin: only 3x3 / stride 1 / pad 1 layers that run on the weights-resident kernel (64 input channels) or on "
                            "the patch-resident kernel's loader-wave forms (frcnn_conv2d_bnin_supported)");
            return FRCNN_EINVAL;
        }
    }
    if (wres_on) {
        FRCNN_CHECK_ARG(!bn_in || d->in_pix_stride == 64, "conv2d_fprop_bnin: the input must be the dense [M][64] output of the previous convolution");
        FRCNN_CHECK_ARG(dry_run || ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(y)) & 15) == 0,
                        "conv2d_fprop(3x3, weights resident): operands must be 16-byte aligned");
        g_last_ws_bytes = g_last_ws_counter_bytes = 0;
        return launch_wres(p, reinterpret_cast<hipStream_t>(stream), d->n);
    }
    if (patch_on) {
        FRCNN_CHECK_ARG(dry_run || ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(y)) & 15) == 0,
                        "conv2d_fprop(patch 3x3): operands must be 16-byte aligned");
        g_last_ws_bytes = g_last_ws_counter_bytes = 0;
        return launch_patch(p, reinterpret_cast<hipStream_t>(stream), d->n, patch_sb, patch_bn);
    }
    return conv_tile_dispatch(p, d, reinterpret_cast<hipStream_t>(stream));
}

}  // namespace

extern "C" const char* frcnn_last_conv_instantiation(void) { return g_last_inst; }
#ifdef FRCNN_STAMPS
extern "C" void frcnn_debug_set_stamp_buffer(unsigned long long* dev_buffer) { g_stamp_buffer = dev_buffer; }
#endif
void frcnn_note_instantiation(const char* s) { snprintf(g_last_inst, sizeof(g_last_inst), "%s", s); }    // (conv_wgrad.hip)

extern "C" const char* frcnn_conv2d_describe(const frcnn_conv_desc* d, int with_bn_reduce) {
    // the dispatch decision for this descriptor without launching anything (no device needed): dummy non-null operands
    static const uint8_t dummy[16] = {0};
    frcnn_bn_reduce red;
    red.z = reinterpret_cast<const frcnn_bf16*>(dummy);
    red.relu_mask = nullptr;
    red.mean = red.invstd = reinterpret_cast<const float*>(dummy);
    red.partial = const_cast<float*>(reinterpret_cast<const float*>(dummy));
    const void* q = dummy;
    g_last_inst[0] = 0;
    const int rc = conv2d_fprop_impl(d, reinterpret_cast<const frcnn_bf16*>(q), reinterpret_cast<const frcnn_bf16*>(q), reinterpret_cast<const float*>(q),
                                     reinterpret_cast<const frcnn_bf16*>(q), nullptr, const_cast<void*>(q),
                                     const_cast<double*>(reinterpret_cast<const double*>(q)), with_bn_reduce ? &red : nullptr, nullptr, true);
    return rc == FRCNN_OK ? g_last_inst : nullptr;
}

// One descriptor serves the bf16 and the fp8 entry points (the models pass the same frcnn_conv_desc to either), and only the tile kernel
// has a workspace form: the answer is what THAT kernel would use -- the patch-resident 3x3 kernel, which takes some of these layers in
// bf16, ignores the workspace it is handed.
extern "C" size_t frcnn_conv2d_workspace_bytes(const frcnn_conv_desc* d) {
    g_last_ws_bytes = g_last_ws_counter_bytes = 0;
    g_ws_query = true;
    const bool ok = d && frcnn_conv2d_describe(d, 0);
    g_ws_query = false;
    return ok ? g_last_ws_bytes : 0;
}

extern "C" size_t frcnn_conv2d_workspace_counter_bytes(const frcnn_conv_desc* d) {
    g_last_ws_bytes = g_last_ws_counter_bytes = 0;
    g_ws_query = true;
    const bool ok = d && frcnn_conv2d_describe(d, 0);
    g_ws_query = false;
    return ok ? g_last_ws_counter_bytes : 0;
}

extern "C" int frcnn_conv2d_stat_tiles(const frcnn_conv_desc* d) {
    if (!d) return FRCNN_EINVAL;
    return FRCNN_STAT_SLOTS;
}

extern "C" int frcnn_conv2d_fprop(const frcnn_conv_desc* d, const frcnn_bf16* x, const frcnn_bf16* w, const float* bias,
                                  const frcnn_bf16* res, void* y, double* stats_partial, frcnn_stream_t stream) {
    return conv2d_fprop_impl(d, x, w, bias, res, nullptr, y, stats_partial, nullptr, stream);
}

extern "C" int frcnn_conv2d_fprop_bnin(const frcnn_conv_desc* d, const frcnn_bf16* z_in, const frcnn_bf16* w, const float* bias, frcnn_bf16* y,
                                       double* stats_partial, const frcnn_bn_in* bn, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(bn, "conv2d_fprop_bnin: null frcnn_bn_in");
    FRCNN_CHECK_ARG(d && !(d->flags & (FRCNN_CONV_ADD_RES | FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC)), "conv2d_fprop_bnin: plain bf16 output only");
    return conv2d_fprop_impl(d, z_in, w, bias, nullptr, nullptr, y, stats_partial, nullptr, stream, false, nullptr, nullptr, 1, bn);
}

extern "C" int frcnn_conv2d_bnin_supported(const frcnn_conv_desc* d) {
    if (!d) return 0;
    static const uint8_t dummy[16] = {0};
    const void* q = dummy;
    frcnn_bn_in bn;
    memset(&bn, 0, sizeof(bn));
    const int rc = conv2d_fprop_impl(d, reinterpret_cast<const frcnn_bf16*>(q), reinterpret_cast<const frcnn_bf16*>(q), reinterpret_cast<const float*>(q), nullptr,
                                     nullptr, const_cast<void*>(q), const_cast<double*>(reinterpret_cast<const double*>(q)), nullptr, nullptr, true, nullptr, nullptr,
                                     1, &bn);
    return rc == FRCNN_OK ? 1 : 0;
}

// fp8 (OCP e4m3) operands: the geometry with every element count halved IS the bf16 kernel's geometry in bytes (conv_tile_kernel, F8)
static int fp8_halved_desc(const frcnn_conv_desc* d, frcnn_conv_desc* h, const char* who, const bool dgrad = false) {
    FRCNN_CHECK_ARG(d, "%s: null descriptor", who);
    FRCNN_CHECK_ARG(d->cin % 128 == 0 && d->in_pix_stride % 128 == 0, "%s: cin=%d / in_pix_stride=%d must be multiples of 128 (128-deep fp8 MFMA steps)",
                    who, d->cin, d->in_pix_stride);
    FRCNN_CHECK_ARG(!(d->flags & (FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC)) && d->split_k <= 1 && (dgrad || !(d->flags & FRCNN_CONV_ADD_RES)),
                    "%s: bf16 output, no split-K (a residual only in the data-gradient form)", who);
    *h = *d;
    h->cin = d->cin / 2;
    h->in_pix_stride = d->in_pix_stride / 2;
    return FRCNN_OK;
}

extern "C" int frcnn_conv2d_fprop_fp8(const frcnn_conv_desc* d, const frcnn_fp8* x8, const frcnn_fp8* w8, const float* x_scale,
                                      const float* w_scale, const float* bias, frcnn_bf16* y, double* stats_partial, frcnn_stream_t stream) {
    frcnn_conv_desc h;
    const int rc = fp8_halved_desc(d, &h, "conv2d_fprop_fp8");
    if (rc != FRCNN_OK) return rc;
    FRCNN_CHECK_ARG(x_scale && w_scale, "conv2d_fprop_fp8: null scale pointer");
    return conv2d_fprop_impl(&h, reinterpret_cast<const frcnn_bf16*>(x8), reinterpret_cast<const frcnn_bf16*>(w8), bias, nullptr, nullptr, y,
                             stats_partial, nullptr, stream, false, x_scale, w_scale);
}

extern "C" int frcnn_conv2d_dgrad_fp8(const frcnn_conv_desc* d, const frcnn_fp8* dz8, const frcnn_fp8* w_t8, const float* dz_scale,
                                      const float* w_scale, const frcnn_bf16* res, const uint8_t* res_mask, frcnn_bf16* gx,
                                      const frcnn_bn_reduce* red, frcnn_stream_t stream) {
    frcnn_conv_desc h;
    const int rc = fp8_halved_desc(d, &h, "conv2d_dgrad_fp8", true);
    if (rc != FRCNN_OK) return rc;
    FRCNN_CHECK_ARG(dz_scale && w_scale, "conv2d_dgrad_fp8: null scale pointer");
    FRCNN_CHECK_ARG(!(d->flags & (FRCNN_CONV_STATS | FRCNN_CONV_BIAS | FRCNN_CONV_RELU)), "conv2d_dgrad_fp8: only ADD_RES may be set");
    FRCNN_CHECK_ARG(!res_mask || (res && (d->flags & FRCNN_CONV_ADD_RES)), "conv2d_dgrad_fp8: res_mask without ADD_RES residual");
    FRCNN_CHECK_ARG(!red || (red->z && red->mean && red->invstd && red->partial), "conv2d_dgrad_fp8: incomplete reduce arguments");
    return conv2d_fprop_impl(&h, reinterpret_cast<const frcnn_bf16*>(dz8), reinterpret_cast<const frcnn_bf16*>(w_t8), nullptr, res, res_mask, gx,
                             nullptr, red, stream, false, dz_scale, w_scale, 2);
}

extern "C" const char* frcnn_conv2d_describe_dgrad_fp8(const frcnn_conv_desc* d, int with_bn_reduce) {
    static const uint8_t dummy[16] = {0};
    frcnn_conv_desc h;
    g_last_inst[0] = 0;
    if (fp8_halved_desc(d, &h, "conv2d_describe_dgrad_fp8", true) != FRCNN_OK) return nullptr;
    frcnn_bn_reduce red;
    red.z = reinterpret_cast<const frcnn_bf16*>(dummy);
    red.relu_mask = nullptr;
    red.mean = red.invstd = reinterpret_cast<const float*>(dummy);
    red.partial = const_cast<float*>(reinterpret_cast<const float*>(dummy));
    const void* q = dummy;
    const int rc = conv2d_fprop_impl(&h, reinterpret_cast<const frcnn_bf16*>(q), reinterpret_cast<const frcnn_bf16*>(q), nullptr,
                                     reinterpret_cast<const frcnn_bf16*>(q), nullptr, const_cast<void*>(q), nullptr, with_bn_reduce ? &red : nullptr,
                                     nullptr, true, reinterpret_cast<const float*>(q), reinterpret_cast<const float*>(q), 2);
    return rc == FRCNN_OK ? g_last_inst : nullptr;
}

extern "C" const char* frcnn_conv2d_describe_fp8(const frcnn_conv_desc* d) {
    static const uint8_t dummy[16] = {0};
    frcnn_conv_desc h;
    g_last_inst[0] = 0;
    if (fp8_halved_desc(d, &h, "conv2d_describe_fp8") != FRCNN_OK) return nullptr;
    const void* q = dummy;
    const int rc = conv2d_fprop_impl(&h, reinterpret_cast<const frcnn_bf16*>(q), reinterpret_cast<const frcnn_bf16*>(q), reinterpret_cast<const float*>(q),
                                     nullptr, nullptr, const_cast<void*>(q), const_cast<double*>(reinterpret_cast<const double*>(q)), nullptr, nullptr,
                                     true, reinterpret_cast<const float*>(q), reinterpret_cast<const float*>(q));
    return rc == FRCNN_OK ? g_last_inst : nullptr;
}

extern "C" int frcnn_conv2d_dgrad_bnreduce(const frcnn_conv_desc* d, const frcnn_bf16* dz, const frcnn_bf16* w_t, const frcnn_bf16* res,
                                           const uint8_t* res_mask, frcnn_bf16* gx, const frcnn_bn_reduce* red, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(!res_mask || (res && d && (d->flags & FRCNN_CONV_ADD_RES)), "conv2d_dgrad_bnreduce: res_mask without ADD_RES residual");
    FRCNN_CHECK_ARG(red && red->z && red->mean && red->invstd && red->partial, "conv2d_dgrad_bnreduce: incomplete reduce arguments");
    FRCNN_CHECK_ARG(d && !(d->flags & (FRCNN_CONV_STATS | FRCNN_CONV_OUT_F32 | FRCNN_CONV_SPLITK_ATOMIC | FRCNN_CONV_BIAS | FRCNN_CONV_RELU)),
                    "conv2d_dgrad_bnreduce: only ADD_RES may be set");
    return conv2d_fprop_impl(d, dz, w_t, nullptr, res, res_mask, gx, nullptr, red, stream);
}
