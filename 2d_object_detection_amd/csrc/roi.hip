// Fused RoI pooling of the reference's ROIPooling layer (fast_rcnn_detector.py:133-177):
// tf.image.crop_and_resize (bilinear, 14x14, extrapolation 0) + MaxPool 2x2 in ONE pass that never
// materialises the 14x14 crop (241 MB / image in the reference): HBM traffic is the feature map
// (L2 / Infinity-Cache resident) plus the pooled output.
//
// Forward: one workgroup per RoI; lanes run over 8-channel vectors (16-byte loads of the NHWC
// feature map), each work item evaluates the 2x2 bilinear samples of one pooled bin and keeps the
// max and its window position.  Backward: one workgroup per *sampled* RoI row, lanes over single
// channels so that every float-atomic wave instruction adds 256 contiguous bytes of one
// feature-map pixel.
#include "common.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace {

struct RoiGeom { float y1s, x1s, hs, ws; };   // in = y1s + i*hs  (TF crop_and_resize coordinate map)

__device__ __forceinline__ RoiGeom roi_geom(const float* roi /*x1,y1,x2,y2 rel*/, int Hf, int Wf, int crop) {
    const float x1 = roi[0], y1 = roi[1], x2 = roi[2], y2 = roi[3];
    const float hm1 = (float)(Hf - 1), wm1 = (float)(Wf - 1);
    RoiGeom g;
    g.hs = (y2 - y1) * hm1 / (float)(crop - 1);
    g.ws = (x2 - x1) * wm1 / (float)(crop - 1);
    g.y1s = y1 * hm1;
    g.x1s = x1 * wm1;
    return g;
}

// KS > 0: pooling window fixed at compile time -- the KS*KS samples of a bin are unrolled, so their 4*KS*KS tap loads are
// all in flight before the first interpolation.  The kernel is VALU-bound (rocprofv3 --pmc: 788 VALU instructions per
// (bin, 8-channel vector) item before this form): crop_and_resize is separable, so the row terms (top / bottom row offset,
// y weight, validity) of the crop's rows and the column terms of its columns are computed once per workgroup into LDS
// instead of once per item and sample, and the three interpolations are single FMAs (fmaf is explicit: the file keeps
// implicit contraction off for the bit-exact box arithmetic elsewhere; the pooled output is bf16 and the oracle comparison
// carries the matching tolerance).
constexpr int kMaxCrop = 64;
template <int KS>
__global__ __launch_bounds__(256) void roi_fwd_kernel(const bf16_t* __restrict__ feat, const float* __restrict__ rois, int P, int Hf, int Wf,
                                                      int C8, int ps, int ks_rt, bf16_t* __restrict__ pooled, uint8_t* __restrict__ amax, int nsplit,
                                                      int npairs, const int* __restrict__ levels, int level) {
    const int ks = KS > 0 ? KS : ks_rt;
    // workgroup -> (RoI, channel slice).  Workgroups b, b + 8, ... share an XCD (one L2 of 4 MiB): the grid is laid out so that an XCD
    // only ever sees ONE (image, channel slice) pair -- 1/8 of all feature maps, 1.9 MB at batch 4 -- instead of every XCD pulling all
    // maps through its L2 (15.3 MB at batch 4: 262 MB fetched from beyond L2 per launch for a 15.3 MB tensor, 17x, measured in round 2).
    // nsplit channel slices per RoI, B * nsplit pairs dealt round-robin over the 8 XCDs (speed only: any placement is correct).
    int row, cv_begin, cv_count;
    {
        const int lanes_x = npairs < 8 ? npairs : 8;           // XCD lanes in use; npairs = B * nsplit
        const int xcd = blockIdx.x % lanes_x, q = blockIdx.x / lanes_x;
        const int pair = xcd + lanes_x * (q / P);              // an XCD lane runs its pairs one after the other
        const int p_ = q % P;
        if (pair >= npairs) return;
        const int b_ = pair / nsplit, slice = pair - b_ * nsplit;
        row = b_ * P + p_;
        cv_count = C8 / nsplit;
        cv_begin = slice * cv_count;
    }
    const int b = row / P;
    const int crop = ps * ks;
    if (levels && levels[row] != level) return;                // feature pyramid: this launch pools another level's RoIs
    __shared__ int4 ys[kMaxCrop], xs[kMaxCrop];               // {first tap byte offset, second tap byte offset, weight bits, valid}
    if ((int)threadIdx.x < 2 * crop) {
        const RoiGeom g = roi_geom(rois + (int64_t)row * 4, Hf, Wf, crop);
        const bool is_x = (int)threadIdx.x >= crop;
        const int k = is_x ? threadIdx.x - crop : threadIdx.x;
        const float in = is_x ? g.x1s + (float)k * g.ws : g.y1s + (float)k * g.hs;
        const float lim = is_x ? (float)(Wf - 1) : (float)(Hf - 1);
        const bool ok = in >= 0.f && in <= lim;                // NaN-safe
        const float lo = floorf(in), hi = ceilf(in), w = in - lo;
        const int pitch = (is_x ? 1 : Wf) * C8 * 16;           // bytes per column / per row of the NHWC map
        int4 e;
        e.x = ok ? (int)lo * pitch : 0;
        e.y = ok ? (int)hi * pitch : 0;
        e.z = __float_as_int(w);
        e.w = ok ? 1 : 0;
        if (is_x) xs[k] = e;
        else ys[k] = e;
    }
    __syncthreads();
    const unsigned char* fb = reinterpret_cast<const unsigned char*>(feat + (int64_t)b * Hf * Wf * C8 * 8);
    const int items = ps * ps * cv_count;
    if (KS == 2) {
        // 2 x 2 pooling window, fully unrolled.  Taps come through a buffer descriptor of this image's map: 32-bit offsets (no
        // 64-bit address arithmetic per tap), and an invalid sample points all four taps beyond the descriptor -- the range check
        // returns zeros and the interpolation of zeros with a zeroed weight is exactly 0, the extrapolation value -- so the loop
        // has no branch.  The maximum is a v_max3 + v_max; the arg-max (first maximum, as the strict '>' scan gives) is
        // recovered afterwards from the four kept samples.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)fb, 0, Hf * Wf * C8 * 16, 0x00020000);
        constexpr unsigned kBeyond = 0xFFFFFFF0u;
        for (int it = threadIdx.x; it < items; it += blockDim.x) {
            const int cv = cv_begin + it % cv_count;
            const int bin = it / cv_count;
            const int ph = bin / ps, pw = bin - ph * ps;
            const unsigned c16 = (unsigned)cv * 16u;
            float v[4][8];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int4 yy = ys[ph * 2 + (s_ >> 1)], xx = xs[pw * 2 + (s_ & 1)];
                const bool ok = (yy.w & xx.w) != 0;
                const float ly = ok ? __int_as_float(yy.z) : 0.f, lx = ok ? __int_as_float(xx.z) : 0.f;
                float tl[8], tr[8], bl[8], br[8];
                unpack8(__builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(yy.x + xx.x) + c16 : kBeyond, 0, 0), tl);
                unpack8(__builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(yy.x + xx.y) + c16 : kBeyond, 0, 0), tr);
                unpack8(__builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(yy.y + xx.x) + c16 : kBeyond, 0, 0), bl);
                unpack8(__builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(yy.y + xx.y) + c16 : kBeyond, 0, 0), br);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float top = fmaf(tr[e] - tl[e], lx, tl[e]);
                    const float bot = fmaf(br[e] - bl[e], lx, bl[e]);
                    v[s_][e] = fmaf(bot - top, ly, top);
                }
            }
            float best[8];
            unsigned arg[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                best[e] = fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0][e], v[1][e]), v[2][e]), v[3][e]);
                arg[e] = v[0][e] == best[e] ? 0u : v[1][e] == best[e] ? 1u : v[2][e] == best[e] ? 2u : 3u;
            }
            const int64_t o = ((int64_t)row * ps * ps + bin) * C8 + cv;
            *reinterpret_cast<u32x4*>(pooled + o * 8) = pack8(best);
            u32x2 a;
            a[0] = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
            a[1] = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
            *reinterpret_cast<u32x2*>(amax + o * 8) = a;
        }
        return;
    }
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int cv = cv_begin + it % cv_count;
        const int bin = it / cv_count;
        const int ph = bin / ps, pw = bin - ph * ps;
        float best[8];
        unsigned char arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = 0; }
        for (int s = 0; s < ks * ks; ++s) {
            const int4 yy = ys[ph * ks + s / ks], xx = xs[pw * ks + s % ks];
            float v[8];
            if (!(yy.w & xx.w)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            } else {
                const float ly = __int_as_float(yy.z), lx = __int_as_float(xx.z);
                const int c16 = cv * 16;
                float tl[8], tr[8], bl[8], br[8];
                unpack8(*reinterpret_cast<const u32x4*>(fb + (yy.x + xx.x + c16)), tl);
                unpack8(*reinterpret_cast<const u32x4*>(fb + (yy.x + xx.y + c16)), tr);
                unpack8(*reinterpret_cast<const u32x4*>(fb + (yy.y + xx.x + c16)), bl);
                unpack8(*reinterpret_cast<const u32x4*>(fb + (yy.y + xx.y + c16)), br);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float top = fmaf(tr[e] - tl[e], lx, tl[e]);
                    const float bot = fmaf(br[e] - bl[e], lx, bl[e]);
                    v[e] = fmaf(bot - top, ly, top);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (v[e] > best[e]) { best[e] = v[e]; arg[e] = (unsigned char)s; }
        }
        const int64_t o = ((int64_t)row * ps * ps + bin) * C8 + cv;
        *reinterpret_cast<u32x4*>(pooled + o * 8) = pack8(best);
        u32x2 a;
        a[0] = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((unsigned)arg[3] << 24);
        a[1] = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((unsigned)arg[7] << 24);
        *reinterpret_cast<u32x2*>(amax + o * 8) = a;
    }
}

// ---- 2 x 2 pooling window, round 4 form.  The round-3 kernel above spends 527 VALU instructions per (bin, 8-channel vector) item and is
// VALU-bound (rocprofv3 --pmc, profiles/r03_c_pmc_counters.txt).  Three changes, same arithmetic per element (bit-identical output):
//   1. A wave owns whole bins (lanes = channel vectors), so everything that depends on the bin alone -- tap offsets, interpolation
//      weights, validity -- is WAVE-UNIFORM: read once per wave from the LDS tables into scalar registers instead of per lane.  (A two-bins-per-wave form for
//      32-vector slices was built and measured SLOWER than the per-item kernel -- 122 vs 85 us on the pyramid's 256-channel maps, 220
//      vs 182 us at 1000 proposals: per-lane selects between two scalar sets and half the tap sharing; such slices stay on roi_fwd_kernel.)
//   2. The 2 x 2 samples of a bin lie a fraction of a cell apart whenever the proposal is small: when both sample rows fall between the
//      same two feature rows (and / or both columns between the same two feature columns) the taps are the same cells -- loaded and
//      unpacked once (4 or 8 taps instead of 16), and the horizontal interpolations of a shared row pair are computed once per sample
//      column instead of once per sample.  The case is a wave-uniform branch.
//   3. The interpolations run on channel PAIRS (v_pk_add_f32 / v_pk_fma_f32).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool RS, bool CS>
__device__ __forceinline__ void roi_bin_ks2(const __amdgpu_buffer_rsrc_t rs, const unsigned c16, const unsigned (&rowoff)[4], const unsigned (&coloff)[4],
                                            const float (&ly)[2], const float (&lx)[2], bf16_t* __restrict__ pooled_o, uint8_t* __restrict__ amax_o) {
    constexpr int NR = RS ? 2 : 4, NC = CS ? 2 : 4;
    f32x2 f[NR][NC][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // (the whole offset in the VECTOR operand: the descriptor's range check -- which turns an outside sample into zeros -- is
            // specified on the vector offset; a scalar offset would be subtracted from the record count instead)
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, rowoff[r] + coloff[c] + c16, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) f[r][c][q] = f32x2{__uint_as_float(raw[q] << 16), __uint_as_float(raw[q] & 0xFFFF0000u)};
        }
    }
    // horizontal interpolation of every (feature row, sample column) pair in use
    f32x2 hz[NR][2][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int clo = CS ? 0 : 2 * j, chi = CS ? 1 : 2 * j + 1;
            const f32x2 w = {lx[j], lx[j]};
#pragma unroll
            for (int q = 0; q < 4; ++q) hz[r][j][q] = __builtin_elementwise_fma(f[r][chi][q] - f[r][clo][q], w, f[r][clo][q]);
        }
    }
    float v[4][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rlo = RS ? 0 : 2 * i, rhi = RS ? 1 : 2 * i + 1;
        const f32x2 w = {ly[i], ly[i]};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 t = __builtin_elementwise_fma(hz[rhi][j][q] - hz[rlo][j][q], w, hz[rlo][j][q]);
                v[i * 2 + j][2 * q] = t[0];
                v[i * 2 + j][2 * q + 1] = t[1];
            }
        }
    }
    float best[8];
    unsigned arg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        best[e] = fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0][e], v[1][e]), v[2][e]), v[3][e]);
        arg[e] = v[0][e] == best[e] ? 0u : v[1][e] == best[e] ? 1u : v[2][e] == best[e] ? 2u : 3u;
    }
    *reinterpret_cast<u32x4*>(pooled_o) = pack8(best);
    u32x2 a;
    a[0] = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
    a[1] = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
    // (the arg-max bytes are read again only by the backward pass, 60 MB per batch-4 step: a non-temporal store keeps them from pushing the
    // pooled rows -- which the head GEMM reads next -- out of the memory-side cache; same-box A/B 3.906 / 3.899 -> 3.887 / 3.893 ms)
    __builtin_nontemporal_store(a, reinterpret_cast<u32x2*>(amax_o));
}

// channel-vector slices in multiples of 64: a wave = one bin x 64 channel vectors
__global__ __launch_bounds__(256) void roi_fwd_ks2_kernel(const bf16_t* __restrict__ feat, const float* __restrict__ rois, int P, int Hf, int Wf, int C8,
                                                          int ps, bf16_t* __restrict__ pooled, uint8_t* __restrict__ amax, int nsplit, int npairs,
                                                          const int* __restrict__ levels, int level) {
    int row, cv_begin, cv_count;
    {   // workgroup -> (RoI, channel slice): as roi_fwd_kernel (the (image, slice) pairs dealt over the XCD lanes)
        const int lanes_x = npairs < 8 ? npairs : 8;
        const int xcd = blockIdx.x % lanes_x, q = blockIdx.x / lanes_x;
        const int pair = xcd + lanes_x * (q / P);
        const int p_ = q % P;
        if (pair >= npairs) return;
        const int b_ = pair / nsplit, slice = pair - b_ * nsplit;
        row = b_ * P + p_;
        cv_count = C8 / nsplit;
        cv_begin = slice * cv_count;
    }
    const int b = row / P;
    const int crop = ps * 2;
    if (levels && levels[row] != level) return;
    // {first tap byte offset, second tap byte offset, weight bits} per crop row / column; a sample outside the map points both taps
    // 1 GiB beyond the descriptor: the range check returns zeros for all four taps and the interpolation of zeros is exactly 0
    __shared__ int4 ys[kMaxCrop], xs[kMaxCrop];
    if ((int)threadIdx.x < 2 * crop) {
        const RoiGeom g = roi_geom(rois + (int64_t)row * 4, Hf, Wf, crop);
        const bool is_x = (int)threadIdx.x >= crop;
        const int k = is_x ? threadIdx.x - crop : threadIdx.x;
        const float in = is_x ? g.x1s + (float)k * g.ws : g.y1s + (float)k * g.hs;
        const float lim = is_x ? (float)(Wf - 1) : (float)(Hf - 1);
        const bool ok = in >= 0.f && in <= lim;                // NaN-safe
        const float lo = floorf(in), hi = ceilf(in), w = in - lo;
        const int pitch = (is_x ? 1 : Wf) * C8 * 16;
        int4 e;
        e.x = ok ? (int)lo * pitch : 0x40000000;
        e.y = ok ? (int)hi * pitch : 0x40000000;
        e.z = __float_as_int(ok ? w : 0.f);
        e.w = 0;
        if (is_x) xs[k] = e;
        else ys[k] = e;
    }
    __syncthreads();
    const unsigned char* fb = reinterpret_cast<const unsigned char*>(feat + (int64_t)b * Hf * Wf * C8 * 8);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)fb, 0, Hf * Wf * C8 * 16, 0x00020000);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int nbins = ps * ps, blocks = cv_count / 64;
    for (int u = wave; u < nbins * blocks; u += nwaves) {
        const int bin = u / blocks;
        const int ph = bin / ps, pw = bin - ph * ps;
        // the bin's scalar set: 4 row offsets, 4 column offsets (lo / hi of its two sample rows / columns), 2 + 2 weights
        unsigned rowoff[4], coloff[4];
        float ly[2], lx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int4 yy = ys[ph * 2 + i], xx = xs[pw * 2 + i];
            rowoff[2 * i] = (unsigned)__builtin_amdgcn_readfirstlane(yy.x);
            rowoff[2 * i + 1] = (unsigned)__builtin_amdgcn_readfirstlane(yy.y);
            ly[i] = __int_as_float(__builtin_amdgcn_readfirstlane(yy.z));
            coloff[2 * i] = (unsigned)__builtin_amdgcn_readfirstlane(xx.x);
            coloff[2 * i + 1] = (unsigned)__builtin_amdgcn_readfirstlane(xx.y);
            lx[i] = __int_as_float(__builtin_amdgcn_readfirstlane(xx.z));
        }
        const bool rsame = rowoff[0] == rowoff[2] && rowoff[1] == rowoff[3];
        const bool csame = coloff[0] == coloff[2] && coloff[1] == coloff[3];
        const int cv = cv_begin + (u - bin * blocks) * 64 + lane;
        const int64_t o = ((int64_t)row * nbins + bin) * C8 + cv;
        const unsigned c16 = (unsigned)cv * 16u;
        if (rsame) {                                             // (shared rows: slots 0 / 1 serve both samples)
            if (csame) roi_bin_ks2<true, true>(rs, c16, rowoff, coloff, ly, lx, pooled + o * 8, amax + o * 8);
            else roi_bin_ks2<true, false>(rs, c16, rowoff, coloff, ly, lx, pooled + o * 8, amax + o * 8);
        } else {
            if (csame) roi_bin_ks2<false, true>(rs, c16, rowoff, coloff, ly, lx, pooled + o * 8, amax + o * 8);
            else roi_bin_ks2<false, false>(rs, c16, rowoff, coloff, ly, lx, pooled + o * 8, amax + o * 8);
        }
    }
}

__global__ __launch_bounds__(256) void roi_bwd_kernel(const bf16_t* __restrict__ gpooled, const uint8_t* __restrict__ amax,
                                                      const float* __restrict__ rois, const int* __restrict__ rows, int P, int Hf, int Wf,
                                                      int C, int ps, int ks, float* __restrict__ gfeat) {
    const int r = blockIdx.x;                 // dense row of gpooled
    const int row = rows[r];                  // RoI row b*P + p
    const int b = row / P;
    const int crop = ps * ks;
    const RoiGeom g = roi_geom(rois + (int64_t)row * 4, Hf, Wf, crop);
    const float hm1 = (float)(Hf - 1), wm1 = (float)(Wf - 1);
    float* gb = gfeat + (int64_t)b * Hf * Wf * C;
    const int items = ps * ps * C;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int c = it % C;
        const int bin = it / C;
        const int ph = bin / ps, pw = bin - ph * ps;
        const float gv = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(gpooled + (int64_t)r * items + it));
        if (gv == 0.f) continue;
        const int s = amax[(int64_t)row * items + it];
        const int i = ph * ks + s / ks, j = pw * ks + s % ks;
        const float in_y = g.y1s + (float)i * g.hs;
        const float in_x = g.x1s + (float)j * g.ws;
        if (!(in_y >= 0.f && in_y <= hm1 && in_x >= 0.f && in_x <= wm1)) continue;
        const float ty = floorf(in_y), by = ceilf(in_y), ly = in_y - ty;
        const float lx_ = floorf(in_x), rx = ceilf(in_x), lx = in_x - lx_;
        const int t = (int)ty, bo = (int)by, l = (int)lx_, rr = (int)rx;
        const float dtop = (1.f - ly) * gv, dbot = ly * gv;
        atomicAdd(gb + ((int64_t)t * Wf + l) * C + c, (1.f - lx) * dtop);
        atomicAdd(gb + ((int64_t)t * Wf + rr) * C + c, lx * dtop);
        atomicAdd(gb + ((int64_t)bo * Wf + l) * C + c, (1.f - lx) * dbot);
        atomicAdd(gb + ((int64_t)bo * Wf + rr) * C + c, lx * dbot);
    }
}

// Backward without global atomics: one workgroup owns one feature-map row of one image and a slab of CS channels, keeps
// that row's gradient in LDS ([Wf][CS] fp32), walks the image's sampled RoI rows, skips every pooled bin row whose 2x2
// sample rows cannot touch its feature row (uniform test), and adds the y-weighted contributions of the rest with LDS
// atomics; the finished row leaves as bf16 with plain coalesced stores.  Every element of gfeat is written exactly once
// (no pre-zeroing, no fp32 intermediate, no cast pass), and the only global traffic is the pooled gradient / argmax rows
// (L2-resident, re-read ~3x: a pooled bin row touches 2-3 feature rows) instead of 4 float atomics per element at the
// memory side's 1.3 TB/s.  Lanes hold channel PAIRS (one 4-byte + one 2-byte load per item).
template <int CS>
__global__ __launch_bounds__(256) void roi_bwd_rows_kernel(const bf16_t* __restrict__ gpooled, const uint8_t* __restrict__ amax,
                                                           const float* __restrict__ rois, const int* __restrict__ rows, int nrows, int P,
                                                           int Hf, int Wf, int C, int ps, int ks, bf16_t* __restrict__ gfeat,
                                                           const int* __restrict__ levels, int level, int accumulate, const frcnn_bn_reduce red) {
    extern __shared__ __attribute__((aligned(16))) float racc[];              // [Wf][CS]
    constexpr int LANES = CS / 2, GROUPS = 256 / LANES;                        // channel pairs per slab, RoI rows in flight
    const int slabs = C / CS;
    int bid = blockIdx.x;
    const int slab = bid % slabs;
    bid /= slabs;
    const int y = bid % Hf, b = bid / Hf;
    if (accumulate) {
        // gfeat already holds another branch's gradient of this map (the RPN's data gradient): the row starts from it
        const bf16_t* in = gfeat + ((int64_t)(b * Hf + y) * Wf) * C + slab * CS;
        for (int i = threadIdx.x; i < Wf * (CS / 2); i += 256) {
            const int x = i / (CS / 2), cp = (i - x * (CS / 2)) * 2;
            const unsigned int v = *reinterpret_cast<const unsigned int*>(in + (int64_t)x * C + cp);
            racc[x * CS + cp] = bf16_bits_to_f32((unsigned short)(v & 0xFFFFu));
            racc[x * CS + cp + 1] = bf16_bits_to_f32((unsigned short)(v >> 16));
        }
    } else {
        for (int i = threadIdx.x; i < Wf * CS; i += 256) racc[i] = 0.f;
    }
    __syncthreads();

    const int cl = (threadIdx.x % LANES) * 2, grp = threadIdx.x / LANES;
    const int c0 = slab * CS + cl;
    const int crop = ps * ks, items = ps * ps * C;
    const float hm1 = (float)(Hf - 1), wm1 = (float)(Wf - 1), fy = (float)y;
    for (int r = grp; r < nrows; r += GROUPS) {
        const int row = rows[r];
        if (row / P != b) continue;                                           // (uniform per group)
        if (levels && levels[row] != level) continue;                         // feature pyramid: pooled from another level
        const RoiGeom g = roi_geom(rois + (int64_t)row * 4, Hf, Wf, crop);
        if (g.hs == 0.f && g.ws == 0.f) {
            // a box of zero extent (the zero padding of an NMS output with fewer detections than slots: all four coordinates
            // equal): every one of its crop samples is the same point, so all ps*ps bins route their gradient to the same <= 4
            // pixels.  Taken bin by bin that is ps*ps LDS atomics per channel on ONE address in the ONE workgroup that owns the
            // box's feature row -- with half the sampled rows padding, that workgroup ran for hundreds of microseconds while the
            // rest of the chip idled (round 2: 48 -> 376 us across launches).  Sum the bins in registers, add once.
            const float in_y = g.y1s, in_x = g.x1s;
            if (!(in_y >= 0.f && in_y <= hm1 && in_x >= 0.f && in_x <= wm1)) continue;
            const float ty = floorf(in_y), by = ceilf(in_y), ly = in_y - ty;
            const float wy = (ty == fy ? (1.f - ly) : 0.f) + (by == fy ? ly : 0.f);
            if (wy == 0.f) continue;
            float s0 = 0.f, s1 = 0.f;
            for (int bin = 0; bin < ps * ps; ++bin) {
                const unsigned int g2 = *reinterpret_cast<const unsigned int*>(gpooled + (int64_t)r * items + (int64_t)bin * C + c0);
                s0 += bf16_bits_to_f32((unsigned short)(g2 & 0xFFFFu));
                s1 += bf16_bits_to_f32((unsigned short)(g2 >> 16));
            }
            const float lxf = floorf(in_x), rxf = ceilf(in_x), lx = in_x - lxf;
            if (s0 != 0.f) {
                atomicAdd(racc + (int)lxf * CS + cl, (1.f - lx) * wy * s0);
                atomicAdd(racc + (int)rxf * CS + cl, lx * wy * s0);
            }
            if (s1 != 0.f) {
                atomicAdd(racc + (int)lxf * CS + cl + 1, (1.f - lx) * wy * s1);
                atomicAdd(racc + (int)rxf * CS + cl + 1, lx * wy * s1);
            }
            continue;
        }
        for (int ph = 0; ph < ps; ++ph) {
            // feature rows reachable from this bin row: floor of the smallest to ceil of the largest valid sample coordinate
            const float ya = g.y1s + (float)(ph * ks) * g.hs, yb = g.y1s + (float)(ph * ks + ks - 1) * g.hs;
            const float ylo = fminf(ya, yb), yhi = fmaxf(ya, yb);
            if (!(floorf(ylo) <= fy && ceilf(yhi) >= fy)) continue;           // NaN-safe: skipped
            // the bin row's loads first (independent, all in flight together), then the arithmetic: the loop is latency-bound
            // when every bin waits for its own two loads
            constexpr int PWMAX = 8;
            for (int pw0 = 0; pw0 < ps; pw0 += PWMAX) {
                unsigned int g2v[PWMAX];
                unsigned short a2v[PWMAX];
#pragma unroll
                for (int u = 0; u < PWMAX; ++u) {
                    const int pw = pw0 + u;
                    const int64_t it = (int64_t)(ph * ps + (pw < ps ? pw : ps - 1)) * C + c0;
                    g2v[u] = pw < ps ? *reinterpret_cast<const unsigned int*>(gpooled + (int64_t)r * items + it) : 0u;
                    a2v[u] = *reinterpret_cast<const unsigned short*>(amax + (int64_t)row * items + it);
                }
                // Consecutive bins of a bin row land on the same or the next column (a proposal is a few cells wide, its 14 sample
                // columns lie ~0.4 cells apart): their contributions are run-length combined in registers -- pending sums for column
                // px and px + 1 -- and leave as ONE LDS atomic per column instead of two per bin (which, back to back on one address,
                // are what this kernel costs when the gradient is dense).
                float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
                int px[2] = {-4, -4};
                auto flush = [&](const int e) {
                    if (a0[e] != 0.f) atomicAdd(racc + px[e] * CS + cl + e, a0[e]);
                    // (finite gradients: non-zero only where column px + 1 exists; a NaN / Inf gradient makes 0 * d non-zero on the last
                    // column, hence the range check -- it must propagate, not write one column past the LDS row)
                    if (a1[e] != 0.f && px[e] + 1 < Wf) atomicAdd(racc + (px[e] + 1) * CS + cl + e, a1[e]);
                };
#pragma unroll
                for (int u = 0; u < PWMAX; ++u) {
                    const int pw = pw0 + u;
                    const unsigned int g2 = g2v[u];
                    if (g2 == 0u) continue;
                    const unsigned short a2 = a2v[u];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float gv = bf16_bits_to_f32((unsigned short)(e ? (g2 >> 16) : (g2 & 0xFFFFu)));
                        if (gv == 0.f) continue;
                        const int sidx = e ? (a2 >> 8) : (a2 & 0xFF);
                        const int i = ph * ks + sidx / ks, j = pw * ks + sidx % ks;
                        const float in_y = g.y1s + (float)i * g.hs;
                        const float in_x = g.x1s + (float)j * g.ws;
                        if (!(in_y >= 0.f && in_y <= hm1 && in_x >= 0.f && in_x <= wm1)) continue;
                        const float ty = floorf(in_y), by = ceilf(in_y), ly = in_y - ty;
                        const float wy = (ty == fy ? (1.f - ly) : 0.f) + (by == fy ? ly : 0.f);
                        if (wy == 0.f) continue;
                        const float lxf = floorf(in_x), lx = in_x - lxf;
                        const float d = wy * gv;
                        const int xi = (int)lxf;
                        const float c_l = (1.f - lx) * d, c_r = lx * d;
                        if (xi == px[e]) {
                            a0[e] += c_l;
                            a1[e] += c_r;
                        } else if (xi == px[e] + 1) {
                            if (a0[e] != 0.f) atomicAdd(racc + px[e] * CS + cl + e, a0[e]);
                            a0[e] = a1[e] + c_l;
                            a1[e] = c_r;
                            px[e] = xi;
                        } else {
                            flush(e);
                            a0[e] = c_l;
                            a1[e] = c_r;
                            px[e] = xi;
                        }
                    }
                }
                flush(0);
                flush(1);
            }
        }
    }
    __syncthreads();
    bf16_t* out = gfeat + ((int64_t)(b * Hf + y) * Wf) * C + slab * CS;
    // red.partial != NULL: gfeat is complete with this launch and is the gradient arriving at a BatchNorm(+ReLU) layer whose raw input
    // was red.z -- the row's contribution to that layer's backward sums (sum g*m, sum g*m*xhat; g as stored, m = the ReLU mask bit) is
    // accumulated while the row is written: what frcnn_bn_bwd_reduce(gout = gfeat, ...) would add after re-reading gfeat, z and the mask
    float sg[2] = {0.f, 0.f}, sgx[2] = {0.f, 0.f};
    const int cp_fixed = (threadIdx.x % (CS / 2)) * 2;                        // (256 % (CS / 2) == 0: a thread keeps its channel pair)
    float mu[2] = {0.f, 0.f}, is[2] = {0.f, 0.f};
    if (red.partial) {
        mu[0] = red.mean[slab * CS + cp_fixed]; mu[1] = red.mean[slab * CS + cp_fixed + 1];
        is[0] = red.invstd[slab * CS + cp_fixed]; is[1] = red.invstd[slab * CS + cp_fixed + 1];
    }
    for (int i = threadIdx.x; i < Wf * (CS / 2); i += 256) {
        const int x = i / (CS / 2), cp = (i - x * (CS / 2)) * 2;
        const unsigned int v = (unsigned int)f32_to_bf16_bits(racc[x * CS + cp]) | ((unsigned int)f32_to_bf16_bits(racc[x * CS + cp + 1]) << 16);
        *reinterpret_cast<unsigned int*>(out + (int64_t)x * C + cp) = v;
        if (red.partial) {
            const int64_t pix = (int64_t)(b * Hf + y) * Wf + x;
            const int c = slab * CS + cp;
            const unsigned int zz = *reinterpret_cast<const unsigned int*>(reinterpret_cast<const bf16_t*>(red.z) + pix * C + c);
            const unsigned int mb = red.relu_mask ? red.relu_mask[pix * (C / 8) + (c >> 3)] : 0xFFu;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float g = bf16_bits_to_f32((unsigned short)(e ? (v >> 16) : (v & 0xFFFFu)));
                const float z = bf16_bits_to_f32((unsigned short)(e ? (zz >> 16) : (zz & 0xFFFFu)));
                const float gm = ((mb >> ((c + e) & 7)) & 1u) ? g : 0.f;
                sg[e] += gm;
                sgx[e] += gm * ((z - mu[e]) * is[e]);
            }
        }
    }
    if (red.partial) {
        // the 256 / (CS / 2) threads that share a channel pair meet in LDS (the row accumulator is free now), then one atomic per
        // (statistic, channel) and workgroup
        __syncthreads();
        constexpr int SH = 256 / (CS / 2);
        float* sc = racc;                                                     // [SH][CS][2]
        const int grp2 = threadIdx.x / (CS / 2);
        sc[(grp2 * CS + cp_fixed) * 2] = sg[0];
        sc[(grp2 * CS + cp_fixed) * 2 + 1] = sgx[0];
        sc[(grp2 * CS + cp_fixed + 1) * 2] = sg[1];
        sc[(grp2 * CS + cp_fixed + 1) * 2 + 1] = sgx[1];
        __syncthreads();
        if (threadIdx.x < 2 * CS) {
            const int stat = threadIdx.x / CS, cl2 = threadIdx.x - stat * CS;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < SH; ++k) t += sc[(k * CS + cl2) * 2 + stat];
            const int slot = blockIdx.x & (FRCNN_STAT_SLOTS - 1);
            atomicAdd(red.partial + ((int64_t)slot * 2 + stat) * C + slab * CS + cl2, t);
        }
    }
}

}  // namespace

static int roi_fwd_impl(const frcnn_bf16* feat, const float* rois, int b, int p, int hf, int wf, int c, int ps, int ks,
                        frcnn_bf16* pooled, uint8_t* argmax, const int32_t* levels, int level, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(feat && rois && pooled && argmax, "roi_crop_pool_fwd: null pointer");
    FRCNN_CHECK_ARG(b > 0 && p > 0 && c % 8 == 0 && ps >= 1 && ks >= 1 && ps * ks >= 2 && ks * ks <= 255 && hf > 1 && wf > 1,
                    "roi_crop_pool_fwd: bad sizes");
    FRCNN_CHECK_ARG(ps * ks <= kMaxCrop && 2 * ps * ks <= 256 && (long long)hf * wf * c * 2 < (1ll << 31), "roi_crop_pool_fwd: crop or feature map too large");
    // channel slices per RoI: enough (image, slice) pairs for the 8 XCDs, slices of at least 8 channel vectors (see the kernel) -- and,
    // for the 2 x 2 window on maps of >= 64 channel vectors, slices in whole multiples of 64 vectors (the wave-uniform form: a wave owns
    // whole bins); with fewer than 8 pairs two XCDs share a pair's slice of the map
    const int c8 = c / 8;
    // (roi_fwd_ks2_kernel marks out-of-map samples with the byte offset 0x40000000 and relies on the buffer descriptor's range check to
    // return zeros for it: the sentinel -- plus any valid row / column offset -- must lie beyond a per-image map, i.e. the map under 1 GiB)
    const bool uniform_ok = ks == 2 && c8 % 64 == 0 && (long long)hf * wf * c * 2 < 0x40000000ll;
    int nsplit = 1;
    if (uniform_ok) {
        while (b * nsplit < 8 && c8 % (2 * nsplit) == 0 && (c8 / (2 * nsplit)) % 64 == 0) nsplit *= 2;
    } else {
        while (b * nsplit < 8 && c8 % (2 * nsplit) == 0 && c8 / (2 * nsplit) >= 8) nsplit *= 2;
    }
    const int pairs = b * nsplit, lanes_x = pairs < 8 ? pairs : 8, per_lane = (pairs + lanes_x - 1) / lanes_x;
    const dim3 grid((unsigned)(lanes_x * per_lane * p));
    int form = uniform_ok ? 1 : 0;
#ifdef FRCNN_SWEEP
    if (const char* e = getenv("FRCNN_ROI_FWD_OLD")) { if (atoi(e)) form = 0; }
#endif
    if (form == 1)
        hipLaunchKernelGGL(roi_fwd_ks2_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(feat), rois,
                           p, hf, wf, c / 8, ps, reinterpret_cast<bf16_t*>(pooled), argmax, nsplit, pairs, levels, level);
    else if (ks == 2)
        hipLaunchKernelGGL(roi_fwd_kernel<2>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16_t*>(feat), rois, p, hf, wf, c / 8, ps, ks, reinterpret_cast<bf16_t*>(pooled), argmax, nsplit, pairs, levels, level);
    else
        hipLaunchKernelGGL(roi_fwd_kernel<0>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16_t*>(feat), rois, p, hf, wf, c / 8, ps, ks, reinterpret_cast<bf16_t*>(pooled), argmax, nsplit, pairs, levels, level);
    FRCNN_CHECK_LAUNCH("roi_crop_pool_fwd");
    return FRCNN_OK;
}

extern "C" int frcnn_roi_crop_pool_fwd(const frcnn_bf16* feat, const float* rois, int b, int p, int hf, int wf, int c, int ps, int ks,
                                       frcnn_bf16* pooled, uint8_t* argmax, frcnn_stream_t stream) {
    return roi_fwd_impl(feat, rois, b, p, hf, wf, c, ps, ks, pooled, argmax, nullptr, 0, stream);
}

extern "C" int frcnn_roi_crop_pool_fwd_level(const frcnn_bf16* feat, const float* rois, int b, int p, int hf, int wf, int c, int ps, int ks,
                                             frcnn_bf16* pooled, uint8_t* argmax, const int32_t* levels, int level, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(levels, "roi_crop_pool_fwd_level: null level table");
    return roi_fwd_impl(feat, rois, b, p, hf, wf, c, ps, ks, pooled, argmax, levels, level, stream);
}

extern "C" int frcnn_roi_crop_pool_bwd(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows, int nrows,
                                       int p, int hf, int wf, int c, int ps, int ks, float* gfeat, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gpooled && argmax && rois && rows && gfeat && nrows > 0, "roi_crop_pool_bwd: bad arguments");
    hipLaunchKernelGGL(roi_bwd_kernel, dim3(nrows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(gpooled), argmax, rois, rows, p, hf, wf, c, ps, ks, gfeat);
    FRCNN_CHECK_LAUNCH("roi_crop_pool_bwd");
    return FRCNN_OK;
}

static int roi_bwd_bf16_impl(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows,
                             int nrows, int b, int p, int hf, int wf, int c, int ps, int ks, frcnn_bf16* gfeat, const int32_t* levels, int level,
                             frcnn_stream_t stream, int accumulate = 0, const frcnn_bn_reduce* red = nullptr) {
    FRCNN_CHECK_ARG(gpooled && argmax && rois && rows && gfeat && nrows > 0 && b > 0, "roi_crop_pool_bwd_bf16: bad arguments");
    FRCNN_CHECK_ARG(c % 64 == 0 && ps >= 1 && ks >= 1 && ks * ks <= 255 && hf > 1 && wf > 1, "roi_crop_pool_bwd_bf16: bad sizes");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    frcnn_bn_reduce rd{};
    if (red) {
        FRCNN_CHECK_ARG(red->z && red->mean && red->invstd && red->partial, "roi_crop_pool_bwd_bf16_add: incomplete BatchNorm-reduce arguments");
        rd = *red;
    }
    // 64-channel slabs: 8 RoI rows in flight per workgroup, 8 workgroups per CU (measured: 118 us; 128 channels 129, 32 channels 127);
    // 32-channel slabs where a row of 64 channels does not fit 64 KB of LDS (the stride-4 level of a feature pyramid: 311 pixels)
    // (LDS: one feature row of the slab in fp32; with a fused reduce at least its 4 KB of scratch)
    const size_t smem = (size_t)wf * 64 * 4 < 4096 ? 4096 : (size_t)wf * 64 * 4;
    if (smem > 64 * 1024) {
        const size_t smem32 = (size_t)wf * 32 * 4;
        FRCNN_CHECK_ARG(smem32 <= 64 * 1024 && c % 32 == 0, "roi_crop_pool_bwd_bf16: feature map too wide (wf=%d)", wf);
        hipLaunchKernelGGL(roi_bwd_rows_kernel<32>, dim3(b * hf * (c / 32)), dim3(256), smem32, s, reinterpret_cast<const bf16_t*>(gpooled),
                           argmax, rois, rows, nrows, p, hf, wf, c, ps, ks, reinterpret_cast<bf16_t*>(gfeat), levels, level, accumulate, rd);
        FRCNN_CHECK_LAUNCH("roi_crop_pool_bwd_bf16");
        return FRCNN_OK;
    }
    hipLaunchKernelGGL(roi_bwd_rows_kernel<64>, dim3(b * hf * (c / 64)), dim3(256), smem, s, reinterpret_cast<const bf16_t*>(gpooled),
                       argmax, rois, rows, nrows, p, hf, wf, c, ps, ks, reinterpret_cast<bf16_t*>(gfeat), levels, level, accumulate, rd);
    FRCNN_CHECK_LAUNCH("roi_crop_pool_bwd_bf16");
    return FRCNN_OK;
}

extern "C" int frcnn_roi_crop_pool_bwd_bf16_add(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows, int nrows,
                                                int b, int p, int hf, int wf, int c, int ps, int ks, frcnn_bf16* gfeat,
                                                const struct frcnn_bn_reduce* red, frcnn_stream_t stream) {
    return roi_bwd_bf16_impl(gpooled, argmax, rois, rows, nrows, b, p, hf, wf, c, ps, ks, gfeat, nullptr, 0, stream, 1, red);
}

extern "C" int frcnn_roi_crop_pool_bwd_bf16(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows,
                                            int nrows, int b, int p, int hf, int wf, int c, int ps, int ks, frcnn_bf16* gfeat,
                                            frcnn_stream_t stream) {
    return roi_bwd_bf16_impl(gpooled, argmax, rois, rows, nrows, b, p, hf, wf, c, ps, ks, gfeat, nullptr, 0, stream);
}

extern "C" int frcnn_roi_crop_pool_bwd_bf16_level(const frcnn_bf16* gpooled, const uint8_t* argmax, const float* rois, const int32_t* rows,
                                                  int nrows, int b, int p, int hf, int wf, int c, int ps, int ks, frcnn_bf16* gfeat,
                                                  const int32_t* levels, int level, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(levels, "roi_crop_pool_bwd_bf16_level: null level table");
    return roi_bwd_bf16_impl(gpooled, argmax, rois, rows, nrows, b, p, hf, wf, c, ps, ks, gfeat, levels, level, stream);
}
