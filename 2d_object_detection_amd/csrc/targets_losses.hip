// Target assignment, sampling, losses and loss gradients of the reference's training path
// (utils/training.py, utils/losses.py, get_training_samples of both detectors).
// One workgroup per image; box arithmetic without FMA contraction so that IoUs (hence the
// discrete fg/bg decisions) are bit-identical to the fp32 oracle.
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kMaxGt = 128;
constexpr int kMaxC1 = 32;

// utils/metrics.py:136-208
__device__ __forceinline__ float ref_iou(const f32x4 a, const float area_a, const f32x4 b, const float area_b) {
    const float dw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
    const float dh = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
    const float inter = fmaxf(0.0f, dw) * fmaxf(0.0f, dh);
    const float uni = area_a + area_b - inter;
    return inter == 0.0f ? 0.0f : inter / uni;
}

struct AssignParams {
    const float* regions; int rpi;
    const float* gt_labels; const float* gt_boxes;
    int R, G, C1g, C1, objectness;
    float W, H, fg_lo, fg_hi, bg_lo, bg_hi;
    float* tl; float* tb;
};

// MODE 0: the whole assignment of one image in one workgroup (pass 1, reduction, pass 2).
// Long region lists (the feature pyramid's ~82 k anchors per image: 400 us in one workgroup) are cut into gridDim.y slices and three
// launches: MODE 1 = pass 1 of a slice (max IoU / first arg-max of every region, stashed in the label rows); MODE 2 = one workgroup
// per image finds the first region attaining the global maximum and marks it in its stash (bit 30 of the arg-max word: no scratch
// memory, nothing a later pass overwrites is read by another workgroup); MODE 3 = pass 2 of a slice.  Same arithmetic, same results.
constexpr int kForceBit = 0x40000000;
template <int MODE>
__global__ __launch_bounds__(1024) void assign_targets_kernel(const AssignParams p) {
    __shared__ f32x4 gbox[kMaxGt];
    __shared__ float garea[kMaxGt];
    __shared__ int gsrc[kMaxGt];          // original gt row of compacted entry
    __shared__ int gobj[kMaxGt];          // objectness class (0/1, -1 = none) of original row
    __shared__ int nvalid;
    __shared__ float red_v[1024];
    __shared__ int red_i[1024];
    const int b = blockIdx.x;
    const float* gl = p.gt_labels + (int64_t)b * p.G * p.C1g;
    const float* gbx = p.gt_boxes + (int64_t)b * p.G * 4;
    const float* regions = p.regions + (p.rpi ? (int64_t)b * p.R * 4 : 0);
    float* tl = p.tl + (int64_t)b * p.R * p.C1;
    float* tb = p.tb + (int64_t)b * p.R * (p.C1 - 1) * 4;
    const int C = p.C1 - 1;
    const int per_slice = (p.R + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_begin = MODE == 1 || MODE == 3 ? (int)blockIdx.y * per_slice : 0;
    const int r_end = MODE == 1 || MODE == 3 ? min(p.R, r_begin + per_slice) : p.R;
    int ng = 0, nlive = 0;
    __shared__ int live_idx[kMaxGt];
    __shared__ int live_cnt[3];
    if (MODE != 2) {
    // ---- compact the non-padding ground truth in order (training.py:43-45), absolute coords (:48): one thread per gt row,
    // order-preserving slots from wave ballots (G <= 128: two waves)
    __shared__ int wave_cnt[2];
    bool keep = false;
    int obj = -1;
    unsigned long long keep_mask = 0ull;
    if (threadIdx.x < kMaxGt) {
        const int g = threadIdx.x;
        if (g < p.G) {
            float s = 0.f;
            for (int c = 0; c < p.C1g; ++c) s += gl[g * p.C1g + c];
            if (p.objectness) {                       // rpn_detector.py:141: one_hot(int(sum), 2)
                const int k = (int)s;
                obj = (k == 0 || k == 1) ? k : -1;
                keep = obj >= 0;                      // one-hot row sums to 1 != 0 (an out-of-range index gives a zero row)
            } else {
                keep = s != 0.0f;
            }
        }
        keep_mask = __ballot(keep);
        if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(keep_mask);
    }
    __syncthreads();
    bool live = false;                    // a kept row whose box has positive width and height (any other row's IoU is 0 with every region)
    if (keep) {
        const int g = threadIdx.x;
        const int n = ((threadIdx.x >> 6) ? wave_cnt[0] : 0) + __popcll(keep_mask & ((1ull << (threadIdx.x & 63)) - 1ull));
        f32x4 bx = {gbx[g * 4] * p.W, gbx[g * 4 + 1] * p.H, gbx[g * 4 + 2] * p.W, gbx[g * 4 + 3] * p.H};
        gbox[n] = bx;
        garea[n] = (bx[2] - bx[0]) * (bx[3] - bx[1]);
        gsrc[n] = g;
        gobj[n] = obj;
        live = (bx[2] - bx[0] > 0.0f) && (bx[3] - bx[1] > 0.0f);
    }
    if (threadIdx.x == 0) nvalid = wave_cnt[0] + wave_cnt[1];
    __syncthreads();
    ng = nvalid;
    // The RPN path keeps all G = 100 rows (the objectness conversion turns zero padding into a "background" row,
    // rpn_detector.py:141 -- SURVEY A.6) although a handful are real boxes.  A row of zero or negative extent has IoU exactly 0
    // with every region (utils/metrics.py:150-208: its intersection width or height clamps to 0), so it can only decide the
    // arg-max when NO row has a positive IoU -- and then the first maximum is row 0 whatever the rows are.  The IoU loop
    // therefore runs over the live rows only (compacted, in order): max / first-arg-max are unchanged, the work drops ~10x.
    {
        const unsigned long long lm = threadIdx.x < kMaxGt ? __ballot(live) : 0ull;
        if (threadIdx.x < kMaxGt && (threadIdx.x & 63) == 0) live_cnt[threadIdx.x >> 6] = __popcll(lm);
        __syncthreads();
        if (live) {
            const int n = ((threadIdx.x >> 6) ? wave_cnt[0] : 0) + __popcll(keep_mask & ((1ull << (threadIdx.x & 63)) - 1ull));   // compacted row
            live_idx[((threadIdx.x >> 6) ? live_cnt[0] : 0) + __popcll(lm & ((1ull << (threadIdx.x & 63)) - 1ull))] = n;
        }
        if (threadIdx.x == 0) live_cnt[2] = live_cnt[0] + live_cnt[1];
        __syncthreads();
    }
    nlive = live_cnt[2];
    }   // MODE != 2

    // ---- pass 1: per-region max IoU / first argmax; stash them in the output buffers
    float best_v = -1.f;
    int best_i = 0x7FFFFFFF;
    if (MODE == 0 || MODE == 1)
    for (int r = r_begin + (int)threadIdx.x; r < r_end; r += blockDim.x) {
        const f32x4 rb = *reinterpret_cast<const f32x4*>(regions + (int64_t)r * 4);
        const float ra = (rb[2] - rb[0]) * (rb[3] - rb[1]);
        float mx = 0.f;                   // (every kept row contributes an IoU >= 0; rows of no extent exactly 0)
        int am = 0;                       // first maximum when all IoUs are 0: row 0
        for (int l = 0; l < nlive; ++l) {
            const int g = live_idx[l];
            const float v = ref_iou(rb, ra, gbox[g], garea[g]);
            if (v > mx) { mx = v; am = g; }
        }
        tl[(int64_t)r * p.C1] = mx;
        tl[(int64_t)r * p.C1 + 1] = __int_as_float(am);
        if (mx > best_v || (mx == best_v && r < best_i)) { best_v = mx; best_i = r; }
    }
    if (MODE == 1) return;
    if (MODE == 2)                        // the stashed maxima of ALL regions of the image
        for (int r = threadIdx.x; r < p.R; r += blockDim.x) {
            const float mx = tl[(int64_t)r * p.C1];
            if (mx > best_v || (mx == best_v && r < best_i)) { best_v = mx; best_i = r; }
        }
    int force_idx = -1;
    if (MODE != 3) {
    red_v[threadIdx.x] = best_v;
    red_i[threadIdx.x] = best_i;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = red_v[threadIdx.x + s];
            const int oi = red_i[threadIdx.x + s];
            if (ov > red_v[threadIdx.x] || (ov == red_v[threadIdx.x] && oi < red_i[threadIdx.x])) {
                red_v[threadIdx.x] = ov;
                red_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    force_idx = red_i[0];                 // first region attaining the global max (training.py:137-138)
    }
    if (MODE == 2) {
        if (threadIdx.x == 0 && force_idx >= 0 && force_idx < p.R) {
            float* w = tl + (int64_t)force_idx * p.C1 + 1;
            *w = __int_as_float(__float_as_int(*w) | kForceBit);
        }
        return;
    }

    // ---- pass 2: labels + encoded target boxes
    for (int r = r_begin + (int)threadIdx.x; r < r_end; r += blockDim.x) {
        const float mx = tl[(int64_t)r * p.C1];
        const int am_raw = __float_as_int(tl[(int64_t)r * p.C1 + 1]);
        const int am = am_raw & ~kForceBit;
        const bool bg = (mx >= p.bg_lo) && (mx < p.bg_hi);
        const bool fg = ((mx >= p.fg_lo) && (mx < p.fg_hi)) || (MODE == 3 ? (am_raw & kForceBit) != 0 : r == force_idx);
        float lab[kMaxC1];
#pragma unroll 1
        for (int c = 0; c < p.C1; ++c) lab[c] = 0.f;
        if (bg) lab[0] = 1.f;
        if (fg && ng > 0) {
            if (p.objectness) {
                for (int c = 0; c < p.C1; ++c) lab[c] = (c == gobj[am]) ? 1.f : 0.f;
            } else {
                const float* src = gl + gsrc[am] * p.C1g;
                for (int c = 0; c < p.C1; ++c) lab[c] = src[c];
            }
        }
        for (int c = 0; c < p.C1; ++c) tl[(int64_t)r * p.C1 + c] = lab[c];
        // encode (utils/boxes.py:44-73) of the argmax gt against this region
        f32x4 enc = {0.f, 0.f, 0.f, 0.f};
        if (ng > 0) {
            const f32x4 rb = *reinterpret_cast<const f32x4*>(regions + (int64_t)r * 4);
            const f32x4 gb = gbox[am];
            const float cxr = (rb[2] + rb[0]) / 2.0f, cyr = (rb[3] + rb[1]) / 2.0f;
            const float wr = rb[2] - rb[0], hr = rb[3] - rb[1];
            const float cx = (gb[2] + gb[0]) / 2.0f, cy = (gb[3] + gb[1]) / 2.0f;
            const float w = gb[2] - gb[0], h = gb[3] - gb[1];
            enc[0] = (cx - cxr) / wr;
            enc[1] = (cy - cyr) / hr;
            enc[2] = logf(w / wr);
            enc[3] = logf(h / hr);
        }
        for (int c = 0; c < C; ++c) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(tb + ((int64_t)r * C + c) * 4) = (lab[c + 1] != 0.f) ? enc : z;
        }
    }
}

// ---------------------------------------------------------------- sampling
struct SampleParams {
    const float* tl; int R, C1, S, max_fg;
    unsigned int k0, k1; const int64_t* step; int stream_base, image_base;
    int* out; int* ws; int* status;
};
constexpr int kFgLds = 8192;

// blockDim.x = 256 (short lists) or 1024 (the pyramid's ~82 k regions per image: the two ordered sweeps are split over 16 waves)
__global__ __launch_bounds__(1024) void sample_kernel(const SampleParams p) {
    __shared__ int wave_fg[16], wave_bg[16];
    __shared__ int fg_lds[kFgLds];
    __shared__ int swap_j[1024];          // partner position of step i of the partial Fisher-Yates (S <= 1024)
    const int b = blockIdx.x;
    const float* tl = p.tl + (int64_t)b * p.R * p.C1;
    int* fg_list = p.ws + (int64_t)b * 2 * p.R;
    int* bg_list = fg_list + p.R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // candidate lists in ascending region order (training.py:99-104).  Every wave owns a contiguous share of the regions and
    // walks it 64 rows at a time (coalesced); ballots give the order-preserving slots.  Two sweeps: count, then write.
    const int nw = blockDim.x >> 6;
    const int per_wave = ((p.R + nw - 1) / nw + 63) / 64 * 64;
    const int w0 = wave * per_wave, w1 = min(p.R, w0 + per_wave);
    auto classify = [&](const int r, bool& is_fg, bool& is_bg) {
        float s = 0.f, l0 = 0.f;
        if (r < w1) {
            l0 = tl[(int64_t)r * p.C1];
            for (int c = 0; c < p.C1; ++c) s += tl[(int64_t)r * p.C1 + c];
        }
        is_fg = r < w1 && s != 0.0f && l0 == 0.0f;
        is_bg = r < w1 && s != 0.0f && l0 == 1.0f;
    };
    int nf = 0, nb = 0;
    for (int r0 = w0; r0 < w1; r0 += 64) {
        bool f, g;
        classify(r0 + lane, f, g);
        nf += __popcll(__ballot(f));
        nb += __popcll(__ballot(g));
    }
    if (lane == 0) { wave_fg[wave] = nf; wave_bg[wave] = nb; }
    __syncthreads();
    int of = 0, ob = 0, nfg = 0, nbg = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wave) { of += wave_fg[w]; ob += wave_bg[w]; }
        nfg += wave_fg[w];
        nbg += wave_bg[w];
    }
    const bool in_lds = nfg <= kFgLds;
    for (int r0 = w0; r0 < w1; r0 += 64) {
        bool f, g;
        classify(r0 + lane, f, g);
        const unsigned long long mf = __ballot(f), mg = __ballot(g);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (f) {
            const int pos = of + __popcll(mf & below);
            fg_list[pos] = r0 + lane;
            if (in_lds) fg_lds[pos] = r0 + lane;
        }
        if (g) bg_list[ob + __popcll(mg & below)] = r0 + lane;
        of += __popcll(mf);
        ob += __popcll(mg);
    }
    const int n_fg = min(nfg, p.max_fg);
    const int n_bg = p.S - n_fg;
    const unsigned int step = (unsigned int)(*p.step);
    int* out = p.out + (int64_t)b * p.S;
    // fg: partial Fisher-Yates (== shuffle then take).  The swap chain is sequential by construction, its random partners are
    // not: all threads draw them first (Philox + the modulo were 3/4 of this kernel's time on the one lane walking the chain)
    for (int i = threadIdx.x; i < n_fg; i += blockDim.x) {
        const unsigned int rnd = philox_first((unsigned)i, (unsigned)(b + p.image_base), step, (unsigned)(p.stream_base + 0), p.k0, p.k1);
        swap_j[i] = i + (int)(rnd % (unsigned)(nfg - i));
    }
    __syncthreads();                       // (also publishes fg_list / fg_lds / bg_list inside the workgroup)
    if (threadIdx.x == 0) {
        int* lst = in_lds ? fg_lds : fg_list;
        for (int i = 0; i < n_fg; ++i) {
            const int j = swap_j[i];
            const int a = lst[i], c = lst[j];
            lst[i] = c; lst[j] = a;
            out[i] = c;
        }
        if (n_bg > 0 && nbg == 0) atomicOr(p.status, 1);
    }
    for (int i = threadIdx.x; i < n_bg; i += blockDim.x) {
        int v = 0;
        if (nbg > 0) {
            const unsigned int rnd = philox_first((unsigned)i, (unsigned)(b + p.image_base), step, (unsigned)(p.stream_base + 1), p.k0, p.k1);
            v = bg_list[rnd % (unsigned)nbg];
        }
        out[n_fg + i] = v;
    }
}

// ---------------------------------------------------------------- losses + per-sample gradients
struct LossParams {
    const float* scores; const float* deltas; const float* tl; const float* tb; const int* idx;
    int B, R, C1, S; float cls_scale, reg_scale;
    float* losses; float* dlog; float* ddel;
    bf16_t* dhead; int ld; int* rows_out;          // optional fused Fast-RCNN head-gradient rows (what rcnn_head_grad_kernel writes)
    float* rpn_dhead; const int* keep; int locs, apl, rpn_ld;   // optional fused RPN scatter-add (what rpn_head_grad_kernel does; C1 == 2)
    float* bias_grad;                              // optional (with dhead, ld <= 64): += column sums of the dhead rows (colsum_kernel's sums)
    int stage;                                     // dhead rows are assembled in (dynamic) LDS and leave as 16-byte vectors
};

// One workgroup, one thread per sampled row (rows beyond 1024: strided).  MAXC bounds C1 at compile time (2 / 8 / 32): the row's
// scores, targets and box terms are loaded into registers in fully unrolled, predicated loops -- every load of a row is in flight
// at once.  (With runtime-bounded loops each class cost a memory round trip: 4 passes x C1 trips made this kernel 19-31 us.)
template <int MAXC>
__global__ __launch_bounds__(1024) void losses_kernel(const LossParams p) {
    __shared__ float red[2][16];
    __shared__ float cs[32][65];
    extern __shared__ __attribute__((aligned(16))) unsigned char hstage[];     // [rows][ld] bf16 when p.stage
    const int C = p.C1 - 1;
    const int rows = p.B * p.S;
    const float inv_rows = 1.0f / (float)rows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float cls = 0.f, reg = 0.f;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
        const int b = i / p.S;
        const int r = p.idx[i];
        const float* sc = p.scores + ((int64_t)b * p.R + r) * p.C1;
        const float* t = p.tl + ((int64_t)b * p.R + r) * p.C1;
        const float* tbr = p.tb + ((int64_t)b * p.R + r) * C * 4;
        const float* pvr = p.deltas + ((int64_t)b * p.R + r) * C * 4;
        float pr[MAXC], tt[MAXC];
        f32x4 tv[MAXC - 1], pv[MAXC - 1];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            pr[c] = c < p.C1 ? sc[c] : 0.f;
            tt[c] = c < p.C1 ? t[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < MAXC - 1; ++c) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            tv[c] = c < C ? *reinterpret_cast<const f32x4*>(tbr + c * 4) : z4;
            pv[c] = c < C ? *reinterpret_cast<const f32x4*>(pvr + c * 4) : z4;
        }
        // the head-gradient row of this sample: 64 two-byte values.  Written straight to global memory they are 64 store instructions
        // per wave, each touching 64 different 128-byte rows; staged (p.stage) they go to LDS and leave below as whole 16-byte vectors
        bf16_t* hrow = p.dhead ? (p.stage ? reinterpret_cast<bf16_t*>(hstage) + (int64_t)i * p.ld : p.dhead + (int64_t)i * p.ld) : nullptr;
        float* arow = nullptr;                       // RPN: the dense head-gradient row of this sample's location, and its anchor slot
        int ak = 0;
        if (p.rpn_dhead) {
            const int a = p.keep ? p.keep[r] : r;
            const int loc = a / p.apl;
            ak = a - loc * p.apl;
            arow = p.rpn_dhead + ((int64_t)b * p.locs + loc) * p.rpn_ld;
        }
        // Keras categorical_crossentropy on probabilities: renormalise, clip, -sum t log p
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < p.C1) s += pr[c];
        float gq[MAXC];
        float row_loss = 0.f, dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            gq[c] = 0.f;
            if (c < p.C1) {
                const float q = pr[c] / s;
                const float qc = fminf(fmaxf(q, 1e-7f), 1.0f - 1e-7f);
                row_loss -= tt[c] * logf(qc);
                const bool pass = (q >= 1e-7f) && (q <= 1.0f - 1e-7f);
                gq[c] = pass ? -tt[c] / qc : 0.f;
                dot += gq[c] * pr[c];
            }
        }
        cls += row_loss;
        if (p.dlog || hrow || arow) {
            // d/dp_j = gq_j/s - dot/s^2 ; softmax backward: dz_i = p_i (dp_i - sum_j p_j dp_j)
            float dp[MAXC];
            float pdot = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                dp[c] = 0.f;
                if (c < p.C1) { dp[c] = gq[c] / s - dot / (s * s); pdot += pr[c] * dp[c]; }
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (c < p.C1) {
                    const float dz = p.cls_scale * inv_rows * pr[c] * (dp[c] - pdot);
                    if (p.dlog) p.dlog[(int64_t)i * p.C1 + c] = dz;
                    if (hrow) hrow[c] = (bf16_t)dz;
                    if (arow) atomicAdd(arow + 2 * ak + c, dz);
                }
            }
        }
        // Huber(delta=1), mean over the 4 coords, rows with sum(target) != 0, summed (losses.py:35-41)
#pragma unroll
        for (int c = 0; c < MAXC - 1; ++c) {
            if (c < C) {
                const bool keep = (tv[c][0] + tv[c][1] + tv[c][2] + tv[c][3]) != 0.0f;
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
                if (keep) {
                    float h = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = pv[c][e] - tv[c][e];
                        const float ad = fabsf(d);
                        h += (ad <= 1.0f) ? 0.5f * d * d : ad - 0.5f;
                        g[e] = p.reg_scale * 0.25f * ((ad <= 1.0f) ? d : (d > 0.f ? 1.f : -1.f));
                    }
                    reg += h * 0.25f;
                }
                if (p.ddel) *reinterpret_cast<f32x4*>(p.ddel + ((int64_t)i * C + c) * 4) = g;
                if (hrow) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) hrow[p.C1 + 4 * c + e] = (bf16_t)g[e];
                }
                if (arow) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(arow + 2 * p.apl + 4 * ak + e, g[e]);
                }
            }
        }
        if (hrow) {
            for (int c = p.C1 + 4 * C; c < p.ld; ++c) hrow[c] = (bf16_t)0.f;
            p.rows_out[i] = b * p.R + r;
        }
    }
    // fixed-order sums: lanes (xor tree), then the 16 waves in ascending order
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        cls += __shfl_xor(cls, sh);
        reg += __shfl_xor(reg, sh);
    }
    if (lane == 0) { red[0][wave] = cls; red[1][wave] = reg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c2 = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[0][w]; c2 += red[1][w]; }
        p.losses[0] = a * inv_rows;
        p.losses[1] = c2;
    }
    if (p.stage) {
        __syncthreads();
        const int nvec = rows * p.ld / 8;                         // (ld % 8 == 0: checked by the host)
        for (int v = threadIdx.x; v < nvec; v += blockDim.x)
            *reinterpret_cast<u32x4*>(p.dhead + (int64_t)v * 8) = *reinterpret_cast<const u32x4*>(hstage + (int64_t)v * 16);
    }
    if (p.bias_grad) {
        // The head's bias gradient = column sums of the bf16 gradient rows written above, in colsum_kernel's order (row chunks of 256;
        // per column 32 partial sums over rows rl, rl + 32, ..., added in ascending rl; one atomic per column and chunk): the bits
        // frcnn_colsum_bf16(dhead_s, B*S, ld, ld, bias_grad) produces, without its launch.
        const bf16_t* hsrc = p.stage ? reinterpret_cast<const bf16_t*>(hstage) : p.dhead;
        __threadfence_block();
        __syncthreads();
        const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
        for (int r0 = 0; r0 < rows; r0 += 256) {
            const int r1 = min(rows, r0 + 256);
            for (int rl = part; rl < 32; rl += (int)(blockDim.x >> 6)) {
                // the (at most 8) rows of this partial sum: loads issued together, added in the loop's order (+0 for a missing row
                // leaves the sum's bits alone)
                float vv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = r0 + rl + 32 * k;
                    vv[k] = (col < p.ld && r < r1) ? bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(hsrc + (int64_t)r * p.ld + col)) : 0.f;
                }
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += vv[k];
                cs[rl][col] = s;
            }
            __syncthreads();
            if (threadIdx.x < 64 && threadIdx.x < p.ld) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) s += cs[k][threadIdx.x];
                atomicAdd(p.bias_grad + threadIdx.x, s);
            }
            __syncthreads();
        }
    }
}

static void launch_losses(LossParams p, hipStream_t stream) {
    // head-gradient rows through LDS when they fit beside the static arrays (48 KB: 384 rows of 64 columns) and every row is owned by one
    // thread of the single pass (rows <= 1024)
    const size_t bytes = p.dhead ? (size_t)p.B * p.S * p.ld * 2 : 0;
    p.stage = (p.dhead && p.ld % 8 == 0 && bytes <= 48 * 1024 && p.B * p.S <= 1024) ? 1 : 0;
    const size_t smem = p.stage ? bytes : 0;
    if (p.C1 <= 2) hipLaunchKernelGGL(losses_kernel<2>, dim3(1), dim3(1024), smem, stream, p);
    else if (p.C1 <= 8) hipLaunchKernelGGL(losses_kernel<8>, dim3(1), dim3(1024), smem, stream, p);
    else hipLaunchKernelGGL(losses_kernel<kMaxC1>, dim3(1), dim3(1024), smem, stream, p);
}

// RPN: scatter-add per-sample gradients into the dense fp32 head-gradient matrix
__global__ void rpn_head_grad_kernel(const float* __restrict__ dlog, const float* __restrict__ ddel, const int* __restrict__ idx,
                                     const int* __restrict__ keep, int B, int S, int locs, int apl, float* __restrict__ dhead, int ld, int win_off,
                                     int win_n) {
    // win_off / win_n: only the samples whose index lies in [win_off, win_off + win_n) belong to this head (one pyramid level's
    // window of the concatenated anchor list); 0 / INT_MAX: a single feature map
    const int total = B * S;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / S;
        int j = idx[i];
        if (j < win_off || j - win_off >= win_n) continue;
        j -= win_off;
        const int a = keep ? keep[j] : j;
        const int loc = a / apl, k = a - loc * apl;
        float* row = dhead + ((int64_t)b * locs + loc) * ld;
        atomicAdd(row + 2 * k, dlog[(int64_t)i * 2]);
        atomicAdd(row + 2 * k + 1, dlog[(int64_t)i * 2 + 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(row + 2 * apl + 4 * k + e, ddel[(int64_t)i * 4 + e]);
    }
}

// RCNN: per-sample gradient rows -> bf16 [B*S, ld] (zero padded) + RoI row ids
__global__ void rcnn_head_grad_kernel(const float* __restrict__ dlog, const float* __restrict__ ddel, const int* __restrict__ idx, int B,
                                      int R, int C1, int S, bf16_t* __restrict__ dhead, int ld, int* __restrict__ rows_out) {
    const int total = B * S * ld;
    const int nreg = 4 * (C1 - 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % ld, row = i / ld;
        float v = 0.f;
        if (col < C1) v = dlog[(int64_t)row * C1 + col];
        else if (col < C1 + nreg) v = ddel[(int64_t)row * nreg + (col - C1)];
        dhead[i] = (bf16_t)v;
        if (col == 0) rows_out[row] = (row / S) * R + idx[row];
    }
}

// RCNN head post: bias + softmax / split.  One wave per row, lane = column (C1 <= kMaxC1 <= 64): coalesced row reads, the
// class maximum and the exponentials' sum by wave reductions.  The sum runs over the classes in ascending order (a serial
// chain of C1 v_readlane adds, not a tree), so the probabilities are the same bits a one-thread-per-row loop produces.
__global__ __launch_bounds__(256) void rcnn_head_post_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ bias, int R,
                                                             int C1, float* __restrict__ scores, float* __restrict__ deltas) {
    const int nreg = 4 * (C1 - 1);
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = logits + (int64_t)r * ld;
    const float l = lane < C1 ? row[lane] + bias[lane] : -INFINITY;
    float mx = l;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
    const float e = lane < C1 ? expf(l - mx) : 0.f;
    float s = 0.f;
    for (int c = 0; c < C1; ++c) s += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), c));
    const float inv = 1.f / s;
    if (lane < C1) scores[(int64_t)r * C1 + lane] = e * inv;
    for (int c = lane; c < nreg; c += 64) deltas[(int64_t)r * nreg + c] = row[C1 + c] + bias[C1 + c];
}

}  // namespace

#define S_(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int frcnn_assign_targets(const float* regions, int regions_per_image, const float* gt_labels, const float* gt_boxes, int b,
                                    int r, int g, int c1g, int objectness, float img_w, float img_h, float fg_lo, float fg_hi,
                                    float bg_lo, float bg_hi, float* target_labels, float* target_boxes, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(regions && gt_labels && gt_boxes && target_labels && target_boxes, "assign_targets: null pointer");
    const int c1 = objectness ? 2 : c1g;
    FRCNN_CHECK_ARG(b > 0 && r > 0 && g > 0 && g <= kMaxGt && c1 >= 2 && c1 <= kMaxC1 && c1g <= kMaxC1, "assign_targets: bad sizes (G<=%d, C+1<=%d)", kMaxGt, kMaxC1);
    AssignParams p;
    p.regions = regions; p.rpi = regions_per_image; p.gt_labels = gt_labels; p.gt_boxes = gt_boxes;
    p.R = r; p.G = g; p.C1g = c1g; p.C1 = c1; p.objectness = objectness;
    p.W = img_w; p.H = img_h; p.fg_lo = fg_lo; p.fg_hi = fg_hi; p.bg_lo = bg_lo; p.bg_hi = bg_hi;
    p.tl = target_labels; p.tb = target_boxes;
    if (r >= 4096) {
        // long lists: slices of ~2 k regions (two per thread), at most 64 per image
        int ns = (r + 2047) / 2048;
        if (ns > 64) ns = 64;
        hipLaunchKernelGGL(assign_targets_kernel<1>, dim3(b, ns), dim3(1024), 0, S_(stream), p);
        hipLaunchKernelGGL(assign_targets_kernel<2>, dim3(b, 1), dim3(1024), 0, S_(stream), p);
        hipLaunchKernelGGL(assign_targets_kernel<3>, dim3(b, ns), dim3(1024), 0, S_(stream), p);
    } else {
        hipLaunchKernelGGL(assign_targets_kernel<0>, dim3(b, 1), dim3(1024), 0, S_(stream), p);
    }
    FRCNN_CHECK_LAUNCH("assign_targets");
    return FRCNN_OK;
}

extern "C" int frcnn_sample_indices(const float* target_labels, int b, int r, int c1, int num_samples, float fg_proportion, uint64_t seed,
                                    const int64_t* step, int stream_base, int32_t* indices, int32_t* workspace, int32_t* status,
                                    int image_base, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(target_labels && step && indices && workspace && status && b > 0 && r > 0 && num_samples > 0, "sample_indices: bad arguments");
    FRCNN_CHECK_ARG(num_samples <= 1024, "sample_indices: num_samples=%d > 1024 unsupported", num_samples);
    SampleParams p;
    p.tl = target_labels; p.R = r; p.C1 = c1; p.S = num_samples;
    p.max_fg = (int)nearbyint((double)num_samples * (double)fg_proportion);   // tf.math.round: half to even
    p.k0 = (unsigned int)(seed & 0xFFFFFFFFull); p.k1 = (unsigned int)(seed >> 32);
    p.step = step; p.stream_base = stream_base; p.image_base = image_base; p.out = indices; p.ws = workspace; p.status = status;
    hipLaunchKernelGGL(sample_kernel, dim3(b), dim3(r >= 4096 ? 1024 : 256), 0, S_(stream), p);
    FRCNN_CHECK_LAUNCH("sample_indices");
    return FRCNN_OK;
}

extern "C" int frcnn_losses_head_grad(const float* scores, const float* deltas, const float* target_labels, const float* target_boxes,
                                      const int32_t* indices, int b, int r, int c1, int s, float cls_scale, float reg_scale,
                                      float* losses, float* dlogits_s, float* ddeltas_s, frcnn_bf16* dhead_s, int ld, int32_t* rows_out,
                                      float* bias_grad, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(scores && deltas && target_labels && target_boxes && indices && losses, "losses: null pointer");
    FRCNN_CHECK_ARG(c1 >= 2 && c1 <= kMaxC1 && b > 0 && s > 0, "losses: bad sizes");
    FRCNN_CHECK_ARG(!dhead_s || (rows_out && ld >= c1 + 4 * (c1 - 1)), "losses: fused head gradient needs rows_out and ld >= 5*(C+1)-4");
    FRCNN_CHECK_ARG(!bias_grad || (dhead_s && ld <= 64), "losses: the fused bias gradient needs dhead_s with ld <= 64");
    LossParams p;
    p.scores = scores; p.deltas = deltas; p.tl = target_labels; p.tb = target_boxes; p.idx = indices;
    p.B = b; p.R = r; p.C1 = c1; p.S = s; p.cls_scale = cls_scale; p.reg_scale = reg_scale;
    p.losses = losses; p.dlog = dlogits_s; p.ddel = ddeltas_s;
    p.dhead = reinterpret_cast<bf16_t*>(dhead_s); p.ld = ld; p.rows_out = rows_out;
    p.rpn_dhead = nullptr; p.keep = nullptr; p.locs = 0; p.apl = 1; p.rpn_ld = 0;
    p.bias_grad = bias_grad;
    launch_losses(p, S_(stream));
    FRCNN_CHECK_LAUNCH("losses");
    return FRCNN_OK;
}

extern "C" int frcnn_losses_rpn_head_grad(const float* scores, const float* deltas, const float* target_labels, const float* target_boxes,
                                          const int32_t* indices, int b, int r, int s, float cls_scale, float reg_scale, float* losses,
                                          float* dlogits_s, float* ddeltas_s, const int32_t* keep, int num_anchors_total, int a_per_loc,
                                          float* dhead, int ld, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(scores && deltas && target_labels && target_boxes && indices && losses && dhead, "losses_rpn_head_grad: null pointer");
    FRCNN_CHECK_ARG(b > 0 && s > 0 && a_per_loc > 0 && num_anchors_total % a_per_loc == 0 && ld >= 6 * a_per_loc, "losses_rpn_head_grad: bad sizes");
    LossParams p;
    p.scores = scores; p.deltas = deltas; p.tl = target_labels; p.tb = target_boxes; p.idx = indices;
    p.B = b; p.R = r; p.C1 = 2; p.S = s; p.cls_scale = cls_scale; p.reg_scale = reg_scale;
    p.losses = losses; p.dlog = dlogits_s; p.ddel = ddeltas_s;
    p.dhead = nullptr; p.ld = 0; p.rows_out = nullptr; p.bias_grad = nullptr;
    p.rpn_dhead = dhead; p.keep = keep; p.locs = num_anchors_total / a_per_loc; p.apl = a_per_loc; p.rpn_ld = ld;
    launch_losses(p, S_(stream));
    FRCNN_CHECK_LAUNCH("losses");
    return FRCNN_OK;
}

extern "C" int frcnn_losses(const float* scores, const float* deltas, const float* target_labels, const float* target_boxes,
                            const int32_t* indices, int b, int r, int c1, int s, float cls_scale, float reg_scale, float* losses,
                            float* dlogits_s, float* ddeltas_s, frcnn_stream_t stream) {
    return frcnn_losses_head_grad(scores, deltas, target_labels, target_boxes, indices, b, r, c1, s, cls_scale, reg_scale, losses, dlogits_s,
                                  ddeltas_s, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int frcnn_rpn_head_grad(const float* dlogits_s, const float* ddeltas_s, const int32_t* indices, const int32_t* keep, int b, int s,
                                   int num_anchors_total, int a_per_loc, float* dhead, int ld, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(dlogits_s && ddeltas_s && indices && dhead && ld >= 6 * a_per_loc, "rpn_head_grad: bad arguments");
    hipLaunchKernelGGL(rpn_head_grad_kernel, dim3(cdiv((int64_t)b * s, 256)), dim3(256), 0, S_(stream), dlogits_s, ddeltas_s, indices, keep,
                       b, s, num_anchors_total / a_per_loc, a_per_loc, dhead, ld, 0, 0x7FFFFFFF);
    FRCNN_CHECK_LAUNCH("rpn_head_grad");
    return FRCNN_OK;
}

extern "C" int frcnn_rpn_head_grad_level(const float* dlogits_s, const float* ddeltas_s, const int32_t* indices, const int32_t* keep, int b, int s,
                                         int num_anchors_level, int a_per_loc, float* dhead, int ld, int offset, int n, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(dlogits_s && ddeltas_s && indices && dhead && ld >= 6 * a_per_loc && offset >= 0 && n > 0, "rpn_head_grad_level: bad arguments");
    hipLaunchKernelGGL(rpn_head_grad_kernel, dim3(cdiv((int64_t)b * s, 256)), dim3(256), 0, S_(stream), dlogits_s, ddeltas_s, indices, keep,
                       b, s, num_anchors_level / a_per_loc, a_per_loc, dhead, ld, offset, n);
    FRCNN_CHECK_LAUNCH("rpn_head_grad_level");
    return FRCNN_OK;
}

extern "C" int frcnn_rcnn_head_grad(const float* dlogits_s, const float* ddeltas_s, const int32_t* indices, int b, int r, int c1, int s,
                                    frcnn_bf16* dhead_s, int ld, int32_t* rows_out, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(dlogits_s && ddeltas_s && indices && dhead_s && rows_out && ld >= c1 + 4 * (c1 - 1), "rcnn_head_grad: bad arguments");
    hipLaunchKernelGGL(rcnn_head_grad_kernel, dim3(cdiv((int64_t)b * s * ld, 256)), dim3(256), 0, S_(stream), dlogits_s, ddeltas_s, indices,
                       b, r, c1, s, reinterpret_cast<bf16_t*>(dhead_s), ld, rows_out);
    FRCNN_CHECK_LAUNCH("rcnn_head_grad");
    return FRCNN_OK;
}

extern "C" int frcnn_rcnn_head_post(const float* logits, int ld, const float* bias, int r, int nc1, float* scores, float* deltas,
                                    frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(logits && bias && scores && deltas && nc1 >= 2 && nc1 <= kMaxC1 && ld >= nc1 + 4 * (nc1 - 1), "rcnn_head_post: bad arguments");
    hipLaunchKernelGGL(rcnn_head_post_kernel, dim3(cdiv(r, 4)), dim3(256), 0, S_(stream), logits, ld, bias, r, nc1, scores, deltas);
    FRCNN_CHECK_LAUNCH("rcnn_head_post");
    return FRCNN_OK;
}
