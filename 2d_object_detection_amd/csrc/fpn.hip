// Feature-pyramid pieces of BASELINE.json configs[4] (Lin et al., "Feature Pyramid Networks for Object Detection", CVPR 2017; the
// reference has no FPN: models/faster_rcnn.py:25-34 wires one conv4 map): the top-down merge (nearest-neighbour upsampling + add) and
// its gradient, the stride-2 subsampling that makes the coarsest RPN level, and the RoI -> level assignment.  The convolutions of the
// neck are ordinary launches of the implicit-GEMM kernel.  All HBM-bound byte movers, 16-byte vectors, NHWC bf16.
#include "common.h"

#pragma clang fp contract(off)

namespace {

#define S_(s) reinterpret_cast<hipStream_t>(s)

// out[b,y,x,:] = lat[b,y,x,:] + top[b, (y*ht)/h, (x*wt)/w, :]      (sec. 3: nearest-neighbour upsampling to the finer map's size)
// (lat and out may be the same buffer -- the neck merges in place -- so neither is __restrict__)
__global__ __launch_bounds__(256) void upsample_add_kernel(const bf16_t* __restrict__ top, int ht, int wt, const bf16_t* lat,
                                                           bf16_t* out, int B, int h, int w, int C8) {
    const int64_t total = (int64_t)B * h * w * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % C8);
        int64_t pix = i / C8;
        const int x = (int)(pix % w);
        pix /= w;
        const int y = (int)(pix % h), b = (int)(pix / h);
        const int ys = (int)(((int64_t)y * ht) / h), xs = (int)(((int64_t)x * wt) / w);
        float a[8], t[8];
        unpack8(*reinterpret_cast<const u32x4*>(lat + i * 8), a);
        unpack8(*reinterpret_cast<const u32x4*>(top + ((((int64_t)b * ht + ys) * wt + xs) * C8 + cv) * 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += t[e];
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8(a);
    }
}

// gtop[b,ys,xs,:] (+)= sum of g over the fine pixels that read (ys, xs): rows [ceil(ys*h/ht), ceil((ys+1)*h/ht)), columns likewise
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const bf16_t* __restrict__ g, int h, int w, bf16_t* __restrict__ gtop, int B, int ht,
                                                               int wt, int C8, int accumulate) {
    const int64_t total = (int64_t)B * ht * wt * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % C8);
        int64_t pix = i / C8;
        const int xs = (int)(pix % wt);
        pix /= wt;
        const int ys = (int)(pix % ht), b = (int)(pix / ht);
        const int y0 = (int)(((int64_t)ys * h + ht - 1) / ht), y1 = (int)(((int64_t)(ys + 1) * h + ht - 1) / ht);
        const int x0 = (int)(((int64_t)xs * w + wt - 1) / wt), x1 = (int)(((int64_t)(xs + 1) * w + wt - 1) / wt);
        float acc[8];
        if (accumulate) unpack8(*reinterpret_cast<const u32x4*>(gtop + i * 8), acc);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        }
        for (int y = y0; y < y1 && y < h; ++y)
            for (int x = x0; x < x1 && x < w; ++x) {
                float v[8];
                unpack8(*reinterpret_cast<const u32x4*>(g + ((((int64_t)b * h + y) * w + x) * C8 + cv) * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        *reinterpret_cast<u32x4*>(gtop + i * 8) = pack8(acc);
    }
}

// y[b,i,j,:] = x[b,2i,2j,:]  (sec. 4.1: the extra RPN level is a stride-2 subsampling of the coarsest output map)
__global__ __launch_bounds__(256) void subsample2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int h, int w, int ho, int wo,
                                                         int C8) {
    const int64_t total = (int64_t)B * ho * wo * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % C8);
        int64_t pix = i / C8;
        const int xo = (int)(pix % wo);
        pix /= wo;
        const int yo = (int)(pix % ho), b = (int)(pix / ho);
        *reinterpret_cast<u32x4*>(y + i * 8) = *reinterpret_cast<const u32x4*>(x + ((((int64_t)b * h + 2 * yo) * w + 2 * xo) * C8 + cv) * 8);
    }
}

// gx[b,2i,2j,:] += gy[b,i,j,:]  (gx already holds the gradient of the map's other consumers)
__global__ __launch_bounds__(256) void subsample2_bwd_add_kernel(const bf16_t* __restrict__ gy, bf16_t* __restrict__ gx, int B, int h, int w, int ho,
                                                                 int wo, int C8) {
    const int64_t total = (int64_t)B * ho * wo * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % C8);
        int64_t pix = i / C8;
        const int xo = (int)(pix % wo);
        pix /= wo;
        const int yo = (int)(pix % ho), b = (int)(pix / ho);
        bf16_t* dst = gx + ((((int64_t)b * h + 2 * yo) * w + 2 * xo) * C8 + cv) * 8;
        float a[8], v[8];
        unpack8(*reinterpret_cast<const u32x4*>(dst), a);
        unpack8(*reinterpret_cast<const u32x4*>(gy + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += v[e];
        *reinterpret_cast<u32x4*>(dst) = pack8(a);
    }
}

// Lin et al. eq. (1), k = floor(k0 + log2(sqrt(w h) / 224)) with k0 = 4 clamped to [2, 4], without logarithms (bit-exact):
// k = 2 + [w h >= 112^2] + [w h >= 224^2]; w, h in input pixels from the relative box (oracle/fpn.py: roi_levels)
__global__ void roi_levels_kernel(const float* __restrict__ rois, int64_t n, float W, float H, int* __restrict__ levels) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(rois + i * 4);
        const float w = (r[2] - r[0]) * W, h = (r[3] - r[1]) * H;
        const float area = w * h;
        levels[i] = 2 + (area >= 112.f * 112.f ? 1 : 0) + (area >= 224.f * 224.f ? 1 : 0);
    }
}

static int grid_for(int64_t total) {
    const int64_t b = (total + 255) / 256;
    return (int)(b < 8192 ? (b < 1 ? 1 : b) : 8192);
}

}  // namespace

extern "C" int frcnn_upsample_add(const frcnn_bf16* top, int ht, int wt, const frcnn_bf16* lat, frcnn_bf16* out, int b, int h, int w, int c,
                                  frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(top && lat && out && b > 0 && h >= ht && w >= wt && ht > 0 && wt > 0 && c % 8 == 0, "upsample_add: bad arguments");
    hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((int64_t)b * h * w * (c / 8))), dim3(256), 0, S_(stream), reinterpret_cast<const bf16_t*>(top),
                       ht, wt, reinterpret_cast<const bf16_t*>(lat), reinterpret_cast<bf16_t*>(out), b, h, w, c / 8);
    FRCNN_CHECK_LAUNCH("upsample_add");
    return FRCNN_OK;
}

extern "C" int frcnn_upsample_add_bwd(const frcnn_bf16* g, int h, int w, frcnn_bf16* gtop, int b, int ht, int wt, int c, int accumulate,
                                      frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(g && gtop && b > 0 && h >= ht && w >= wt && ht > 0 && wt > 0 && c % 8 == 0, "upsample_add_bwd: bad arguments");
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(grid_for((int64_t)b * ht * wt * (c / 8))), dim3(256), 0, S_(stream),
                       reinterpret_cast<const bf16_t*>(g), h, w, reinterpret_cast<bf16_t*>(gtop), b, ht, wt, c / 8, accumulate);
    FRCNN_CHECK_LAUNCH("upsample_add_bwd");
    return FRCNN_OK;
}

extern "C" int frcnn_subsample2(const frcnn_bf16* x, frcnn_bf16* y, int b, int h, int w, int c, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(x && y && b > 0 && h > 0 && w > 0 && c % 8 == 0, "subsample2: bad arguments");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for((int64_t)b * ho * wo * (c / 8))), dim3(256), 0, S_(stream), reinterpret_cast<const bf16_t*>(x),
                       reinterpret_cast<bf16_t*>(y), b, h, w, ho, wo, c / 8);
    FRCNN_CHECK_LAUNCH("subsample2");
    return FRCNN_OK;
}

extern "C" int frcnn_subsample2_bwd_add(const frcnn_bf16* gy, frcnn_bf16* gx, int b, int h, int w, int c, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(gy && gx && b > 0 && h > 0 && w > 0 && c % 8 == 0, "subsample2_bwd_add: bad arguments");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    hipLaunchKernelGGL(subsample2_bwd_add_kernel, dim3(grid_for((int64_t)b * ho * wo * (c / 8))), dim3(256), 0, S_(stream),
                       reinterpret_cast<const bf16_t*>(gy), reinterpret_cast<bf16_t*>(gx), b, h, w, ho, wo, c / 8);
    FRCNN_CHECK_LAUNCH("subsample2_bwd_add");
    return FRCNN_OK;
}

extern "C" int frcnn_roi_assign_levels(const float* rois_rel, int64_t n, float img_w, float img_h, int32_t* levels, frcnn_stream_t stream) {
    FRCNN_CHECK_ARG(rois_rel && levels && n > 0, "roi_assign_levels: bad arguments");
    hipLaunchKernelGGL(roi_levels_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), rois_rel, n, img_w, img_h, levels);
    FRCNN_CHECK_LAUNCH("roi_assign_levels");
    return FRCNN_OK;
}
