/* CPU oracle (TEST INFRASTRUCTURE) -- a SECOND, independent restatement of the two TensorFlow-internal ops of the hot
 * path that have no other cross-check in this container (torchvision and TensorFlow are absent):
 *
 *   tf.image.crop_and_resize (bilinear, extrapolation_value 0) as called at reference
 *       models/detectors/fast_rcnn_detector.py:160-166, and its gradient w.r.t. the image
 *       (CropAndResizeGradImage: what tf.GradientTape runs for it at models/faster_rcnn.py:103);
 *   tf.image.combined_non_max_suppression as called at reference utils/post_processing.py:53-55.
 *
 * Written from the published contracts in a deliberately different style from oracle/roi.py (vectorised
 * torch gathers + autograd) and oracle/nms.c (sorted walk against a kept list): plain scalar loops, one output element at a
 * time; NMS in the textbook form "repeat: take the best live candidate, keep it, kill everything that overlaps it", the final
 * top-k by repeated arg-max.  tests/test_oracle_golden.py cross-checks the pairs on random inputs including out-of-range and
 * zero-area boxes, inverted corners, thresholds and per-class caps.  Two restatements that agree do not PIN either against
 * TensorFlow (parity stays unpinned, see oracle/__init__.py); they remove transcription errors of one author's single reading.
 *
 * Only tests/ may load this.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* image [B,H,W,C]; boxes [n,4] = normalised (y1, x1, y2, x2); box_index [n]; out [n,ch,cw,C] */
void indep_crop_and_resize(const float* image, int B, int H, int W, int C, const float* boxes, const int* box_index, int n, int ch, int cw,
                           float* out) {
    (void)B;
    for (int b = 0; b < n; ++b) {
        const float y1 = boxes[4 * b + 0], x1 = boxes[4 * b + 1], y2 = boxes[4 * b + 2], x2 = boxes[4 * b + 3];
        const float* img = image + (size_t)box_index[b] * H * W * C;
        const float height_scale = ch > 1 ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.0f;
        const float width_scale = cw > 1 ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.0f;
        for (int y = 0; y < ch; ++y) {
            const float in_y = ch > 1 ? y1 * (float)(H - 1) + (float)y * height_scale : 0.5f * (y1 + y2) * (float)(H - 1);
            for (int x = 0; x < cw; ++x) {
                float* o = out + (((size_t)b * ch + y) * cw + x) * C;
                const float in_x = cw > 1 ? x1 * (float)(W - 1) + (float)x * width_scale : 0.5f * (x1 + x2) * (float)(W - 1);
                if (in_y < 0.0f || in_y > (float)(H - 1) || in_x < 0.0f || in_x > (float)(W - 1)) {
                    for (int c = 0; c < C; ++c) o[c] = 0.0f;            /* extrapolation_value */
                    continue;
                }
                const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
                const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
                const float y_lerp = in_y - (float)top, x_lerp = in_x - (float)left;
                for (int c = 0; c < C; ++c) {
                    const float tl = img[((size_t)top * W + left) * C + c], tr = img[((size_t)top * W + right) * C + c];
                    const float bl = img[((size_t)bottom * W + left) * C + c], br = img[((size_t)bottom * W + right) * C + c];
                    const float t = tl + (tr - tl) * x_lerp;
                    const float bo = bl + (br - bl) * x_lerp;
                    o[c] = t + (bo - t) * y_lerp;
                }
            }
        }
    }
}

/* gradient of the above w.r.t. the image: grads [n,ch,cw,C] scattered (+=) into grad_image [B,H,W,C], which is zeroed here */
void indep_crop_and_resize_grad_image(const float* grads, int B, int H, int W, int C, const float* boxes, const int* box_index, int n, int ch,
                                      int cw, float* grad_image) {
    memset(grad_image, 0, sizeof(float) * (size_t)B * H * W * C);
    for (int b = 0; b < n; ++b) {
        const float y1 = boxes[4 * b + 0], x1 = boxes[4 * b + 1], y2 = boxes[4 * b + 2], x2 = boxes[4 * b + 3];
        float* gi = grad_image + (size_t)box_index[b] * H * W * C;
        const float height_scale = ch > 1 ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.0f;
        const float width_scale = cw > 1 ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.0f;
        for (int y = 0; y < ch; ++y) {
            const float in_y = ch > 1 ? y1 * (float)(H - 1) + (float)y * height_scale : 0.5f * (y1 + y2) * (float)(H - 1);
            if (in_y < 0.0f || in_y > (float)(H - 1)) continue;
            const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
            const float y_lerp = in_y - (float)top;
            for (int x = 0; x < cw; ++x) {
                const float in_x = cw > 1 ? x1 * (float)(W - 1) + (float)x * width_scale : 0.5f * (x1 + x2) * (float)(W - 1);
                if (in_x < 0.0f || in_x > (float)(W - 1)) continue;
                const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
                const float x_lerp = in_x - (float)left;
                const float* g = grads + (((size_t)b * ch + y) * cw + x) * C;
                for (int c = 0; c < C; ++c) {
                    const float dtop = (1.0f - y_lerp) * g[c], dbottom = y_lerp * g[c];
                    gi[((size_t)top * W + left) * C + c] += (1.0f - x_lerp) * dtop;
                    gi[((size_t)top * W + right) * C + c] += x_lerp * dtop;
                    gi[((size_t)bottom * W + left) * C + c] += (1.0f - x_lerp) * dbottom;
                    gi[((size_t)bottom * W + right) * C + c] += x_lerp * dbottom;
                }
            }
        }
    }
}

static float lo2(float a, float b) { return a < b ? a : b; }
static float hi2(float a, float b) { return a > b ? a : b; }

/* overlap ratio of two boxes given as two opposite corners in any order */
static float overlap(const float* p, const float* q) {
    const float pa0 = lo2(p[0], p[2]), pa1 = hi2(p[0], p[2]), pb0 = lo2(p[1], p[3]), pb1 = hi2(p[1], p[3]);
    const float qa0 = lo2(q[0], q[2]), qa1 = hi2(q[0], q[2]), qb0 = lo2(q[1], q[3]), qb1 = hi2(q[1], q[3]);
    const float area_p = (pa1 - pa0) * (pb1 - pb0), area_q = (qa1 - qa0) * (qb1 - qb0);
    if (area_p <= 0.0f || area_q <= 0.0f) return 0.0f;
    const float da = hi2(lo2(pa1, qa1) - hi2(pa0, qa0), 0.0f);
    const float db = hi2(lo2(pb1, qb1) - hi2(pb0, qb0), 0.0f);
    const float inter = da * db;
    return inter / (area_p + area_q - inter);
}

/* boxes [B,N,q,4] (q = 1 or C), scores [B,N,C]; outputs [B,T,4], [B,T], int [B,T], int [B] */
void indep_combined_nms(const float* boxes, const float* scores, int B, int N, int q, int C, int max_per_class, int T, float iou_thr,
                        float score_thr, float* out_boxes, float* out_scores, int* out_classes, int* out_valid) {
    char* live = (char*)malloc((size_t)(N > 0 ? N : 1));
    const int cap = C * max_per_class;
    int* pool_box = (int*)malloc(sizeof(int) * (size_t)(cap + 1));
    int* pool_cls = (int*)malloc(sizeof(int) * (size_t)(cap + 1));
    float* pool_score = (float*)malloc(sizeof(float) * (size_t)(cap + 1));
    char* taken = (char*)malloc((size_t)(cap + 1));
    for (int b = 0; b < B; ++b) {
        int pooled = 0;
        for (int c = 0; c < C; ++c) {
            const int slot = q == 1 ? 0 : c;
            for (int i = 0; i < N; ++i) live[i] = scores[((size_t)b * N + i) * C + c] > score_thr;
            for (int kept = 0; kept < max_per_class; ++kept) {
                int best = -1;
                for (int i = 0; i < N; ++i)                          /* best live candidate; the FIRST one among equals */
                    if (live[i] && (best < 0 || scores[((size_t)b * N + i) * C + c] > scores[((size_t)b * N + best) * C + c])) best = i;
                if (best < 0) break;
                live[best] = 0;
                pool_box[pooled] = best;
                pool_cls[pooled] = c;
                pool_score[pooled] = scores[((size_t)b * N + best) * C + c];
                ++pooled;
                const float* kb = boxes + (((size_t)b * N + best) * q + slot) * 4;
                for (int i = 0; i < N; ++i)
                    if (live[i] && overlap(boxes + (((size_t)b * N + i) * q + slot) * 4, kb) > iou_thr) live[i] = 0;
            }
        }
        /* the image's detections: the T best of the pool, earlier pool entries first among equals */
        memset(taken, 0, (size_t)(cap + 1));
        int nv = 0;
        for (int t = 0; t < T; ++t) {
            float* ob = out_boxes + ((size_t)b * T + t) * 4;
            int best = -1;
            for (int k = 0; k < pooled; ++k)
                if (!taken[k] && (best < 0 || pool_score[k] > pool_score[best])) best = k;
            if (best < 0) {
                ob[0] = ob[1] = ob[2] = ob[3] = 0.0f;
                out_scores[(size_t)b * T + t] = 0.0f;
                out_classes[(size_t)b * T + t] = 0;
                continue;
            }
            taken[best] = 1;
            ++nv;
            const float* src = boxes + (((size_t)b * N + pool_box[best]) * q + (q == 1 ? 0 : pool_cls[best])) * 4;
            for (int k = 0; k < 4; ++k) ob[k] = lo2(hi2(src[k], 0.0f), 1.0f);           /* clip_boxes=True */
            out_scores[(size_t)b * T + t] = pool_score[best];
            out_classes[(size_t)b * T + t] = pool_cls[best];
        }
        out_valid[b] = nv;
    }
    free(live); free(pool_box); free(pool_cls); free(pool_score); free(taken);
}
