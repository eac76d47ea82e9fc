/* CPU oracle (TEST INFRASTRUCTURE) -- plain-C restatement of
 * tf.image.combined_non_max_suppression as called at reference
 * utils/post_processing.py:53-55 (pad_per_class=False, clip_boxes=True).
 *
 * [TF-ext] The arithmetic lives in TensorFlow 2.x (tensorflow/core/kernels/image/
 * non_max_suppression_op.cc, version unpinned by the reference, source not present here);
 * restated from its public contract, SURVEY.md A.5.  PARITY UNPINNED against TensorFlow.
 *
 * Contract restated:
 *   per (image b, class c): candidates = boxes with score > score_threshold, visited in
 *   descending score (ties: lower box index first -- TF's heap order is implementation
 *   defined); a candidate is kept iff IoU with every already kept box of that class is
 *   <= iou_threshold; stop at max_output_size_per_class.
 *   per image: concatenate kept (class-major), sort by score descending (stable), keep first
 *   max_total_size, clip coords to [0,1], zero-pad.
 *   IoU: corner pairs (0,2) and (1,3) are min/max-normalised; 0 if either area <= 0.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC  (see oracle/Makefile).
 */
#include <stdlib.h>
#include <string.h>

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

float oracle_nms_iou(const float* bi, const float* bj) {
    const float y0i = fminf_(bi[0], bi[2]), x0i = fminf_(bi[1], bi[3]);
    const float y1i = fmaxf_(bi[0], bi[2]), x1i = fmaxf_(bi[1], bi[3]);
    const float y0j = fminf_(bj[0], bj[2]), x0j = fminf_(bj[1], bj[3]);
    const float y1j = fmaxf_(bj[0], bj[2]), x1j = fmaxf_(bj[1], bj[3]);
    const float area_i = (y1i - y0i) * (x1i - x0i);
    const float area_j = (y1j - y0j) * (x1j - x0j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iy0 = fmaxf_(y0i, y0j), ix0 = fmaxf_(x0i, x0j);
    const float iy1 = fminf_(y1i, y1j), ix1 = fminf_(x1i, x1j);
    const float inter = fmaxf_(iy1 - iy0, 0.0f) * fmaxf_(ix1 - ix0, 0.0f);
    return inter / (area_i + area_j - inter);
}

typedef struct { float score; int idx; int cls; int ord; } cand_t;

static int cmp_desc(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    if (x->cls != y->cls) return x->cls < y->cls ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static int cmp_desc_ord(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->ord > y->ord) - (x->ord < y->ord);
}

/* boxes [B,N,q,4], scores [B,N,C] (row stride score_stride floats, first class at scores+0).
 * q is 1 or C.  Outputs: out_boxes [B,T,4], out_scores [B,T], out_classes [B,T] (int32),
 * out_valid [B] (int32).  Returns 0. */
int oracle_combined_nms(const float* boxes, const float* scores, int B, int N, int q, int C,
                        int score_stride, int max_per_class, int max_total, float iou_thr,
                        float score_thr, float* out_boxes, float* out_scores, int* out_classes,
                        int* out_valid) {
    cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * (size_t)(N > 0 ? N : 1));
    cand_t* kept = (cand_t*)malloc(sizeof(cand_t) * (size_t)(C * max_per_class + 1));
    int* sel = (int*)malloc(sizeof(int) * (size_t)(max_per_class + 1));
    for (int b = 0; b < B; ++b) {
        int nk = 0;
        for (int c = 0; c < C; ++c) {
            const int bc = (q == 1) ? 0 : c;
            int nc = 0;
            for (int i = 0; i < N; ++i) {
                const float s = scores[((size_t)b * N + i) * score_stride + c];
                if (s > score_thr) { cand[nc].score = s; cand[nc].idx = i; cand[nc].cls = c; cand[nc].ord = 0; ++nc; }
            }
            qsort(cand, (size_t)nc, sizeof(cand_t), cmp_desc);
            int ns = 0;
            for (int t = 0; t < nc && ns < max_per_class; ++t) {
                const float* bi = boxes + (((size_t)b * N + cand[t].idx) * q + bc) * 4;
                int ok = 1;
                for (int j = ns - 1; j >= 0; --j) {
                    const float* bj = boxes + (((size_t)b * N + sel[j]) * q + bc) * 4;
                    if (oracle_nms_iou(bi, bj) > iou_thr) { ok = 0; break; }
                }
                if (ok) { sel[ns++] = cand[t].idx; kept[nk] = cand[t]; kept[nk].ord = nk; ++nk; }
            }
        }
        qsort(kept, (size_t)nk, sizeof(cand_t), cmp_desc_ord);
        const int nv = nk < max_total ? nk : max_total;
        out_valid[b] = nv;
        for (int t = 0; t < max_total; ++t) {
            float* ob = out_boxes + ((size_t)b * max_total + t) * 4;
            if (t < nv) {
                const int bc = (q == 1) ? 0 : kept[t].cls;
                const float* bi = boxes + (((size_t)b * N + kept[t].idx) * q + bc) * 4;
                for (int k = 0; k < 4; ++k) ob[k] = fminf_(fmaxf_(bi[k], 0.0f), 1.0f);
                out_scores[(size_t)b * max_total + t] = kept[t].score;
                out_classes[(size_t)b * max_total + t] = kept[t].cls;
            } else {
                ob[0] = ob[1] = ob[2] = ob[3] = 0.0f;
                out_scores[(size_t)b * max_total + t] = 0.0f;
                out_classes[(size_t)b * max_total + t] = 0;
            }
        }
    }
    free(cand); free(kept); free(sel);
    return 0;
}
