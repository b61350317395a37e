/*
 * oracle/postproc_ref.c — CPU restatement (plain C, scalar, single thread) of the reference's
 * detection post-processing.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker; the product path never links or calls it.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 * torchvision is a third-party dependency absent from the reference tree (unpinned; contemporaneous
 * with "pytorch 1.10" => torchvision 0.11.x, README.md:5-6): ref_batched_nms restates its published
 * algorithm (ops/boxes.py batched_nms -> _batched_nms_coordinate_trick for <= 1000 boxes, and
 * csrc/ops/cpu/nms_kernel.cpp nms_kernel_impl).  Its greedy core is pinned by the reference's own live
 * DataEncoder._box_nms on integer boxes (tests/golden, see DESIGN.md "Oracle").
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).  No FMA contraction: the
 * fp32 rounding sequence is part of the contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* utill/utills.py:58-73 coords_origin_fcos: (x*s + s//2, y*s + s//2), x fastest */
void ref_coords(int H, int W, int stride, float* out /* [H*W][2] */) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            out[2 * (y * W + x) + 0] = (float)(x * stride) + (float)(stride / 2);
            out[2 * (y * W + x) + 1] = (float)(y * stride) + (float)(stride / 2);
        }
}

static float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

/* model/modules/head.py:52-66 for one image: cls [L][C], cnt [L], reg [L][4], coords [L][2]
 * score = sqrt(max_c sigmoid(cls) * sigmoid(cnt)); class = first argmax + 1; box = c -/+ ltrb */
void ref_decode(const float* cls, const float* cnt, const float* reg, const float* coords, int L, int C,
                float* scores, int32_t* classes, float* boxes) {
    for (int i = 0; i < L; ++i) {
        float best = -1.0f;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float s = sigmoidf_(cls[(size_t)i * C + c]);
            if (s > best) { best = s; bi = c; }
        }
        scores[i] = sqrtf(best * sigmoidf_(cnt[i]));
        classes[i] = bi + 1;
        boxes[4 * i + 0] = coords[2 * i + 0] - reg[4 * i + 0];
        boxes[4 * i + 1] = coords[2 * i + 1] - reg[4 * i + 1];
        boxes[4 * i + 2] = coords[2 * i + 0] + reg[4 * i + 2];
        boxes[4 * i + 3] = coords[2 * i + 1] + reg[4 * i + 3];
    }
}

/* stable descending argsort by score (ties: lower index first) */
typedef struct { float s; int32_t i; } si_t;
static int cmp_desc(const void* a, const void* b) {
    const si_t* x = (const si_t*)a; const si_t* y = (const si_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
void ref_argsort_desc(const float* scores, int n, int32_t* order) {
    si_t* t = (si_t*)malloc(sizeof(si_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { t[i].s = scores[i]; t[i].i = i; }
    qsort(t, (size_t)n, sizeof(si_t), cmp_desc);
    for (int i = 0; i < n; ++i) order[i] = t[i].i;
    free(t);
}

/* head.py:69-80: torch.topk(score, K, largest, sorted) -> indices; tie order defined as lower index first */
void ref_topk(const float* scores, int L, int K, int32_t* idx) {
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)L);
    ref_argsort_desc(scores, L, order);
    memcpy(idx, order, sizeof(int32_t) * (size_t)K);
    free(order);
}

/* torchvision nms_kernel_impl (CPU): boxes [n][4] xyxy, returns kept indices in score-descending order.
 * suppress j when ovr > iou_threshold, the comparison done in double as in the C++ source
 * (scalar_t ovr vs const double iou_threshold). */
int ref_nms(const float* boxes, const float* scores, int n, double iou_thr, int32_t* keep) {
    if (n == 0) return 0;
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    uint8_t* sup = (uint8_t*)calloc((size_t)n, 1);
    float* areas = (float*)malloc(sizeof(float) * (size_t)n);
    ref_argsort_desc(scores, n, order);
    for (int i = 0; i < n; ++i) areas[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    int nk = 0;
    for (int _i = 0; _i < n; ++_i) {
        const int i = order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        const float iarea = areas[i];
        for (int _j = _i + 1; _j < n; ++_j) {
            const int j = order[_j];
            if (sup[j]) continue;
            const float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
            const float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float ovr = inter / ((iarea + areas[j]) - inter);
            if ((double)ovr > iou_thr) sup[j] = 1;
        }
    }
    free(order); free(sup); free(areas);
    return nk;
}

/* torchvision batched_nms, coordinate-trick branch (head.py:94 call site; the authors' own commented
 * restatement at head.py:104-149 documents the same offsets = idxs * (max_coordinate + 1)). */
int ref_batched_nms(const float* boxes, const float* scores, const int64_t* classes, int n, double iou_thr,
                    int32_t* keep) {
    if (n == 0) return 0;
    float mx = -INFINITY;
    for (int i = 0; i < 4 * n; ++i) mx = fmaxf(mx, boxes[i]);
    float* ob = (float*)malloc(sizeof(float) * 4 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const float off = (float)classes[i] * (mx + 1.0f);
        for (int k = 0; k < 4; ++k) ob[4 * i + k] = boxes[4 * i + k] + off;
    }
    const int nk = ref_nms(ob, scores, n, iou_thr, keep);
    free(ob);
    return nk;
}

/* head.py:84-102 FCOSHead.post_process for one image on top-k rows (score-descending):
 * mask score >= thr, batched_nms, gather.  Returns count; keep = indices into the K input rows. */
int ref_post_process(const float* scores, const int64_t* classes, const float* boxes, int K, float score_thr,
                     double iou_thr, int32_t* keep) {
    int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * (size_t)(K > 0 ? K : 1));
    int n = 0;
    for (int i = 0; i < K; ++i) if (scores[i] >= score_thr) sel[n++] = i;
    float* b = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    float* s = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int64_t* c = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { memcpy(b + 4 * i, boxes + 4 * sel[i], 16); s[i] = scores[sel[i]]; c[i] = classes[sel[i]]; }
    const int nk = ref_batched_nms(b, s, c, n, iou_thr, keep);
    for (int i = 0; i < nk; ++i) keep[i] = sel[keep[i]];
    free(sel); free(b); free(s); free(c);
    return nk;
}

/* utill/utills.py:221-255 DataEncoder._box_nms: "+1" areas, keep while ovr <= threshold (fp32 compare),
 * mode 0 'union' / 1 'min'; returns kept original indices, score-descending. */
int ref_box_nms_plus1(const float* boxes, const float* scores, int n, float thr, int mode, int32_t* keep) {
    if (n == 0) return 0;
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    float* areas = (float*)malloc(sizeof(float) * (size_t)n);
    ref_argsort_desc(scores, n, order);
    for (int i = 0; i < n; ++i)
        areas[i] = ((boxes[4 * i + 2] - boxes[4 * i + 0]) + 1.0f) * ((boxes[4 * i + 3] - boxes[4 * i + 1]) + 1.0f);
    int nk = 0, len = n;
    while (len > 0) {
        const int i = order[0];
        keep[nk++] = i;
        if (len == 1) break;
        int m = 0;
        for (int q = 1; q < len; ++q) {
            const int j = order[q];
            const float xx1 = fmaxf(boxes[4 * j], boxes[4 * i]), yy1 = fmaxf(boxes[4 * j + 1], boxes[4 * i + 1]);
            const float xx2 = fminf(boxes[4 * j + 2], boxes[4 * i + 2]), yy2 = fminf(boxes[4 * j + 3], boxes[4 * i + 3]);
            const float w = fmaxf((xx2 - xx1) + 1.0f, 0.0f), h = fmaxf((yy2 - yy1) + 1.0f, 0.0f);
            const float inter = w * h;
            float ovr;
            if (mode == 0) ovr = inter / ((areas[i] + areas[j]) - inter);
            else ovr = inter / fminf(areas[j], areas[i]);
            if (ovr <= thr) order[m++] = j;
        }
        len = m;
    }
    free(order); free(areas);
    return nk;
}

/* utill/utills.py:201-218 DataEncoder._box_iou (plus_one=1) / test.py:23-53 iou_2d (plus_one=0) */
void ref_pairwise_iou(const float* a, const float* b, int Na, int Nb, int plus_one, float* out) {
    const float p = plus_one ? 1.0f : 0.0f;
    for (int i = 0; i < Na; ++i)
        for (int j = 0; j < Nb; ++j) {
            const float ltx = fmaxf(a[4 * i], b[4 * j]), lty = fmaxf(a[4 * i + 1], b[4 * j + 1]);
            const float rbx = fminf(a[4 * i + 2], b[4 * j + 2]), rby = fminf(a[4 * i + 3], b[4 * j + 3]);
            const float w = fmaxf((rbx - ltx) + p, 0.0f), h = fmaxf((rby - lty) + p, 0.0f);
            const float inter = w * h;
            const float a1 = ((a[4 * i + 2] - a[4 * i]) + p) * ((a[4 * i + 3] - a[4 * i + 1]) + p);
            const float a2 = ((b[4 * j + 2] - b[4 * j]) + p) * ((b[4 * j + 3] - b[4 * j + 1]) + p);
            out[(size_t)i * Nb + j] = inter / ((a1 + a2) - inter);
        }
}

/* head.py:152-162 ClipBoxes */
void ref_clip_boxes(float* boxes, int n, int img_h, int img_w) {
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) boxes[4 * i + k] = fmaxf(boxes[4 * i + k], 0.0f);
        boxes[4 * i + 0] = fminf(boxes[4 * i + 0], (float)(img_w - 1));
        boxes[4 * i + 2] = fminf(boxes[4 * i + 2], (float)(img_w - 1));
        boxes[4 * i + 1] = fminf(boxes[4 * i + 1], (float)(img_h - 1));
        boxes[4 * i + 3] = fminf(boxes[4 * i + 3], (float)(img_h - 1));
    }
}

/* dataset/voc.py:57-58,104,155: transforms.ToTensor (u8/255) + Normalize ((v-mean)/std); out [H*W][3] */
void ref_normalize_u8(const uint8_t* img, int n_pix, const float* mean3, const float* std3, float* out) {
    for (int i = 0; i < n_pix; ++i)
        for (int c = 0; c < 3; ++c) out[3 * i + c] = ((float)img[3 * i + c] / 255.0f - mean3[c]) / std3[c];
}

/* Test_coco.py:147-151: boxes /= scale; xyxy -> xywh */
void ref_boxes_rescale_xywh(float* boxes, int n, float scale) {
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) boxes[4 * i + k] = boxes[4 * i + k] / scale;
        boxes[4 * i + 2] = boxes[4 * i + 2] - boxes[4 * i + 0];
        boxes[4 * i + 3] = boxes[4 * i + 3] - boxes[4 * i + 1];
    }
}
