/*
 * Scalar C restatement of the detection-head box pipeline -- TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the algorithms live in torchvision 0.16.2 (un-vendored dependency of the
 * reference, see reference Pipfile:8), which is absent from the build container; the reference
 * holds no tests or golden vectors for them.  This file restates the published algorithms
 * independently of oracle/tv_reference.py (plain loops, no tensor library) so that the two
 * restatements can cross-check each other.  Reference call sites that put these functions on
 * the path: object_detection/retinanet/retinanet_model.py:22-27,42-54 (AnchorGenerator,
 * RetinaNet), object_detection/train.py:62,84 (compute_loss / postprocess_detections via
 * model(...)), object_detection/predict.py:61,72 (postprocess + plain nms).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * Every float expression below is a sequence of single IEEE-754 binary32 operations.
 * Nothing in the product (object_detection_amd/) links or loads this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- spec exp / sigmoid (same sequence as oracle/spec_math.py) --------------------------- */
static inline float pow2i(int k) {
    union { uint32_t u; float f; } v;
    v.u = (uint32_t)(k + 127) << 23;
    return v.f;
}

float orc_expf(float x) {
    const float LOG2E = 1.44269504088896341f, LN2_HI = 0.693359375f, LN2_LO = -2.12194440e-4f;
    const float EXP_HI = 88.72283935546875f, EXP_LO = -103.97208404541015625f;
    if (x != x) return x;
    if (x > EXP_HI) return INFINITY;
    if (x < EXP_LO) return 0.0f;
    float k = rintf(x * LOG2E);
    float r = x - k * LN2_HI;
    r = r - k * LN2_LO;
    float p = 1.9875691500e-4f * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    float rr = r * r;
    p = p * rr + r;
    p = p + 1.0f;
    int ki = (int)k;
    int k1 = ki / 2;
    int k2 = ki - k1;
    float t = p * pow2i(k1);
    return t * pow2i(k2);
}

float orc_sigmoidf(float x) {
    float e = orc_expf(-x);
    float d = 1.0f + e;
    return 1.0f / d;
}

void orc_expf_array(const float* x, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = orc_expf(x[i]);
}

void orc_sigmoidf_array(const float* x, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = orc_sigmoidf(x[i]);
}

/* ---- torchvision/ops/boxes.py: box_iou ---------------------------------------------------- */
static inline float iou_pair(const float* g, float garea, const float* a, float aarea) {
    float ltx = g[0] > a[0] ? g[0] : a[0];
    float lty = g[1] > a[1] ? g[1] : a[1];
    float rbx = g[2] < a[2] ? g[2] : a[2];
    float rby = g[3] < a[3] ? g[3] : a[3];
    float w = rbx - ltx;
    float h = rby - lty;
    if (w < 0.0f) w = 0.0f;
    if (h < 0.0f) h = 0.0f;
    float inter = w * h;
    float uni = garea + aarea;
    uni = uni - inter;
    return inter / uni;
}

static inline float area_of(const float* b) {
    float w = b[2] - b[0];
    float h = b[3] - b[1];
    return w * h;
}

/* out[m*n]: row g, column a */
void orc_box_iou(const float* gt, int64_t m, const float* anchors, int64_t n, float* out) {
    for (int64_t g = 0; g < m; ++g) {
        float ga = area_of(gt + 4 * g);
        for (int64_t a = 0; a < n; ++a)
            out[g * n + a] = iou_pair(gt + 4 * g, ga, anchors + 4 * a, area_of(anchors + 4 * a));
    }
}

/* ---- box_iou + Matcher(high, low, allow_low_quality_matches) fused, O(M) extra memory ----- */
void orc_match_retinanet(const float* gt, int64_t m, const float* anchors, int64_t n, float high,
                         float low, int allow_low_quality, int64_t* out) {
    if (m == 0) { /* RetinaNet.compute_loss empty-target guard */
        for (int64_t a = 0; a < n; ++a) out[a] = -1;
        return;
    }
    float* garea = (float*)malloc(sizeof(float) * (size_t)m);
    float* gmax = (float*)malloc(sizeof(float) * (size_t)m);
    int64_t* all = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t g = 0; g < m; ++g) { garea[g] = area_of(gt + 4 * g); gmax[g] = -INFINITY; }
    for (int64_t a = 0; a < n; ++a) {
        float aa = area_of(anchors + 4 * a);
        float best = 0.0f; int64_t bi = 0;
        for (int64_t g = 0; g < m; ++g) {
            float v = iou_pair(gt + 4 * g, garea[g], anchors + 4 * a, aa);
            if (g == 0 || v > best) { best = v; bi = g; }   /* first maximal index */
            if (v > gmax[g]) gmax[g] = v;
        }
        all[a] = bi;
        if (best < low) out[a] = -1;
        else if (best < high) out[a] = -2;
        else out[a] = bi;
    }
    if (allow_low_quality) {
        for (int64_t a = 0; a < n; ++a) {
            float aa = area_of(anchors + 4 * a);
            for (int64_t g = 0; g < m; ++g) {
                float v = iou_pair(gt + 4 * g, garea[g], anchors + 4 * a, aa);
                if (v == gmax[g]) { out[a] = all[a]; break; }
            }
        }
    }
    free(garea); free(gmax); free(all);
}

/* ---- Matcher on an explicit [m,n] quality matrix ------------------------------------------ */
void orc_matcher(const float* q, int64_t m, int64_t n, float high, float low, int allow_low_quality,
                 int64_t* out) {
    float* gmax = (float*)malloc(sizeof(float) * (size_t)m);
    int64_t* all = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t g = 0; g < m; ++g) {
        float b = q[g * n];
        for (int64_t a = 1; a < n; ++a) if (q[g * n + a] > b) b = q[g * n + a];
        gmax[g] = b;
    }
    for (int64_t a = 0; a < n; ++a) {
        float best = q[a]; int64_t bi = 0;
        for (int64_t g = 1; g < m; ++g) if (q[g * n + a] > best) { best = q[g * n + a]; bi = g; }
        all[a] = bi;
        out[a] = best < low ? -1 : (best < high ? -2 : bi);
    }
    if (allow_low_quality)
        for (int64_t a = 0; a < n; ++a)
            for (int64_t g = 0; g < m; ++g)
                if (q[g * n + a] == gmax[g]) { out[a] = all[a]; break; }
    free(gmax); free(all);
}

/* ---- torchvision/models/detection/fcos.py: matching half of FCOS.compute_loss ------------- */
void orc_match_fcos(const float* gt, int64_t m, const float* anchors, int64_t n, int64_t n_first_level,
                    int64_t n_last_level, float radius, int64_t* out) {
    if (m == 0) { for (int64_t a = 0; a < n; ++a) out[a] = -1; return; }
    for (int64_t a = 0; a < n; ++a) {
        const float* A = anchors + 4 * a;
        float cx = (A[0] + A[2]) / 2.0f, cy = (A[1] + A[3]) / 2.0f;
        float size = A[2] - A[0];
        float lower = (a < n_first_level) ? 0.0f : size * 4.0f;
        float upper = (a >= n - n_last_level) ? INFINITY : size * 8.0f;
        float rs = radius * size;
        float best = 0.0f; int64_t bi = 0;
        for (int64_t g = 0; g < m; ++g) {
            const float* G = gt + 4 * g;
            float gcx = (G[0] + G[2]) / 2.0f, gcy = (G[1] + G[3]) / 2.0f;
            float dx = fabsf(cx - gcx), dy = fabsf(cy - gcy);
            float dmax = dx > dy ? dx : dy;
            int match = dmax < rs;
            float l = cx - G[0], t = cy - G[1], r = G[2] - cx, b = G[3] - cy;
            float mn = l; if (t < mn) mn = t; if (r < mn) mn = r; if (b < mn) mn = b;
            float mx = l; if (t > mx) mx = t; if (r > mx) mx = r; if (b > mx) mx = b;
            match = match && (mn > 0.0f) && (mx > lower) && (mx < upper);
            float garea = (G[2] - G[0]) * (G[3] - G[1]);
            float v = (match ? 1.0f : 0.0f) * (1e8f - garea);
            if (g == 0 || v > best) { best = v; bi = g; }
        }
        out[a] = (best < 1e-5f) ? -1 : bi;
    }
}

/* ---- torchvision/csrc/ops/cpu/nms_kernel.cpp (float_rule = 0) and the decision rule of
 *      torchvision/csrc/ops/cuda/nms_kernel.cu (float_rule = 1: the fp32 IoU is compared with the threshold
 *      rounded to float; the greedy order and the IoU formula are the same) ------------------ */
typedef struct { float s; int64_t i; } sidx_t;
static int cmp_desc_stable(const void* a, const void* b) {
    const sidx_t* x = (const sidx_t*)a; const sidx_t* y = (const sidx_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int64_t orc_nms_rule(const float* boxes, const float* scores, int64_t n, double thr, int float_rule, int64_t* keep) {
    if (n == 0) return 0;
    sidx_t* order = (sidx_t*)malloc(sizeof(sidx_t) * (size_t)n);
    float* areas = (float*)malloc(sizeof(float) * (size_t)n);
    unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) {
        order[i].s = scores[i]; order[i].i = i;
        areas[i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    }
    qsort(order, (size_t)n, sizeof(sidx_t), cmp_desc_stable);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j].i;
            if (sup[j]) continue;
            float xx1 = ix1 < boxes[4 * j] ? boxes[4 * j] : ix1;           /* std::max(ix1, x1[j]) */
            float yy1 = iy1 < boxes[4 * j + 1] ? boxes[4 * j + 1] : iy1;
            float xx2 = boxes[4 * j + 2] < ix2 ? boxes[4 * j + 2] : ix2;   /* std::min(ix2, x2[j]) */
            float yy2 = boxes[4 * j + 3] < iy2 ? boxes[4 * j + 3] : iy2;
            float dw = xx2 - xx1, dh = yy2 - yy1;
            float w = 0.0f < dw ? dw : 0.0f;
            float h = 0.0f < dh ? dh : 0.0f;
            float inter = w * h;
            float den = iarea + areas[j];
            den = den - inter;
            float ovr = inter / den;
            if (float_rule ? (ovr > (float)thr) : ((double)ovr > thr)) sup[j] = 1;
        }
    }
    free(order); free(areas); free(sup);
    return nk;
}

int64_t orc_nms(const float* boxes, const float* scores, int64_t n, double thr, int64_t* keep) {
    return orc_nms_rule(boxes, scores, n, thr, 0, keep);
}

/* mode 0 = _batched_nms_coordinate_trick, 1 = _batched_nms_vanilla */
int64_t orc_batched_nms_rule(const float* boxes, const float* scores, const int64_t* labels, int64_t n,
                             double thr, int mode, int float_rule, int64_t* keep) {
    if (n == 0) return 0;
    if (mode == 0) {
        float mx = boxes[0];
        for (int64_t i = 1; i < 4 * n; ++i) if (boxes[i] > mx) mx = boxes[i];
        float step = mx + 1.0f;
        float* ob = (float*)malloc(sizeof(float) * 4 * (size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            float off = (float)labels[i] * step;
            for (int c = 0; c < 4; ++c) ob[4 * i + c] = boxes[4 * i + c] + off;
        }
        int64_t nk = orc_nms_rule(ob, scores, n, thr, float_rule, keep);
        free(ob);
        return nk;
    }
    unsigned char* mask = (unsigned char*)calloc((size_t)n, 1);
    unsigned char* done = (unsigned char*)calloc((size_t)n, 1);
    float* cb = (float*)malloc(sizeof(float) * 4 * (size_t)n);
    float* cs = (float*)malloc(sizeof(float) * (size_t)n);
    int64_t* ci = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* ck = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t s = 0; s < n; ++s) {
        if (done[s]) continue;
        int64_t cls = labels[s], c = 0;
        for (int64_t i = s; i < n; ++i)
            if (labels[i] == cls) {
                done[i] = 1; ci[c] = i; cs[c] = scores[i];
                memcpy(cb + 4 * c, boxes + 4 * i, 4 * sizeof(float)); ++c;
            }
        int64_t nk = orc_nms_rule(cb, cs, c, thr, float_rule, ck);
        for (int64_t k = 0; k < nk; ++k) mask[ci[ck[k]]] = 1;
    }
    sidx_t* order = (sidx_t*)malloc(sizeof(sidx_t) * (size_t)n);
    int64_t nk = 0;
    for (int64_t i = 0; i < n; ++i) if (mask[i]) { order[nk].s = scores[i]; order[nk].i = i; ++nk; }
    qsort(order, (size_t)nk, sizeof(sidx_t), cmp_desc_stable);
    for (int64_t k = 0; k < nk; ++k) keep[k] = order[k].i;
    free(mask); free(done); free(cb); free(cs); free(ci); free(ck); free(order);
    return nk;
}

int64_t orc_batched_nms(const float* boxes, const float* scores, const int64_t* labels, int64_t n,
                        double thr, int mode, int64_t* keep) {
    return orc_batched_nms_rule(boxes, scores, labels, n, thr, mode, 0, keep);
}

/* ---- torchvision/models/detection/_utils.py: BoxCoder.decode_single (weights 1,1,1,1) ----- */
void orc_decode_single(const float* rel, const float* anchors, int64_t n, float clip, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        const float* A = anchors + 4 * i; const float* R = rel + 4 * i;
        float w = A[2] - A[0], h = A[3] - A[1];
        float hw = 0.5f * w, hh = 0.5f * h;
        float cx = A[0] + hw, cy = A[1] + hh;
        float dx = R[0] / 1.0f, dy = R[1] / 1.0f, dw = R[2] / 1.0f, dh = R[3] / 1.0f;
        if (dw > clip) dw = clip;
        if (dh > clip) dh = clip;
        float pcx = dx * w; pcx = pcx + cx;
        float pcy = dy * h; pcy = pcy + cy;
        float pw = orc_expf(dw) * w;
        float ph = orc_expf(dh) * h;
        float cw = 0.5f * pw, ch = 0.5f * ph;
        out[4 * i + 0] = pcx - cw; out[4 * i + 1] = pcy - ch;
        out[4 * i + 2] = pcx + cw; out[4 * i + 3] = pcy + ch;
    }
}

/* BoxLinearCoder(normalize_by_size=True).decode */
void orc_decode_linear(const float* rel, const float* anchors, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        const float* A = anchors + 4 * i; const float* R = rel + 4 * i;
        float sx = A[0] + A[2], sy = A[1] + A[3];
        float cx = 0.5f * sx, cy = 0.5f * sy;
        float w = A[2] - A[0], h = A[3] - A[1];
        float r0 = R[0] * w, r1 = R[1] * h, r2 = R[2] * w, r3 = R[3] * h;
        out[4 * i + 0] = cx - r0; out[4 * i + 1] = cy - r1;
        out[4 * i + 2] = cx + r2; out[4 * i + 3] = cy + r3;
    }
}

/* clip_boxes_to_image: clamp(min=0, max=size) = min(max(x, 0), size) */
void orc_clip(float* boxes, int64_t n, float height, float width) {
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < 4; ++c) {
            float lim = (c & 1) ? height : width;
            float v = boxes[4 * i + c];
            v = v < 0.0f ? 0.0f : v;
            v = v > lim ? lim : v;
            boxes[4 * i + c] = v;
        }
}

/* score filter + top-k of one (image, level): spec sigmoid, > thresh, k best, ties -> lower index.
 * ctr == NULL: RetinaNet score; ctr != NULL (one per anchor): FCOS sqrt(sig(cls) * sig(ctr)). */
/* "a ranks before b": higher score first, lower index first among equal scores */
static inline int ranks_before(const sidx_t* a, const sidx_t* b) {
    return a->s > b->s || (a->s == b->s && a->i < b->i);
}

int64_t orc_filter_topk(const float* logits, const float* ctr, int64_t n_anchors, int64_t num_classes,
                        float thresh, int64_t k, float* out_scores, int64_t* out_idx) {
    /* the k best candidates in a binary heap whose root is the worst of them (O(n log k), memory O(k)) */
    int64_t n = n_anchors * num_classes, c = 0;
    if (k <= 0) return 0;
    sidx_t* heap = (sidx_t*)malloc(sizeof(sidx_t) * (size_t)k);
    for (int64_t i = 0; i < n; ++i) {
        float s = orc_sigmoidf(logits[i]);
        if (ctr) { float t = s * orc_sigmoidf(ctr[i / num_classes]); s = sqrtf(t); }
        if (!(s > thresh)) continue;
        sidx_t e; e.s = s; e.i = i;
        if (c < k) {                                   /* sift up: a parent must not rank before its child */
            int64_t p = c++;
            while (p > 0) {
                int64_t q = (p - 1) / 2;
                if (!ranks_before(&heap[q], &e)) break;
                heap[p] = heap[q]; p = q;
            }
            heap[p] = e;
        } else if (ranks_before(&e, &heap[0])) {       /* better than the worst kept: replace the root, sift down */
            int64_t p = 0;
            for (;;) {
                int64_t l = 2 * p + 1, r = l + 1, w = l;
                if (l >= c) break;
                if (r < c && ranks_before(&heap[l], &heap[r])) w = r;    /* the worse of the two children */
                if (!ranks_before(&e, &heap[w])) break;
                heap[p] = heap[w]; p = w;
            }
            heap[p] = e;
        }
    }
    qsort(heap, (size_t)c, sizeof(sidx_t), cmp_desc_stable);
    for (int64_t j = 0; j < c; ++j) { out_scores[j] = heap[j].s; out_idx[j] = heap[j].i; }
    free(heap);
    return c;
}

/* AnchorGenerator.grid_anchors for one level: order (y, x, a); int32 shifts + fp32 base */
void orc_grid_anchors(const float* base, int64_t A, int64_t gh, int64_t gw, int64_t sh, int64_t sw,
                      float* out) {
    for (int64_t y = 0; y < gh; ++y)
        for (int64_t x = 0; x < gw; ++x)
            for (int64_t a = 0; a < A; ++a) {
                float fx = (float)(int32_t)(x * sw), fy = (float)(int32_t)(y * sh);
                float* o = out + 4 * ((y * gw + x) * A + a);
                o[0] = fx + base[4 * a + 0]; o[1] = fy + base[4 * a + 1];
                o[2] = fx + base[4 * a + 2]; o[3] = fy + base[4 * a + 3];
            }
}
