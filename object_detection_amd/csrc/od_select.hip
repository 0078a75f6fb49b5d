// Score filter + top-k + box decode + clip for every (image, level) of the pyramid.
// Replaces the inner loop of torchvision RetinaNet.postprocess_detections / FCOS.postprocess_detections:
//   scores = sigmoid(logits).flatten(); keep = scores > score_thresh; topk(min(#keep, K));
//   anchor = idx // C; label = idx % C; decode_single(reg[anchor], anchors[anchor]); clip_boxes_to_image
// (reference trigger object_detection/train.py:84, object_detection/predict.py:61; knobs predict.py:44-45).
//
// Kernel 1 (filter_kernel, HBM-bound stream): every workgroup owns one 8192-element chunk of one
// (image, level) logit slab, reads it once with 16-byte loads, and appends the survivors as unique
// 64-bit keys  (score_bits << 32) | (0xFFFFFFFF - flat_index)  to the chunk's own segment of the
// workspace (no global atomics; one LDS counter per workgroup).  Larger key = better candidate, and
// the low word makes the adopted tie rule (equal score -> lower flat index first) part of the order.
//
// Kernel 2 (select_kernel, one 1024-thread workgroup per (image, level)): radix-descends on the key
// with 4096-bin LDS histograms until the keys above the cut fit in LDS, gathers them, bitonic-sorts
// them in LDS, and decodes + clips the first K straight into the padded candidate arrays.
#include "od_common.h"
#include "../../include/od_boxes.h"

namespace {

using namespace od;

constexpr int ST = 256;            // filter workgroup
constexpr int CHUNK = 8192;        // logits per filter workgroup (32 KB)
constexpr int SEL_T = 1024;        // select workgroup
constexpr int CAP = 8192;          // keys held in LDS by the select kernel
constexpr int HBITS = 12;
constexpr int HBINS = 1 << HBITS;  // histogram bins per radix level
constexpr int DIRECT = 2048;       // survivors <= DIRECT: skip the histogram, sort everything

struct FilterParams {
    int num_levels;
    int num_classes;
    int has_ctr;
    int total_chunks;                       // per image
    int chunk_start[OD_MAX_LEVELS + 1];     // first chunk id of each level
    long long level_elems[OD_MAX_LEVELS];   // N_l * C
    const float* cls[OD_MAX_LEVELS];
    long long cls_stride[OD_MAX_LEVELS];
    const float* ctr[OD_MAX_LEVELS];
    long long ctr_stride[OD_MAX_LEVELS];
    float thresh;
    float pre_logit;                        // logits <= pre_logit can never pass the threshold
    int prefilter;                          // 0: threshold <= 0, every logit is scored
};

struct SelectParams {
    int num_classes;
    int topk;
    int coder;
    float clip;                             // bbox_xform_clip
    unsigned long long lo0;                 // initial radix interval [lo0, lo0 + 2^w0)
    int w0;
    const float* reg[OD_MAX_LEVELS];
    long long reg_stride[OD_MAX_LEVELS];
    const float* anchors[OD_MAX_LEVELS];
    long long anchor_stride[OD_MAX_LEVELS];
    int chunk_start[OD_MAX_LEVELS + 1];
};

__device__ __forceinline__ float score_of(const FilterParams& p, int l, int b, float x, long long flat) {
    float s = spec_sigmoidf(x);
    if (p.has_ctr) {
        long long a = flat / p.num_classes;
        float c = p.ctr[l][(long long)b * p.ctr_stride[l] + a];
        s = sqrtf(s * spec_sigmoidf(c));
    }
    return s;
}

__global__ __launch_bounds__(ST) void filter_kernel(FilterParams p, unsigned long long* __restrict__ seg_keys,
                                                    int* __restrict__ seg_count) {
    __shared__ int s_count;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int chunk = blockIdx.x;
    const int b = blockIdx.y;
    int l = 0;
#pragma unroll
    for (int i = 1; i < OD_MAX_LEVELS; ++i)
        if (i < p.num_levels && chunk >= p.chunk_start[i]) l = i;
    const long long E = p.level_elems[l];
    const long long e0 = (long long)(chunk - p.chunk_start[l]) * CHUNK;
    const float* src = p.cls[l] + (long long)b * p.cls_stride[l];
    unsigned long long* seg = seg_keys + ((long long)b * p.total_chunks + chunk) * CHUNK;
    if (tid == 0) s_count = 0;
    __syncthreads();

    const bool vec_ok = (((uintptr_t)(src + e0)) & 15) == 0;
    const long long rem = E - e0;
    const int len = rem < CHUNK ? (int)rem : CHUNK;

#pragma unroll 2
    for (int it = 0; it < CHUNK / (ST * 4); ++it) {
        float x[4];
        int off[4];
        const int blk = it * ST * 4;
        if (vec_ok && blk + ST * 4 <= len) {
            const float4 v = *reinterpret_cast<const float4*>(src + e0 + blk + tid * 4);
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) off[k] = blk + tid * 4 + k;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                off[k] = blk + k * ST + tid;
                x[k] = (off[k] < len) ? src[e0 + off[k]] : -__builtin_inff();
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool pass = false;
            float s = 0.0f;
            if (off[k] < len && (!p.prefilter || !(x[k] <= p.pre_logit))) {
                s = score_of(p, l, b, x[k], e0 + off[k]);
                pass = s > p.thresh;
            }
            const uint64_t bal = __ballot(pass);
            if (bal) {
                int base = 0;
                const int first = __ffsll((unsigned long long)bal) - 1;
                if (lane == first) base = atomicAdd(&s_count, __popcll(bal));
                base = __builtin_amdgcn_readlane(base, first);
                if (pass) {
                    const unsigned long long flat = (unsigned long long)(e0 + off[k]);
                    seg[base + mbcnt(bal)] = ((unsigned long long)f2u(s) << 32) | (0xFFFFFFFFull - flat);
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) seg_count[(long long)b * p.total_chunks + chunk] = s_count;
}

// ---- block-wide helpers for the select kernel ----------------------------------------------------
__device__ __forceinline__ int block_sum(int v, int* s_tmp) {   // s_tmp: 17 ints
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int w = (int)wave_sum_all((uint32_t)v);
    __syncthreads();
    if (lane == 0) s_tmp[wave] = w;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int i = 0; i < SEL_T / 64; ++i) t += s_tmp[i];
    return t;
}

template <typename F>
__device__ __forceinline__ void for_each_key(const unsigned long long* keys, const int* counts, int nseg, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < nseg; s += SEL_T / 64) {
        const int c = counts[s];
        const unsigned long long* k = keys + (long long)s * CHUNK;
        for (int e = lane; e < c; e += 64) f(k[e]);
    }
}

__device__ __forceinline__ void decode_store(const SelectParams& p, int l, int b, unsigned long long key, int rank,
                                             const int* __restrict__ image_hw, int num_levels,
                                             float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
                                             int* __restrict__ cand_labels, long long* __restrict__ cand_index) {
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)key;
    const float score = u2f((uint32_t)(key >> 32));
    const uint32_t a = flat / (uint32_t)p.num_classes;
    const int label = (int)(flat - a * (uint32_t)p.num_classes);
    const float4 r = *reinterpret_cast<const float4*>(p.reg[l] + ((long long)b * p.reg_stride[l] + a) * 4);
    const float4 an = *reinterpret_cast<const float4*>(p.anchors[l] + ((long long)b * p.anchor_stride[l] + a) * 4);
    float x1, y1, x2, y2;
    if (p.coder == OD_CODER_BOXCODER) {
        // BoxCoder.decode_single, weights (1,1,1,1): every product and sum rounded separately
        const float w = an.z - an.x, h = an.w - an.y;
        const float cx = an.x + 0.5f * w, cy = an.y + 0.5f * h;
        const float dw = (r.z > p.clip) ? p.clip : r.z;
        const float dh = (r.w > p.clip) ? p.clip : r.w;
        const float pcx = r.x * w + cx, pcy = r.y * h + cy;
        const float pw = spec_expf(dw) * w, ph = spec_expf(dh) * h;
        const float cw = 0.5f * pw, ch = 0.5f * ph;
        x1 = pcx - cw; y1 = pcy - ch; x2 = pcx + cw; y2 = pcy + ch;
    } else {
        // BoxLinearCoder(normalize_by_size=True).decode
        const float cx = 0.5f * (an.x + an.z), cy = 0.5f * (an.y + an.w);
        const float w = an.z - an.x, h = an.w - an.y;
        x1 = cx - r.x * w; y1 = cy - r.y * h; x2 = cx + r.z * w; y2 = cy + r.w * h;
    }
    // clip_boxes_to_image: clamp(min=0, max=size)
    const float H = (float)image_hw[2 * b], W = (float)image_hw[2 * b + 1];
    x1 = fminf(fmaxf(x1, 0.0f), W); x2 = fminf(fmaxf(x2, 0.0f), W);
    y1 = fminf(fmaxf(y1, 0.0f), H); y2 = fminf(fmaxf(y2, 0.0f), H);
    const long long slot = ((long long)b * num_levels + l) * p.topk + rank;
    *reinterpret_cast<float4*>(cand_boxes + slot * 4) = make_float4(x1, y1, x2, y2);
    cand_scores[slot] = score;
    cand_labels[slot] = label;
    if (cand_index) cand_index[slot] = (long long)flat;
}

__global__ __launch_bounds__(SEL_T) void select_kernel(SelectParams p, int num_levels, int total_chunks,
                                                       const unsigned long long* __restrict__ seg_keys,
                                                       const int* __restrict__ seg_count,
                                                       const int* __restrict__ image_hw,
                                                       float* __restrict__ cand_boxes,
                                                       float* __restrict__ cand_scores,
                                                       int* __restrict__ cand_labels,
                                                       long long* __restrict__ cand_index,
                                                       int* __restrict__ cand_count) {
    __shared__ unsigned long long s_keys[CAP];
    __shared__ unsigned int s_hist[HBINS];
    __shared__ int s_tmp[32];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l = blockIdx.x, b = blockIdx.y;
    const int c_first = p.chunk_start[l];
    const int nseg = p.chunk_start[l + 1] - c_first;
    const int* counts = seg_count + (long long)b * total_chunks + c_first;
    const unsigned long long* keys = seg_keys + ((long long)b * total_chunks + c_first) * CHUNK;
    const int K = p.topk;

    int mine = 0;
    for (int s = tid; s < nseg; s += SEL_T) mine += counts[s];
    const int total = block_sum(mine, s_tmp);
    if (total == 0) {
        if (tid == 0) cand_count[b * num_levels + l] = 0;
        return;
    }

    unsigned long long lo = 0;   // gather threshold: keys >= lo are taken into LDS
    int n_ge = total;
    if (total > DIRECT && total > K) {
        lo = p.lo0;
        int w = p.w0;
        int need = K;            // how many of the keys inside [lo, lo + 2^w) are still wanted
        while (true) {
            const int d = w < HBITS ? w : HBITS;
            const int shift = w - d;
            const int nb = 1 << d;
            for (int i = tid; i < nb; i += SEL_T) s_hist[i] = 0;
            __syncthreads();
            const unsigned long long lo_c = lo;
            const unsigned long long span = (w >= 64) ? ~0ull : ((1ull << w) - 1);
            for_each_key(keys, counts, nseg, [&](unsigned long long k) {
                if (k >= lo_c && (k - lo_c) <= span) atomicAdd(&s_hist[(unsigned int)((k - lo_c) >> shift)], 1u);
            });
            __syncthreads();
            // suffix scan: thread t owns bins [t*per, t*per+per); find the bin where the count from the top reaches `need`
            const int per = (nb + SEL_T - 1) / SEL_T;     // 4 when nb = 4096, else 1 (nb <= 1024)
            int local = 0;
            for (int i = 0; i < per; ++i) { int bin = tid * per + i; if (bin < nb) local += (int)s_hist[bin]; }
            // inclusive suffix sum over threads: wave-level then across waves
            int suf = local;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { int t = __shfl_down(suf, o); if (lane + o < 64) suf += t; }
            if (lane == 0) s_tmp[wave] = suf;             // wave total
            __syncthreads();
            int higher = 0;                               // keys in waves above this one
            for (int i = wave + 1; i < SEL_T / 64; ++i) higher += s_tmp[i];
            const int above_me = higher + suf - local;    // keys in bins strictly above my bins
            __syncthreads();
            if (above_me < need && need <= above_me + local) {
                int acc = above_me;
                for (int i = per - 1; i >= 0; --i) {
                    const int bin = tid * per + i;
                    if (bin >= nb) continue;
                    const int h = (int)s_hist[bin];
                    if (acc < need && need <= acc + h) { s_tmp[16] = bin; s_tmp[17] = acc; s_tmp[18] = h; }
                    acc += h;
                }
            }
            __syncthreads();
            const int bstar = s_tmp[16], above = s_tmp[17], h = s_tmp[18];
            __syncthreads();
            lo = lo + ((unsigned long long)bstar << shift);
            w = shift;
            n_ge = (K - need) + above + h;
            need -= above;
            if (n_ge <= CAP || w == 0) break;
        }
    }

    // ---- gather keys >= lo into LDS ------------------------------------------------------------------
    if (tid == 0) s_tmp[20] = 0;
    __syncthreads();
    {
        const unsigned long long lo_c = lo;
        const int lane_ = lane;
        for (int s = wave; s < nseg; s += SEL_T / 64) {
            const int c = counts[s];
            const unsigned long long* kp = keys + (long long)s * CHUNK;
            for (int e0 = 0; e0 < c; e0 += 64) {
                const int e = e0 + lane_;
                unsigned long long k = 0;
                bool take = false;
                if (e < c) { k = kp[e]; take = k >= lo_c; }
                const uint64_t bal = __ballot(take);
                if (bal) {
                    int base = 0;
                    const int first = __ffsll((unsigned long long)bal) - 1;
                    if (lane_ == first) base = atomicAdd(&s_tmp[20], __popcll(bal));
                    base = __builtin_amdgcn_readlane(base, first);
                    const int pos = base + mbcnt(bal);
                    if (take && pos < CAP) s_keys[pos] = k;
                }
            }
        }
    }
    __syncthreads();
    int cnt = s_tmp[20];
    if (cnt > CAP) cnt = CAP;   // unreachable for unique keys; keeps LDS accesses in bounds
    int n_pad = 64;
    while (n_pad < cnt) n_pad <<= 1;
    for (int i = cnt + tid; i < n_pad; i += SEL_T) s_keys[i] = 0ull;
    __syncthreads();

    // ---- bitonic sort, descending -------------------------------------------------------------------
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pad; i += SEL_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_keys[i], c2 = s_keys[ixj];
                    const bool desc = (i & k) == 0;
                    if ((a < c2) == desc) { s_keys[i] = c2; s_keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }

    const int n_out = cnt < K ? cnt : K;
    for (int r = tid; r < n_out; r += SEL_T)
        decode_store(p, l, b, s_keys[r], r, image_hw, num_levels, cand_boxes, cand_scores, cand_labels, cand_index);
    if (tid == 0) cand_count[b * num_levels + l] = n_out;
}

__global__ __launch_bounds__(256) void decode_kernel(const float4* __restrict__ rel, const float4* __restrict__ boxes,
                                                     long long n, int coder, float clip, float clip_h, float clip_w,
                                                     float4* __restrict__ out) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 r = rel[i], an = boxes[i];
    float x1, y1, x2, y2;
    if (coder == OD_CODER_BOXCODER) {
        const float w = an.z - an.x, h = an.w - an.y;
        const float cx = an.x + 0.5f * w, cy = an.y + 0.5f * h;
        const float dw = (r.z > clip) ? clip : r.z;
        const float dh = (r.w > clip) ? clip : r.w;
        const float pcx = r.x * w + cx, pcy = r.y * h + cy;
        const float pw = spec_expf(dw) * w, ph = spec_expf(dh) * h;
        const float cw = 0.5f * pw, ch = 0.5f * ph;
        x1 = pcx - cw; y1 = pcy - ch; x2 = pcx + cw; y2 = pcy + ch;
    } else {
        const float cx = 0.5f * (an.x + an.z), cy = 0.5f * (an.y + an.w);
        const float w = an.z - an.x, h = an.w - an.y;
        x1 = cx - r.x * w; y1 = cy - r.y * h; x2 = cx + r.z * w; y2 = cy + r.w * h;
    }
    if (clip_h > 0.0f) {
        x1 = fminf(fmaxf(x1, 0.0f), clip_w); x2 = fminf(fmaxf(x2, 0.0f), clip_w);
        y1 = fminf(fmaxf(y1, 0.0f), clip_h); y2 = fminf(fmaxf(y2, 0.0f), clip_h);
    }
    out[i] = make_float4(x1, y1, x2, y2);
}

struct Layout {
    int total_chunks;
    int chunk_start[OD_MAX_LEVELS + 1];
    size_t keys_bytes, counts_bytes;
};

bool make_layout(const int64_t* level_size, int num_levels, int batch, int num_classes, Layout* lay) {
    long long chunks = 0;
    for (int l = 0; l < num_levels; ++l) {
        if (level_size[l] < 0) return false;
        long long e = level_size[l] * (long long)num_classes;
        if (e > 0xFFFFFFFFll) return false;
        lay->chunk_start[l] = (int)chunks;
        chunks += (e + CHUNK - 1) / CHUNK;
        if (chunks > 0x3fffffff) return false;
    }
    for (int l = num_levels; l <= OD_MAX_LEVELS; ++l) lay->chunk_start[l] = (int)chunks;
    lay->total_chunks = (int)chunks;
    lay->keys_bytes = ((size_t)batch * (size_t)chunks * CHUNK * 8 + 255) & ~(size_t)255;
    lay->counts_bytes = ((size_t)batch * (size_t)chunks * 4 + 255) & ~(size_t)255;
    return true;
}

}  // namespace

extern "C" size_t od_select_candidates_workspace_bytes(const int64_t* level_size, int32_t num_levels,
                                                       int32_t batch, int32_t num_classes) {
    Layout lay;
    if (!level_size || num_levels < 1 || num_levels > OD_MAX_LEVELS || batch < 0 || num_classes < 1) return 0;
    if (!make_layout(level_size, num_levels, batch, num_classes, &lay)) return 0;
    return lay.keys_bytes + lay.counts_bytes;
}

extern "C" int od_select_candidates(const float* const* cls_logits, const int64_t* cls_batch_stride,
                                    const float* const* bbox_ctrness, const int64_t* ctr_batch_stride,
                                    const float* const* bbox_regression, const int64_t* reg_batch_stride,
                                    const float* const* anchors, const int64_t* anchor_batch_stride,
                                    const int64_t* level_size, int32_t num_levels, int32_t batch,
                                    int32_t num_classes, const int32_t* image_hw, float score_thresh,
                                    int32_t topk, int32_t coder, float bbox_xform_clip, float* cand_boxes,
                                    float* cand_scores, int32_t* cand_labels, int64_t* cand_index,
                                    int32_t* cand_count, void* workspace, size_t workspace_bytes, void* stream) {
    if (!cls_logits || !cls_batch_stride || !bbox_regression || !reg_batch_stride || !anchors ||
        !anchor_batch_stride || !level_size || !image_hw || !cand_boxes || !cand_scores || !cand_labels ||
        !cand_count || !workspace)
        return OD_ERR_ARG;
    if (bbox_ctrness && !ctr_batch_stride) return OD_ERR_ARG;
    if (num_levels < 1 || num_levels > OD_MAX_LEVELS || num_classes < 1 || batch < 0) return OD_ERR_ARG;
    if (topk < 1 || topk > OD_MAX_TOPK) return OD_ERR_LIMIT;
    if (coder != OD_CODER_BOXCODER && coder != OD_CODER_LINEAR) return OD_ERR_ARG;
    if (batch == 0) return OD_OK;
    if (batch > 65535) return OD_ERR_LIMIT;
    Layout lay;
    if (!make_layout(level_size, num_levels, batch, num_classes, &lay)) return OD_ERR_LIMIT;
    if (workspace_bytes < lay.keys_bytes + lay.counts_bytes) return OD_ERR_WORKSPACE;
    if (((uintptr_t)workspace & 255) || ((uintptr_t)cand_boxes & 15)) return OD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* seg_keys = (unsigned long long*)workspace;
    int* seg_count = (int*)((char*)workspace + lay.keys_bytes);

    FilterParams fp;
    SelectParams sp;
    fp.num_levels = num_levels;
    fp.num_classes = num_classes;
    fp.has_ctr = bbox_ctrness ? 1 : 0;
    fp.total_chunks = lay.total_chunks;
    for (int l = 0; l <= OD_MAX_LEVELS; ++l) { fp.chunk_start[l] = lay.chunk_start[l]; sp.chunk_start[l] = lay.chunk_start[l]; }
    for (int l = 0; l < OD_MAX_LEVELS; ++l) {
        const bool on = l < num_levels;
        fp.level_elems[l] = on ? level_size[l] * (long long)num_classes : 0;
        fp.cls[l] = on ? cls_logits[l] : nullptr;
        fp.cls_stride[l] = on ? cls_batch_stride[l] : 0;
        fp.ctr[l] = (on && bbox_ctrness) ? bbox_ctrness[l] : nullptr;
        fp.ctr_stride[l] = (on && bbox_ctrness) ? ctr_batch_stride[l] : 0;
        sp.reg[l] = on ? bbox_regression[l] : nullptr;
        sp.reg_stride[l] = on ? reg_batch_stride[l] / 4 : 0;     // in boxes
        sp.anchors[l] = on ? anchors[l] : nullptr;
        sp.anchor_stride[l] = on ? anchor_batch_stride[l] / 4 : 0;
        if (on && level_size[l] > 0) {
            if (!cls_logits[l] || !bbox_regression[l] || !anchors[l]) return OD_ERR_ARG;
            if (((uintptr_t)bbox_regression[l] & 15) || ((uintptr_t)anchors[l] & 15)) return OD_ERR_ARG;
            if ((reg_batch_stride[l] & 3) || (anchor_batch_stride[l] & 3)) return OD_ERR_ARG;
            if (((uintptr_t)cls_logits[l] & 3)) return OD_ERR_ARG;
        }
    }
    fp.thresh = score_thresh;
    // Prefilter bound on the logit (host double math, generous margin): a logit at or below it cannot
    // produce score > thresh.  FCOS: sqrt(sig(cls)*sig(ctr)) > t implies sig(cls) > t^2 (sig(ctr) <= 1).
    {
        double t = (double)score_thresh;
        if (bbox_ctrness && t > 0) t = t * t;
        fp.prefilter = 1;
        if (!(t > 0.0)) { fp.pre_logit = 0.0f; fp.prefilter = 0; }
        else if (t >= 1.0) fp.pre_logit = __builtin_inff();
        else {
            double lg = __builtin_log(t / (1.0 - t));
            fp.pre_logit = (float)(lg - 1e-3 * (1.0 + __builtin_fabs(lg)));
        }
    }
    // initial radix interval: all keys of scores in (thresh, 1.0] lie in [lo0, lo0 + 2^w0)
    {
        float tpos = score_thresh > 0.0f ? score_thresh : 0.0f;
        uint32_t tb;
        __builtin_memcpy(&tb, &tpos, 4);
        if (tb > 0x3F800000u) tb = 0x3F800000u;
        const unsigned long long hi_excl = (0x3F800000ull + 1ull) << 32;
        const unsigned long long span = hi_excl - ((unsigned long long)tb << 32);
        int w0 = 0;
        while (w0 < 63 && (1ull << w0) < span) ++w0;
        sp.w0 = w0;
        sp.lo0 = ((1ull << w0) > hi_excl) ? 0ull : hi_excl - (1ull << w0);
    }
    sp.num_classes = num_classes;
    sp.topk = topk;
    sp.coder = coder;
    sp.clip = bbox_xform_clip;

    if (lay.total_chunks > 0) {
        OD_LAUNCH(OD_KERNEL_FILTER, s,
                  hipLaunchKernelGGL(filter_kernel, dim3((unsigned)lay.total_chunks, (unsigned)batch), dim3(ST), 0, s,
                                     fp, seg_keys, seg_count));
    }
    OD_LAUNCH(OD_KERNEL_SELECT, s,
              hipLaunchKernelGGL(select_kernel, dim3((unsigned)num_levels, (unsigned)batch), dim3(SEL_T), 0, s, sp,
                                 (int)num_levels, lay.total_chunks, (const unsigned long long*)seg_keys,
                                 (const int*)seg_count, image_hw, cand_boxes, cand_scores, cand_labels,
                                 (long long*)cand_index, cand_count));
    return OD_OK;
}

extern "C" int od_decode_boxes(const float* rel_codes, const float* boxes, int64_t n, int32_t coder,
                               float bbox_xform_clip, float clip_h, float clip_w, float* out, void* stream) {
    if (n < 0) return OD_ERR_ARG;
    if (n == 0) return OD_OK;
    if (!rel_codes || !boxes || !out) return OD_ERR_ARG;
    if (((uintptr_t)rel_codes & 15) || ((uintptr_t)boxes & 15) || ((uintptr_t)out & 15)) return OD_ERR_ARG;
    if (coder != OD_CODER_BOXCODER && coder != OD_CODER_LINEAR) return OD_ERR_ARG;
    OD_LAUNCH(OD_KERNEL_DECODE, (hipStream_t)stream,
              hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                 (const float4*)rel_codes, (const float4*)boxes, (long long)n, (int)coder,
                                 bbox_xform_clip, clip_h, clip_w, (float4*)out));
    return OD_OK;
}
