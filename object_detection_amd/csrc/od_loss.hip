// Classification loss of the RetinaNet head, forward and backward, fused with the construction of its targets.
// Replaces torchvision RetinaNetClassificationHead.compute_loss (head built at reference
// object_detection/retinanet/retinanet_model.py:47-54, loss taken at object_detection/train.py:62-63):
//   per image:  fg = matched >= 0;  target = zeros_like(logits);  target[fg, labels[matched[fg]]] = 1
//               valid = matched != -2;  loss_b = sigmoid_focal_loss(logits[valid], target[valid], "sum") / max(1, #fg)
//   result = sum_b loss_b / B
// with sigmoid_focal_loss(x, t; alpha, gamma) = alpha_t * (1 - p_t)^gamma * BCEWithLogits(x, t).
//
// The reference path materialises target [N,C], two boolean masks, the gathered logits[valid] / target[valid]
// copies and about ten element-wise temporaries per image.  Here the logits are streamed once (forward) or
// read once and the gradient written once (backward); the target of an element is derived on the fly from
// the anchor's matched index.  Both kernels are HBM streams: 4*B*N*C + 8*B*N bytes forward, twice the logits
// plus the matches backward.
//
// Arithmetic: e = exp(-|x|) gives p, 1-p and log1p(e) without cancellation; the fast hardware exp/log are
// used where their absolute error is harmless (relative error of every element's loss stays below 3e-6), so
// the sum agrees with the fp32 torch evaluation to about 1e-6; partial sums are fp32 inside a workgroup
// and double across workgroups (deterministic: no atomics).
#include "od_common.h"
#include "../../include/od_boxes.h"

namespace {

using namespace od;

constexpr int LT = 256;                 // threads per workgroup
constexpr int LPT = 16;                 // elements per thread (four 16-byte groups)
constexpr int LCHUNK = LT * LPT;        // elements per workgroup

typedef float vec4f __attribute__((ext_vector_type(4)));

// Element types of the head outputs and of their gradients: fp32, or the 16-bit types a head produces under autocast
// (the reference trains with use_amp, reference object_detection/configs/config_all.py:16, train.py:61).  All
// arithmetic is fp32; a 16-bit gradient is rounded once, to nearest even.
struct bf16_t { unsigned short bits; };
template <typename T> struct Elem;
template <> struct Elem<float> {
    typedef vec4f V;
    static __device__ __forceinline__ float to_float(float v) { return v; }
    static __device__ __forceinline__ float from_float(float v) { return v; }
    static __device__ __forceinline__ float4 unpack(V v) { return make_float4(v.x, v.y, v.z, v.w); }
    static __device__ __forceinline__ V pack(float4 f) { V v; v.x = f.x; v.y = f.y; v.z = f.z; v.w = f.w; return v; }
};
template <> struct Elem<_Float16> {
    typedef _Float16 V __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float to_float(_Float16 v) { return (float)v; }
    static __device__ __forceinline__ _Float16 from_float(float v) { return (_Float16)v; }
    static __device__ __forceinline__ float4 unpack(V v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
    static __device__ __forceinline__ V pack(float4 f) {
        V v; v.x = (_Float16)f.x; v.y = (_Float16)f.y; v.z = (_Float16)f.z; v.w = (_Float16)f.w; return v;
    }
};
template <> struct Elem<bf16_t> {
    typedef unsigned short V __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float up(unsigned short h) { return u2f((uint32_t)h << 16); }
    static __device__ __forceinline__ unsigned short down(float f) {
        const uint32_t u = f2u(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);      // NaN stays NaN
        return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    static __device__ __forceinline__ float to_float(bf16_t v) { return up(v.bits); }
    static __device__ __forceinline__ bf16_t from_float(float v) { bf16_t r; r.bits = down(v); return r; }
    static __device__ __forceinline__ float4 unpack(V v) { return make_float4(up(v.x), up(v.y), up(v.z), up(v.w)); }
    static __device__ __forceinline__ V pack(float4 f) { V v; v.x = down(f.x); v.y = down(f.y); v.z = down(f.z); v.w = down(f.w); return v; }
};
template <typename T> constexpr uintptr_t vec_mask() { return 4 * sizeof(T) - 1; }   // alignment of a group of four

struct FocalTerms {
    float loss;                         // alpha_t * (1 - p_t)^gamma * ce
    float dldx;                         // its derivative with respect to the logit
};

// t: target of the element (true = 1.0).  G2: gamma == 2 (the reference's value) is a product; any other gamma
// goes through pow, which costs more than the rest of the element together and is therefore compiled apart.
template <bool G2>
__device__ __forceinline__ FocalTerms focal_terms(float x, bool t, float alpha, float gamma, bool want_grad) {
    const float ax = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(ax * -1.44269504f);      // exp(-|x|) in [0, 1]; 0 once |x| > 87
    const float inv = __builtin_amdgcn_rcpf(1.0f + e);              // 1 ulp; 1 + e is in [1, 2]
    const float ei = e * inv;
    const float p = (x >= 0.0f) ? inv : ei;                         // sigmoid(x)
    const float q = (x >= 0.0f) ? ei : inv;                         // 1 - sigmoid(x), without cancellation
    // log1p(e): both forms are evaluated and one selected, which is cheaper than a divergent branch
    const float series = e * fmaf(e, fmaf(e, fmaf(e, fmaf(e, 0.2f, -0.25f), 0.33333334f), -0.5f), 1.0f);
    const float l1p = (e < 0.1f) ? series : __builtin_amdgcn_logf(1.0f + e) * 0.69314718f;
    const float sp = fmaxf(x, 0.0f) + l1p;                          // softplus(x) = -log(1 - p)
    const float ce = t ? sp - x : sp;                               // BCEWithLogits
    const float w = t ? q : p;                                      // 1 - p_t
    float wg;                                                       // (1 - p_t)^gamma
    if (G2) wg = w * w;
    else wg = (w > 0.0f) ? __powf(w, gamma) : (gamma == 0.0f ? 1.0f : 0.0f);
    const float at = (alpha >= 0.0f) ? (t ? alpha : 1.0f - alpha) : 1.0f;
    FocalTerms r;
    r.loss = at * wg * ce;
    r.dldx = 0.0f;
    if (want_grad) {
        // t = 1: alpha (1-p)^g [ -g p ce - (1-p) ];   t = 0: (1-alpha) p^g [ g (1-p) ce + p ]
        r.dldx = t ? at * wg * (-gamma * p * ce - q) : at * wg * (gamma * q * ce + p);
    }
    return r;
}

struct LossArgs {
    const void* logits;                 // [batch, n, C] of the element type, image stride batch_stride (elements)
    long long batch_stride;
    long long n;
    int num_classes;
    const long long* matched;           // [batch, n]
    const long long* gt_labels;         // [sum M]
    const int* gt_offsets;              // [batch + 1]
    float alpha, gamma;
    int blocks_per_image;
    int norm;                           // OD_LOSS_NORM_*
};

// anchor and class of flat element e (one division per group of four elements; 32-bit when E < 2^32)
__device__ __forceinline__ void split_index(long long e, int num_classes, bool small, long long& a, int& c) {
    if (small) {
        const uint32_t q = (uint32_t)e / (uint32_t)num_classes;
        a = (long long)q;
        c = (int)((uint32_t)e - q * (uint32_t)num_classes);
    } else {
        a = e / num_classes;
        c = (int)(e - a * num_classes);
    }
}

// class of an anchor from its matched index: -1 = background (all targets 0), -2 = ignored anchor
__device__ __forceinline__ int class_of(const LossArgs& p, int b, long long m) {
    if (m == -2) return -2;
    const int m0 = p.gt_offsets[b];
    if (m < 0 || m >= (long long)(p.gt_offsets[b + 1] - m0)) return -1;   // (an index past the image's GT list is not followed)
    return (int)p.gt_labels[m0 + m];
}

// The elements a thread owns: LPT/4 groups of four consecutive logits, group `it` at e0 + it*1024 + tid*4.
// All logits and the matched index of every group's first anchor are requested before anything is used.
struct ThreadGroups {
    float x[LPT / 4][4];
    long long a[LPT / 4];               // anchor of the group's first element
    int c[LPT / 4];                     // its class position
    long long m[LPT / 4];               // matched index of that anchor
};

template <typename T>
__device__ __forceinline__ void load_groups(const LossArgs& p, int b, const T* src, long long e0, long long E,
                                            bool vec_ok, bool small, ThreadGroups& g) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < LPT / 4; ++it) {
        const long long eb = e0 + it * (LT * 4) + tid * 4;
        if (vec_ok) {
            const float4 v = Elem<T>::unpack(*reinterpret_cast<const typename Elem<T>::V*>(src + eb));
            g.x[it][0] = v.x; g.x[it][1] = v.y; g.x[it][2] = v.z; g.x[it][3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) g.x[it][k] = (eb + k < E) ? Elem<T>::to_float(src[eb + k]) : 0.0f;
        }
        split_index(eb < E ? eb : 0, p.num_classes, small, g.a[it], g.c[it]);
        g.m[it] = (eb < E) ? p.matched[(long long)b * p.n + g.a[it]] : -2;
    }
}

// Fast form of the two functions below for a workgroup whose chunk is complete and 16-byte aligned, with at least
// four classes: a group of four elements then spans at most two anchors, so two matched indices cover it, nothing
// needs a bounds test and the element loop is free of branches.  f(it, k, x, is_target, valid) is called for every
// element (valid = its anchor is not ignored); returns the thread's number of foreground anchors.
template <typename T, typename F>
__device__ __forceinline__ int for_each_element_fast(const LossArgs& p, int b, const T* src, long long e0, bool small,
                                                     F f) {
    const int tid = threadIdx.x;
    const int C = p.num_classes;
    const bool straddle = (C & 3) != 0;                             // otherwise a group never crosses an anchor
    const long long* mrow = p.matched + (long long)b * p.n;
    float4 x[LPT / 4];
    long long m0[LPT / 4], m1[LPT / 4];
    int c0[LPT / 4];
#pragma unroll
    for (int it = 0; it < LPT / 4; ++it) {
        const long long eb = e0 + it * (LT * 4) + tid * 4;
        // streamed once: non-temporal loads keep these lines from displacing others in the caches
        x[it] = Elem<T>::unpack(__builtin_nontemporal_load(reinterpret_cast<const typename Elem<T>::V*>(src + eb)));
        long long a;
        split_index(eb, C, small, a, c0[it]);
        m0[it] = __builtin_nontemporal_load(mrow + a);
        m1[it] = straddle ? mrow[a + 1 < p.n ? a + 1 : a] : 0;
    }
    int cls0[LPT / 4], cls1[LPT / 4];
#pragma unroll
    for (int it = 0; it < LPT / 4; ++it) {
        cls0[it] = class_of(p, b, m0[it]);
        cls1[it] = straddle ? class_of(p, b, m1[it]) : cls0[it];
    }
    int fg = 0;
#pragma unroll
    for (int it = 0; it < LPT / 4; ++it) {
        const float xs[4] = {x[it].x, x[it].y, x[it].z, x[it].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool next = c0[it] + k >= C;
            const int cls = next ? cls1[it] : cls0[it];
            const int c = c0[it] + k - (next ? C : 0);
            f(it, k, xs[k], cls == c, cls != -2);
            fg += (c == 0 && cls >= 0) ? 1 : 0;
        }
    }
    return fg;
}

// f(it, k, x, is_target) for every valid (not ignored, in range) element of the thread, in element order;
// also counts the thread's foreground anchors (at their class-0 element).
template <typename F>
__device__ __forceinline__ int for_each_element(const LossArgs& p, int b, long long e0, long long E, const ThreadGroups& g,
                                                F f) {
    const int tid = threadIdx.x;
    int fg = 0;
#pragma unroll
    for (int it = 0; it < LPT / 4; ++it) {
        const long long eb = e0 + it * (LT * 4) + tid * 4;
        long long a = g.a[it];
        int c = g.c[it];
        int cls = class_of(p, b, g.m[it]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (eb + k < E) {
                if (cls != -2) {
                    f(it, k, g.x[it][k], cls == c);
                    fg += (c == 0 && cls >= 0) ? 1 : 0;
                }
                if (++c == p.num_classes && k < 3) {          // the next element belongs to the next anchor
                    c = 0;
                    ++a;
                    cls = (eb + k + 1 < E) ? class_of(p, b, p.matched[(long long)b * p.n + a]) : -2;
                }
            }
        }
    }
    return fg;
}

template <bool G2, typename T>
__global__ __launch_bounds__(LT) void focal_forward_kernel(LossArgs p, double* __restrict__ part_loss,
                                                           int* __restrict__ part_fg) {
    __shared__ float s_sum[LT / 64];
    __shared__ int s_fg[LT / 64];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const long long E = p.n * p.num_classes;
    const bool small = E <= 0xffffffffll;
    const long long e0 = (long long)blockIdx.x * LCHUNK;
    const T* src = (const T*)p.logits + (long long)b * p.batch_stride;
    const bool vec_ok = ((((uintptr_t)(src + e0)) & vec_mask<T>()) == 0) && (e0 + LCHUNK <= E);
    float acc = 0.0f;
    int fg;
    if (vec_ok && p.num_classes >= 4) {
        fg = for_each_element_fast(p, b, src, e0, small, [&](int, int, float x, bool t, bool valid) {
            const float l = focal_terms<G2>(x, t, p.alpha, p.gamma, false).loss;
            acc += valid ? l : 0.0f;
        });
    } else {
        ThreadGroups g;
        load_groups(p, b, src, e0, E, vec_ok, small, g);
        fg = for_each_element(p, b, e0, E, g, [&](int, int, float x, bool t) {
            acc += focal_terms<G2>(x, t, p.alpha, p.gamma, false).loss;
        });
    }
    // workgroup sums: shuffles inside the wave, LDS across the four waves
    float wsum = acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
    const int wfg = (int)wave_sum_all((uint32_t)fg);
    if ((tid & 63) == 0) { s_sum[tid >> 6] = wsum; s_fg[tid >> 6] = wfg; }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        int f = 0;
#pragma unroll
        for (int w = 0; w < LT / 64; ++w) { t += (double)s_sum[w]; f += s_fg[w]; }
        part_loss[(long long)b * p.blocks_per_image + blockIdx.x] = t;
        part_fg[(long long)b * p.blocks_per_image + blockIdx.x] = f;
    }
}

// one workgroup: per-image totals and the loss.
//   norm = OD_LOSS_NORM_PER_IMAGE: loss = mean_b( sum_b / max(1, #fg_b) )            (RetinaNet heads)
//   norm = OD_LOSS_NORM_BATCH    : loss = sum_b sum_b / max(1, sum_b #fg_b)          (FCOS head); num_fg[b] then holds
//                                  the batch total for every b, which is what the backward kernels divide by
// per_image may be null.
__global__ __launch_bounds__(1024) void focal_finalize_kernel(const double* __restrict__ part_loss,
                                                              const int* __restrict__ part_fg, int blocks_per_image,
                                                              int batch, int norm, float* __restrict__ loss_out,
                                                              float* __restrict__ per_image, int* __restrict__ num_fg) {
    __shared__ double s_d[1024 / 64];
    __shared__ int s_i[1024 / 64];
    __shared__ double s_total;
    __shared__ long long s_fg_total;
    const int tid = threadIdx.x;
    if (tid == 0) { s_total = 0.0; s_fg_total = 0; }
    for (int b = 0; b < batch; ++b) {
        double t = 0.0;
        int f = 0;
        for (int i = tid; i < blocks_per_image; i += 1024) {
            t += part_loss[(long long)b * blocks_per_image + i];
            f += part_fg[(long long)b * blocks_per_image + i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o); f += __shfl_xor(f, o); }
        __syncthreads();
        if ((tid & 63) == 0) { s_d[tid >> 6] = t; s_i[tid >> 6] = f; }
        __syncthreads();
        if (tid == 0) {
            double tt = 0.0;
            int ff = 0;
            for (int w = 0; w < 1024 / 64; ++w) { tt += s_d[w]; ff += s_i[w]; }
            if (norm == OD_LOSS_NORM_PER_IMAGE) {
                const double lb = tt / (double)(ff > 1 ? ff : 1);
                if (per_image) per_image[b] = (float)lb;
                num_fg[b] = ff;
                s_total += lb;
            } else {
                if (per_image) per_image[b] = (float)tt;         // divided by the batch's count below
                s_total += tt;
                s_fg_total += ff;
            }
        }
    }
    __syncthreads();
    if (norm == OD_LOSS_NORM_PER_IMAGE) {
        if (tid == 0) *loss_out = (float)(s_total / (double)batch);
        return;
    }
    const long long ft = s_fg_total;
    const double denom = (double)(ft > 1 ? ft : 1);
    for (int b = tid; b < batch; b += 1024) {
        if (per_image) per_image[b] = (float)((double)per_image[b] / denom);
        num_fg[b] = (int)(ft < 0x7fffffff ? ft : 0x7fffffff);
    }
    if (tid == 0) *loss_out = (float)(s_total / denom);
}

template <bool G2, typename T>
__global__ __launch_bounds__(LT) void focal_backward_kernel(LossArgs p, const int* __restrict__ num_fg,
                                                            const float* __restrict__ grad_out, int batch,
                                                            T* __restrict__ grad_logits) {
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const long long E = p.n * p.num_classes;
    const bool small = E <= 0xffffffffll;
    const long long e0 = (long long)blockIdx.x * LCHUNK;
    const T* src = (const T*)p.logits + (long long)b * p.batch_stride;
    T* dst = grad_logits + (long long)b * E;                         // the gradient is written contiguously
    const bool vec_ok = ((((uintptr_t)(src + e0)) & vec_mask<T>()) == 0) &&
                        ((((uintptr_t)(dst + e0)) & vec_mask<T>()) == 0) && (e0 + LCHUNK <= E);
    const int nf = num_fg[b];
    const float scale = *grad_out / ((float)(nf > 1 ? nf : 1) * (p.norm == OD_LOSS_NORM_PER_IMAGE ? (float)batch : 1.0f));
    float gr[LPT / 4][4];
    if (vec_ok && p.num_classes >= 4) {
        for_each_element_fast(p, b, src, e0, small, [&](int it, int k, float x, bool t, bool valid) {
            const float d = scale * focal_terms<G2>(x, t, p.alpha, p.gamma, true).dldx;
            gr[it][k] = valid ? d : 0.0f;                             // ignored anchors get a zero gradient
        });
    } else {
        ThreadGroups g;
        load_groups(p, b, src, e0, E, vec_ok, small, g);
#pragma unroll
        for (int it = 0; it < LPT / 4; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) gr[it][k] = 0.0f;
        for_each_element(p, b, e0, E, g, [&](int it, int k, float x, bool t) {
            gr[it][k] = scale * focal_terms<G2>(x, t, p.alpha, p.gamma, true).dldx;
        });
    }
#pragma unroll
    for (int it = 0; it < LPT / 4; ++it) {
        const long long eb = e0 + it * (LT * 4) + tid * 4;
        if (vec_ok) {
            *reinterpret_cast<typename Elem<T>::V*>(dst + eb) =
                Elem<T>::pack(make_float4(gr[it][0], gr[it][1], gr[it][2], gr[it][3]));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (eb + k < E) dst[eb + k] = Elem<T>::from_float(gr[it][k]);
        }
    }
}

bool loss_layout(int64_t n, int32_t num_classes, int32_t batch, int* blocks, size_t* part_bytes, size_t* total) {
    if (n < 0 || num_classes < 1 || batch < 0) return false;
    const long long E = n * (long long)num_classes;
    const long long blk = (E + LCHUNK - 1) / LCHUNK;
    if (blk > 0x7fffffff) return false;
    *blocks = (int)blk;
    *part_bytes = (((size_t)batch * (size_t)blk * sizeof(double)) + 255) & ~(size_t)255;
    *total = *part_bytes + ((((size_t)batch * (size_t)blk * sizeof(int)) + 255) & ~(size_t)255);
    return true;
}

size_t elem_size(int32_t dtype) {        // 0 for an unknown element type
    return dtype == OD_DTYPE_F32 ? 4 : (dtype == OD_DTYPE_F16 || dtype == OD_DTYPE_BF16) ? 2 : 0;
}

int fill_args(LossArgs* a, const void* cls_logits, int32_t dtype, int64_t batch_stride, int64_t n, int32_t num_classes,
              const int64_t* matched, const int64_t* gt_labels, const int32_t* gt_offsets, float alpha, float gamma,
              int32_t normalization, int blocks) {
    if (!cls_logits || !matched || !gt_offsets || elem_size(dtype) == 0) return OD_ERR_ARG;
    if (normalization != OD_LOSS_NORM_PER_IMAGE && normalization != OD_LOSS_NORM_BATCH) return OD_ERR_ARG;
    if (batch_stride < n * (long long)num_classes) return OD_ERR_ARG;
    if (((uintptr_t)cls_logits & (elem_size(dtype) - 1)) || ((uintptr_t)matched & 7)) return OD_ERR_ARG;
    a->logits = cls_logits;
    a->batch_stride = batch_stride;
    a->n = n;
    a->num_classes = num_classes;
    a->matched = (const long long*)matched;
    a->gt_labels = (const long long*)gt_labels;
    a->gt_offsets = gt_offsets;
    a->alpha = alpha;
    a->gamma = gamma;
    a->blocks_per_image = blocks;
    a->norm = normalization;
    return OD_OK;
}

// ------------------------------------------------------------------------------------------------
// Regression loss of the RetinaNet head (torchvision RetinaNetRegressionHead.compute_loss -> _utils._box_loss):
// per image, over the foreground anchors (matched >= 0):
//   "giou"      : sum generalized_box_iou_loss(BoxCoder.decode_single(regression, anchor), gt, eps)
//   "l1"        : sum |regression - BoxCoder.encode_single(gt, anchor)|
//   "smooth_l1" : the same difference through smooth_l1(beta)
// divided by max(1, #foreground), then averaged over the batch.  The reference selects "giou"
// (reference object_detection/retinanet/retinanet_model.py:53).  The kernels stream the matched indices and touch
// regression / anchor / target box only where an anchor is foreground; the arithmetic there uses the full-precision
// exp / log / division, since it is a tiny share of the work.
constexpr int RT = 256;                 // threads per workgroup
constexpr int RPT = 8;                  // anchors per thread
constexpr int RCHUNK = RT * RPT;

struct BoxLossArgs {
    const void* reg;                    // [batch, n, 4] of the element type
    long long reg_stride;               // elements between images
    const float* anchors;               // [n, 4]; images anchor_stride elements apart (0: shared by all images)
    long long anchor_stride;
    const long long* matched;           // [batch, n]
    const float* gt_boxes;              // [sum M, 4]
    const int* gt_offsets;              // [batch + 1]
    long long n;
    int loss_type;
    float beta, wx, wy, ww, wh, clip, eps;
    int blocks_per_image;
    int coder;                          // OD_CODER_BOXCODER, or OD_CODER_LINEAR (FCOS; "giou" only)
    int norm;                           // OD_LOSS_NORM_*
    const void* ctr;                    // FCOS centre-ness logits [batch, n] of the element type, or null
    long long ctr_stride;
};

struct BoxTerms {
    float loss;
    float4 grad;                        // d loss / d regression
    float ctr_loss, ctr_grad;           // centre-ness: BCE-with-logits against the encoded target, and d / d logit
};

// derivative of max(a, b) (sel = a > b) or min(a, b) (sel = a < b) with respect to a, torch's tie rule (one half)
__device__ __forceinline__ float pick(bool sel, float a, float b) { return sel ? 1.0f : (a == b ? 0.5f : 0.0f); }

__device__ __forceinline__ BoxTerms giou_terms(const BoxLossArgs& p, float4 r, float4 a, float4 g) {
    const float w = a.z - a.x, h = a.w - a.y;
    float x1, y1, x2, y2;
    float dw0 = 0.0f, dh0 = 0.0f, pw = 0.0f, ph = 0.0f;
    if (p.coder == OD_CODER_BOXCODER) {                 // BoxCoder.decode_single
        const float cx = a.x + 0.5f * w, cy = a.y + 0.5f * h;
        const float dx = r.x / p.wx, dy = r.y / p.wy;
        dw0 = r.z / p.ww; dh0 = r.w / p.wh;
        const float dw = fminf(dw0, p.clip), dh = fminf(dh0, p.clip);
        const float pcx = dx * w + cx, pcy = dy * h + cy;
        pw = expf(dw) * w; ph = expf(dh) * h;
        const float hw = 0.5f * pw, hh = 0.5f * ph;
        x1 = pcx - hw; y1 = pcy - hh; x2 = pcx + hw; y2 = pcy + hh;
    } else {                                            // BoxLinearCoder(normalize_by_size=True).decode
        const float cx = 0.5f * (a.x + a.z), cy = 0.5f * (a.y + a.w);
        x1 = cx - r.x * w; y1 = cy - r.y * h; x2 = cx + r.z * w; y2 = cy + r.w * h;
    }
    // generalized_box_iou_loss
    const float ix1 = fmaxf(x1, g.x), iy1 = fmaxf(y1, g.y), ix2 = fminf(x2, g.z), iy2 = fminf(y2, g.w);
    const bool hit = (iy2 > iy1) && (ix2 > ix1);
    const float iw = ix2 - ix1, ih = iy2 - iy1;
    const float inter = hit ? iw * ih : 0.0f;
    const float bw = x2 - x1, bh = y2 - y1;
    const float uni = bw * bh + (g.z - g.x) * (g.w - g.y) - inter;
    const float ue = uni + p.eps;
    const float iou = inter / ue;
    const float ex1 = fminf(x1, g.x), ey1 = fminf(y1, g.y), ex2 = fmaxf(x2, g.z), ey2 = fmaxf(y2, g.w);
    const float ew = ex2 - ex1, eh = ey2 - ey1;
    const float area_c = ew * eh;
    const float ce = area_c + p.eps;
    BoxTerms t;
    t.loss = 1.0f - (iou - (area_c - uni) / ce);
    // backward: loss = 1 - I/(U+eps) + (C-U)/(C+eps),  U = A_pred + A_gt - I
    const float dI = -(1.0f / ue + inter / (ue * ue)) + 1.0f / ce;
    const float dU = inter / (ue * ue) - 1.0f / ce;                 // reaches the box through A_pred
    const float dC = ue / (ce * ce);
    float gx1 = dU * -bh + dC * -eh * pick(x1 < g.x, x1, g.x);
    float gy1 = dU * -bw + dC * -ew * pick(y1 < g.y, y1, g.y);
    float gx2 = dU * bh + dC * eh * pick(x2 > g.z, x2, g.z);
    float gy2 = dU * bw + dC * ew * pick(y2 > g.w, y2, g.w);
    if (hit) {
        gx1 += dI * -ih * pick(x1 > g.x, x1, g.x);
        gy1 += dI * -iw * pick(y1 > g.y, y1, g.y);
        gx2 += dI * ih * pick(x2 < g.z, x2, g.z);
        gy2 += dI * iw * pick(y2 < g.w, y2, g.w);
    }
    t.ctr_loss = t.ctr_grad = 0.0f;
    if (p.coder == OD_CODER_BOXCODER) {
        // through decode_single; clamp passes the gradient where the value is not above the bound
        t.grad.x = (gx1 + gx2) * w / p.wx;
        t.grad.y = (gy1 + gy2) * h / p.wy;
        t.grad.z = (dw0 <= p.clip) ? 0.5f * (gx2 - gx1) * pw / p.ww : 0.0f;
        t.grad.w = (dh0 <= p.clip) ? 0.5f * (gy2 - gy1) * ph / p.wh : 0.0f;
    } else {
        t.grad.x = -gx1 * w; t.grad.y = -gy1 * h; t.grad.z = gx2 * w; t.grad.w = gy2 * h;
    }
    return t;
}

// FCOS centre-ness of an anchor matched to GT g: BoxLinearCoder.encode(anchor, gt) = (l, t, r, b) / (w, h, w, h),
// target = sqrt( min(l, r) / max(l, r) * min(t, b) / max(t, b) ); loss = BCE-with-logits(x, target).
__device__ __forceinline__ void ctrness_terms(float x, float4 a, float4 g, float& loss, float& grad) {
    const float cx = 0.5f * (a.x + a.z), cy = 0.5f * (a.y + a.w);
    const float w = a.z - a.x, h = a.w - a.y;
    const float l = (cx - g.x) / w, tp = (cy - g.y) / h, r = (g.z - cx) / w, bt = (g.w - cy) / h;
    const float z = sqrtf((fminf(l, r) / fmaxf(l, r)) * (fminf(tp, bt) / fmaxf(tp, bt)));
    const float e = expf(-fabsf(x));
    loss = (fmaxf(x, 0.0f) - x * z) + log1pf(e);
    const float sg = (x >= 0.0f) ? 1.0f / (1.0f + e) : e / (1.0f + e);
    grad = sg - z;
}

__device__ __forceinline__ void l1_term(float d, float beta, float& loss, float& grad) {
    const float ad = fabsf(d);
    if (ad < beta) {                                                // smooth_l1 only (beta == 0 for plain l1)
        loss = 0.5f * d * d / beta;
        grad = d / beta;
    } else {
        loss = ad - 0.5f * beta;
        grad = (d > 0.0f) ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    }
}

__device__ __forceinline__ BoxTerms l1_terms(const BoxLossArgs& p, float4 r, float4 a, float4 g) {
    // BoxCoder.encode_single(gt, anchor)
    const float aw = a.z - a.x, ah = a.w - a.y;
    const float acx = a.x + 0.5f * aw, acy = a.y + 0.5f * ah;
    const float gw = g.z - g.x, gh = g.w - g.y;
    const float gcx = g.x + 0.5f * gw, gcy = g.y + 0.5f * gh;
    const float tx = p.wx * (gcx - acx) / aw, ty = p.wy * (gcy - acy) / ah;
    const float tw = p.ww * logf(gw / aw), th = p.wh * logf(gh / ah);
    BoxTerms t;
    float l0, l1, l2, l3;
    l1_term(r.x - tx, p.beta, l0, t.grad.x);
    l1_term(r.y - ty, p.beta, l1, t.grad.y);
    l1_term(r.z - tw, p.beta, l2, t.grad.z);
    l1_term(r.w - th, p.beta, l3, t.grad.w);
    t.loss = ((l0 + l1) + l2) + l3;
    t.ctr_loss = t.ctr_grad = 0.0f;
    return t;
}

// Terms of anchor `a` of image b if it is foreground; returns false (and touches nothing) otherwise.
template <typename T>
__device__ __forceinline__ bool box_terms_of(const BoxLossArgs& p, int b, long long a, long long m, BoxTerms& t) {
    const int m0 = p.gt_offsets[b];
    if (m < 0 || m >= (long long)(p.gt_offsets[b + 1] - m0)) return false;
    const float4 r = Elem<T>::unpack(
        *reinterpret_cast<const typename Elem<T>::V*>((const T*)p.reg + (long long)b * p.reg_stride + a * 4));
    const float4 an = *reinterpret_cast<const float4*>(p.anchors + (long long)b * p.anchor_stride + a * 4);
    const float4 g = *reinterpret_cast<const float4*>(p.gt_boxes + (long long)(m0 + m) * 4);
    t = (p.loss_type == OD_BOX_LOSS_GIOU) ? giou_terms(p, r, an, g) : l1_terms(p, r, an, g);
    if (p.ctr) {
        const float x = Elem<T>::to_float(((const T*)p.ctr)[(long long)b * p.ctr_stride + a]);
        ctrness_terms(x, an, g, t.ctr_loss, t.ctr_grad);
    }
    return true;
}

template <typename T>
__global__ __launch_bounds__(RT) void box_loss_forward_kernel(BoxLossArgs p, double* __restrict__ part_loss,
                                                              double* __restrict__ part_ctr, int* __restrict__ part_fg) {
    __shared__ float s_sum[RT / 64];
    __shared__ float s_ctr[RT / 64];
    __shared__ int s_fg[RT / 64];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const long long a0 = (long long)blockIdx.x * RCHUNK;
    const long long* mrow = p.matched + (long long)b * p.n;
    long long m[RPT];
#pragma unroll
    for (int it = 0; it < RPT; ++it) {
        const long long a = a0 + it * RT + tid;
        m[it] = (a < p.n) ? __builtin_nontemporal_load(mrow + a) : -1;
    }
    float acc = 0.0f, acc_ctr = 0.0f;
    int fg = 0;
#pragma unroll
    for (int it = 0; it < RPT; ++it) {
        BoxTerms t;
        if (box_terms_of<T>(p, b, a0 + it * RT + tid, m[it], t)) {
            acc += t.loss;
            acc_ctr += t.ctr_loss;
            ++fg;
        }
    }
    float wsum = acc, wctr = acc_ctr;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { wsum += __shfl_xor(wsum, o); wctr += __shfl_xor(wctr, o); }
    const int wfg = (int)wave_sum_all((uint32_t)fg);
    if ((tid & 63) == 0) { s_sum[tid >> 6] = wsum; s_ctr[tid >> 6] = wctr; s_fg[tid >> 6] = wfg; }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0, c = 0.0;
        int f = 0;
#pragma unroll
        for (int w = 0; w < RT / 64; ++w) { t += (double)s_sum[w]; c += (double)s_ctr[w]; f += s_fg[w]; }
        part_loss[(long long)b * p.blocks_per_image + blockIdx.x] = t;
        if (part_ctr) part_ctr[(long long)b * p.blocks_per_image + blockIdx.x] = c;
        part_fg[(long long)b * p.blocks_per_image + blockIdx.x] = f;
    }
}

template <typename T>
__global__ __launch_bounds__(RT) void box_loss_backward_kernel(BoxLossArgs p, const int* __restrict__ num_fg,
                                                               const float* __restrict__ grad_out,
                                                               const float* __restrict__ grad_out_ctr, int batch,
                                                               T* __restrict__ grad_reg, T* __restrict__ grad_ctr) {
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const long long a0 = (long long)blockIdx.x * RCHUNK;
    const long long* mrow = p.matched + (long long)b * p.n;
    T* dst = grad_reg + (long long)b * p.n * 4;                      // the gradient is written contiguously
    long long m[RPT];
#pragma unroll
    for (int it = 0; it < RPT; ++it) {
        const long long a = a0 + it * RT + tid;
        m[it] = (a < p.n) ? __builtin_nontemporal_load(mrow + a) : -1;
    }
    const int nf = num_fg[b];
    const float denom = (float)(nf > 1 ? nf : 1) * (p.norm == OD_LOSS_NORM_PER_IMAGE ? (float)batch : 1.0f);
    const float scale = *grad_out / denom;
    const float scale_ctr = grad_ctr ? *grad_out_ctr / denom : 0.0f;
#pragma unroll
    for (int it = 0; it < RPT; ++it) {
        const long long a = a0 + it * RT + tid;
        float4 gr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float gc = 0.0f;
        BoxTerms t;
        if (box_terms_of<T>(p, b, a, m[it], t)) {
            gr = make_float4(scale * t.grad.x, scale * t.grad.y, scale * t.grad.z, scale * t.grad.w);
            gc = scale_ctr * t.ctr_grad;
        }
        if (a < p.n) {
            *reinterpret_cast<typename Elem<T>::V*>(dst + a * 4) = Elem<T>::pack(gr);
            if (grad_ctr) grad_ctr[(long long)b * p.n + a] = Elem<T>::from_float(gc);
        }
    }
}

bool box_loss_layout(int64_t n, int32_t batch, int* blocks, size_t* part_bytes, size_t* total) {
    if (n < 0 || batch < 0) return false;
    const long long blk = (n + RCHUNK - 1) / RCHUNK;
    if (blk > 0x7fffffff) return false;
    *blocks = (int)blk;
    *part_bytes = (((size_t)batch * (size_t)blk * sizeof(double)) + 255) & ~(size_t)255;
    // box partial sums, centre-ness partial sums, foreground counts
    *total = 2 * *part_bytes + ((((size_t)batch * (size_t)blk * sizeof(int)) + 255) & ~(size_t)255);
    return true;
}

int fill_box_args(BoxLossArgs* a, const void* reg, int32_t dtype, int64_t reg_stride, const float* anchors,
                  int64_t anchor_stride,
                  int64_t n, const int64_t* matched, const float* gt_boxes, const int32_t* gt_offsets, int32_t loss_type,
                  float beta, const float* weights, float clip, float eps, int blocks) {
    if (!reg || !anchors || !matched || !gt_offsets || !weights || elem_size(dtype) == 0) return OD_ERR_ARG;
    if (reg_stride < n * 4 || (anchor_stride != 0 && anchor_stride < n * 4)) return OD_ERR_ARG;
    // rows are read as vectors of four
    if (((uintptr_t)reg & (4 * elem_size(dtype) - 1)) || (reg_stride & 3) || ((uintptr_t)anchors & 15) || (anchor_stride & 3) ||
        ((uintptr_t)gt_boxes & 15) || ((uintptr_t)matched & 7))
        return OD_ERR_ARG;
    if (loss_type != OD_BOX_LOSS_L1 && loss_type != OD_BOX_LOSS_SMOOTH_L1 && loss_type != OD_BOX_LOSS_GIOU)
        return OD_ERR_ARG;
    if (loss_type == OD_BOX_LOSS_SMOOTH_L1 && !(beta >= 0.0f)) return OD_ERR_ARG;
    for (int k = 0; k < 4; ++k)
        if (!(weights[k] != 0.0f)) return OD_ERR_ARG;
    a->reg = reg;
    a->reg_stride = reg_stride;
    a->anchors = anchors;
    a->anchor_stride = anchor_stride;
    a->matched = (const long long*)matched;
    a->gt_boxes = gt_boxes;
    a->gt_offsets = gt_offsets;
    a->n = n;
    a->loss_type = loss_type;
    a->beta = (loss_type == OD_BOX_LOSS_SMOOTH_L1) ? beta : 0.0f;
    a->wx = weights[0]; a->wy = weights[1]; a->ww = weights[2]; a->wh = weights[3];
    a->clip = clip;
    a->eps = eps;
    a->blocks_per_image = blocks;
    a->coder = OD_CODER_BOXCODER;       // the FCOS entry points change these four
    a->norm = OD_LOSS_NORM_PER_IMAGE;
    a->ctr = nullptr;
    a->ctr_stride = 0;
    return OD_OK;
}

}  // namespace

extern "C" size_t od_focal_loss_workspace_bytes(int64_t n, int32_t num_classes, int32_t batch) {
    int blocks;
    size_t part, total;
    if (!loss_layout(n, num_classes, batch, &blocks, &part, &total)) return 0;
    return total > 0 ? total : 256;
}

extern "C" int od_focal_loss_forward(const void* cls_logits, int32_t dtype, int64_t batch_stride, int64_t n, int32_t num_classes,
                                     const int64_t* matched, const int64_t* gt_labels, const int32_t* gt_offsets,
                                     int32_t batch, float alpha, float gamma, int32_t normalization, float* loss,
                                     float* per_image_loss,
                                     int32_t* num_foreground, void* workspace, size_t workspace_bytes, void* stream) {
    int blocks;
    size_t part, total;
    if (!loss_layout(n, num_classes, batch, &blocks, &part, &total)) return OD_ERR_ARG;
    if (batch < 1 || batch > 65535) return batch == 0 ? OD_ERR_ARG : OD_ERR_LIMIT;
    if (!loss || !per_image_loss || !num_foreground || !workspace) return OD_ERR_ARG;
    if (workspace_bytes < total || ((uintptr_t)workspace & 255)) return OD_ERR_WORKSPACE;
    LossArgs a;
    const int rc = fill_args(&a, cls_logits, dtype, batch_stride, n, num_classes, matched, gt_labels, gt_offsets, alpha,
                             gamma, normalization, blocks);
    if (rc != OD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    double* part_loss = (double*)workspace;
    int* part_fg = (int*)((char*)workspace + part);
    if (blocks > 0) {
        const dim3 grid((unsigned)blocks, (unsigned)batch);
#define OD_FOCAL_FWD(G2, T)                                                                                  \
    OD_LAUNCH_KERNEL(OD_KERNEL_FOCAL_FWD, (focal_forward_kernel<G2, T>), grid, dim3(LT), 0, s, a, part_loss, \
                     part_fg)
        if (gamma == 2.0f) {
            if (dtype == OD_DTYPE_F32) { OD_FOCAL_FWD(true, float); }
            else if (dtype == OD_DTYPE_F16) { OD_FOCAL_FWD(true, _Float16); }
            else { OD_FOCAL_FWD(true, bf16_t); }
        } else {
            if (dtype == OD_DTYPE_F32) { OD_FOCAL_FWD(false, float); }
            else if (dtype == OD_DTYPE_F16) { OD_FOCAL_FWD(false, _Float16); }
            else { OD_FOCAL_FWD(false, bf16_t); }
        }
#undef OD_FOCAL_FWD
    }
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(1024), 0, s, (const double*)part_loss, (const int*)part_fg,
                       blocks, (int)batch, (int)normalization, loss, per_image_loss, num_foreground);
    OD_LAUNCH_CHECK();
    return OD_OK;
}

extern "C" int od_focal_loss_backward(const void* cls_logits, int32_t dtype, int64_t batch_stride, int64_t n,
                                      int32_t num_classes, const int64_t* matched, const int64_t* gt_labels,
                                      const int32_t* gt_offsets, int32_t batch, float alpha, float gamma,
                                      int32_t normalization, const int32_t* num_foreground, const float* grad_loss,
                                      void* grad_logits, void* stream) {
    int blocks;
    size_t part, total;
    if (!loss_layout(n, num_classes, batch, &blocks, &part, &total)) return OD_ERR_ARG;
    if (batch < 1 || batch > 65535) return batch == 0 ? OD_ERR_ARG : OD_ERR_LIMIT;
    if (!num_foreground || !grad_loss || !grad_logits || elem_size(dtype) == 0) return OD_ERR_ARG;
    if ((uintptr_t)grad_logits & (elem_size(dtype) - 1)) return OD_ERR_ARG;
    LossArgs a;
    const int rc = fill_args(&a, cls_logits, dtype, batch_stride, n, num_classes, matched, gt_labels, gt_offsets, alpha,
                             gamma, normalization, blocks);
    if (rc != OD_OK) return rc;
    if (blocks == 0) return OD_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks, (unsigned)batch);
#define OD_FOCAL_BWD(G2, T)                                                                                      \
    OD_LAUNCH_KERNEL(OD_KERNEL_FOCAL_BWD, (focal_backward_kernel<G2, T>), grid, dim3(LT), 0, s, a, num_foreground, \
                     grad_loss, (int)batch, (T*)grad_logits)
    if (gamma == 2.0f) {
        if (dtype == OD_DTYPE_F32) { OD_FOCAL_BWD(true, float); }
        else if (dtype == OD_DTYPE_F16) { OD_FOCAL_BWD(true, _Float16); }
        else { OD_FOCAL_BWD(true, bf16_t); }
    } else {
        if (dtype == OD_DTYPE_F32) { OD_FOCAL_BWD(false, float); }
        else if (dtype == OD_DTYPE_F16) { OD_FOCAL_BWD(false, _Float16); }
        else { OD_FOCAL_BWD(false, bf16_t); }
    }
#undef OD_FOCAL_BWD
    return OD_OK;
}

extern "C" size_t od_box_loss_workspace_bytes(int64_t n, int32_t batch) {
    int blocks;
    size_t part, total;
    if (!box_loss_layout(n, batch, &blocks, &part, &total)) return 0;
    return total > 0 ? total : 256;
}

namespace {

// Forward of the regression-type losses: `a` complete; loss_ctr null unless a.ctr is set.
int box_loss_forward_impl(const BoxLossArgs& a, int32_t dtype, int32_t batch, int blocks, size_t part, float* loss,
                          float* loss_ctr, float* per_image_loss, int32_t* num_foreground, void* workspace,
                          hipStream_t s) {
    double* part_loss = (double*)workspace;
    double* part_ctr = (double*)((char*)workspace + part);
    int* part_fg = (int*)((char*)workspace + 2 * part);
    if (blocks > 0) {
        const dim3 grid((unsigned)blocks, (unsigned)batch);
#define OD_BOX_FWD(T)                                                                                         \
    OD_LAUNCH_KERNEL(OD_KERNEL_BOX_LOSS_FWD, (box_loss_forward_kernel<T>), grid, dim3(RT), 0, s, a, part_loss, \
                     a.ctr ? part_ctr : (double*)nullptr, part_fg)
        if (dtype == OD_DTYPE_F32) { OD_BOX_FWD(float); }
        else if (dtype == OD_DTYPE_F16) { OD_BOX_FWD(_Float16); }
        else { OD_BOX_FWD(bf16_t); }
#undef OD_BOX_FWD
    }
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(1024), 0, s, (const double*)part_loss, (const int*)part_fg,
                       blocks, (int)batch, a.norm, loss, per_image_loss, num_foreground);
    if (a.ctr)
        hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(1024), 0, s, (const double*)part_ctr,
                           (const int*)part_fg, blocks, (int)batch, a.norm, loss_ctr, (float*)nullptr, num_foreground);
    OD_LAUNCH_CHECK();
    return OD_OK;
}

int box_loss_backward_impl(const BoxLossArgs& a, int32_t dtype, int32_t batch, int blocks, const int32_t* num_foreground,
                           const float* grad_loss, const float* grad_loss_ctr, void* grad_regression,
                           void* grad_ctrness, hipStream_t s) {
    if (blocks == 0) return OD_OK;
    const dim3 grid((unsigned)blocks, (unsigned)batch);
#define OD_BOX_BWD(T)                                                                                             \
    OD_LAUNCH_KERNEL(OD_KERNEL_BOX_LOSS_BWD, (box_loss_backward_kernel<T>), grid, dim3(RT), 0, s, a, num_foreground, \
                     grad_loss, grad_loss_ctr, (int)batch, (T*)grad_regression, (T*)grad_ctrness)
    if (dtype == OD_DTYPE_F32) { OD_BOX_BWD(float); }
    else if (dtype == OD_DTYPE_F16) { OD_BOX_BWD(_Float16); }
    else { OD_BOX_BWD(bf16_t); }
#undef OD_BOX_BWD
    return OD_OK;
}

const float kUnitWeights[4] = {1.0f, 1.0f, 1.0f, 1.0f};

}  // namespace

extern "C" int od_box_loss_forward(const void* bbox_regression, int32_t dtype, int64_t batch_stride, const float* anchors,
                                   int64_t anchor_batch_stride, int64_t n, const int64_t* matched, const float* gt_boxes,
                                   const int32_t* gt_offsets, int32_t batch, int32_t loss_type, float beta,
                                   const float* weights, float bbox_xform_clip, float eps, float* loss,
                                   float* per_image_loss, int32_t* num_foreground, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    int blocks;
    size_t part, total;
    if (!box_loss_layout(n, batch, &blocks, &part, &total)) return OD_ERR_ARG;
    if (batch < 1 || batch > 65535) return batch == 0 ? OD_ERR_ARG : OD_ERR_LIMIT;
    if (!loss || !per_image_loss || !num_foreground || !workspace) return OD_ERR_ARG;
    if (workspace_bytes < total || ((uintptr_t)workspace & 255)) return OD_ERR_WORKSPACE;
    BoxLossArgs a;
    const int rc = fill_box_args(&a, bbox_regression, dtype, batch_stride, anchors, anchor_batch_stride, n, matched, gt_boxes,
                                 gt_offsets, loss_type, beta, weights, bbox_xform_clip, eps, blocks);
    if (rc != OD_OK) return rc;
    return box_loss_forward_impl(a, dtype, batch, blocks, part, loss, nullptr, per_image_loss, num_foreground, workspace,
                                 (hipStream_t)stream);
}

extern "C" int od_box_loss_backward(const void* bbox_regression, int32_t dtype, int64_t batch_stride, const float* anchors,
                                    int64_t anchor_batch_stride, int64_t n, const int64_t* matched, const float* gt_boxes,
                                    const int32_t* gt_offsets, int32_t batch, int32_t loss_type, float beta,
                                    const float* weights, float bbox_xform_clip, float eps,
                                    const int32_t* num_foreground, const float* grad_loss, void* grad_regression,
                                    void* stream) {
    int blocks;
    size_t part, total;
    if (!box_loss_layout(n, batch, &blocks, &part, &total)) return OD_ERR_ARG;
    if (batch < 1 || batch > 65535) return batch == 0 ? OD_ERR_ARG : OD_ERR_LIMIT;
    if (!num_foreground || !grad_loss || !grad_regression || elem_size(dtype) == 0) return OD_ERR_ARG;
    if ((uintptr_t)grad_regression & (4 * elem_size(dtype) - 1)) return OD_ERR_ARG;
    BoxLossArgs a;
    const int rc = fill_box_args(&a, bbox_regression, dtype, batch_stride, anchors, anchor_batch_stride, n, matched, gt_boxes,
                                 gt_offsets, loss_type, beta, weights, bbox_xform_clip, eps, blocks);
    if (rc != OD_OK) return rc;
    return box_loss_backward_impl(a, dtype, batch, blocks, num_foreground, grad_loss, nullptr, grad_regression, nullptr,
                                  (hipStream_t)stream);
}

extern "C" int od_fcos_box_loss_forward(const void* bbox_regression, const void* bbox_ctrness, int32_t dtype,
                                        int64_t reg_batch_stride, int64_t ctr_batch_stride, const float* anchors,
                                        int64_t anchor_batch_stride, int64_t n, const int64_t* matched,
                                        const float* gt_boxes, const int32_t* gt_offsets, int32_t batch, float eps,
                                        float* loss_regression, float* loss_ctrness, int32_t* num_foreground,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    int blocks;
    size_t part, total;
    if (!box_loss_layout(n, batch, &blocks, &part, &total)) return OD_ERR_ARG;
    if (batch < 1 || batch > 65535) return batch == 0 ? OD_ERR_ARG : OD_ERR_LIMIT;
    if (!loss_regression || !loss_ctrness || !num_foreground || !workspace || !bbox_ctrness) return OD_ERR_ARG;
    if (workspace_bytes < total || ((uintptr_t)workspace & 255)) return OD_ERR_WORKSPACE;
    if (ctr_batch_stride < n || elem_size(dtype) == 0 || ((uintptr_t)bbox_ctrness & (elem_size(dtype) - 1))) return OD_ERR_ARG;
    BoxLossArgs a;
    const int rc = fill_box_args(&a, bbox_regression, dtype, reg_batch_stride, anchors, anchor_batch_stride, n, matched,
                                 gt_boxes, gt_offsets, OD_BOX_LOSS_GIOU, 0.0f, kUnitWeights, 0.0f, eps, blocks);
    if (rc != OD_OK) return rc;
    a.coder = OD_CODER_LINEAR;
    a.norm = OD_LOSS_NORM_BATCH;
    a.ctr = bbox_ctrness;
    a.ctr_stride = ctr_batch_stride;
    return box_loss_forward_impl(a, dtype, batch, blocks, part, loss_regression, loss_ctrness, nullptr, num_foreground,
                                 workspace, (hipStream_t)stream);
}

extern "C" int od_fcos_box_loss_backward(const void* bbox_regression, const void* bbox_ctrness, int32_t dtype,
                                         int64_t reg_batch_stride, int64_t ctr_batch_stride, const float* anchors,
                                         int64_t anchor_batch_stride, int64_t n, const int64_t* matched,
                                         const float* gt_boxes, const int32_t* gt_offsets, int32_t batch, float eps,
                                         const int32_t* num_foreground, const float* grad_loss_regression,
                                         const float* grad_loss_ctrness, void* grad_regression, void* grad_ctrness,
                                         void* stream) {
    int blocks;
    size_t part, total;
    if (!box_loss_layout(n, batch, &blocks, &part, &total)) return OD_ERR_ARG;
    if (batch < 1 || batch > 65535) return batch == 0 ? OD_ERR_ARG : OD_ERR_LIMIT;
    if (!num_foreground || !grad_loss_regression || !grad_loss_ctrness || !grad_regression || !grad_ctrness || !bbox_ctrness)
        return OD_ERR_ARG;
    if (elem_size(dtype) == 0 || ((uintptr_t)grad_regression & (4 * elem_size(dtype) - 1)) ||
        ((uintptr_t)grad_ctrness & (elem_size(dtype) - 1)) || ctr_batch_stride < n)
        return OD_ERR_ARG;
    BoxLossArgs a;
    const int rc = fill_box_args(&a, bbox_regression, dtype, reg_batch_stride, anchors, anchor_batch_stride, n, matched,
                                 gt_boxes, gt_offsets, OD_BOX_LOSS_GIOU, 0.0f, kUnitWeights, 0.0f, eps, blocks);
    if (rc != OD_OK) return rc;
    a.coder = OD_CODER_LINEAR;
    a.norm = OD_LOSS_NORM_BATCH;
    a.ctr = bbox_ctrness;
    a.ctr_stride = ctr_batch_stride;
    return box_loss_backward_impl(a, dtype, batch, blocks, num_foreground, grad_loss_regression, grad_loss_ctrness,
                                  grad_regression, grad_ctrness, (hipStream_t)stream);
}
