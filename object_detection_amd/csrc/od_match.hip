// Anchor <-> ground-truth matching kernels.
//
// od_match_retinanet: fused torchvision box_iou + Matcher(high, low, allow_low_quality_matches)
// (the per-image loop body of RetinaNet.compute_loss; reference trigger object_detection/train.py:62,
// thresholds object_detection/configs/config_all.py:10-11).  The [M,N] IoU matrix is never stored.
//
// Work decomposition: one workgroup = one tile of 1024 consecutive anchors of one image
// (256 threads x 4 anchors, anchors live in registers).  GT boxes are staged through LDS in chunks;
// a GT whose box cannot overlap the tile's bounding box is culled (its IoU with every anchor of the
// tile is exactly +0, which can neither win a strict '>' argmax nor raise a per-GT maximum), and the
// survivors keep their ascending order so "first maximal index" tie-breaking is unchanged.
//
// Two passes over the same code path (so recomputed IoUs are bit-identical):
//   pass 1: per-GT maximum over all anchors  (wave DPP max -> LDS atomic max -> global atomic max)
//   pass 2: per-anchor max/argmax, thresholds, and the low-quality restore
//           (anchor a is restored to its own argmax when iou[g,a] == max_a' iou[g,a'] for some g).
// If some GT has maximum 0 (it overlaps no anchor) torchvision restores EVERY anchor (all of them tie
// at 0 for that GT); pass 2 detects that case from the per-GT maxima and does the same.
#include "od_common.h"
#include "../../include/od_boxes.h"

namespace {

using namespace od;

constexpr int MT = 256;        // threads per workgroup
constexpr int R = 4;           // anchors per thread
constexpr int TILE = MT * R;   // anchors per workgroup
constexpr int GCH = 512;       // GT boxes staged per chunk (2 per thread)

struct GtAux {                 // second 16-byte LDS record per staged GT
    float area;
    float gmax;                // pass 2: global maximum IoU of this GT
    int index;                 // GT index inside its image
    int pad;
};

template <int PASS>
__global__ __launch_bounds__(MT) void match_retinanet_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n,
    const float4* __restrict__ gt, const int* __restrict__ gt_offsets, uint32_t* gt_max_key,
    float high, float low, int allow_lq, long long* __restrict__ matched) {
    __shared__ float4 s_g[GCH];
    __shared__ GtAux s_aux[GCH];
    __shared__ uint32_t s_tilemax[GCH];
    __shared__ float s_red[4][4];
    __shared__ float s_bbox[4];
    __shared__ int s_wcnt[4];
    __shared__ int s_flag;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.y;
    const long long tile_base = (long long)blockIdx.x * TILE;
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    long long* out = matched + (long long)b * n;

    if (M <= 0) {                                   // empty-target guard of compute_loss
        if (PASS == 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                long long idx = tile_base + tid + r * MT;
                if (idx < n) out[idx] = -1;
            }
        }
        return;
    }

    // ---- anchors of this tile into registers (out-of-range lanes replay the last anchor) --------
    const float4* A = anchors + (long long)b * anchor_batch_stride;
    float ax1[R], ay1[R], ax2[R], ay2[R], aarea[R];
    float bx1 = __builtin_inff(), by1 = __builtin_inff(), bx2 = -__builtin_inff(), by2 = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long long idx = tile_base + tid + r * MT;
        if (idx > n - 1) idx = n - 1;
        float4 a = A[idx];
        ax1[r] = a.x; ay1[r] = a.y; ax2[r] = a.z; ay2[r] = a.w;
        aarea[r] = box_area(a.x, a.y, a.z, a.w);
        bx1 = fminf(bx1, a.x); by1 = fminf(by1, a.y);
        bx2 = fmaxf(bx2, a.z); by2 = fmaxf(by2, a.w);
    }
    // tile bounding box: wave DPP reductions, then across the 4 waves through LDS
    bx1 = wave_fmin_all(bx1); by1 = wave_fmin_all(by1);
    bx2 = wave_fmax_all(bx2); by2 = wave_fmax_all(by2);
    if (lane == 0) { s_red[wave][0] = bx1; s_red[wave][1] = by1; s_red[wave][2] = bx2; s_red[wave][3] = by2; }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (tid == 0) {
        s_bbox[0] = fminf(fminf(s_red[0][0], s_red[1][0]), fminf(s_red[2][0], s_red[3][0]));
        s_bbox[1] = fminf(fminf(s_red[0][1], s_red[1][1]), fminf(s_red[2][1], s_red[3][1]));
        s_bbox[2] = fmaxf(fmaxf(s_red[0][2], s_red[1][2]), fmaxf(s_red[2][2], s_red[3][2]));
        s_bbox[3] = fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3]));
    }
    // pass 2: does some GT of this image have maximum IoU == 0 ?  (then every anchor is restored)
    if (PASS == 2 && allow_lq) {
        int any0 = 0;
        for (int g = tid; g < M; g += MT) any0 |= (ordered_key_inv(gt_max_key[m0 + g]) == 0.0f);
        if (any0) s_flag = 1;   // benign race: every writer stores 1
    }
    __syncthreads();
    const float tb_x1 = s_bbox[0], tb_y1 = s_bbox[1], tb_x2 = s_bbox[2], tb_y2 = s_bbox[3];
    const bool restore_all = (PASS == 2) && allow_lq && (s_flag != 0);

    float best[R];
    int bidx[R];
    bool restore[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0.0f; bidx[r] = 0; restore[r] = false; }

    for (int c0 = 0; c0 < M; c0 += GCH) {
        // ---- stage the GT of this chunk that can overlap the tile, order preserved --------------
        int base = 0;
#pragma unroll
        for (int sr = 0; sr < GCH / MT; ++sr) {
            int g = c0 + sr * MT + tid;
            bool ok = g < M;
            float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                gb = gt[m0 + g];
                ok = !(gb.z <= tb_x1 || gb.x >= tb_x2 || gb.w <= tb_y1 || gb.y >= tb_y2);
            }
            uint64_t bal = __ballot(ok);
            if (lane == 0) s_wcnt[wave] = __popcll(bal);
            __syncthreads();
            int pos = base + mbcnt(bal);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                int c = s_wcnt[w];
                if (w < wave) pos += c;
                base += c;
            }
            if (ok) {
                s_g[pos] = gb;
                GtAux aux;
                aux.area = box_area(gb.x, gb.y, gb.z, gb.w);
                aux.gmax = (PASS == 2) ? ordered_key_inv(gt_max_key[m0 + g]) : 0.0f;
                aux.index = g;
                aux.pad = 0;
                s_aux[pos] = aux;
                if (PASS == 1) s_tilemax[pos] = 0u;
            }
            __syncthreads();
        }
        const int cnt = base;

        // ---- IoU of the 4 register anchors against every staged GT ------------------------------
        for (int j = 0; j < cnt; ++j) {
            const float4 gb = s_g[j];
            const GtAux aux = s_aux[j];
            float v[R];
#pragma unroll
            for (int r = 0; r < R; ++r)
                v[r] = iou_pair(gb.x, gb.y, gb.z, gb.w, aux.area, ax1[r], ay1[r], ax2[r], ay2[r], aarea[r]);
            if (PASS == 1) {
                float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                uint32_t k = wave_umax_lane63(ordered_key(m));
                if (lane == 63) atomicMax(&s_tilemax[j], k);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (v[r] > best[r]) { best[r] = v[r]; bidx[r] = aux.index; }
                    restore[r] = restore[r] || (v[r] == aux.gmax);
                }
            }
        }
        if (PASS == 1) {
            __syncthreads();
            const uint32_t zero_key = 0x80000000u;   // ordered_key(+0.0f), the initial per-GT maximum
            for (int j = tid; j < cnt; j += MT) {
                uint32_t k = s_tilemax[j];
                if (k > zero_key) atomicMax(&gt_max_key[m0 + s_aux[j].index], k);
            }
        }
        __syncthreads();
    }

    if (PASS == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            long long idx = tile_base + tid + r * MT;
            if (idx < n) {
                long long m = (best[r] < low) ? -1 : ((best[r] < high) ? -2 : (long long)bidx[r]);
                if (allow_lq && (restore[r] || restore_all)) m = bidx[r];
                out[idx] = m;
            }
        }
    }
}

// ---- FCOS centre sampling: one thread per anchor, GT staged in LDS --------------------------------
constexpr int FT = 256;
constexpr int FGCH = 1024;

__global__ __launch_bounds__(FT) void match_fcos_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n, long long n_first,
    long long n_last, const float4* __restrict__ gt, const int* __restrict__ gt_offsets, float radius,
    long long* __restrict__ matched) {
    __shared__ float4 s_g[FGCH];
    const int b = blockIdx.y;
    const long long idx = (long long)blockIdx.x * FT + threadIdx.x;
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    const bool valid = idx < n;
    float4 a = make_float4(0.f, 0.f, 1.f, 1.f);
    if (valid) a = anchors[(long long)b * anchor_batch_stride + idx];
    // anchor_centers = (a[:, :2] + a[:, 2:]) / 2 ; anchor_sizes = a[:, 2] - a[:, 0]
    const float cx = (a.x + a.z) / 2.0f, cy = (a.y + a.w) / 2.0f;
    const float size = a.z - a.x;
    const float lower = (idx < n_first) ? 0.0f : size * 4.0f;
    const float upper = (idx >= n - n_last) ? __builtin_inff() : size * 8.0f;
    const float rs = radius * size;
    float best = 0.0f;
    int bi = 0;
    for (int c0 = 0; c0 < M; c0 += FGCH) {
        __syncthreads();
        for (int g = threadIdx.x; g < FGCH && c0 + g < M; g += FT) s_g[g] = gt[m0 + c0 + g];
        __syncthreads();
        const int cnt = min(FGCH, M - c0);
        for (int j = 0; j < cnt; ++j) {
            const float4 G = s_g[j];
            float gcx = (G.x + G.z) / 2.0f, gcy = (G.y + G.w) / 2.0f;
            float dmax = fmaxf(fabsf(cx - gcx), fabsf(cy - gcy));
            float l = cx - G.x, t = cy - G.y, r = G.z - cx, bt = G.w - cy;
            float mn = fminf(fminf(l, t), fminf(r, bt));
            float mx = fmaxf(fmaxf(l, t), fmaxf(r, bt));
            bool match = (dmax < rs) && (mn > 0.0f) && (mx > lower) && (mx < upper);
            float garea = (G.z - G.x) * (G.w - G.y);
            float v = (match ? 1.0f : 0.0f) * (1e8f - garea);
            if ((c0 + j == 0) || v > best) { best = v; bi = c0 + j; }
        }
    }
    if (valid) matched[(long long)b * n + idx] = (M <= 0 || best < 1e-5f) ? -1 : (long long)bi;
}

// ---- explicit IoU matrix and Matcher on a given matrix -------------------------------------------
__global__ __launch_bounds__(256) void box_iou_kernel(const float4* __restrict__ b1, long long m,
                                                      const float4* __restrict__ b2, long long n,
                                                      float* __restrict__ out) {
    long long a = (long long)blockIdx.x * 256 + threadIdx.x;
    long long g = blockIdx.y;
    if (a >= n || g >= m) return;
    float4 G = b1[g], A = b2[a];
    out[g * n + a] = iou_pair(G.x, G.y, G.z, G.w, box_area(G.x, G.y, G.z, G.w), A.x, A.y, A.z, A.w,
                              box_area(A.x, A.y, A.z, A.w));
}

// torch.max semantics: NaN is the maximum, first index among equals.
__device__ __forceinline__ bool torch_gt(float v, float best) { return (v > best) || (v != v && best == best); }

__global__ __launch_bounds__(256) void matcher_rowmax_kernel(const float* __restrict__ q, long long n,
                                                             float* __restrict__ rowmax) {
    __shared__ float s[256];
    long long g = blockIdx.x;
    const float* row = q + g * n;
    float best = row[0];
    for (long long a = threadIdx.x; a < n; a += 256) { float v = row[a]; if (torch_gt(v, best)) best = v; }
    s[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { float v = s[threadIdx.x + o]; if (torch_gt(v, s[threadIdx.x])) s[threadIdx.x] = v; }
        __syncthreads();
    }
    if (threadIdx.x == 0) rowmax[g] = s[0];
}

__global__ __launch_bounds__(256) void matcher_col_kernel(const float* __restrict__ q, long long m, long long n,
                                                          const float* __restrict__ rowmax, float high, float low,
                                                          int allow_lq, long long* __restrict__ matches) {
    long long a = (long long)blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    float best = q[a];
    long long bi = 0;
    bool restore = false;
    for (long long g = 0; g < m; ++g) {
        float v = q[g * n + a];
        if (g > 0 && torch_gt(v, best)) { best = v; bi = g; }
        restore = restore || (v == rowmax[g]);
    }
    long long r = (best < low) ? -1 : (((best >= low) && (best < high)) ? -2 : bi);
    if (allow_lq && restore) r = bi;
    matches[a] = r;
}

}  // namespace

extern "C" size_t od_match_retinanet_workspace_bytes(int64_t m_total, int32_t batch) {
    (void)batch;
    size_t b = (size_t)(m_total > 0 ? m_total : 0) * sizeof(uint32_t);
    return (b + 255) & ~(size_t)255;
}

extern "C" int od_match_retinanet(const float* anchors, int64_t anchor_batch_stride, int64_t n,
                                  const float* gt_boxes, const int32_t* gt_offsets, int64_t m_total,
                                  int32_t batch, float high, float low, int32_t allow_low_quality,
                                  int64_t* matched, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || batch < 0 || m_total < 0) return OD_ERR_ARG;
    if (n == 0 || batch == 0) return OD_OK;
    if (!anchors || !gt_offsets || !matched || (m_total > 0 && !gt_boxes)) return OD_ERR_ARG;
    if (anchor_batch_stride != 0 && anchor_batch_stride < n) return OD_ERR_ARG;
    if (((uintptr_t)anchors & 15) || ((uintptr_t)gt_boxes & 15)) return OD_ERR_ARG;
    if (m_total > 0x7fffffff || batch > 65535) return OD_ERR_LIMIT;
    if (workspace_bytes < od_match_retinanet_workspace_bytes(m_total, batch)) return OD_ERR_WORKSPACE;
    if (m_total > 0 && !workspace) return OD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    uint32_t* gt_max_key = (uint32_t*)workspace;
    dim3 grid((unsigned)((n + TILE - 1) / TILE), (unsigned)batch);
    if (allow_low_quality && m_total > 0) {
        OD_HIP(hipMemsetD32Async((hipDeviceptr_t)gt_max_key, (int)0x80000000u, (size_t)m_total, s));
        OD_LAUNCH(OD_KERNEL_MATCH_GTMAX, s,
                  hipLaunchKernelGGL(match_retinanet_kernel<1>, grid, dim3(MT), 0, s, (const float4*)anchors,
                                     (long long)anchor_batch_stride, (long long)n, (const float4*)gt_boxes,
                                     gt_offsets, gt_max_key, high, low, 1, (long long*)matched));
    }
    OD_LAUNCH(OD_KERNEL_MATCH_ASSIGN, s,
              hipLaunchKernelGGL(match_retinanet_kernel<2>, grid, dim3(MT), 0, s, (const float4*)anchors,
                                 (long long)anchor_batch_stride, (long long)n, (const float4*)gt_boxes, gt_offsets,
                                 gt_max_key, high, low, allow_low_quality ? 1 : 0, (long long*)matched));
    return OD_OK;
}

extern "C" int od_match_fcos(const float* anchors, int64_t anchor_batch_stride, int64_t n, int64_t n_first_level,
                             int64_t n_last_level, const float* gt_boxes, const int32_t* gt_offsets,
                             int32_t batch, float center_sampling_radius, int64_t* matched, void* stream) {
    if (n < 0 || batch < 0) return OD_ERR_ARG;
    if (n == 0 || batch == 0) return OD_OK;
    if (!anchors || !gt_offsets || !matched) return OD_ERR_ARG;
    if (anchor_batch_stride != 0 && anchor_batch_stride < n) return OD_ERR_ARG;
    if (((uintptr_t)anchors & 15) || ((uintptr_t)gt_boxes & 15)) return OD_ERR_ARG;
    if (n_first_level < 0 || n_last_level < 0 || n_first_level > n || n_last_level > n) return OD_ERR_ARG;
    if (batch > 65535) return OD_ERR_LIMIT;
    dim3 grid((unsigned)((n + FT - 1) / FT), (unsigned)batch);
    OD_LAUNCH(OD_KERNEL_MATCH_FCOS, (hipStream_t)stream,
              hipLaunchKernelGGL(match_fcos_kernel, grid, dim3(FT), 0, (hipStream_t)stream, (const float4*)anchors,
                                 (long long)anchor_batch_stride, (long long)n, (long long)n_first_level,
                                 (long long)n_last_level, (const float4*)gt_boxes, gt_offsets,
                                 center_sampling_radius, (long long*)matched));
    return OD_OK;
}

extern "C" int od_box_iou(const float* boxes1, int64_t m, const float* boxes2, int64_t n, float* out, void* stream) {
    if (m < 0 || n < 0) return OD_ERR_ARG;
    if (m == 0 || n == 0) return OD_OK;
    if (!boxes1 || !boxes2 || !out) return OD_ERR_ARG;
    if (((uintptr_t)boxes1 & 15) || ((uintptr_t)boxes2 & 15)) return OD_ERR_ARG;
    if (m > 65535) return OD_ERR_LIMIT;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)m);
    OD_LAUNCH(OD_KERNEL_BOX_IOU, (hipStream_t)stream,
              hipLaunchKernelGGL(box_iou_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)boxes1,
                                 (long long)m, (const float4*)boxes2, (long long)n, out));
    return OD_OK;
}

extern "C" size_t od_matcher_workspace_bytes(int64_t m, int64_t n) {
    (void)n;
    size_t b = (size_t)(m > 0 ? m : 0) * sizeof(float);
    return (b + 255) & ~(size_t)255;
}

extern "C" int od_matcher(const float* quality, int64_t m, int64_t n, float high, float low,
                          int32_t allow_low_quality, int64_t* matches, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (m <= 0 || n <= 0) return OD_ERR_ARG;   // torchvision raises ValueError on an empty matrix
    if (!quality || !matches || !workspace) return OD_ERR_ARG;
    if (workspace_bytes < od_matcher_workspace_bytes(m, n)) return OD_ERR_WORKSPACE;
    if (m > 0x7fffffff) return OD_ERR_LIMIT;
    hipStream_t s = (hipStream_t)stream;
    float* rowmax = (float*)workspace;
    OD_LAUNCH(OD_KERNEL_MATCHER, s,
              hipLaunchKernelGGL(matcher_rowmax_kernel, dim3((unsigned)m), dim3(256), 0, s, quality, (long long)n,
                                 rowmax));
    OD_LAUNCH(OD_KERNEL_MATCHER, s,
              hipLaunchKernelGGL(matcher_col_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, quality,
                                 (long long)m, (long long)n, rowmax, high, low, allow_low_quality ? 1 : 0,
                                 (long long*)matches));
    return OD_OK;
}
