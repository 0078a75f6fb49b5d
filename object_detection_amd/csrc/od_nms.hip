// Class-batched non-maximum suppression, one 1024-thread workgroup per image.
// Replaces torchvision.ops.boxes.batched_nms (_batched_nms_coordinate_trick / _batched_nms_vanilla),
// the native torchvision::nms op (csrc/ops/cpu/nms_kernel.cpp semantics), `keep[:detections_per_img]`
// and the gather of the kept boxes/scores/labels at the tail of postprocess_detections
// (reference trigger object_detection/train.py:84, object_detection/predict.py:61, and the plain
// nms call at object_detection/predict.py:72).
//
// Algorithm (greedy NMS is inherently ordered; the work around the order is made parallel):
//   1. keys (ordered score bits << 32 | ~slot) are bitonic-sorted in LDS: stable descending order.
//   2. the sorted boxes are gathered once into an SoA workspace (coordinate-trick offsets applied).
//   3. candidates are consumed in strips of 256 in sorted order, stopping as soon as max_keep boxes
//      are kept:
//        A. every strip candidate is tested against the boxes kept so far (kept boxes are loaded 64
//           at a time into registers and broadcast lane by lane; 4 thread groups split the kept list);
//        B. the still-alive candidates are compacted and their pairwise suppression bits
//           (row i suppresses column j, i < j) are computed by all 16 waves into an LDS bit matrix
//           stored per column;
//        C. one wave resolves the strip 64 candidates at a time: a column is dead if its bit column
//           meets the kept set of earlier blocks; inside a block the greedy result is the fixed
//           point of  kept = alive & ~(exists kept i<j suppressing j), reached by ballot iteration.
//   4. kept boxes are gathered to the outputs in order; rows past the count are zero-filled.
#include "od_common.h"
#include "../../include/od_boxes.h"

namespace {

using namespace od;

constexpr int NT = 1024;              // threads per workgroup
constexpr int S = 256;                // strip length
constexpr int PARTS = NT / S;         // thread groups splitting the kept list in phase A
constexpr int SB = S / 64;            // 64-candidate blocks per strip
constexpr int MAXN = OD_MAX_NMS_BOXES;
constexpr int CW = SB * 2 + 1;        // u32 words per column of the bit matrix (+1 pad against bank conflicts)
constexpr int MAX_SEG = 64;

struct NmsArgs {
    const float4* boxes;
    const float* scores;
    const int* labels;
    const int* seg_count;
    int num_seg, seg_stride, n_slots, n_pad, segments_sorted;
    float thr_up;          // smallest float f with (double)f > iou_threshold; compare ovr >= thr_up
    int can_skip_zero;     // iou_threshold >= 0: a pair with zero intersection can never suppress
    int mode, max_keep;
    float* ws;             // per image 7 * n_pad words: x1,y1,x2,y2,area (float), label, slot (int)
    long long* keep_idx;
    float4* out_boxes;
    float* out_scores;
    long long* out_labels;
    int* out_count;
};

// intersection area exactly as nms_kernel.cpp: max(0, min(x2) - max(x1)) * max(0, min(y2) - max(y1))
__device__ __forceinline__ float intersection(float ix1, float iy1, float ix2, float iy2, float jx1, float jy1,
                                              float jx2, float jy2) {
    const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
    const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    return w * h;
}

// ovr = inter / (iarea + jarea - inter);  (double)ovr > thr  <=>  ovr >= thr_up
__device__ __forceinline__ bool over_threshold(float inter, float iarea, float jarea, float thr_up) {
    const float ovr = inter / ((iarea + jarea) - inter);
    return ovr >= thr_up;
}

__global__ __launch_bounds__(NT) void nms_kernel(NmsArgs a) {
    __shared__ unsigned long long s_keys[MAXN];
    __shared__ int s_kept[MAXN];
    __shared__ unsigned int s_col[S * CW];
    __shared__ int s_dead[S];
    __shared__ int s_alive[S];
    __shared__ int s_segpre[MAX_SEG + 1];
    __shared__ int s_tmp[40];
    __shared__ float s_ftmp[NT / 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float4* boxes = a.boxes + (long long)b * a.n_slots;
    const float* scores = a.scores + (long long)b * a.n_slots;
    const int* labels = a.labels ? a.labels + (long long)b * a.n_slots : nullptr;
    const int* segc = a.seg_count + (long long)b * a.num_seg;
    float* sx1 = a.ws + (long long)b * 7 * a.n_pad;
    float* sy1 = sx1 + a.n_pad;
    float* sx2 = sy1 + a.n_pad;
    float* sy2 = sx2 + a.n_pad;
    float* sar = sy2 + a.n_pad;
    int* slab = reinterpret_cast<int*>(sar + a.n_pad);
    int* sslot = slab + a.n_pad;

    // ---- 1. keys, valid count, max coordinate ------------------------------------------------------
    if (tid == 0) {
        int acc = 0;
        for (int s = 0; s < a.num_seg; ++s) {
            s_segpre[s] = acc;
            int c = segc[s];
            c = c < 0 ? 0 : (c > a.seg_stride ? a.seg_stride : c);
            acc += c;
        }
        s_segpre[a.num_seg] = acc;
    }
    __syncthreads();
    const int n_valid = s_segpre[a.num_seg];
    float maxc = -__builtin_inff();
    for (int i = tid; i < a.n_pad; i += NT) {
        unsigned long long key = 0ull;
        if (i < a.n_slots) {
            const int seg = i / a.seg_stride;
            const int r = i - seg * a.seg_stride;
            if (r < s_segpre[seg + 1] - s_segpre[seg]) {
                key = ((unsigned long long)ordered_key(scores[i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
                const float4 bx = boxes[i];
                maxc = fmaxf(maxc, fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
            }
        }
        s_keys[i] = key;
    }
    maxc = wave_fmax_all(maxc);
    if (lane == 0) s_ftmp[wave] = maxc;
    __syncthreads();
    maxc = s_ftmp[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) maxc = fmaxf(maxc, s_ftmp[w]);

    int mode = a.mode;
    if (mode == OD_NMS_AUTO_CPU) mode = (4 * n_valid > 4000) ? OD_NMS_VANILLA : OD_NMS_TRICK;
    if (mode == OD_NMS_AUTO_GPU) mode = (4 * n_valid > 20000) ? OD_NMS_VANILLA : OD_NMS_TRICK;
    const bool vanilla = (mode == OD_NMS_VANILLA) && labels != nullptr;
    const bool trick = (mode == OD_NMS_TRICK) && labels != nullptr;

    // ---- 2. bitonic sort (descending): (score desc, slot asc) --------------------------------------
    for (int k = 2; k <= a.n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < a.n_pad; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = s_keys[i], y = s_keys[ixj];
                    const bool desc = (i & k) == 0;
                    if ((x < y) == desc) { s_keys[i] = y; s_keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }

    // ---- 3. gather sorted boxes into the SoA workspace --------------------------------------------
    const float step = maxc + 1.0f;   // offsets = labels.to(boxes) * (boxes.max() + 1)
    for (int p = tid; p < n_valid; p += NT) {
        const int slot = (int)(0xFFFFFFFFu - (uint32_t)s_keys[p]);
        float4 bx = boxes[slot];
        const int lab = labels ? labels[slot] : 0;
        if (trick) {
            const float off = (float)lab * step;
            bx.x += off; bx.y += off; bx.z += off; bx.w += off;
        }
        sx1[p] = bx.x; sy1[p] = bx.y; sx2[p] = bx.z; sy2[p] = bx.w;
        sar[p] = (bx.z - bx.x) * (bx.w - bx.y);
        slab[p] = lab;
        sslot[p] = slot;
    }
    if (tid == 0) { s_tmp[0] = 0; s_tmp[1] = 0; }   // [0] = kept count, [1] = done flag
    __syncthreads();

    const int max_keep = a.max_keep < n_valid ? a.max_keep : n_valid;
    const float thr_up = a.thr_up;
    const bool can_skip = a.can_skip_zero != 0;

    // ---- 4. strips ------------------------------------------------------------------------------------
    for (int s0 = 0; s0 < n_valid && max_keep > 0; s0 += S) {
        const int nk = s_tmp[0];
        if (tid < S) s_dead[tid] = 0;
        for (int i = tid; i < S * CW; i += NT) s_col[i] = 0u;
        __syncthreads();

        // -- A: against the kept list ------------------------------------------------------------------
        {
            const int c = tid & (S - 1);
            const int part = tid / S;
            const int p = s0 + c;
            const bool valid = p < n_valid;
            const int pp = valid ? p : n_valid - 1;
            const float jx1 = sx1[pp], jy1 = sy1[pp], jx2 = sx2[pp], jy2 = sy2[pp], jar = sar[pp];
            const int jlab = slab[pp];
            bool dead = false;
            for (int kb = part; kb * 64 < nk; kb += PARTS) {
                const int e = kb * 64 + lane;
                const int kp = s_kept[e < nk ? e : nk - 1];
                const float kx1 = sx1[kp], ky1 = sy1[kp], kx2 = sx2[kp], ky2 = sy2[kp], kar = sar[kp];
                const int klab = slab[kp];
                const int cnt = (nk - kb * 64) < 64 ? (nk - kb * 64) : 64;
                for (int i = 0; i < cnt; ++i) {
                    const float ix1 = bcast_lane(kx1, i), iy1 = bcast_lane(ky1, i);
                    const float ix2 = bcast_lane(kx2, i), iy2 = bcast_lane(ky2, i);
                    const float iar = bcast_lane(kar, i);
                    const float inter = intersection(ix1, iy1, ix2, iy2, jx1, jy1, jx2, jy2);
                    if (can_skip && __ballot(inter > 0.0f) == 0) continue;
                    bool sup = over_threshold(inter, iar, jar, thr_up);
                    if (vanilla) sup = sup && (bcast_lane(klab, i) == jlab);
                    dead = dead || sup;
                }
            }
            if (dead && valid) s_dead[c] = 1;
        }
        __syncthreads();

        // -- compact the alive candidates of the strip (order preserved) -----------------------------
        if (tid < S) {
            const bool alive = (s0 + tid < n_valid) && (s_dead[tid] == 0);
            const uint64_t bal = __ballot(alive);
            if (lane == 0) s_tmp[8 + wave] = __popcll(bal);
            s_dead[tid] = alive ? mbcnt(bal) : -1;   // reuse: rank inside the wave, or -1
        }
        __syncthreads();
        int na = 0;
        {
            int pre = 0;
#pragma unroll
            for (int w = 0; w < SB; ++w) {
                const int c = s_tmp[8 + w];
                if (w < wave) pre += c;
                na += c;
            }
            if (tid < S && s_dead[tid] >= 0) s_alive[pre + s_dead[tid]] = tid;
        }
        __syncthreads();
        const int nab = (na + 63) >> 6;

        // -- B: pairwise bits among alive candidates, 16 rows x 64 columns per work item --------------
        {
            const int n_tiles = nab * (nab + 1) / 2;
            for (int item = wave; item < n_tiles * 4; item += NT / 64) {
                const int tile = item >> 2, quarter = item & 3;
                int rb = 0, t = tile;
                while (t >= nab - rb) { t -= nab - rb; ++rb; }   // row block rb, column block cb >= rb
                const int cb = rb + t;
                const int col = cb * 64 + lane;
                const bool cvalid = col < na;
                const int cp = s0 + s_alive[cvalid ? col : na - 1];
                const float jx1 = sx1[cp], jy1 = sy1[cp], jx2 = sx2[cp], jy2 = sy2[cp], jar = sar[cp];
                const int jlab = slab[cp];
                const int row0 = rb * 64 + quarter * 16;
                const int rrow = row0 + (lane & 15);
                const int rp = s0 + s_alive[rrow < na ? rrow : na - 1];
                const float rx1 = sx1[rp], ry1 = sy1[rp], rx2 = sx2[rp], ry2 = sy2[rp], rar = sar[rp];
                const int rlab = slab[rp];
                const int rcnt = (na - row0) < 16 ? (na - row0) : 16;
                unsigned int bits = 0u;
                for (int i = 0; i < rcnt; ++i) {
                    const float ix1 = bcast_lane(rx1, i), iy1 = bcast_lane(ry1, i);
                    const float ix2 = bcast_lane(rx2, i), iy2 = bcast_lane(ry2, i);
                    const float iar = bcast_lane(rar, i);
                    const float inter = intersection(ix1, iy1, ix2, iy2, jx1, jy1, jx2, jy2);
                    if (can_skip && __ballot(inter > 0.0f) == 0) continue;
                    bool sup = over_threshold(inter, iar, jar, thr_up);
                    if (vanilla) sup = sup && (bcast_lane(rlab, i) == jlab);
                    sup = sup && (row0 + i < col);
                    bits |= sup ? (1u << i) : 0u;
                }
                if (cvalid && bits) atomicOr(&s_col[col * CW + rb * 2 + (quarter >> 1)], bits << ((quarter & 1) * 16));
            }
        }
        __syncthreads();

        // -- C: resolve the strip in order (one wave) ---------------------------------------------------
        if (wave == 0) {
            unsigned long long keptw[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) keptw[i] = 0ull;
            int nkept = nk;
            bool done = false;
#pragma unroll
            for (int blk = 0; blk < SB; ++blk) {
                if (blk < nab && !done) {
                    const int col = blk * 64 + lane;
                    const bool cvalid = col < na;
                    const unsigned int* cw = &s_col[(cvalid ? col : 0) * CW];
                    bool dead = false;
#pragma unroll
                    for (int c = 0; c < SB; ++c) {
                        if (c < blk) {
                            const unsigned long long m = ((unsigned long long)cw[2 * c + 1] << 32) | cw[2 * c];
                            dead = dead || ((m & keptw[c]) != 0ull);
                        }
                    }
                    const unsigned long long own = ((unsigned long long)cw[2 * blk + 1] << 32) | cw[2 * blk];
                    const bool base = cvalid && !dead;
                    unsigned long long K = __ballot(base);
                    for (int it = 0; it < 64; ++it) {
                        const unsigned long long K2 = __ballot(base && ((own & K) == 0ull));
                        if (K2 == K) break;
                        K = K2;
                    }
                    keptw[blk] = K;
                    const bool kept = (K >> lane) & 1ull;
                    const int pos = nkept + mbcnt(K);
                    if (kept && pos < max_keep) s_kept[pos] = s0 + s_alive[col];
                    nkept += __popcll(K);
                    if (nkept >= max_keep) { nkept = max_keep; done = true; }
                }
            }
            if (lane == 0) { s_tmp[0] = nkept; s_tmp[1] = done ? 1 : 0; }
        }
        __syncthreads();
        if (s_tmp[1]) break;
    }
    __syncthreads();

    // ---- 5. outputs -----------------------------------------------------------------------------------
    const int nk = s_tmp[0];
    for (int k = tid; k < a.max_keep; k += NT) {
        const long long o = (long long)b * a.max_keep + k;
        if (k < nk) {
            const int slot = sslot[s_kept[k]];
            a.out_boxes[o] = boxes[slot];
            a.out_scores[o] = scores[slot];
            if (a.out_labels) a.out_labels[o] = labels ? (long long)labels[slot] : 0ll;
            if (a.keep_idx) {
                const int seg = slot / a.seg_stride;
                a.keep_idx[o] = (long long)(s_segpre[seg] + (slot - seg * a.seg_stride));
            }
        } else {
            a.out_boxes[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            a.out_scores[o] = 0.0f;
            if (a.out_labels) a.out_labels[o] = 0ll;
            if (a.keep_idx) a.keep_idx[o] = -1ll;
        }
    }
    if (tid == 0) a.out_count[b] = nk;
}

int next_pow2(int v) {
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

extern "C" size_t od_batched_nms_workspace_bytes(int32_t batch, int32_t num_seg, int32_t seg_stride) {
    if (batch < 0 || num_seg < 1 || seg_stride < 1) return 0;
    long long n = (long long)num_seg * seg_stride;
    if (n > MAXN) return 0;
    size_t b = (size_t)batch * 7 * (size_t)next_pow2((int)n) * 4;
    return (b + 255) & ~(size_t)255;
}

extern "C" int od_batched_nms(const float* boxes, const float* scores, const int32_t* labels,
                              const int32_t* seg_count, int32_t num_seg, int32_t seg_stride,
                              int32_t segments_sorted, int32_t batch, double iou_threshold, int32_t mode,
                              int32_t max_keep, int64_t* keep_idx, float* out_boxes, float* out_scores,
                              int64_t* out_labels, int32_t* out_count, void* workspace, size_t workspace_bytes,
                              void* stream) {
    if (batch < 0 || num_seg < 1 || seg_stride < 1 || max_keep < 0) return OD_ERR_ARG;
    if (batch == 0) return OD_OK;
    if (!boxes || !scores || !seg_count || !out_boxes || !out_scores || !out_count || !workspace) return OD_ERR_ARG;
    if (num_seg > MAX_SEG || (long long)num_seg * seg_stride > MAXN) return OD_ERR_LIMIT;
    if (mode < OD_NMS_TRICK || mode > OD_NMS_AUTO_GPU) return OD_ERR_ARG;
    if (((uintptr_t)boxes & 15) || ((uintptr_t)out_boxes & 15)) return OD_ERR_ARG;
    if (workspace_bytes < od_batched_nms_workspace_bytes(batch, num_seg, seg_stride)) return OD_ERR_WORKSPACE;
    NmsArgs a;
    a.boxes = (const float4*)boxes;
    a.scores = scores;
    a.labels = labels;
    a.seg_count = seg_count;
    a.num_seg = num_seg;
    a.seg_stride = seg_stride;
    a.n_slots = num_seg * seg_stride;
    a.n_pad = next_pow2(a.n_slots);
    a.segments_sorted = segments_sorted;
    // (double)ovr > thr  <=>  ovr >= thr_up, thr_up = smallest float whose double value exceeds thr
    float f = (float)iou_threshold;
    if (!((double)f > iou_threshold)) f = __builtin_nextafterf(f, __builtin_inff());
    if (iou_threshold != iou_threshold) f = __builtin_nanf("");   // NaN threshold: nothing suppresses
    a.thr_up = f;
    a.can_skip_zero = (iou_threshold >= 0.0) ? 1 : 0;
    a.mode = mode;
    a.max_keep = max_keep;
    a.ws = (float*)workspace;
    a.keep_idx = (long long*)keep_idx;
    a.out_boxes = (float4*)out_boxes;
    a.out_scores = out_scores;
    a.out_labels = (long long*)out_labels;
    a.out_count = out_count;
    if (max_keep == 0) {
        OD_HIP(hipMemsetAsync(out_count, 0, sizeof(int) * (size_t)batch, (hipStream_t)stream));
        return OD_OK;
    }
    OD_LAUNCH(OD_KERNEL_NMS, (hipStream_t)stream,
              hipLaunchKernelGGL(nms_kernel, dim3((unsigned)batch), dim3(NT), 0, (hipStream_t)stream, a));
    return OD_OK;
}
