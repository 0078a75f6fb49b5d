// Class-batched non-maximum suppression, one 1024-thread workgroup per image.
// Replaces torchvision.ops.boxes.batched_nms (_batched_nms_coordinate_trick / _batched_nms_vanilla),
// the native torchvision::nms op (csrc/ops/cpu/nms_kernel.cpp semantics), `keep[:detections_per_img]`
// and the gather of the kept boxes/scores/labels at the tail of postprocess_detections
// (reference trigger object_detection/train.py:84, object_detection/predict.py:61, and the plain
// nms call at object_detection/predict.py:72).
//
// Algorithm (greedy NMS is inherently ordered; the work around the order is made parallel):
//   1. order: unique keys (ordered score bits << 32 | ~slot) give the stable descending order.
//      Pre-sorted segments (the per-level candidate lists of the pipeline) are merged with a
//      merge-path tree in LDS (O(n) work per round); arbitrary input is bitonic-sorted in LDS.
//   2. the sorted boxes are gathered once into an SoA workspace (coordinate-trick offsets applied).
//   3. candidates are consumed in strips of 256 in sorted order, stopping as soon as max_keep boxes
//      are kept:
//        A. every strip candidate is tested against the boxes kept so far (kept boxes live in LDS
//           and are read as broadcasts; 4 thread groups split the kept list);
//        B. the still-alive candidates are compacted into LDS and their pairwise suppression bits
//           (row i suppresses column j, i < j) are computed by all 16 waves into an LDS bit matrix
//           stored per column;
//        C. one wave resolves the strip 64 candidates at a time: a column is dead if its bit column
//           meets the kept set of earlier blocks; inside a block the greedy result is the fixed
//           point of  kept = alive & ~(exists kept i<j suppressing j), reached by ballot iteration.
//   4. kept boxes are gathered to the outputs in order; rows past the count are zero-filled.
#include "od_common.h"
#include "../../include/od_boxes.h"

#include <mutex>

namespace {

using namespace od;

constexpr int NT = 1024;              // threads per workgroup
constexpr int S = 256;                // strip length
constexpr int SB = S / 64;            // 64-candidate blocks per strip
constexpr int MAXN = OD_MAX_NMS_BOXES;
constexpr int MERGE_MAX = 6144;       // merge-path ordering needs a second key buffer of this many keys
constexpr int KC = 1024;              // kept boxes cached in LDS (later ones are read from the workspace)
constexpr int CW = SB * 2 + 1;        // u32 words per column of the bit matrix (+1 pad against bank conflicts)
constexpr int MAX_SEG = 64;

// LDS carve-up (bytes).  Region B holds the second key buffer while ordering, afterwards the strip state.
constexpr int OFF_A = 0;                               // u64 keys[MAXN]
constexpr int OFF_B = OFF_A + MAXN * 8;                // u64 keys2[MERGE_MAX]  |  strip state:
constexpr int OFF_KBOX = OFF_B;                        //   float4 k_box[KC]
constexpr int OFF_KAR = OFF_KBOX + KC * 16;            //   float   k_ar[KC]
constexpr int OFF_KLAB = OFF_KAR + KC * 4;             //   int     k_lab[KC]
constexpr int OFF_KEPT = OFF_KLAB + KC * 4;            //   u16     kept[MAXN]   (sorted positions)
constexpr int OFF_ABOX = OFF_KEPT + MAXN * 2;          //   float4  a_box[S]     (alive strip candidates)
constexpr int OFF_AAR = OFF_ABOX + S * 16;             //   float   a_ar[S]
constexpr int OFF_ALAB = OFF_AAR + S * 4;              //   int     a_lab[S]
constexpr int OFF_APOS = OFF_ALAB + S * 4;             //   int     a_pos[S]     (sorted position)
constexpr int END_B = OFF_APOS + S * 4;
static_assert(END_B <= OFF_B + MERGE_MAX * 8, "strip state must fit in the second key buffer");
constexpr int OFF_ORDER = OFF_B + MERGE_MAX * 8;       // u16 order[MAXN]
constexpr int OFF_COL = OFF_ORDER + MAXN * 2;          // u32 col[S*CW]
constexpr int OFF_DEAD = OFF_COL + S * CW * 4;         // int dead[S]
constexpr int LDS_BYTES = OFF_DEAD + S * 4;

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
// (bare v_min / v_max on the coordinates: fminf / fmaxf would first quiet every operand read from memory)
__device__ __forceinline__ float intersection(const float4 i, const float4 j) {
    const float xx1 = hw_max(i.x, j.x), yy1 = hw_max(i.y, j.y);
    const float xx2 = hw_min(i.z, j.z), yy2 = hw_min(i.w, j.w);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    return w * h;
}

// ovr = inter / (iarea + jarea - inter);  (double)ovr > thr  <=>  ovr >= thr_up
__device__ __forceinline__ bool over_threshold_exact(float inter, float iarea, float jarea, float thr_up) {
    const float ovr = inter / ((iarea + jarea) - inter);
    return ovr >= thr_up;
}

// Same decision without the IEEE division in the common case: ovr >= thr_up is decided by comparing inter with
// thr_up*(1 +- 1e-5)*den (two multiplications, no reciprocal).  Outside that band the rounding of the products
// (2^-24 relative) cannot change the outcome of the exact division; anything inside it, a product that is not a
// normal finite positive number (degenerate or huge boxes, NaN) and thresholds for which the band is not valid
// take the exact division.  The branch is wave-uniform, so the exact path costs nothing unless some lane needs it.
struct ThrBand { float up, lo, hi; bool usable; };
__device__ __forceinline__ bool over_threshold(float inter, float iarea, float jarea, const ThrBand& t) {
    const float den = (iarea + jarea) - inter;
    const float ph = t.hi * den, pl = t.lo * den;
    const bool above = inter >= ph, below = inter <= pl;
    const bool safe = (pl > 1e-30f) && (ph < 1e38f);
    bool res = above;
    if (__ballot(!(above || below) || !safe || !t.usable) != 0) res = (inter / den) >= t.up;
    return res;
}

#ifdef OD_STAMPS
#define STAMP(k) do { if (threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(NT) void nms_kernel(NmsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LDS_BYTES];
    __shared__ int s_segpre[MAX_SEG + 1];
    __shared__ int s_run[3][MAX_SEG + 1];     // merge rounds: [0] src start, [1] src length, [2] first chunk of pair
    __shared__ int s_tmp[40];
    __shared__ float s_ftmp[NT / 64];
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_raw + OFF_A);
    unsigned long long* s_keys2 = reinterpret_cast<unsigned long long*>(s_raw + OFF_B);
    float4* k_box = reinterpret_cast<float4*>(s_raw + OFF_KBOX);
    float* k_ar = reinterpret_cast<float*>(s_raw + OFF_KAR);
    int* k_lab = reinterpret_cast<int*>(s_raw + OFF_KLAB);
    unsigned short* s_kept = reinterpret_cast<unsigned short*>(s_raw + OFF_KEPT);
    float4* a_box = reinterpret_cast<float4*>(s_raw + OFF_ABOX);
    float* a_ar = reinterpret_cast<float*>(s_raw + OFF_AAR);
    int* a_lab = reinterpret_cast<int*>(s_raw + OFF_ALAB);
    int* a_pos = reinterpret_cast<int*>(s_raw + OFF_APOS);
    unsigned short* s_order = reinterpret_cast<unsigned short*>(s_raw + OFF_ORDER);
    unsigned int* s_col = reinterpret_cast<unsigned int*>(s_raw + OFF_COL);
    int* s_dead = reinterpret_cast<int*>(s_raw + OFF_DEAD);

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
#ifdef OD_STAMPS
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.ws + (long long)gridDim.x * 7 * a.n_pad) + blockIdx.x * 32;
    int stamp_i = 4;
#endif
    STAMP(0);

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
    STAMP(1);

    // ---- 2. order: s_order[rank] = slot, ranks by (score desc, slot asc) --------------------------
    if (a.segments_sorted && a.n_slots <= MERGE_MAX) {
        // merge-path tree over the pre-sorted segments: each round merges adjacent runs pairwise
        unsigned long long* src = s_keys;
        unsigned long long* dst = s_keys2;
        int nruns = a.num_seg;
        if (tid < nruns) { s_run[0][tid] = tid * a.seg_stride; s_run[1][tid] = s_segpre[tid + 1] - s_segpre[tid]; }
        __syncthreads();
        const int CH = (n_valid + NT - 1) / NT > 0 ? (n_valid + NT - 1) / NT : 1;   // outputs per chunk
        while (nruns > 1) {
            const int npairs = (nruns + 1) >> 1;
            if (tid == 0) {
                int chunk = 0;
                for (int p = 0; p < npairs; ++p) {
                    const int la = s_run[1][2 * p];
                    const int lb = (2 * p + 1 < nruns) ? s_run[1][2 * p + 1] : 0;
                    s_run[2][p] = chunk;
                    chunk += (la + lb + CH - 1) / CH;
                }
                s_run[2][npairs] = chunk;
            }
            __syncthreads();
            const int nchunks = s_run[2][npairs];
            for (int ch = tid; ch < nchunks; ch += NT) {
                int p = 0;
                while (p + 1 < npairs && ch >= s_run[2][p + 1]) ++p;
                const int sa = s_run[0][2 * p], na = s_run[1][2 * p];
                const bool hasb = 2 * p + 1 < nruns;
                const int sb = hasb ? s_run[0][2 * p + 1] : 0, nb = hasb ? s_run[1][2 * p + 1] : 0;
                int outbase = 0;
                for (int q = 0; q < p; ++q)
                    outbase += s_run[1][2 * q] + ((2 * q + 1 < nruns) ? s_run[1][2 * q + 1] : 0);
                const int k0 = (ch - s_run[2][p]) * CH;
                const int k1 = (k0 + CH < na + nb) ? k0 + CH : na + nb;
                // co-rank: i = number of A elements among the first k0 outputs
                int lo = k0 - nb > 0 ? k0 - nb : 0, hi = k0 < na ? k0 : na;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (src[sa + mid] > src[sb + (k0 - mid - 1)]) lo = mid + 1; else hi = mid;
                }
                int i = lo, j = k0 - lo;
                for (int k = k0; k < k1; ++k) {
                    const unsigned long long va = (i < na) ? src[sa + i] : 0ull;
                    const unsigned long long vb = (j < nb) ? src[sb + j] : 0ull;
                    const bool takea = (j >= nb) || (i < na && va > vb);
                    dst[outbase + k] = takea ? va : vb;
                    i += takea ? 1 : 0;
                    j += takea ? 0 : 1;
                }
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0;
                for (int p = 0; p < npairs; ++p) {
                    const int len = s_run[1][2 * p] + ((2 * p + 1 < nruns) ? s_run[1][2 * p + 1] : 0);
                    s_run[0][p] = acc;
                    s_run[1][p] = len;
                    acc += len;
                }
            }
            __syncthreads();
            nruns = npairs;
            unsigned long long* t = src; src = dst; dst = t;
        }
        const int start = s_run[0][0];
        for (int p = tid; p < n_valid; p += NT) s_order[p] = (unsigned short)(0xFFFFFFFFu - (uint32_t)src[start + p]);
        __syncthreads();
    } else {
        // bitonic sort of the keys, descending; each thread owns n_pad/2/NT disjoint pairs per stage,
        // loads them all, then stores (the LDS latency is paid once per stage, not once per pair)
        constexpr int PQ = MAXN / 2 / NT;
        for (int k = 2; k <= a.n_pad; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                unsigned long long x[PQ], y[PQ];
                int lo[PQ];
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
                    const int pr = tid + q * NT;                          // pair index
                    lo[q] = ((pr & ~(j - 1)) << 1) | (pr & (j - 1));      // insert a 0 bit at log2(j)
                    const bool on = pr < (a.n_pad >> 1);
                    x[q] = on ? s_keys[lo[q]] : 0ull;
                    y[q] = on ? s_keys[lo[q] | j] : 0ull;
                }
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
                    const int pr = tid + q * NT;
                    const bool desc = (lo[q] & k) == 0;
                    if (pr < (a.n_pad >> 1) && ((x[q] < y[q]) == desc)) { s_keys[lo[q]] = y[q]; s_keys[lo[q] | j] = x[q]; }
                }
                __syncthreads();
            }
        }
        for (int p = tid; p < n_valid; p += NT) s_order[p] = (unsigned short)(0xFFFFFFFFu - (uint32_t)s_keys[p]);
        __syncthreads();
    }
    STAMP(2);

    // ---- 3. gather sorted boxes into the SoA workspace --------------------------------------------
    const float step = maxc + 1.0f;   // offsets = labels.to(boxes) * (boxes.max() + 1)
    for (int p = tid; p < n_valid; p += NT) {
        const int slot = (int)s_order[p];
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
    STAMP(3);

    const int max_keep = a.max_keep < n_valid ? a.max_keep : n_valid;
    ThrBand thr;
    thr.up = a.thr_up;
    thr.lo = a.thr_up * (1.0f - 1e-5f);
    thr.hi = a.thr_up * (1.0f + 1e-5f);
    thr.usable = (a.thr_up > 1e-30f) && (a.thr_up < 1e30f);
    const bool can_skip = a.can_skip_zero != 0;

    // ---- 4. strips ------------------------------------------------------------------------------------
    for (int s0 = 0, SL = S; s0 < n_valid && max_keep > 0; s0 += SL) {
        const int nk = s_tmp[0];
        // strip length: the full 256 while many boxes are still wanted, shorter (a multiple of 64) once
        // only a few are missing (64 or 128), so that the last strip does not test 256 candidates to keep a handful
        {
            const int want = max_keep - nk;
            const int target = want + (want >> 1);
            SL = target <= 64 ? 64 : (target <= 128 ? 128 : S);      // power of two: c = tid & (SL - 1)
        }
        const int parts = NT / SL;
        if (tid < S) s_dead[tid] = 0;
        for (int i = tid; i < S * CW; i += NT) s_col[i] = 0u;
        __syncthreads();

        // -- A: against the kept list ------------------------------------------------------------------
        const int c = tid & (SL - 1);
        const int part = tid / SL;
        const int p = s0 + c;
        const bool valid = p < n_valid;
        const int pp = valid ? p : n_valid - 1;
        const float4 jb = make_float4(sx1[pp], sy1[pp], sx2[pp], sy2[pp]);
        const float jar = sar[pp];
        const int jlab = slab[pp];
        {
            bool dead = false;
            const int nk_lds = nk < KC ? nk : KC;
            for (int e0 = part * 4; e0 < nk_lds; e0 += parts * 4) {
                float inter[4], iar[4];
                bool anypos = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = (e0 + u < nk_lds) ? e0 + u : nk_lds - 1;     // tail repeats the last kept box
                    inter[u] = intersection(k_box[e], jb);
                    iar[u] = k_ar[e];
                    anypos = anypos || (inter[u] > 0.0f);
                }
                if (can_skip && __ballot(anypos) == 0) continue;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = (e0 + u < nk_lds) ? e0 + u : nk_lds - 1;
                    bool sup = over_threshold(inter[u], iar[u], jar, thr);
                    if (vanilla) sup = sup && (k_lab[e] == jlab);
                    dead = dead || sup;
                }
            }
            // kept boxes beyond the LDS cache come from the workspace, four at a time so that their loads overlap
            for (int e0 = KC + part * 4; e0 < nk; e0 += parts * 4) {
                float4 kb[4];
                float kar[4];
                int klab[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kp = s_kept[(e0 + u < nk) ? e0 + u : nk - 1];     // tail repeats the last kept box
                    kb[u] = make_float4(sx1[kp], sy1[kp], sx2[kp], sy2[kp]);
                    kar[u] = sar[kp];
                    klab[u] = slab[kp];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float inter = intersection(kb[u], jb);
                    bool sup = over_threshold_exact(inter, kar[u], jar, thr.up);
                    if (vanilla) sup = sup && (klab[u] == jlab);
                    dead = dead || sup;
                }
            }
            if (dead && valid) s_dead[c] = 1;
        }
        __syncthreads();
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif

        // -- compact the alive candidates of the strip into LDS (order preserved) ---------------------
        bool alive = false;
        int wrank = 0;
        if (tid < SL) {
            alive = valid && (s_dead[tid] == 0);
            const uint64_t bal = __ballot(alive);
            if (lane == 0) s_tmp[8 + wave] = __popcll(bal);
            wrank = mbcnt(bal);
        }
        __syncthreads();
        int na = 0;
        {
            int pre = 0;
#pragma unroll
            for (int w = 0; w < SB; ++w) {
                const int cw = (w * 64 < SL) ? s_tmp[8 + w] : 0;
                if (w < wave) pre += cw;
                na += cw;
            }
            if (alive) {
                const int q = pre + wrank;
                a_box[q] = jb; a_ar[q] = jar; a_lab[q] = jlab; a_pos[q] = p;
            }
        }
        __syncthreads();
        const int nab = (na + 63) >> 6;
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif

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
                const int cq = cvalid ? col : na - 1;
                const float4 cbx = a_box[cq];
                const float car = a_ar[cq];
                const int clab = a_lab[cq];
                const int row0 = rb * 64 + quarter * 16;
                const int rcnt = (na - row0) < 16 ? (na - row0) : 16;
                unsigned int bits = 0u;
                for (int i0 = 0; i0 < rcnt; i0 += 4) {
                    float inter[4], iar[4];
                    bool anypos = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = (row0 + i0 + u < na) ? row0 + i0 + u : na - 1;   // masked below
                        inter[u] = intersection(a_box[r], cbx);
                        iar[u] = a_ar[r];
                        anypos = anypos || (inter[u] > 0.0f);
                    }
                    if (can_skip && __ballot(anypos) == 0) continue;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u;
                        const int r = (row0 + i < na) ? row0 + i : na - 1;
                        bool sup = over_threshold(inter[u], iar[u], car, thr);
                        if (vanilla) sup = sup && (a_lab[r] == clab);
                        sup = sup && (i < rcnt) && (row0 + i < col);
                        bits |= sup ? (1u << i) : 0u;
                    }
                }
                if (cvalid && bits) atomicOr(&s_col[col * CW + rb * 2 + (quarter >> 1)], bits << ((quarter & 1) * 16));
            }
        }
        __syncthreads();
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif

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
                    for (int cc = 0; cc < SB; ++cc) {
                        if (cc < blk) {
                            const unsigned long long m = ((unsigned long long)cw[2 * cc + 1] << 32) | cw[2 * cc];
                            dead = dead || ((m & keptw[cc]) != 0ull);
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
                    if (kept && pos < max_keep) {
                        s_kept[pos] = (unsigned short)a_pos[col];
                        if (pos < KC) { k_box[pos] = a_box[col]; k_ar[pos] = a_ar[col]; k_lab[pos] = a_lab[col]; }
                    }
                    nkept += __popcll(K);
                    if (nkept >= max_keep) { nkept = max_keep; done = true; }
                }
            }
            if (lane == 0) { s_tmp[0] = nkept; s_tmp[1] = done ? 1 : 0; }
        }
        __syncthreads();
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif
        if (s_tmp[1]) break;
    }
    __syncthreads();
    STAMP(30);

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
    STAMP(31);
}

// one thread per padded detection slot: six floats of the message row; thread 0 of an image appends the count
__global__ __launch_bounds__(256) void pack_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                   const long long* __restrict__ labels, const int* __restrict__ counts,
                                                   int max_det, float* __restrict__ message) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    float* row = message + (long long)b * (6ll * max_det + 1);
    if (k < max_det) {
        const long long i = (long long)b * max_det + k;
        const float4 bx = boxes[i];
        float* o = row + 6ll * k;
        o[0] = bx.x; o[1] = bx.y; o[2] = bx.z; o[3] = bx.w;
        o[4] = scores[i];
        o[5] = (float)labels[i];
    }
    if (k == 0) row[6ll * max_det] = (float)counts[b];
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
    size_t b = (size_t)batch * 7 * (size_t)next_pow2((int)n) * 4 + (size_t)batch * 256;   // + debug stamps
    return (b + 255) & ~(size_t)255;
}

static int batched_nms_launch(const float* boxes, const float* scores, const int32_t* labels,
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
    OD_LAUNCH_KERNEL(OD_KERNEL_NMS, nms_kernel, dim3((unsigned)batch), dim3(NT), 0, (hipStream_t)stream, a);
    return OD_OK;
}

extern "C" int od_batched_nms(const float* boxes, const float* scores, const int32_t* labels,
                              const int32_t* seg_count, int32_t num_seg, int32_t seg_stride,
                              int32_t segments_sorted, int32_t batch, double iou_threshold, int32_t mode,
                              int32_t max_keep, int64_t* keep_idx, float* out_boxes, float* out_scores,
                              int64_t* out_labels, int32_t* out_count, void* workspace, size_t workspace_bytes,
                              void* stream) {
    return batched_nms_launch(boxes, scores, labels, seg_count, num_seg, seg_stride, segments_sorted, batch,
                              iou_threshold, mode, max_keep, keep_idx, out_boxes, out_scores, out_labels, out_count,
                              workspace, workspace_bytes, stream);
}

// ---- od_batched_nms for callers that cannot allocate ---------------------------------------------------------
// The scratch memory of the NMS kernels (the sorted boxes of every image in SoA form) lives in a device array that
// is part of the library's code object, so the entry point needs no workspace argument and allocates nothing.
// Every GPU holds its own copy of the array at its own address: the address is looked up once per device
// (hipGetSymbolAddress with that device current) and host code only ever uses that pointer -- the array's name is
// not a device address on the host side.  All calls on one GPU share the array: calls that overlap in time on one
// GPU (different streams, different threads) race.  Nothing in it has to start at zero: the kernels write every
// element they later read.
constexpr size_t SCRATCH_BYTES = OD_NMS_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[MAX_DEVICES];

// Device address of `scratch` on the current device (looked up on the first call per device).
static int scratch_pointer(void** out) {
    int dev = -1;
    OD_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return OD_ERR_LIMIT;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        OD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (p == nullptr) return OD_ERR_ARG;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return OD_OK;
}

extern "C" int od_batched_nms_noalloc(const float* boxes, const float* scores, const int32_t* labels,
                                      const int32_t* seg_count, int32_t num_seg, int32_t seg_stride,
                                      int32_t segments_sorted, int32_t batch, double iou_threshold, int32_t mode,
                                      int32_t max_keep, int64_t* keep_idx, float* out_boxes, float* out_scores,
                                      int64_t* out_labels, int32_t* out_count, void* stream) {
    if (batch < 0 || num_seg < 1 || seg_stride < 1 || max_keep < 0) return OD_ERR_ARG;
    if (batch == 0) return OD_OK;
    const size_t need = od_batched_nms_workspace_bytes(batch, num_seg, seg_stride);
    if (need == 0 || need > SCRATCH_BYTES) return OD_ERR_LIMIT;      // beyond the built-in scratch array
    void* ws = nullptr;
    const int rc = scratch_pointer(&ws);
    if (rc != OD_OK) return rc;
    return batched_nms_launch(boxes, scores, labels, seg_count, num_seg, seg_stride, segments_sorted, batch,
                              iou_threshold, mode, max_keep, keep_idx, out_boxes, out_scores, out_labels, out_count,
                              ws, SCRATCH_BYTES, stream);
}

extern "C" int od_pack_detections(const float* boxes, const float* scores, const int64_t* labels, const int32_t* counts,
                                  int32_t batch, int32_t max_det, float* message, void* stream) {
    if (batch < 0 || max_det < 0) return OD_ERR_ARG;
    if (batch == 0) return OD_OK;
    if (!counts || !message || (max_det > 0 && (!boxes || !scores || !labels))) return OD_ERR_ARG;
    if (batch > 65535) return OD_ERR_LIMIT;
    if ((uintptr_t)boxes & 15) return OD_ERR_ARG;
    const unsigned gx = (unsigned)((max_det + 255) / 256);
    OD_LAUNCH_KERNEL(OD_KERNEL_PACK, pack_kernel, dim3(gx > 0 ? gx : 1u, (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)boxes, scores, (const long long*)labels, counts, max_det, message);
    return OD_OK;
}
