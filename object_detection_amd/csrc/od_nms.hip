// Class-batched non-maximum suppression, one 1024-thread workgroup per image.
// Replaces torchvision.ops.boxes.batched_nms (_batched_nms_coordinate_trick / _batched_nms_vanilla),
// the native torchvision::nms op (csrc/ops/cpu/nms_kernel.cpp or csrc/ops/cuda/nms_kernel.cu semantics),
// `keep[:detections_per_img]` and the gather of the kept boxes/scores/labels at the tail of
// postprocess_detections (reference trigger object_detection/train.py:84, object_detection/predict.py:61, and the
// plain nms call at object_detection/predict.py:72).
//
// Greedy NMS is ordered by score; the kernel consumes the candidates in that order in strips of up to 256 and
// stops as soon as max_keep boxes are kept, so only the head of the order is ever produced:
//   1. order, lazily.  The pipeline hands over one pre-sorted segment per pyramid level.  The next strip of the
//      global order lies inside the next 256 entries of every segment: those windows are loaded into LDS, every
//      window entry finds its rank among all windows by binary search in the other windows, and the entries of
//      rank < strip length form the strip.  No global sort or merge of all candidates is ever made.  Unsorted
//      input (or more than 8 segments) is ranked beforehand by nms_order_kernel and read as one sorted segment.
//   2. classes never interact (vanilla mode by definition; coordinate-trick mode because the offsets
//      label * (max + 1) move the classes to disjoint coordinate ranges -- verified per image, see `independent`).
//      A strip is therefore re-ordered by (class, rank); waves then hold candidates of one class, and
//        A. every candidate is tested against the boxes kept so far, which sit in LDS grouped by strip and class:
//           a group of four kept boxes without the wave's class is skipped after one label comparison;
//        B. the candidates still alive are compacted and each tests only the earlier alive candidates of its own
//           class (suppression bits into an LDS bit matrix, one column per candidate);
//        C. one wave resolves the strip 64 columns at a time: a column is dead if its bit column meets the kept
//           set of earlier blocks; inside a block the greedy result is the fixed point of
//           kept = alive & ~(exists kept i<j suppressing j), reached by ballot iteration;
//      the newly kept boxes are appended to the kept list in class order and to the output list in score order.
//   3. kept boxes are gathered to the outputs in order; rows past the count are zero-filled.
#include "od_common.h"
#include "../../include/od_boxes.h"

#include <mutex>
#include <stdlib.h>
#include <type_traits>

namespace {

using namespace od;

constexpr int NT = 1024;              // threads per workgroup
constexpr int S = 256;                // strip length
static_assert(S == 256, "a merge of the ordering tree is one wave x 4 keys per lane; window entries are decoded with >> 8");
constexpr int SB = S / 64;            // 64-candidate blocks per strip
constexpr int MAXN = OD_MAX_NMS_BOXES;
constexpr int WSEG = 8;               // pre-sorted segments that are merged lazily (more: ranked beforehand)
constexpr int KC = 1024;              // kept boxes cached in LDS (later ones are read through the workspace)
constexpr int CW = SB * 2 + 1;        // u32 words per column of the bit matrix (+1 pad against bank conflicts)
constexpr int MAX_SEG = 64;
constexpr int NMS_MODE_MASK = 0xff;
constexpr int RK = 896;               // candidates at the head of the order that are ranked in one go (a multiple of 64; with the
                                      // 16 * WSEG entries behind the last samples one pass of the workgroup ranks them) ...
constexpr int RSEG = 1024;            // ... when the (at most WSEG) sorted segments hold at most this many entries each
static_assert(WSEG * RSEG * 4 >= (WSEG * S + (WSEG / 2) * S) * 8 + WSEG * S * 4, "the window arrays share the memory of the segment keys");
static_assert(RK % 64 == 0 && RK + 16 * WSEG <= NT, "strips start at multiples of 64; one pass of the workgroup ranks the head");

struct NmsArgs {
    const float4* boxes;
    const float* scores;
    const int* labels;
    const int* seg_count;
    int num_seg, seg_stride, n_slots;
    int lazy;              // 1: the segments are sorted and merged strip by strip (2: the head of the order is ranked in one go first); 0: `order` holds the global order
    float thr_up;          // suppress when ovr >= thr_up (the threshold rule folded into one float, see od_batched_nms)
    int can_skip_zero;     // iou_threshold >= 0: a pair with zero intersection can never suppress
    int mode, max_keep;
    int* order;            // [batch, n_slots] slots in (score desc, slot asc) order          (lazy == 0)
    int* kept_out;         // [batch, n_slots] kept slots in output order                      (max_keep > KC)
    int* kept_cls;         // [batch, n_slots] kept slots in kept-list order beyond the LDS cache (max_keep > KC)
    unsigned long long* stamps;
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

// ovr = inter / (iarea + jarea - inter);  suppressed  <=>  ovr >= thr_up
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
#define STAMP(k) do { const int k__ = (k); if (threadIdx.x == 0 && k__ < 32) stamps[k__] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long nms_key(float score, int slot) {
    return ((unsigned long long)ordered_key(score) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)slot);
}

// number of keys of the descending-sorted arr[0, len) that are greater than `key`
__device__ __forceinline__ int count_greater(const unsigned long long* arr, int len, unsigned long long key) {
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (arr[mid] > key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(NT) void nms_kernel(NmsArgs a) {
    // 32 KB used in two ways: while the head of the order is ranked in one go (see RK), the score keys of up to WSEG
    // segments of up to RSEG entries; afterwards the windows of the lazy merge:
    __shared__ unsigned long long s_raw[WSEG * RSEG / 2];
    unsigned long long* const s_wk = s_raw;                          // [WSEG * S] the next S keys of every segment: score key << 32 | ~window entry (0: past the end)
    unsigned long long* const s_mk = s_raw + WSEG * S;               // [WSEG / 2 * S] second buffer of the merge tree
    int* const s_wslot = reinterpret_cast<int*>(s_raw + WSEG * S + (WSEG / 2) * S);   // [WSEG * S] the window entries' slots
    uint32_t* const s_key = reinterpret_cast<uint32_t*>(s_raw);      // [WSEG * RSEG] score keys of whole segments (0: past the end)
    __shared__ int s_order[RK];                         // slot of the r-th candidate of the image, r < RK
    __shared__ int s_olab[RK];                          // its label
    __shared__ __attribute__((aligned(16))) float4 s_obox[RK];       // its box
    __shared__ int s_cut[WSEG];                         // entries at the head of every segment that may rank below RK
    __shared__ int s_wlab[WSEG * S];                    // their labels
    __shared__ __attribute__((aligned(16))) float4 s_sbox[S];   // box of the strip's r-th candidate (as given)
    __shared__ int s_sraw[S];                           // its label as given
    __shared__ __attribute__((aligned(16))) float4 k_box[KC];   // kept boxes in kept-list order (strip, class, rank)
    __shared__ float k_ar[KC];
    __shared__ __attribute__((aligned(16))) int k_lab[KC];
    __shared__ int s_kept_lds[KC];                      // kept slots in output order (max_keep <= KC)
    __shared__ __attribute__((aligned(16))) float4 a_box[S];    // alive candidates of the strip, class order
    __shared__ float a_ar[S];
    __shared__ int a_lab[S];
    __shared__ int a_slot[S];
    __shared__ int a_rank[S];                           // rank of the alive candidate inside the strip (score order)
    __shared__ unsigned int s_col[S * CW];
    __shared__ int s_strip[S];                          // slot of the strip's r-th candidate (score order)
    __shared__ __attribute__((aligned(16))) uint32_t s_ckey[S];   // its (class, rank) key
    __shared__ int s_crank[S];                          // its position in (class, rank) order
    __shared__ int s_cpos[S];                           // inverse: rank of the candidate at class-order position c
    __shared__ int s_flag[S];                           // A: dead; later: kept, by rank
    __shared__ int s_segpre[MAX_SEG + 1];
    __shared__ int s_head[WSEG], s_taken[WSEG];
    __shared__ unsigned long long s_keptw[SB];
    __shared__ int s_tmp[40];
    __shared__ float s_ftmp[2 * (NT / 64)];
    __shared__ int s_itmp[2 * (NT / 64)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float4* boxes = a.boxes + (long long)b * a.n_slots;
    const float* scores = a.scores + (long long)b * a.n_slots;
    const int* labels = a.labels ? a.labels + (long long)b * a.n_slots : nullptr;
    const int* segc = a.seg_count + (long long)b * a.num_seg;
    const int* order = a.order ? a.order + (long long)b * a.n_slots : nullptr;
    int* kept_out = (a.max_keep > KC) ? a.kept_out + (long long)b * a.n_slots : s_kept_lds;
    int* kept_cls = (a.max_keep > KC) ? a.kept_cls + (long long)b * a.n_slots : nullptr;
#ifdef OD_STAMPS
    unsigned long long* stamps = a.stamps + blockIdx.x * 32;
    int stamp_i = 4;
#endif
    STAMP(0);

    // ---- 1. valid count --------------------------------------------------------------------------------------
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
    // ---- 2. segment heads; the first windows are requested before anything waits for them ----------------------
    const int nseg = a.lazy ? a.num_seg : 1;
    if (tid < WSEG) { s_head[tid] = 0; s_taken[tid] = 0; }
    if (tid == 0) { s_tmp[0] = 0; s_tmp[1] = 0; }   // [0] = kept count, [1] = done flag
    // Segments that break the caller's promise of being sorted make ranks collide, which leaves entries of these two
    // tables unwritten; starting from valid values keeps every later index inside its array (the result is then
    // unspecified, but nothing is read or written out of bounds).
    if (tid < S) { s_strip[tid] = 0; s_cpos[tid] = 0; }
    __syncthreads();
    const int n_valid = s_segpre[a.num_seg];

    // Window entry e = segment * S + i of the strip that starts at the current heads.  A thread owns the entries
    // tid and tid + NT; their scores and labels are requested one strip ahead (the loads are in flight while the
    // previous strip is processed) and written to LDS at the top of the strip.
    float pf_score[2];
    int pf_slot[2], pf_lab[2];
    auto prefetch_windows = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * NT;
            pf_slot[u] = -1;
            pf_score[u] = 0.0f;
            pf_lab[u] = 0;
            if (e < nseg * S) {
                const int l = e >> 8, i = e & (S - 1);
                const int len = a.lazy ? (s_segpre[l + 1] - s_segpre[l]) : n_valid;
                const int pos = s_head[l] + i;
                if (pos < len) {
                    const int slot = a.lazy ? l * a.seg_stride + pos : order[pos];
                    pf_slot[u] = slot;
                    pf_score[u] = scores[slot];
                    if (labels) pf_lab[u] = labels[slot];
                }
            }
        }
    };
    // The head of the order in one go.  The pipeline hands over up to WSEG sorted segments of up to RSEG entries (one per
    // pyramid level, topk_candidates each) and stops after a few hundred candidates: the first RK candidates of the
    // image are ranked once -- all score keys in LDS, rank of an entry = its position + the entries of the other
    // segments that precede it, found by binary searches that advance together (one LDS round trip per level); ranks
    // grow along a segment, so an entry whose preceding sample (every 16th entry is ranked first) already has rank >=
    // RK is skipped -- and their boxes and labels are brought into LDS.  The strips of the first RK candidates then
    // need no memory access and no ordering step (ordering every strip of 256 from windows of the segments, which is
    // what happens beyond RK and for longer segments, took a third of the kernel: 5-6 us per strip).
    const bool ranked = a.lazy == 2 && nseg > 1 && a.seg_stride <= RSEG;
    const int ranked_n = ranked ? (n_valid < RK ? n_valid : RK) : 0;
    if (ranked) {
        static_assert(RSEG == NT, "thread t holds entry t of every segment");
        float sc[WSEG];                              // all requests first: one memory round trip, not one per segment
        bool ok[WSEG];
#pragma unroll
        for (int l = 0; l < WSEG; ++l) {
            ok[l] = l < nseg && tid < s_segpre[l + 1] - s_segpre[l];
            sc[l] = ok[l] ? scores[l * a.seg_stride + tid] : 0.0f;
        }
#pragma unroll
        for (int l = 0; l < WSEG; ++l) {
            const uint32_t sk = ordered_key(sc[l]);
            // a valid entry: at least 1 (only the two lowest NaN patterns share a key)
            s_key[l * RSEG + tid] = ok[l] ? (sk > 1u ? sk : 1u) : 0u;
        }
    } else {
        prefetch_windows();
    }

    int mode = a.mode & NMS_MODE_MASK;
    if (mode == OD_NMS_AUTO_CPU) mode = (4 * n_valid > 4000) ? OD_NMS_VANILLA : OD_NMS_TRICK;
    if (mode == OD_NMS_AUTO_GPU) mode = (4 * n_valid > 20000) ? OD_NMS_VANILLA : OD_NMS_TRICK;
    const bool vanilla = (mode == OD_NMS_VANILLA) && labels != nullptr;
    const bool trick = (mode == OD_NMS_TRICK) && labels != nullptr;
    float step = 0.0f;                // offsets = labels.to(boxes) * (boxes.max() + 1)
    bool independent = vanilla;       // boxes of different classes never suppress each other
    bool one_class = labels == nullptr;
    int label_base = 0;               // smallest label of the image
    bool packable = false;            // label - label_base fits in 23 bits: (class, rank) sorts as one 32-bit key
    if (labels) {
        float maxc = -__builtin_inff(), minc = __builtin_inff();
        int maxl = (int)0x80000000, minl = 0x7fffffff;
        for (int i0 = tid; i0 < a.n_slots; i0 += 4 * NT) {       // the loads of four slots are in flight together
            float4 bx[4];
            int lab[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * NT;
                const int ic = i < a.n_slots ? i : 0;
                const int seg = ic / a.seg_stride;
                const int r = ic - seg * a.seg_stride;
                ok[u] = i < a.n_slots && r < s_segpre[seg + 1] - s_segpre[seg];
                bx[u] = boxes[ic];
                lab[u] = labels[ic];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ok[u]) {
                    maxc = fmaxf(maxc, fmaxf(fmaxf(bx[u].x, bx[u].y), fmaxf(bx[u].z, bx[u].w)));
                    minc = fminf(minc, fminf(fminf(bx[u].x, bx[u].y), fminf(bx[u].z, bx[u].w)));
                    maxl = lab[u] > maxl ? lab[u] : maxl;
                    minl = lab[u] < minl ? lab[u] : minl;
                }
            }
        }
        maxc = wave_fmax_all(maxc);
        minc = wave_fmin_all(minc);
        maxl = (int)(wave_umax_all((uint32_t)maxl ^ 0x80000000u) ^ 0x80000000u);
        minl = (int)(~wave_umax_all(~((uint32_t)minl ^ 0x80000000u)) ^ 0x80000000u);
        if (lane == 0) { s_ftmp[wave] = maxc; s_ftmp[NT / 64 + wave] = minc; s_itmp[wave] = maxl; s_itmp[NT / 64 + wave] = minl; }
        __syncthreads();
        maxc = s_ftmp[0]; minc = s_ftmp[NT / 64]; maxl = s_itmp[0]; minl = s_itmp[NT / 64];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) {
            maxc = fmaxf(maxc, s_ftmp[w]);
            minc = fminf(minc, s_ftmp[NT / 64 + w]);
            maxl = s_itmp[w] > maxl ? s_itmp[w] : maxl;
            minl = s_itmp[NT / 64 + w] < minl ? s_itmp[NT / 64 + w] : minl;
        }
        step = maxc + 1.0f;
        // A box of class c ends at or below max + c*step, a box of class c' > c starts at or above c'*step; with
        // max + c*step <= c'*step after rounding (ensured by 2*c'*step*2^-24 <= 1), non-negative coordinates and
        // labels, the two cannot intersect, their overlap is 0 (or NaN) and never exceeds a threshold >= 0.
        if (trick) independent = a.can_skip_zero && minc >= 0.0f && minl >= 0 && ((float)maxl * step <= 8388608.0f);
        one_class = (minl == maxl);
        label_base = minl;
        packable = ((uint32_t)maxl - (uint32_t)minl) < (1u << 23);
    }
    STAMP(1);

    if (ranked) {
        static_assert(RSEG == 1024, "entries are decoded with >> 10");
        __syncthreads();                                 // the keys are in place
        // Entry (segment l, position i, score key k) is preceded by the entries of segment o with a greater score key
        // and, for o < l, by those with the same key (equal scores: lower slot first, and slots grow with the segment).
        auto segment_rank = [&](int l, int i, uint32_t k) -> int {
            int cnt[WSEG];
            uint32_t bar[WSEG];                          // entries above bar[o] precede (k >= 1)
#pragma unroll
            for (int o = 0; o < WSEG; ++o) { cnt[o] = 0; bar[o] = k - (o < l ? 1u : 0u); }
            for (int st = RSEG >> 1; st >= 1; st >>= 1) {
                uint32_t v[WSEG];
#pragma unroll
                for (int o = 0; o < WSEG; ++o) v[o] = o < nseg ? s_key[o * RSEG + cnt[o] + st - 1] : 0u;
#pragma unroll
                for (int o = 0; o < WSEG; ++o) cnt[o] += (v[o] > bar[o]) ? st : 0;
            }
            int rank = i;
#pragma unroll
            for (int o = 0; o < WSEG; ++o) {             // the search stops at RSEG - 1: the last entry decides
                cnt[o] += (cnt[o] == RSEG - 1 && s_key[o * RSEG + RSEG - 1] > bar[o]) ? 1 : 0;
                rank += (o != l) ? cnt[o] : 0;
            }
            return rank;
        };
        if (tid < nseg * (RSEG / 16)) {
            const int l = tid >> 6, i = ((tid & 63) << 4) | 15;
            const uint32_t key = s_key[l * RSEG + i];
            const int sr = key != 0u ? segment_rank(l, i, key) : 0x7fffffff;
            // ranks grow along a segment: the samples below RK are a prefix, and only the entries up to the block of
            // 16 behind the last of them can have a rank below RK
            const int below = __popcll(__ballot(sr < RK));          // wave l holds the samples of segment l
            if (lane == 0) {
                const int len = s_segpre[l + 1] - s_segpre[l];
                const int cut = 16 * (below + 1);
                s_cut[l] = cut < len ? cut : len;
            }
        }
        __syncthreads();
        // the entries to rank, dealt to all threads: entry w of the concatenated heads of the segments
        int cpre[WSEG + 1];
        cpre[0] = 0;
#pragma unroll
        for (int o = 0; o < WSEG; ++o) cpre[o + 1] = cpre[o] + (o < nseg ? s_cut[o] : 0);
        for (int w0 = 0; w0 < cpre[WSEG]; w0 += NT) {
            const int w = w0 + tid;
            const bool on = w < cpre[WSEG];
            int l = 0;
#pragma unroll
            for (int o = 1; o < WSEG; ++o) l += (on && w >= cpre[o]) ? 1 : 0;
            int base = 0;
#pragma unroll
            for (int o = 0; o < WSEG; ++o) base = (o == l) ? cpre[o] : base;
            const int i = on ? w - base : 0;
            const int rank = on ? segment_rank(l, i, s_key[l * RSEG + i]) : RK;
            const bool in = rank < RK;
            if (in) s_order[rank] = l * a.seg_stride + i;
#pragma unroll
            for (int o = 0; o < WSEG; ++o) {                        // entries consumed from every segment
                const unsigned long long m = __ballot(in && l == o);
                if (lane == 0 && m != 0ull) atomicAdd(&s_head[o], __popcll(m));
            }
        }
        __syncthreads();
        if (tid < ranked_n) {
            const int slot = s_order[tid];
            s_obox[tid] = boxes[slot];
            s_olab[tid] = labels ? labels[slot] : 0;
        }
        // (the first strip reads them after its own barriers)
    }

    const int max_keep = a.max_keep < n_valid ? a.max_keep : n_valid;
    ThrBand thr;
    thr.up = a.thr_up;
    thr.lo = a.thr_up * (1.0f - 1e-5f);
    thr.hi = a.thr_up * (1.0f + 1e-5f);
    thr.usable = (a.thr_up > 1e-30f) && (a.thr_up < 1e30f);
    const bool can_skip = a.can_skip_zero != 0;
    const bool by_class = independent && !one_class;   // only boxes of equal label can suppress each other
    const bool group = by_class && packable;           // candidates are grouped by class inside a strip

    // ---- 3. strips ------------------------------------------------------------------------------------------
    int nk_cls = 0;                                    // entries of the kept list (== kept count until the cap binds)
    for (int s0 = 0, SL = S; s0 < n_valid && max_keep > 0; s0 += SL) {
        const int nk = s_tmp[0];
        // strip length: the full 256 while many boxes are still wanted, shorter (64 or 128) once only a few are
        // missing, so that the last strip does not test 256 candidates to keep a handful
        {
            const int want = max_keep - nk;
            const int target = want + (want >> 1);
            SL = target <= 64 ? 64 : (target <= 128 ? 128 : S);      // power of two
        }
        const bool from_order = s0 < ranked_n;               // this strip lies in the head that was ranked in one go
        while (from_order && s0 + SL > ranked_n && ranked_n < n_valid) SL >>= 1;   // ... entirely (s0 and RK are multiples of 64)
        const int slog = SL == 64 ? 6 : (SL == 128 ? 7 : 8);
        const int ns = (n_valid - s0) < SL ? (n_valid - s0) : SL;    // candidates of this strip
        const int parts = NT / SL;

        if (from_order) {
            for (int i = tid; i < S * CW; i += NT) s_col[i] = 0u;
            if (tid < S) {
                s_flag[tid] = 0; s_crank[tid] = 0; s_ckey[tid] = 0xFFFFFFFFu;
                if (tid < ns) {
                    const int lab = s_olab[s0 + tid];
                    s_strip[tid] = s_order[s0 + tid];
                    s_sraw[tid] = lab;
                    if (group) s_ckey[tid] = ((uint32_t)(lab - label_base) << 8) | (uint32_t)tid;
                }
            }
            __syncthreads();
        } else {
        // -- windows: the next S keys of every segment (requested during the previous strip) -------------------
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * NT;
            if (e < WSEG * S) {
                // score key of a valid entry: at least 1 (only the two lowest NaN patterns share a key).  Equal scores:
                // the lower window entry first, i.e. the lower slot (slots grow with the segment and the position).
                const uint32_t sk = ordered_key(pf_score[u]);
                const bool ok = e < nseg * S && pf_slot[u] >= 0;
                s_wk[e] = ok ? (((unsigned long long)(sk > 1u ? sk : 1u) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)e)) : 0ull;
                if (e < nseg * S) { s_wslot[e] = pf_slot[u]; s_wlab[e] = pf_lab[u]; }
            }
        }
        for (int i = tid; i < S * CW; i += NT) s_col[i] = 0u;
        if (tid < S) { s_flag[tid] = 0; s_crank[tid] = 0; s_ckey[tid] = 0xFFFFFFFFu; }
        __syncthreads();
#ifdef OD_STAMPS_FINE
        if (stamp_i < 30) STAMP(stamp_i++);
#endif
        // -- the S best keys of all windows, in order: a tree of merges, one wave per pair of sorted runs ----------
        // max(A[i], B[S-1-i]) holds the S largest keys of two descending runs A and B as a bitonic sequence, which
        // eight compare-exchange stages sort; a lane holds four consecutive elements, so the two finest stages stay
        // in its registers and the other six exchange with lanes 1 ... 32 away (DPP moves and shuffles): no barrier
        // inside a merge, one between the levels of the tree.  (Ranking every window entry by binary searches in the
        // other windows, as this kernel did before, took 5-6 us per strip: 50 dependent LDS reads per entry.)
        const unsigned long long* top = s_wk;                 // one segment: its window is the order
        {
            unsigned long long* src = s_wk;
            unsigned long long* dst = s_mk;
            for (int runs = nseg; runs > 1; runs = (runs + 1) >> 1) {
                const int pairs = (runs + 1) >> 1;
                if (wave < pairs) {
                    const unsigned long long* A = src + (2 * wave) * S;
                    const bool has_b = 2 * wave + 1 < runs;
                    const unsigned long long* B = src + (2 * wave + 1) * S;
                    unsigned long long v[4];
                    const int base = lane * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned long long x = A[base + e];
                        const unsigned long long y = has_b ? B[S - 1 - base - e] : 0ull;
                        v[e] = x > y ? x : y;
                    }
                    if (has_b) {
                        bitonic_lane_stage<32, 4>(v, base, S);
                        bitonic_lane_stage<16, 4>(v, base, S);
                        bitonic_lane_stage<8, 4>(v, base, S);
                        bitonic_lane_stage<4, 4>(v, base, S);
                        bitonic_lane_stage<2, 4>(v, base, S);
                        bitonic_lane_stage<1, 4>(v, base, S);
#pragma unroll
                        for (int jj = 2; jj >= 1; jj >>= 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if ((e & jj) == 0) {
                                    const unsigned long long x = v[e], y = v[e | jj];
                                    if (x < y) { v[e] = y; v[e | jj] = x; }
                                }
                            }
                        }
                    }
                    unsigned long long* D = dst + wave * S;
#pragma unroll
                    for (int e = 0; e < 4; ++e) D[base + e] = v[e];
                }
                __syncthreads();
                top = dst;
                unsigned long long* t = src; src = dst; dst = t;
            }
        }
#ifdef OD_STAMPS_FINE
        if (stamp_i < 30) STAMP(stamp_i++);
#endif
        if (tid < S) {
            const unsigned long long key = tid < ns ? top[tid] : 0ull;
            const bool ok = (key >> 32) != 0ull;
            const int e = ok ? (int)(0xFFFFFFFFu - (uint32_t)key) : 0;
            if (ok) {
                const int lab = s_wlab[e];
                s_strip[tid] = s_wslot[e];
                s_sraw[tid] = lab;
                if (group) s_ckey[tid] = ((uint32_t)(lab - label_base) << 8) | (uint32_t)tid;
            }
            // entries taken from every window: they are a prefix of it
#pragma unroll
            for (int l = 0; l < WSEG; ++l) {
                const unsigned long long m = __ballot(ok && (e >> 8) == l);
                if (lane == 0 && m != 0ull) atomicAdd(&s_taken[l], __popcll(m));
            }
        }
        __syncthreads();
#ifdef OD_STAMPS_FINE
        if (stamp_i < 30) STAMP(stamp_i++);
#endif
        if (tid < nseg) { s_head[tid] += s_taken[tid]; s_taken[tid] = 0; }
        }   // (windows)
        // the boxes of the strip are requested now and stored after the next loop
        float4 my_box = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < ns) my_box = from_order ? s_obox[s0 + tid] : boxes[s_strip[tid]];
        // -- position of every candidate in (class, rank) order: number of smaller (class, rank) keys --------------
        if (group) {
            const int r = tid & (S - 1), quarter = tid >> 8;
            const uint32_t mine = s_ckey[r];
            const uint2* keys2 = reinterpret_cast<const uint2*>(s_ckey + quarter * (SL >> 2));
            int cnt = 0;
#pragma unroll 8
            for (int k = 0; k < (SL >> 3); ++k) {
                const uint2 kk = keys2[k];
                cnt += (kk.x < mine ? 1 : 0) + (kk.y < mine ? 1 : 0);
            }
            if (r < ns && cnt) atomicAdd(&s_crank[r], cnt);
        }
        if (tid < ns) s_sbox[tid] = my_box;
        __syncthreads();
#ifdef OD_STAMPS_FINE
        if (stamp_i < 30) STAMP(stamp_i++);
#endif
        // for the next strip: the heads are final (inside the ranked head of the order there is nothing to request;
        // its last strip requests the first windows, from the heads the ranking counted)
        if (!from_order || (s0 + SL >= ranked_n && ranked_n < n_valid)) prefetch_windows();
        if (tid < ns) s_cpos[group ? s_crank[tid] : tid] = tid;
        __syncthreads();
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif

        // -- A: against the kept list ------------------------------------------------------------------------------
        const int c = tid & (SL - 1);
        const int part = tid >> slog;
        const bool valid = c < ns;
        const int rnk = s_cpos[valid ? c : 0];
        const int slot = s_strip[rnk];
        const int jraw = s_sraw[rnk];
        const int jlab = by_class ? jraw : 0;                // labels are compared only when classes cannot interact
        float4 jb = s_sbox[rnk];
        if (trick) {
            const float off = (float)jraw * step;
            jb.x += off; jb.y += off; jb.z += off; jb.w += off;
        }
        const float jar = (jb.z - jb.x) * (jb.w - jb.y);
        {
            bool dead = false;
            const int nk_lds = nk_cls < KC ? nk_cls : KC;
            for (int e0 = part * 4; e0 < nk_lds; e0 += parts * 4) {
                const int4 kl = *reinterpret_cast<const int4*>(&k_lab[e0]);
                const int kls[4] = {kl.x, kl.y, kl.z, kl.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool same = (e0 + u < nk_lds) && (kls[u] == jlab);
                    if (__ballot(same) == 0ull) continue;                     // no candidate of this wave has that class
                    const float inter = intersection(k_box[e0 + u], jb);
                    if (can_skip && __ballot(same && inter > 0.0f) == 0ull) continue;
                    const bool sup = over_threshold(inter, k_ar[e0 + u], jar, thr);
                    dead = dead || (sup && same);
                }
            }
            // kept boxes beyond the LDS cache come from memory, four at a time so that their loads overlap
            for (int e0 = KC + part * 4; e0 < nk_cls; e0 += parts * 4) {
                float4 kb[4];
                int klab[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ks = kept_cls[(e0 + u < nk_cls) ? e0 + u : nk_cls - 1];     // tail repeats the last kept box
                    kb[u] = boxes[ks];
                    const int raw = labels ? labels[ks] : 0;
                    klab[u] = by_class ? raw : 0;
                    if (trick) {
                        const float off = (float)raw * step;
                        kb[u].x += off; kb[u].y += off; kb[u].z += off; kb[u].w += off;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float inter = intersection(kb[u], jb);
                    const float kar = (kb[u].z - kb[u].x) * (kb[u].w - kb[u].y);
                    const bool sup = over_threshold_exact(inter, kar, jar, thr.up);
                    dead = dead || (sup && klab[u] == jlab);
                }
            }
            if (dead && valid) s_flag[c] = 1;
        }
        __syncthreads();
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif

        // -- compact the alive candidates of the strip into LDS (class order preserved) ---------------------------
        bool alive = false;
        int wrank = 0;
        if (tid < SL) {
            alive = valid && (s_flag[tid] == 0);
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
                a_box[q] = jb; a_ar[q] = jar; a_lab[q] = jlab; a_slot[q] = slot; a_rank[q] = rnk;
            }
        }
        if (tid < S) s_flag[tid] = 0;                        // from here on: kept, indexed by rank
        __syncthreads();
        const int nab = (na + 63) >> 6;
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif

        // -- B: every alive candidate against the earlier alive candidates of its class ---------------------------
        // thread = (column j, quarter of the rows); the waves of one quarter take the column blocks in rotated
        // order so that every SIMD receives short and long columns
        {
            const int bpart = wave >> 2;                          // rows i with (i - run start) % 4 == bpart
            const int j = (((wave + bpart) & 3) << 6) | lane;
            const bool jv = j < na;
            const int jq = jv ? j : 0;
            const float4 cbx = a_box[jq];
            const float car = a_ar[jq];
            const int mine = a_lab[jq];
            // rows i = j - 1 - bpart, j - 5 - bpart, ...: with grouped candidates the walk ends where the class ends
            // (earlier candidates of the strip have smaller classes); two rows per step, their reads together
            bool open = jv;                                       // this column may still meet rows of its class
            for (int i = j - 1 - bpart; __ballot(open && i >= 0) != 0ull; i -= 8) {
                const int i2 = i - 4;
                const bool in1 = open && i >= 0, in2 = open && i2 >= 0;
                const int q1 = in1 ? i : 0, q2 = in2 ? i2 : 0;
                const float4 r1 = a_box[q1], r2 = a_box[q2];
                const float ar1 = a_ar[q1], ar2 = a_ar[q2];
                const int l1 = a_lab[q1], l2 = a_lab[q2];
                const bool on1 = in1 && l1 == mine, on2 = in2 && l2 == mine;
                if (group) open = open && on2;                   // sorted by class: a different class ends the walk
                const float inter1 = intersection(r1, cbx), inter2 = intersection(r2, cbx);
                if (can_skip && __ballot((on1 && inter1 > 0.0f) || (on2 && inter2 > 0.0f)) == 0ull) continue;
                const bool sup1 = over_threshold(inter1, ar1, car, thr) && on1;
                const bool sup2 = over_threshold(inter2, ar2, car, thr) && on2;
                if (sup1) atomicOr(&s_col[j * CW + (i >> 5)], 1u << (i & 31));
                if (sup2) atomicOr(&s_col[j * CW + (i2 >> 5)], 1u << (i2 & 31));
            }
        }
        __syncthreads();
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif

        // -- C: resolve the strip in class order (one wave) ---------------------------------------------------------
        if (wave == 0) {
            unsigned long long keptw[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) keptw[i] = 0ull;
#pragma unroll
            for (int blk = 0; blk < SB; ++blk) {
                if (blk < nab) {
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
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < SB; ++i) s_keptw[i] = keptw[i];
            }
        }
        __syncthreads();
        // -- append: kept list in class order, output list in score order ---------------------------------------
        int kept_total = 0;
        {
            int before = 0;                                  // kept columns before this thread's column
            bool kept = false;
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const unsigned long long m = s_keptw[i];
                kept_total += __popcll(m);
                if (i < (tid >> 6)) before += __popcll(m);
                else if (i == (tid >> 6)) { before += __popcll(m & ((1ull << lane) - 1ull)); kept = (m >> lane) & 1ull; }
            }
            if (tid < na && kept) {
                const int q = nk_cls + before;
                if (q < KC) { k_box[q] = a_box[tid]; k_ar[q] = a_ar[tid]; k_lab[q] = a_lab[tid]; }
                else if (kept_cls) kept_cls[q] = a_slot[tid];
                s_flag[a_rank[tid]] = 1;
            }
        }
        __syncthreads();
        if (tid < SL) {
            const bool kept = s_flag[tid] != 0;
            const uint64_t bal = __ballot(kept);
            if (lane == 0) s_tmp[16 + wave] = __popcll(bal);
            wrank = mbcnt(bal);
            alive = kept;
        }
        __syncthreads();
        if (tid < SL && alive) {
            int pre = 0;
#pragma unroll
            for (int w = 0; w < SB; ++w) if (w < wave && w * 64 < SL) pre += s_tmp[16 + w];
            const int pos = nk + pre + wrank;
            if (pos < max_keep) kept_out[pos] = s_strip[tid];
        }
        nk_cls += kept_total;
        if (tid == 0) {
            int t = nk + kept_total;
            if (t >= max_keep) { t = max_keep; s_tmp[1] = 1; }
            s_tmp[0] = t;
        }
        __syncthreads();
#ifdef OD_STAMPS
        if (stamp_i < 30) STAMP(stamp_i++);
#endif
        if (s_tmp[1]) break;
    }
    __syncthreads();
    STAMP(30);

    // ---- 4. outputs -----------------------------------------------------------------------------------
    const int nk = s_tmp[0];
    for (int k = tid; k < a.max_keep; k += NT) {
        const long long o = (long long)b * a.max_keep + k;
        if (k < nk) {
            const int slot = kept_out[k];
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

// Global order of unsorted candidates (or of more than WSEG segments): rank of a candidate = number of valid
// candidates with a greater key (score, then lower slot).  Keys are unique, so the ranks are a permutation:
// order[rank] = slot.  O(n^2) comparisons spread over the whole chip: a workgroup ranks 32 candidates against all
// candidates, staged through LDS in tiles of 1024; eight threads share a candidate, each counting in an eighth of every
// tile (with one thread per candidate and 256 candidates per workgroup a few thousand candidates made 17 workgroups
// of 4000 serial comparisons: 62 us at n = 4145, now 13).
constexpr int OT = 256;
constexpr int OTILE = 1024;
constexpr int OCAND = 32;                 // candidates per workgroup
constexpr int OPARTS = OT / OCAND;        // threads per candidate
__global__ __launch_bounds__(OT) void nms_order_kernel(const float* __restrict__ scores_all, const int* __restrict__ seg_count,
                                                       int num_seg, int seg_stride, int n_slots, int* __restrict__ order_all) {
    __shared__ unsigned long long s_key[OTILE];
    __shared__ int s_cnt[MAX_SEG];
    __shared__ int s_part[OPARTS][OCAND];
    const int b = blockIdx.y;
    const float* scores = scores_all + (long long)b * n_slots;
    int* order = order_all + (long long)b * n_slots;
    if ((int)threadIdx.x < num_seg) {
        int c = seg_count[(long long)b * num_seg + threadIdx.x];
        s_cnt[threadIdx.x] = c < 0 ? 0 : (c > seg_stride ? seg_stride : c);
    }
    __syncthreads();
    auto key_of = [&](int i) -> unsigned long long {
        if (i >= n_slots) return 0ull;
        const int seg = i / seg_stride;
        return (i - seg * seg_stride < s_cnt[seg]) ? nms_key(scores[i], i) : 0ull;
    };
    const int cand = threadIdx.x & (OCAND - 1), part = threadIdx.x / OCAND;
    const int me = blockIdx.x * OCAND + cand;
    const unsigned long long mine = key_of(me);
    int rank = 0;
    for (int t0 = 0; t0 < n_slots; t0 += OTILE) {
        __syncthreads();
        for (int i = threadIdx.x; i < OTILE; i += OT) s_key[i] = key_of(t0 + i);     // past the end: 0, greater than no key
        __syncthreads();
        const unsigned long long* mykeys = s_key + part * (OTILE / OPARTS);
#pragma unroll 8
        for (int i = 0; i < OTILE / OPARTS; ++i) rank += (mykeys[i] > mine) ? 1 : 0;
    }
    s_part[part][cand] = rank;
    __syncthreads();
    if (part == 0 && mine != 0ull) {
#pragma unroll
        for (int q = 1; q < OPARTS; ++q) rank += s_part[q][cand];
        order[rank] = me;
    }
}

// ---- NMS as a bit matrix over the whole chip ----------------------------------------------------------------
// The form torchvision's CUDA kernel has (csrc/ops/cuda/nms_kernel.cu: an n x ceil(n / 64) matrix of "row i suppresses
// column j" words, then one ordered pass over the rows), for the calls the strip kernel above is slow at: a plain nms
// or batched_nms that KEEPS thousands of boxes (the strip kernel tests every strip against the whole kept list, beyond
// 1024 kept boxes through memory: 1.8 ms to keep all of 4145).  Three launches after the global order:
//   prepare  one workgroup per image: the mode decisions of nms_kernel (trick / vanilla, are the classes independent,
//            the offset step), then boxes (offsets applied), areas and labels written in score order;
//   matrix   one workgroup per 64 x 64 tile of the upper triangle, a wave per 16 rows, lane = column: the 64
//            decisions of a row become one word through a ballot;
//   scan     one workgroup per image walks the row blocks in order: a block's alive columns are what the kept rows of
//            earlier blocks left, the greedy choice inside the block is a scalar loop over its set bits, and the words
//            of the newly kept rows are or-ed into the later blocks' masks (the rows of the NEXT block are requested
//            before the current one is resolved, whether they will be kept or not: one CU's latency, not its
//            bandwidth, is what a dependent walk would wait for).  It stops at max_keep and writes the outputs.
// Same decisions as nms_kernel (same intersection and division code), hence the same results bit for bit.
constexpr int MX_MAXN = 8192;             // candidate slots per image this path takes (128 words per row)
constexpr int MX_WMAX = MX_MAXN / 64;

struct MatrixArgs {
    NmsArgs a;
    int4* header;               // [batch]: n_valid, compare labels?, -, -
    float4* sbox;               // [batch, n_slots] boxes in score order (trick offsets applied)
    float* sarea;               // [batch, n_slots]
    int* slab;                  // [batch, n_slots] labels in score order (0 where labels are not compared)
    unsigned long long* mask;   // [batch, n_slots, words]
    unsigned long long* dcol;   // [batch, n_slots]: for column j, the rows of ITS OWN block of 64 that suppress it
    int words;                  // ceil(n_slots / 64)
};

__global__ __launch_bounds__(NT) void nms_matrix_prepare_kernel(MatrixArgs m) {
    __shared__ int s_segpre[MAX_SEG + 1];
    __shared__ float s_ftmp[2 * (NT / 64)];
    __shared__ int s_itmp[2 * (NT / 64)];
    const NmsArgs& a = m.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float4* boxes = a.boxes + (long long)b * a.n_slots;
    const int* labels = a.labels ? a.labels + (long long)b * a.n_slots : nullptr;
    const int* segc = a.seg_count + (long long)b * a.num_seg;
    const int* order = a.order + (long long)b * a.n_slots;
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
    // the decisions of nms_kernel, step by step the same
    int mode = a.mode & NMS_MODE_MASK;
    if (mode == OD_NMS_AUTO_CPU) mode = (4 * n_valid > 4000) ? OD_NMS_VANILLA : OD_NMS_TRICK;
    if (mode == OD_NMS_AUTO_GPU) mode = (4 * n_valid > 20000) ? OD_NMS_VANILLA : OD_NMS_TRICK;
    const bool vanilla = (mode == OD_NMS_VANILLA) && labels != nullptr;
    const bool trick = (mode == OD_NMS_TRICK) && labels != nullptr;
    float step = 0.0f;
    bool independent = vanilla;
    bool one_class = labels == nullptr;
    if (labels) {
        float maxc = -__builtin_inff(), minc = __builtin_inff();
        int maxl = (int)0x80000000, minl = 0x7fffffff;
        for (int i = tid; i < a.n_slots; i += NT) {
            const int seg = i / a.seg_stride;
            const int r = i - seg * a.seg_stride;
            if (r < s_segpre[seg + 1] - s_segpre[seg]) {
                const float4 bx = boxes[i];
                const int lab = labels[i];
                maxc = fmaxf(maxc, fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
                minc = fminf(minc, fminf(fminf(bx.x, bx.y), fminf(bx.z, bx.w)));
                maxl = lab > maxl ? lab : maxl;
                minl = lab < minl ? lab : minl;
            }
        }
        maxc = wave_fmax_all(maxc);
        minc = wave_fmin_all(minc);
        maxl = (int)(wave_umax_all((uint32_t)maxl ^ 0x80000000u) ^ 0x80000000u);
        minl = (int)(~wave_umax_all(~((uint32_t)minl ^ 0x80000000u)) ^ 0x80000000u);
        if (lane == 0) { s_ftmp[wave] = maxc; s_ftmp[NT / 64 + wave] = minc; s_itmp[wave] = maxl; s_itmp[NT / 64 + wave] = minl; }
        __syncthreads();
        maxc = s_ftmp[0]; minc = s_ftmp[NT / 64]; maxl = s_itmp[0]; minl = s_itmp[NT / 64];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) {
            maxc = fmaxf(maxc, s_ftmp[w]);
            minc = fminf(minc, s_ftmp[NT / 64 + w]);
            maxl = s_itmp[w] > maxl ? s_itmp[w] : maxl;
            minl = s_itmp[NT / 64 + w] < minl ? s_itmp[NT / 64 + w] : minl;
        }
        step = maxc + 1.0f;
        if (trick) independent = a.can_skip_zero && minc >= 0.0f && minl >= 0 && ((float)maxl * step <= 8388608.0f);
        one_class = (minl == maxl);
    }
    const bool by_class = independent && !one_class;
    if (tid == 0) m.header[b] = make_int4(n_valid, by_class ? 1 : 0, 0, 0);
    float4* sbox = m.sbox + (long long)b * a.n_slots;
    float* sarea = m.sarea + (long long)b * a.n_slots;
    int* slab = m.slab + (long long)b * a.n_slots;
    for (int r = tid; r < n_valid; r += NT) {
        const int slot = order[r];
        float4 bx = boxes[slot];
        const int raw = labels ? labels[slot] : 0;
        if (trick) {
            const float off = (float)raw * step;
            bx.x += off; bx.y += off; bx.z += off; bx.w += off;
        }
        sbox[r] = bx;
        sarea[r] = (bx.z - bx.x) * (bx.w - bx.y);
        slab[r] = by_class ? raw : 0;
    }
}

constexpr int MXT = 256;                  // threads of a matrix workgroup: four waves x 16 rows
__global__ __launch_bounds__(MXT) void nms_matrix_kernel(MatrixArgs m) {
    const int cb = blockIdx.x, rb = blockIdx.y, b = blockIdx.z;
    if (rb > cb) return;                                   // upper triangle only
    const int4 h = m.header[b];
    const int n_valid = h.x;
    if (rb * 64 >= n_valid || cb * 64 >= n_valid) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* sbox = m.sbox + (long long)b * m.a.n_slots;
    const float* sarea = m.sarea + (long long)b * m.a.n_slots;
    const int* slab = m.slab + (long long)b * m.a.n_slots;
    unsigned long long* mask = m.mask + (long long)b * m.a.n_slots * m.words;
    const int j = cb * 64 + lane;
    const bool jv = j < n_valid;
    const int jq = jv ? j : n_valid - 1;
    const float4 jb = sbox[jq];
    const float jar = sarea[jq];
    const int jlab = slab[jq];
    const float thr_up = m.a.thr_up;
    const int i0 = rb * 64 + wave * 16;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int i = i0 + u;                              // wave-uniform
        if (i >= n_valid) break;
        const float4 ib = sbox[i];
        const float inter = intersection(ib, jb);
        const bool sup = jv && j > i && slab[i] == jlab && over_threshold_exact(inter, sarea[i], jar, thr_up);
        const unsigned long long word = __ballot(sup);
        if (lane == 0) mask[(long long)i * m.words + cb] = word;
    }
    if (rb == cb) {
        // The diagonal tile once more, transposed: lane = row, a wave takes 16 columns, and a ballot gives the rows of
        // the block that suppress a column -- the form in which the ordered pass resolves a block without a serial loop.
        unsigned long long* dcol = m.dcol + (long long)b * m.a.n_slots;
        const int c0 = cb * 64 + wave * 16;
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int c = c0 + u;                          // wave-uniform column; this lane's row is j
            if (c >= n_valid) break;
            const float4 cbx = sbox[c];
            const float inter = intersection(jb, cbx);
            const bool sup = jv && c > j && slab[c] == jlab && over_threshold_exact(inter, jar, sarea[c], thr_up);
            const unsigned long long word = __ballot(sup);
            if (lane == 0) dcol[c] = word;
        }
    }
}

__global__ __launch_bounds__(NT) void nms_matrix_scan_kernel(MatrixArgs m) {
    __shared__ unsigned long long s_removed[MX_WMAX];
    __shared__ unsigned long long s_keptw;
    __shared__ int s_segpre[MAX_SEG + 1];
    __shared__ int s_nk;
    const NmsArgs& a = m.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int n_valid = m.header[b].x;
    const int W = m.words;
    const unsigned long long* mask = m.mask + (long long)b * a.n_slots * W;
    const unsigned long long* dcol = m.dcol + (long long)b * a.n_slots;
    const int* order = a.order + (long long)b * a.n_slots;
    int* kept_out = a.kept_out + (long long)b * a.n_slots;           // kept ranks in order
    const int max_keep = a.max_keep < n_valid ? a.max_keep : n_valid;
    if (tid < MX_WMAX) s_removed[tid] = 0ull;
    if (tid == 0) {
        s_nk = 0;
        int acc = 0;
        for (int s = 0; s < a.num_seg; ++s) {
            s_segpre[s] = acc;
            int c = a.seg_count[(long long)b * a.num_seg + s];
            c = c < 0 ? 0 : (c > a.seg_stride ? a.seg_stride : c);
            acc += c;
        }
        s_segpre[a.num_seg] = acc;
    }
    __syncthreads();
    const int nblk = (n_valid + 63) >> 6;
    // thread (g, w): word w of the rows 8 q + g of a block, q = 0 ... 7 (W <= 128 words, 1024 threads)
    const int w = tid & (MX_WMAX - 1), g = tid >> 7;
    // Four blocks are in flight: the words of block blk + 4 are requested when block blk is taken up (a block's turn is
    // about a microsecond of scalar work and two barriers, a request a few microseconds of latency).
    constexpr int DEPTH = 4;
    unsigned long long nxt[DEPTH][8];
    unsigned long long dnxt[DEPTH];                      // wave 0: lane l holds the rows of the block that suppress its column l
    auto request = [&](auto slot_tag, int blk) {         // the words right of the diagonal of every row of the block
        constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = blk * 64 + q * 8 + g;
            // (no column block beyond nblk is ever written)
            nxt[SLOT][q] = (blk < nblk && row < n_valid && w > blk && w < nblk) ? mask[(long long)row * W + w] : 0ull;
        }
        const int dc = blk * 64 + lane;
        dnxt[SLOT] = (wave == 0 && blk < nblk && dc < n_valid) ? dcol[dc] : 0ull;
    };
    auto turn = [&](auto slot_tag, int blk) -> bool {    // returns true when max_keep is reached
        constexpr int SLOT = decltype(slot_tag)::value;
        unsigned long long cur[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cur[q] = nxt[SLOT][q];
        const unsigned long long diag = dnxt[SLOT];
        request(slot_tag, blk + DEPTH);
        if (wave == 0) {
            // greedy choice inside the block = the fixed point of kept = alive & ~(some kept earlier row suppresses me),
            // reached by ballot iteration from "every alive column is kept" (as in nms_kernel); then the cap
            const bool alive = ((~s_removed[blk] >> lane) & 1ull) != 0ull && blk * 64 + lane < n_valid;
            unsigned long long K = __ballot(alive);
            for (int it = 0; it < 64; ++it) {
                const unsigned long long K2 = __ballot(alive && ((diag & K) == 0ull));
                if (K2 == K) break;
                K = K2;
            }
            const int nk0 = s_nk;
            const int room = max_keep - nk0;               // > 0: the walk stops when the cap is reached
            const int before = __popcll(K & ((1ull << lane) - 1ull));
            const bool kept_me = ((K >> lane) & 1ull) != 0ull && before < room;
            const unsigned long long kept = __ballot(kept_me);
            if (kept_me) kept_out[nk0 + before] = blk * 64 + lane;
            if (lane == 0) { s_keptw = kept; s_nk = nk0 + __popcll(kept); }
        }
        __syncthreads();
        const unsigned long long kept = s_keptw;
        unsigned long long acc = 0ull;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc |= ((kept >> (q * 8 + g)) & 1ull) ? cur[q] : 0ull;
        if (acc) atomicOr(&s_removed[w], acc);
        const bool done = s_nk >= max_keep;
        __syncthreads();
        return done;
    };
    request(std::integral_constant<int, 0>{}, 0);
    request(std::integral_constant<int, 1>{}, 1);
    request(std::integral_constant<int, 2>{}, 2);
    request(std::integral_constant<int, 3>{}, 3);
    for (int blk = 0; blk < nblk; blk += DEPTH) {
        if (turn(std::integral_constant<int, 0>{}, blk)) break;
        if (blk + 1 >= nblk || turn(std::integral_constant<int, 1>{}, blk + 1)) break;
        if (blk + 2 >= nblk || turn(std::integral_constant<int, 2>{}, blk + 2)) break;
        if (blk + 3 >= nblk || turn(std::integral_constant<int, 3>{}, blk + 3)) break;
    }
    // outputs, as nms_kernel writes them
    const int nk = s_nk;
    const float4* boxes = a.boxes + (long long)b * a.n_slots;
    const float* scores = a.scores + (long long)b * a.n_slots;
    const int* labels = a.labels ? a.labels + (long long)b * a.n_slots : nullptr;
    for (int k = tid; k < a.max_keep; k += NT) {
        const long long o = (long long)b * a.max_keep + k;
        if (k < nk) {
            const int slot = order[kept_out[k]];
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

// ---- tail of the reference's prediction loop (object_detection/predict.py:70-76) for a whole batch -----------
// Per image: boxes.mul_(ratio); keep = nms(boxes, scores, thr); boxes.to(int32)[keep], scores[keep], labels[keep].
// 1024-thread workgroups, at most PT_MAX detections per image (what postprocess_detections leaves: <= 1024); one
// workgroup per image up to 128 detections, above that up to 16 that share step 2 and leave the rest to the last one.
//   1. every thread scales one box and finds its place in the stable descending score order (detections arrive
//      sorted from batched_nms, which one neighbour comparison per thread confirms; otherwise every thread counts
//      the keys above its own);
//   2. suppression bits: a wave takes a tile of 64 rows x 64 columns of the upper triangle, lane = column, and
//      turns the 64 decisions "row i suppresses column j" into one 64-bit word per row with a ballot;
//   3. one wave walks the blocks of 64 in order: a block's alive mask is what earlier kept rows left of it, the
//      greedy choice inside the block is a scalar loop over its set bits, and the words of the newly kept rows are
//      or-ed into the masks of the later blocks; kept detections are written in order as they are found.
constexpr int PT_MAX = 1024;
constexpr int PT_NB = PT_MAX / 64;
// Row block R (rows 64R .. 64R+63) holds the words of column blocks R .. PT_NB-1, with an odd number of words per row:
// the walk reads one row per lane, and an even pitch would put all 64 lanes on the same LDS banks.
__host__ __device__ __forceinline__ constexpr int pt_pitch(int R) { return (PT_NB - R) | 1; }
// first word of row block R = 64 * sum of the pitches of the blocks before it
__host__ __device__ __forceinline__ constexpr int pt_block_base(int R) { return 64 * (PT_NB * R - R * (R - 1) / 2 + (R + 1) / 2); }
constexpr int PT_WORDS = pt_block_base(PT_NB);               // words of the upper block triangle

struct PredictTailArgs {
    const float4* boxes;        // [batch, max_det]
    const float* scores;
    const long long* labels;
    const int* counts;
    const float4* ratio;        // [batch]: factors of x1, y1, x2, y2
    int max_det;
    float thr_up;
    float4* scaled_boxes;       // may alias boxes; may be null
    int4* out_boxes;
    float* out_scores;
    long long* out_labels;
    long long* keep_idx;        // may be null
    int* out_count;
    unsigned long long* g_mask; // [batch, PT_WORDS]: suppression words of an image whose workgroups share the work
    int* done;                  // [batch]: workgroups of the image that have written their words (zero at the launch)
};

__global__ __launch_bounds__(NT) void predict_tail_kernel(PredictTailArgs a) {
    __shared__ __attribute__((aligned(16))) float4 s_box[PT_MAX];   // scaled boxes in score order
    __shared__ float s_area[PT_MAX];
    __shared__ float s_score[PT_MAX];                               // scores as given
    __shared__ int s_src[PT_MAX];                                   // index as given of the r-th best
    __shared__ long long s_lab[PT_MAX];                             // labels in score order
    __shared__ unsigned long long s_mask[PT_WORDS];
    __shared__ unsigned long long s_keptw[PT_NB];                   // kept rows of every block
    __shared__ int s_pre[PT_NB + 1];                                // kept rows before every block

    __shared__ int s_last;
#ifdef PT_STAMPS
    unsigned long long pst[8];
#define PSTAMP(k) pst[k] = __builtin_amdgcn_s_memrealtime()
#else
#define PSTAMP(k) do { } while (0)
#endif
    PSTAMP(0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int G = gridDim.x;                    // workgroups that share the image's suppression matrix
    const long long base = (long long)b * a.max_det;
    int n = a.counts[b];
    n = n < 0 ? 0 : (n > a.max_det ? a.max_det : n);

    // ---- 1. scale, order ------------------------------------------------------------------------------------
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = 0.0f;
    long long lab = 0;
    if (tid < n) {
        const float4 r = a.ratio[b];
        const float4 raw = a.boxes[base + tid];
        sc = a.scores[base + tid];
        lab = a.labels[base + tid];
        bx = make_float4(raw.x * r.x, raw.y * r.y, raw.z * r.z, raw.w * r.w);
        s_score[tid] = sc;
    }
    __syncthreads();
    const bool in_place = !(tid > 0 && tid < n) || s_score[tid - 1] >= sc;
    int rank = tid;
    if (!__syncthreads_and(in_place)) {
        const unsigned long long mine = nms_key(sc, tid);
        rank = 0;
        for (int j = 0; j < n; ++j) rank += nms_key(s_score[j], j) > mine ? 1 : 0;
    }
    if (tid < n) {
        s_box[rank] = bx;
        s_area[rank] = (bx.z - bx.x) * (bx.w - bx.y);
        s_src[rank] = tid;
        s_lab[rank] = lab;
    }
    __syncthreads();

    PSTAMP(1);
    // ---- 2. suppression words of the upper triangle ----------------------------------------------------------
    // unit = 16 rows of a (row block R, column block w) tile, handed out round-robin to the sixteen waves
    ThrBand thr;
    thr.up = a.thr_up;
    thr.lo = a.thr_up * (1.0f - 1e-5f);
    thr.hi = a.thr_up * (1.0f + 1e-5f);
    thr.usable = (a.thr_up > 1e-30f) && (a.thr_up < 1e30f);
    const int nb = (n + 63) >> 6;
    const int units = 4 * (nb * (nb + 1) / 2);
    // one workgroup per image keeps the words in LDS; several write them to the workspace, and the last one to finish
    // collects them
    unsigned long long* mask = G == 1 ? s_mask : a.g_mask + (long long)b * PT_WORDS;
    for (int u = (int)blockIdx.x * (NT / 64) + wave; u < units; u += G * (NT / 64)) {
        int R = 0, rest = u >> 2;                  // tile -> (row block R, column block w), R <= w < nb
        while (rest >= nb - R) { rest -= nb - R; ++R; }
        const int w = R + rest;
        const int j = 64 * w + lane;
        const bool jv = j < n;
        const float4 cj = s_box[jv ? j : 0];
        const float aj = s_area[jv ? j : 0];
        const int r0 = 16 * (u & 3);
        const int pitch = pt_pitch(R);
        unsigned long long* out = mask + pt_block_base(R) + rest + r0 * pitch;
        const int i0 = 64 * R + r0;
        if (i0 >= n) continue;
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            float4 bi[4];
            float ai[4], inter[4], den[4];
            bool above[4];
            unsigned long long unsure = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {           // rows past n read row n-1 again; their words are never used
                const int i = (i0 + q + k) < n ? (i0 + q + k) : (n - 1);
                bi[k] = s_box[i];
                ai[k] = s_area[i];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                inter[k] = intersection(bi[k], cj);
                den[k] = (ai[k] + aj) - inter[k];
                const float ph = thr.hi * den[k], pl = thr.lo * den[k];
                above[k] = inter[k] >= ph;
                const bool below = inter[k] <= pl;
                const bool safe = (pl > 1e-30f) && (ph < 1e38f);
                unsure |= __ballot(!(above[k] || below) || !safe);
            }
            if (unsure != 0ull || !thr.usable) {    // some pair is too close to the threshold for the products to decide
#pragma unroll
                for (int k = 0; k < 4; ++k) above[k] = (inter[k] / den[k]) >= thr.up;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + q + k;
                const unsigned long long word = __ballot(above[k] && jv && j > i);
                if (lane == 0) {
                    if (G == 1) out[(q + k) * pitch] = word;
                    else __hip_atomic_store(out + (q + k) * pitch, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    PSTAMP(2);
    if (G > 1) {
        // The words travel as device-scope stores and loads (they pass the per-XCD L2s: the workgroups of an image run
        // on different XCDs), and a wave has its stores acknowledged before the workgroup takes its ticket.  A
        // device-scope fence pair instead (write-back and invalidation of whole L2s) was measured at 35 us.
        // This is NOT the release / acquire pairing of the HIP memory model; it relies on what gfx942 / gfx950 do:
        // an agent-scope store is issued with sc1 and written through the XCD's L2 to memory, vector stores are
        // counted by vmcnt on these targets (s_waitcnt 0 returns once they are acknowledged), the ticket is a
        // memory-side atomic, and an agent-scope load misses the reader's L2.  A target that counts stores
        // elsewhere (vscnt on gfx10+) needs the wait changed, hence the guard.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "predict_tail_kernel's cross-workgroup hand-off is written for gfx942 / gfx950 (see the comment above)"
#endif
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&a.done[b], 1) == G - 1 ? 1 : 0;
        __syncthreads();
        if (!s_last) return;                       // the last workgroup of the image goes on alone
        const int words = pt_block_base(nb);       // the row blocks in use, whole
        for (int i = tid; i < words; i += NT)
            s_mask[i] = __hip_atomic_load(mask + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // the rescaled boxes go back in the given order (possibly over the input: every workgroup has read it by now)
    if (a.scaled_boxes && tid < n) a.scaled_boxes[base + tid] = bx;

    PSTAMP(3);
    // ---- 3. greedy walk over the blocks (one wave) -----------------------------------------------------------
    if (wave == 0) {
        unsigned long long pending = 0;            // lane w: suppression bits that kept rows left on block w
        int nk = 0;
        for (int c = 0; c < nb; ++c) {
            const int left = n - 64 * c;
            const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pending, c);
            const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pending >> 32), c);
            unsigned long long alive = valid & ~(((unsigned long long)phi << 32) | plo);
            const int i = 64 * c + lane;
            const int pitch = pt_pitch(c);
            const unsigned long long* row = s_mask + pt_block_base(c) + lane * pitch;
            const bool iv = i < n;
            // the row's words for this and all later blocks are requested together
            unsigned long long rw[PT_NB];
#pragma unroll
            for (int k = 0; k < PT_NB; ++k) rw[k] = (iv && c + k < nb) ? row[k] : 0ull;
            const unsigned long long diag = rw[0];
            // only rows that suppress something inside the block need their turn; the others are kept if alive
            unsigned long long rem = alive & __ballot(diag != 0ull);
            while (rem) {
                const int r = __builtin_ctzll(rem);
                const uint32_t dlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)diag, r);
                const uint32_t dhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(diag >> 32), r);
                alive &= ~(((unsigned long long)dhi << 32) | dlo);
                rem &= alive & ~((2ull << r) - 1ull);
            }
            const unsigned long long kept = alive;
            const bool mine = (kept >> lane) & 1ull;
#pragma unroll
            for (int k = 1; k < PT_NB; ++k) {      // what the newly kept rows suppress in later blocks
                if (c + k >= nb) break;
                const unsigned long long v = mine ? rw[k] : 0ull;
                if (__ballot(v != 0ull) == 0ull) continue;
                const unsigned long long all = ((unsigned long long)wave_or_all((uint32_t)(v >> 32)) << 32) |
                                               wave_or_all((uint32_t)v);
                if (lane == c + k) pending |= all;
            }
            if (lane == 0) { s_keptw[c] = kept; s_pre[c] = nk; }
            nk += __popcll(kept);
        }
        if (lane == 0) { s_pre[nb] = nk; a.out_count[b] = nk; }
    }
    __syncthreads();
    PSTAMP(4);
    // ---- 4. outputs: kept rows in order, zeros past the count --------------------------------------------------
    const int nk = s_pre[nb];
    if (tid < n) {
        const unsigned long long kw = s_keptw[wave];
        if ((kw >> lane) & 1ull) {
            const int pos = s_pre[wave] + mbcnt(kw);
            const int src = s_src[tid];
            const float4 k = s_box[tid];
            a.out_boxes[base + pos] = make_int4((int)k.x, (int)k.y, (int)k.z, (int)k.w);
            a.out_scores[base + pos] = s_score[src];
            a.out_labels[base + pos] = s_lab[tid];
            if (a.keep_idx) a.keep_idx[base + pos] = src;
        }
    }
    for (int k = nk + tid; k < a.max_det; k += NT) {
        a.out_boxes[base + k] = make_int4(0, 0, 0, 0);
        a.out_scores[base + k] = 0.0f;
        a.out_labels[base + k] = 0;
        if (a.keep_idx) a.keep_idx[base + k] = 0;
    }
#ifdef PT_STAMPS
    PSTAMP(5);
    if (tid == 0 && b == 0)
        printf("tail stamps (10 ns): order %llu mask %llu collect %llu walk %llu out %llu\n", pst[1] - pst[0], pst[2] - pst[1],
               pst[3] - pst[2], pst[4] - pst[3], pst[5] - pst[4]);
#endif
}

}  // namespace

static size_t nms_align256(size_t v) { return (v + 255) & ~(size_t)255; }

// workspace: order | kept_out | kept_cls (int32 [batch, n_slots] each) | time stamps of a debug build
extern "C" size_t od_batched_nms_workspace_bytes(int32_t batch, int32_t num_seg, int32_t seg_stride) {
    if (batch < 0 || num_seg < 1 || seg_stride < 1) return 0;
    long long n = (long long)num_seg * seg_stride;
    if (n > MAXN || num_seg > MAX_SEG) return 0;
    const size_t part = nms_align256((size_t)batch * (size_t)n * sizeof(int));
    return 3 * part + nms_align256((size_t)batch * 256);
}

static int rank_min_keep() {
    static const int v = [] {
        const char* e = getenv("OD_NMS_RANK_MIN_KEEP");
        return e ? atoi(e) : 256;
    }();
    return v;
}

// Regions behind the strip kernel's workspace: header, boxes / areas / labels in score order, the bit matrix.
struct MatrixLayout { size_t header, sbox, sarea, slab, dcol, mask, total; int words; };
static MatrixLayout matrix_layout(int32_t batch, int32_t num_seg, int32_t seg_stride) {
    MatrixLayout l{};
    const size_t base = od_batched_nms_workspace_bytes(batch, num_seg, seg_stride);
    const long long n = (long long)num_seg * seg_stride;
    if (base == 0 || n > MX_MAXN) return l;
    l.words = (int)((n + 63) / 64);
    l.header = base;
    l.sbox = l.header + nms_align256((size_t)batch * sizeof(int4));
    l.sarea = l.sbox + nms_align256((size_t)batch * (size_t)n * sizeof(float4));
    l.slab = l.sarea + nms_align256((size_t)batch * (size_t)n * sizeof(float));
    l.dcol = l.slab + nms_align256((size_t)batch * (size_t)n * sizeof(int));
    l.mask = l.dcol + nms_align256((size_t)batch * (size_t)n * sizeof(unsigned long long));
    l.total = l.mask + nms_align256((size_t)batch * (size_t)n * (size_t)l.words * sizeof(unsigned long long));
    return l;
}

extern "C" size_t od_batched_nms_matrix_workspace_bytes(int32_t batch, int32_t num_seg, int32_t seg_stride) {
    if (batch < 0 || num_seg < 1 || seg_stride < 1) return 0;
    return matrix_layout(batch, num_seg, seg_stride).total;
}

// Threshold rule folded into one float: a pair is suppressed when ovr >= the returned value.
//   torchvision's CPU kernel (default):  (double)ovr > thr   <=>  ovr >= smallest float whose double value exceeds thr
//   torchvision's CUDA kernel (OD_NMS_THRESHOLD_FLOAT):  ovr > (float)thr  <=>  ovr >= the float after (float)thr
static float fold_threshold(double iou_threshold, int rule) {
    float f = (float)iou_threshold;
    if (rule & OD_NMS_THRESHOLD_FLOAT) f = __builtin_nextafterf(f, __builtin_inff());
    else if (!((double)f > iou_threshold)) f = __builtin_nextafterf(f, __builtin_inff());
    if (iou_threshold != iou_threshold) f = __builtin_nanf("");   // NaN threshold: nothing suppresses
    return f;
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
    if (num_seg > MAX_SEG || (long long)num_seg * seg_stride > MAXN || batch > 65535) return OD_ERR_LIMIT;
    const int rule = mode & ~NMS_MODE_MASK;
    const int base_mode = mode & NMS_MODE_MASK;
    if (base_mode < OD_NMS_TRICK || base_mode > OD_NMS_AUTO_GPU || (rule & ~(OD_NMS_THRESHOLD_FLOAT | OD_NMS_PREFER_MATRIX)))
        return OD_ERR_ARG;
    if (((uintptr_t)boxes & 15) || ((uintptr_t)out_boxes & 15) || ((uintptr_t)workspace & 255)) return OD_ERR_ARG;
    if (workspace_bytes < od_batched_nms_workspace_bytes(batch, num_seg, seg_stride)) return OD_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    NmsArgs a;
    a.boxes = (const float4*)boxes;
    a.scores = scores;
    a.labels = labels;
    a.seg_count = seg_count;
    a.num_seg = num_seg;
    a.seg_stride = seg_stride;
    a.n_slots = num_seg * seg_stride;
    a.lazy = (segments_sorted && num_seg <= WSEG) ? 1 : 0;
    // Ranking the first 896 candidates in one go pays when several strips will be needed; a call that keeps few boxes
    // is done after one or two strips of the lazy merge (max_keep = 100: 15 -> 19 us with the ranked head).
    if (a.lazy && max_keep >= rank_min_keep()) a.lazy = 2;
    a.thr_up = fold_threshold(iou_threshold, rule & OD_NMS_THRESHOLD_FLOAT);
    a.can_skip_zero = (iou_threshold >= 0.0) ? 1 : 0;
    a.mode = base_mode;
    a.max_keep = max_keep;
    const size_t part = nms_align256((size_t)batch * (size_t)a.n_slots * sizeof(int));
    char* w = (char*)workspace;
    a.order = (int*)w;
    a.kept_out = (int*)(w + part);
    a.kept_cls = (int*)(w + 2 * part);
    a.stamps = (unsigned long long*)(w + 3 * part);
    a.keep_idx = (long long*)keep_idx;
    a.out_boxes = (float4*)out_boxes;
    a.out_scores = out_scores;
    a.out_labels = (long long*)out_labels;
    a.out_count = out_count;
    if (max_keep == 0) {
        OD_HIP(hipMemsetAsync(out_count, 0, sizeof(int) * (size_t)batch, s));
        return OD_OK;
    }
    if (rule & OD_NMS_PREFER_MATRIX) {
        const MatrixLayout l = matrix_layout(batch, num_seg, seg_stride);
        if (l.total != 0 && workspace_bytes >= l.total) {
            MatrixArgs m;
            m.a = a;
            m.header = (int4*)(w + l.header);
            m.sbox = (float4*)(w + l.sbox);
            m.sarea = (float*)(w + l.sarea);
            m.slab = (int*)(w + l.slab);
            m.mask = (unsigned long long*)(w + l.mask);
            m.dcol = (unsigned long long*)(w + l.dcol);
            m.words = l.words;
            OD_LAUNCH_KERNEL(OD_KERNEL_NMS_ORDER, nms_order_kernel, dim3((unsigned)((a.n_slots + OCAND - 1) / OCAND), (unsigned)batch),
                             dim3(OT), 0, s, scores, seg_count, (int)num_seg, (int)seg_stride, a.n_slots, a.order);
            OD_LAUNCH_KERNEL(OD_KERNEL_NMS_ORDER, nms_matrix_prepare_kernel, dim3((unsigned)batch), dim3(NT), 0, s, m);
            OD_LAUNCH_KERNEL(OD_KERNEL_NMS, nms_matrix_kernel, dim3((unsigned)l.words, (unsigned)l.words, (unsigned)batch),
                             dim3(MXT), 0, s, m);
            OD_LAUNCH_KERNEL(OD_KERNEL_NMS, nms_matrix_scan_kernel, dim3((unsigned)batch), dim3(NT), 0, s, m);
            return OD_OK;
        }
    }
    if (!a.lazy) {
        OD_LAUNCH_KERNEL(OD_KERNEL_NMS_ORDER, nms_order_kernel, dim3((unsigned)((a.n_slots + OCAND - 1) / OCAND), (unsigned)batch),
                         dim3(OT), 0, s, scores, seg_count, (int)num_seg, (int)seg_stride, a.n_slots, a.order);
    }
    OD_LAUNCH_KERNEL(OD_KERNEL_NMS, nms_kernel, dim3((unsigned)batch), dim3(NT), 0, s, a);
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
// The scratch memory of the NMS kernels (the global order of unsorted candidates and the kept lists beyond the LDS
// cache) lives in a device array that is part of the library's code object, so the entry point needs no workspace argument and allocates nothing.
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

// workgroups per image: about 24 units of 16 rows x 64 columns each (one and a half per wave)
static int predict_tail_groups(int max_det) {
    const int nb = (max_det + 63) / 64;
    const int units = 4 * (nb * (nb + 1) / 2);
    const int g = (units + 23) / 24;
    return g < 1 ? 1 : (g > 16 ? 16 : g);
}

// workspace: done counters (int32 [batch]) | suppression words (uint64 [batch, PT_WORDS]) when an image is shared
extern "C" size_t od_predict_tail_workspace_bytes(int32_t batch, int32_t max_det) {
    if (batch < 0 || max_det < 0 || max_det > PT_MAX) return 0;
    size_t bytes = nms_align256((size_t)batch * sizeof(int));
    if (predict_tail_groups(max_det) > 1) bytes += nms_align256((size_t)batch * PT_WORDS * sizeof(unsigned long long));
    return bytes > 0 ? bytes : 256;
}

extern "C" int od_predict_tail(const float* boxes, const float* scores, const int64_t* labels, const int32_t* counts,
                               const float* ratio, int32_t batch, int32_t max_det, double iou_threshold, int32_t mode,
                               float* scaled_boxes, int32_t* out_boxes, float* out_scores, int64_t* out_labels,
                               int64_t* keep_idx, int32_t* out_count, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (batch < 0 || max_det < 0) return OD_ERR_ARG;
    if (batch == 0) return OD_OK;
    if (!counts || !out_count) return OD_ERR_ARG;
    if (mode & ~OD_NMS_THRESHOLD_FLOAT) return OD_ERR_ARG;
    if (max_det > PT_MAX || batch > 65535) return OD_ERR_LIMIT;
    if (max_det == 0) {
        OD_HIP(hipMemsetAsync(out_count, 0, sizeof(int) * (size_t)batch, (hipStream_t)stream));
        return OD_OK;
    }
    if (!boxes || !scores || !labels || !ratio || !out_boxes || !out_scores || !out_labels) return OD_ERR_ARG;
    if (((uintptr_t)boxes & 15) || ((uintptr_t)ratio & 15) || ((uintptr_t)out_boxes & 15) ||
        ((uintptr_t)scaled_boxes & 15)) return OD_ERR_ARG;
    PredictTailArgs a;
    a.boxes = (const float4*)boxes;
    a.scores = scores;
    a.labels = (const long long*)labels;
    a.counts = counts;
    a.ratio = (const float4*)ratio;
    a.max_det = max_det;
    a.thr_up = fold_threshold(iou_threshold, mode);
    a.scaled_boxes = (float4*)scaled_boxes;
    a.out_boxes = (int4*)out_boxes;
    a.out_scores = out_scores;
    a.out_labels = (long long*)out_labels;
    a.keep_idx = (long long*)keep_idx;
    a.out_count = out_count;
    const int groups = predict_tail_groups(max_det);
    a.done = nullptr;
    a.g_mask = nullptr;
    if (groups > 1) {
        if (!workspace || ((uintptr_t)workspace & 255)) return OD_ERR_ARG;
        if (workspace_bytes < od_predict_tail_workspace_bytes(batch, max_det)) return OD_ERR_WORKSPACE;
        a.done = (int*)workspace;
        a.g_mask = (unsigned long long*)((char*)workspace + nms_align256((size_t)batch * sizeof(int)));
        OD_HIP(hipMemsetAsync(a.done, 0, sizeof(int) * (size_t)batch, (hipStream_t)stream));
    }
    OD_LAUNCH_KERNEL(OD_KERNEL_PREDICT_TAIL, predict_tail_kernel, dim3((unsigned)groups, (unsigned)batch), dim3(NT), 0,
                     (hipStream_t)stream, a);
    return OD_OK;
}
