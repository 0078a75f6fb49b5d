// Score filter + top-k + box decode + clip for every (image, level) of the pyramid.
// Replaces the inner loop of torchvision RetinaNet.postprocess_detections / FCOS.postprocess_detections:
//   scores = sigmoid(logits).flatten(); keep = scores > score_thresh; topk(min(#keep, K));
//   anchor = idx // C; label = idx % C; decode_single(reg[anchor], anchors[anchor]); clip_boxes_to_image
// (reference trigger object_detection/train.py:84, object_detection/predict.py:61; knobs predict.py:44-45).
//
// Kernel 1 (filter_kernel, a read-once stream): every workgroup owns one 8192-element chunk of one
// (image, level) logit slab and requests all of it with 16-byte loads before the first use.  A logit that
// cannot reach the threshold is rejected by one compare; the rest get the specified sigmoid.  Survivors
// become unique 64-bit keys  (score_bits << 32) | (0xFFFFFFFF - flat_index): larger key = better
// candidate, and the low word makes the adopted tie rule (equal score -> lower flat index first) part of
// the order.  Each wave stages its survivors in LDS, reserves that many slots of the (image, level) key
// array with one global atomic, and copies them there, so the array is dense (its order is irrelevant:
// keys are unique and get sorted).
//
// Kernel 2 (select_kernel, one 1024-thread workgroup per (image, level)): a radix descent on the key,
// 12 bits per step with a 4096-bin LDS histogram, narrows the interval that holds the K-th best key.  While
// more than 8192 keys are at or above the cut the steps stream the dense global array (coalesced); then
// the keys move into LDS and the descent continues there, compacting in place, until about K keys are
// left.  Those are bitonic-sorted (registers + lane shuffles + LDS), and the first K are decoded and
// clipped straight into the padded candidate arrays.
#include "od_common.h"
#include <type_traits>
#include "../../include/od_boxes.h"

namespace {

using namespace od;

constexpr int ST = 256;            // filter workgroup
constexpr int CHUNK = 8192;        // logits per filter workgroup (32 KB)
constexpr int SEL_T = 1024;        // select workgroup
constexpr int CAP = 8192;          // keys held in LDS by the select kernel (16384 measured slower: two more keys per thread in every pass)
constexpr int HBITS = 12;
constexpr int HBINS = 1 << HBITS;  // histogram bins per radix level
constexpr int CSTRIDE = 64;        // one key counter per 256 bytes: atomics on one cache line serialise
#ifndef OD_SAMPLE_MAX
#define OD_SAMPLE_MAX 131072
#endif
constexpr long long SAMPLE_MAX = OD_SAMPLE_MAX;   // largest sample of a level
constexpr long long SAMPLE_MIN = 262144;   // levels with at least this many logits get a sampled score cut
constexpr int SAMPLE_RUN = 16;     // consecutive logits per sample position (one 64-byte sector)

// Per (image, level): a lower bound on the scores that can be among the best K, estimated from a sample of the level
// (sample_kernel).  The filter emits only keys with score >= cut; if fewer than K valid keys come out, the estimate
// was too high and the level is filtered again without the cut (filter_kernel in redo mode), so the result never
// depends on the sample.
struct CutInfo {
    float cut;                     // score bound (only meaningful when active)
    float pre;                     // logits <= pre cannot reach it
    int active;
    int pad;
};

struct FilterParams {
    int num_levels;
    int num_classes;
    int has_ctr;
    int topk;
    int total_chunks;                       // per image
    int chunk_start[OD_MAX_LEVELS + 1];     // first chunk id of each level
    long long level_elems[OD_MAX_LEVELS];   // N_l * C
    long long key_off[OD_MAX_LEVELS];       // first key slot of the level inside an image's key array
    long long keys_per_image;
    const float* cls[OD_MAX_LEVELS];
    long long cls_stride[OD_MAX_LEVELS];
    const float* ctr[OD_MAX_LEVELS];
    long long ctr_stride[OD_MAX_LEVELS];
    float thresh;
    float pre_logit;                        // logits <= pre_logit can never pass the threshold
    int prefilter;                          // 0: threshold <= 0, every logit is scored
    // Channel-major levels (the convolution's own [batch, A*C, H*W] layout): plane = H*W, chans = A*C; the element
    // at memory position ch*plane + hw has flat index hw*chans + ch.  plane = 0: anchor-major [batch, N_l, C].
    unsigned int plane[OD_MAX_LEVELS];
    unsigned int chans[OD_MAX_LEVELS];
    int defer[OD_MAX_LEVELS];               // 1: the level is filtered by filter_defer_kernel (DEFER_MIN logits or more)
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
    long long key_off[OD_MAX_LEVELS];
    long long keys_per_image;
    unsigned int plane[OD_MAX_LEVELS];      // channel-major regression [batch, A*4, plane]; 0: [batch, N_l, 4]
    unsigned int apl[OD_MAX_LEVELS];        // A: anchors per location
};

// Flat index (anchor * C + class, anchors in (y, x, a) order) of the logit at memory position `pos` of its level.
__device__ __forceinline__ uint32_t flat_of(const FilterParams& p, int l, uint32_t pos) {
    const uint32_t plane = p.plane[l];
    if (plane == 0) return pos;
    const uint32_t ch = pos / plane;
    return (pos - ch * plane) * p.chans[l] + ch;
}

__device__ __forceinline__ float score_of(const FilterParams& p, int l, int b, float x, uint32_t flat) {
    float s = spec_sigmoidf(x);
    if (p.has_ctr) {
        const uint32_t a = flat / (uint32_t)p.num_classes;
        long long at = a;                                        // position of the anchor's centre-ness value
        if (p.plane[l]) {
            const uint32_t apl = p.chans[l] / (uint32_t)p.num_classes;
            const uint32_t hw = a / apl;
            at = (long long)(a - hw * apl) * p.plane[l] + hw;
        }
        float c = p.ctr[l][(long long)b * p.ctr_stride[l] + at];
        s = sqrtf(s * spec_sigmoidf(c));
    }
    return s;
}

__device__ __forceinline__ int wave_inclusive_sum(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// Every wave tests its 2048 logits (in registers) against the prefilter bound.  The workgroup reserves one
// slot of the (image, level) key array per remaining logit with ONE global atomic -- issued before the
// scoring, so that its round trip overlaps it; each wave moves its logits to a queue in LDS, scores the
// queue 64 entries at a time at a single sigmoid site, and copies the keys out.  A queued logit that fails
// the exact threshold leaves a zero key in its slot (rare: the prefilter bound is within 0.1 % of the
// threshold); the select kernel ignores zero keys.  The kernel is a stream: everything a wave does after its
// loads is kept free of further memory round trips, which under load cost as much as the loads themselves.
// One 8192-logit chunk of image b, by one workgroup of ST threads (all of them call it with the same arguments).
__device__ __forceinline__ void filter_chunk(const FilterParams& p, unsigned long long* __restrict__ keys,
                                             unsigned int* __restrict__ key_count,
                                             unsigned int* __restrict__ key_valid,
                                             unsigned int* __restrict__ key_count_redo,
                                             const CutInfo* __restrict__ cuts, const int redo, const int chunk,
                                             const int b) {
    constexpr int ITERS = CHUNK / (ST * 4);
    constexpr int QCAP = 512;                      // queue entries per wave and pass
    static_assert(ITERS == 8, "eight 16-byte loads per thread; the select tree picks one of 32 values");
    __shared__ unsigned long long s_queue[ST / 64][QCAP];
    __shared__ int s_tot[ST / 64];
    __shared__ unsigned int s_base;
    __shared__ unsigned int s_valid;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int l = 0;
#pragma unroll
    for (int i = 1; i < OD_MAX_LEVELS; ++i)
        if (i < p.num_levels && chunk >= p.chunk_start[i]) l = i;
    if (p.defer[l]) return;                        // very large levels: filter_defer_kernel
    const long long slot = (long long)b * p.num_levels + l;
    const CutInfo ci = cuts[slot];
    float pre = p.pre_logit, cut = -__builtin_inff();
    int prefilter = p.prefilter;
    if (redo) {
        // second pass: only levels whose sampled cut left fewer than K valid keys, this time without the cut
        if (!(ci.active && key_valid[slot * CSTRIDE] < (unsigned int)p.topk)) return;
    } else if (ci.active) {
        pre = ci.pre;
        cut = ci.cut;
        prefilter = 1;
    }
    if (tid == 0) s_valid = 0u;
    const long long E = p.level_elems[l];
    const long long e0 = (long long)(chunk - p.chunk_start[l]) * CHUNK;
    const float* src = p.cls[l] + (long long)b * p.cls_stride[l];
    const bool vec_ok = (((uintptr_t)(src + e0)) & 15) == 0;
    const long long rem = E - e0;
    const int len = rem < CHUNK ? (int)rem : CHUNK;

    // thread t owns elements it*1024 + t*4 + k, held in v<it>.<k>; all loads are requested before the first
    // use.  (Named registers, no array: an array read through selects would be placed in scratch memory.)
    typedef float vec4f __attribute__((ext_vector_type(4)));
    vec4f v0, v1, v2, v3, v4, v5, v6, v7;
    if (vec_ok && len == CHUNK) {
        const vec4f* vp = reinterpret_cast<const vec4f*>(src + e0 + tid * 4);
        // non-temporal: the logits are read once; ordinary loads would push the other streams' working set (the
        // anchors every image shares, the key arrays) out of the caches -- alone the kernel is 3 us slower this way
        // (re-read logits no longer come from the Infinity Cache), inside the overlapped step everything else gains
        v0 = __builtin_nontemporal_load(vp + 0 * ST); v1 = __builtin_nontemporal_load(vp + 1 * ST);
        v2 = __builtin_nontemporal_load(vp + 2 * ST); v3 = __builtin_nontemporal_load(vp + 3 * ST);
        v4 = __builtin_nontemporal_load(vp + 4 * ST); v5 = __builtin_nontemporal_load(vp + 5 * ST);
        v6 = __builtin_nontemporal_load(vp + 6 * ST); v7 = __builtin_nontemporal_load(vp + 7 * ST);
    } else {
        auto ld4 = [&](int it) {
            vec4f r;
            const int off = it * ST * 4 + tid * 4;
            r.x = (off + 0 < len) ? src[e0 + off + 0] : -__builtin_inff();
            r.y = (off + 1 < len) ? src[e0 + off + 1] : -__builtin_inff();
            r.z = (off + 2 < len) ? src[e0 + off + 2] : -__builtin_inff();
            r.w = (off + 3 < len) ? src[e0 + off + 3] : -__builtin_inff();
            return r;
        };
        v0 = ld4(0); v1 = ld4(1); v2 = ld4(2); v3 = ld4(3); v4 = ld4(4); v5 = ld4(5); v6 = ld4(6); v7 = ld4(7);
    }
    // Which elements may reach the threshold is decided per element for the whole wave: one compare gives the lane mask
    // of element e (a ballot, in scalar registers), its population count is scalar arithmetic, and an element that no
    // lane passes -- nearly all of them under a sampled cut -- costs nothing more; the lanes that pass write (logit,
    // position) straight to the wave's queue.  (Collecting a bit mask per lane first and then picking each lane's
    // elements out of its 32 registers with a select tree cost 250-370 vector instructions per wave.)
    const bool every = !prefilter;                  // threshold <= 0: every element inside the chunk is scored
    const bool partial = len < CHUNK;               // the level's last chunk: elements past its end are never candidates
    unsigned long long* queue = s_queue[wave];
    int pos = 0;                                    // wave-uniform: candidates met so far
    // FAST: the usual chunk (a bound on the logit exists, the chunk is whole) tests one compare per element
    auto emit_all = [&](auto fast) {
        constexpr bool FAST = decltype(fast)::value;
        auto emit1 = [&](float x, int e) {
            bool c;
            if constexpr (FAST) {
                c = !(x <= pre);
            } else {
                c = every || !(x <= pre);
                if (partial) c = c && ((e >> 2) * (ST * 4) + tid * 4 + (e & 3) < len);
            }
            const unsigned long long m = __ballot(c);
            if (m != 0ull) {
                const int n = __popcll(m);
                if (c && pos + n <= QCAP) {         // a queue that overflows is refilled group by group below
                    const unsigned int off = (unsigned int)((e >> 2) * (ST * 4) + tid * 4 + (e & 3));
                    queue[pos + mbcnt(m)] = ((unsigned long long)f2u(x) << 32) | off;     // (logit, position in the chunk)
                }
                pos += n;
            }
        };
        auto emit4 = [&](const vec4f& v, int it) {
            emit1(v.x, it * 4 + 0); emit1(v.y, it * 4 + 1); emit1(v.z, it * 4 + 2); emit1(v.w, it * 4 + 3);
        };
        emit4(v0, 0); emit4(v1, 1); emit4(v2, 2); emit4(v3, 3); emit4(v4, 4); emit4(v5, 5); emit4(v6, 6); emit4(v7, 7);
    };
    if (!every && !partial) emit_all(std::true_type{}); else emit_all(std::false_type{});
    const int all_total = pos;
    if (lane == 0) s_tot[wave] = all_total;
    __syncthreads();
    int wg_total = 0, before_me = 0;
    bool one_pass = true;
#pragma unroll
    for (int w = 0; w < ST / 64; ++w) {
        const int t = s_tot[w];
        wg_total += t;
        if (w < wave) before_me += t;
        one_pass = one_pass && t <= QCAP;
    }
    if (wg_total == 0) return;                     // nothing in this chunk can pass

    unsigned long long* dst = keys + (long long)b * p.keys_per_image + p.key_off[l];
    unsigned int* counter = (redo ? key_count_redo : key_count) + slot * CSTRIDE;
    // every lane scores its own queue entry and overwrites it with the key; returns the valid keys of the wave
    auto score_queue = [&](int total) -> int {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the queue is read by other lanes of this wave
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int nvalid = 0;
        for (int i0 = 0; i0 < total; i0 += 64) {
            bool ok = false;
            if (i0 + lane < total) {
                const unsigned long long ent = queue[i0 + lane];
                const uint32_t flat = flat_of(p, l, (uint32_t)e0 + (uint32_t)ent);
                const float sc = score_of(p, l, b, u2f((uint32_t)(ent >> 32)), flat);
                ok = sc > p.thresh && sc >= cut;
                queue[i0 + lane] = ok ? (((unsigned long long)f2u(sc) << 32) | (0xFFFFFFFFull - (unsigned long long)flat)) : 0ull;
            }
            nvalid += __popcll(__ballot(ok));
        }
        return nvalid;
    };
    if (one_pass) {
        // Usual case: every wave's queue holds all its candidates; the workgroup reserves its slots with ONE atomic,
        // requested before the scoring, so that its round trip overlaps it.
        unsigned int wg_base = 0;
        if (tid == 0) wg_base = atomicAdd(counter, (unsigned int)wg_total);
        const int nvalid = score_queue(all_total);
        // valid keys per (image, level): decides after the kernel whether the sampled cut was low enough
        if (lane == 0 && nvalid) atomicAdd(&s_valid, (unsigned int)nvalid);
        if (tid == 0) s_base = wg_base;
        __syncthreads();
        const unsigned int base = s_base + (unsigned int)before_me;
        if (tid == 0 && !redo && s_valid) atomicAdd(key_valid + slot * CSTRIDE, s_valid);
        for (int i = lane; i < all_total; i += 64) dst[base + i] = queue[i];   // own entries: no cross-lane hazard
        return;
    }
    // Dense chunk (no cut, low threshold): some wave has more candidates than its queue holds.  Every wave then makes
    // four passes of eight elements (at most 64 * 8 = QCAP entries each) with a slot reservation of its own per pass.
    // (the two registers of a pass are named at the call: picking them by a run-time pass number would move all
    // eight to scratch memory)
    auto dense_pass = [&](const vec4f& ga, const vec4f& gc, int it0) {
        __builtin_amdgcn_wave_barrier();           // the previous pass has copied its keys out of the queue
        pos = 0;
        auto e1 = [&](float x, int e) {
            const unsigned int off = (unsigned int)((e >> 2) * (ST * 4) + tid * 4 + (e & 3));
            bool cnd = every || !(x <= pre);
            if (partial) cnd = cnd && ((int)off < len);
            const unsigned long long m = __ballot(cnd);
            if (cnd) queue[pos + mbcnt(m)] = ((unsigned long long)f2u(x) << 32) | off;
            pos += __popcll(m);
        };
        e1(ga.x, it0 * 4 + 0); e1(ga.y, it0 * 4 + 1); e1(ga.z, it0 * 4 + 2); e1(ga.w, it0 * 4 + 3);
        e1(gc.x, it0 * 4 + 4); e1(gc.y, it0 * 4 + 5); e1(gc.z, it0 * 4 + 6); e1(gc.w, it0 * 4 + 7);
        const int total = pos;
        unsigned int base = 0;
        if (total > 0 && lane == 0) base = atomicAdd(counter, (unsigned int)total);
        const int nvalid = score_queue(total);
        base = (unsigned int)__builtin_amdgcn_readlane((int)base, 0);
        if (lane == 0 && !redo && nvalid) atomicAdd(key_valid + slot * CSTRIDE, (unsigned int)nvalid);
        for (int i = lane; i < total; i += 64) dst[base + i] = queue[i];
    };
    dense_pass(v0, v1, 0); dense_pass(v2, v3, 2); dense_pass(v4, v5, 4); dense_pass(v6, v7, 6);
}

__global__ __launch_bounds__(ST) void filter_kernel(FilterParams p, unsigned long long* __restrict__ keys,
                                                    unsigned int* __restrict__ key_count,
                                                    unsigned int* __restrict__ key_valid,
                                                    unsigned int* __restrict__ key_count_redo,
                                                    const CutInfo* __restrict__ cuts, int chunk_base) {
    // chunk_base: the chunks of leading deferred levels are not launched
    filter_chunk(p, keys, key_count, key_valid, key_count_redo, cuts, 0, (int)blockIdx.x + chunk_base, (int)blockIdx.y);
}

// Second pass over the levels whose sampled cut left fewer than K valid keys, without the cut.  That almost never
// happens, so the launch is a narrow one -- REDO_WG workgroups per image, which look at the image's levels and leave at
// once when none needs the pass (a launch of one workgroup per chunk cost the chip 4-6 us per call to find that out);
// when a level does need it, the image's workgroups walk that level's chunks in turn.
constexpr int REDO_WG = 24;
__global__ __launch_bounds__(ST) void filter_redo_kernel(FilterParams p, unsigned long long* __restrict__ keys,
                                                         unsigned int* __restrict__ key_count,
                                                         unsigned int* __restrict__ key_valid,
                                                         unsigned int* __restrict__ key_count_redo,
                                                         const CutInfo* __restrict__ cuts, int total_chunks) {
    const int b = blockIdx.y;
    bool need[OD_MAX_LEVELS];
    bool any = false;
#pragma unroll
    for (int l = 0; l < OD_MAX_LEVELS; ++l) {
        need[l] = false;
        if (l < p.num_levels && !p.defer[l]) {
            const long long slot = (long long)b * p.num_levels + l;
            need[l] = cuts[slot].active && key_valid[slot * CSTRIDE] < (unsigned int)p.topk;
        }
        any = any || need[l];
    }
    if (!any) return;
#pragma unroll
    for (int l = 0; l < OD_MAX_LEVELS; ++l) {
        if (!need[l]) continue;
        const int end = (l + 1 < p.num_levels) ? p.chunk_start[l + 1] : total_chunks;
        for (int chunk = p.chunk_start[l] + (int)blockIdx.x; chunk < end; chunk += REDO_WG) {
            filter_chunk(p, keys, key_count, key_valid, key_count_redo, cuts, 1, chunk, b);
            __syncthreads();                       // the queues and counters in LDS are reused by the next chunk
        }
    }
}

// The same filter for very large levels (DEFER_MIN logits or more; one launch per such level).  There the sampled
// cut leaves a key or two per 8192-logit chunk, and one slot reservation per chunk -- a returning atomic on the
// level's single counter line, ~13 ns apart -- takes longer than reading the level (14 746 chunks of a 1536 x 1536,
// 91-class finest level: ~190 us).  A workgroup here walks SUBS consecutive chunks, scores each chunk's candidates
// at once but keeps the keys in the per-wave LDS queues, and reserves slots for all of them together: when a queue
// could not take the next chunk's candidates, and at the end.  A chunk with more candidates than a queue holds
// (dense input, no cut) is handled like in filter_kernel: per wave, four passes, a reservation each.
constexpr long long DEFER_MIN = 8ll << 20;
constexpr int SUBS = 8;

__global__ __launch_bounds__(ST) __attribute__((amdgpu_waves_per_eu(4))) void filter_defer_kernel(FilterParams p, int l, unsigned long long* __restrict__ keys,
                                                          unsigned int* __restrict__ key_count,
                                                          unsigned int* __restrict__ key_valid,
                                                          unsigned int* __restrict__ key_count_redo,
                                                          const CutInfo* __restrict__ cuts, int redo) {
    constexpr int NWV = ST / 64;
    constexpr int QCAP = 512;
    static_assert(CHUNK / (ST * 4) == 8, "eight 16-byte loads per thread; the select tree picks one of 32 values");
    __shared__ unsigned long long s_queue[NWV][QCAP];
    __shared__ int s_tot[2][NWV];
    __shared__ unsigned int s_base;
    __shared__ unsigned int s_valid;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const long long slot = (long long)b * p.num_levels + l;
    const CutInfo ci = cuts[slot];
    float pre = p.pre_logit, cut = -__builtin_inff();
    int prefilter = p.prefilter;
    if (redo) {
        if (!(ci.active && key_valid[slot * CSTRIDE] < (unsigned int)p.topk)) return;
    } else if (ci.active) {
        pre = ci.pre;
        cut = ci.cut;
        prefilter = 1;
    }
    if (tid == 0) s_valid = 0u;
    const long long E = p.level_elems[l];
    const float* src = p.cls[l] + (long long)b * p.cls_stride[l];
    unsigned long long* dst = keys + (long long)b * p.keys_per_image + p.key_off[l];
    unsigned int* counter = (redo ? key_count_redo : key_count) + slot * CSTRIDE;
    unsigned long long* queue = s_queue[wave];
    typedef float vec4f __attribute__((ext_vector_type(4)));

    int held[NWV];                                 // keys waiting in every wave's queue (the same values in all waves)
#pragma unroll
    for (int w = 0; w < NWV; ++w) held[w] = 0;
    int nvalid_acc = 0;                            // valid keys of this wave since the last flush

    // one reservation for everything the four queues hold, then the copy-out (all threads of the workgroup)
    auto flush = [&]() {
        int sum = 0, before_me = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { sum += held[w]; if (w < wave) before_me += held[w]; }
        if (sum == 0) return;
        if (lane == 0 && nvalid_acc) atomicAdd(&s_valid, (unsigned int)nvalid_acc);
        nvalid_acc = 0;
        __syncthreads();
        if (tid == 0) {
            s_base = atomicAdd(counter, (unsigned int)sum);
            const unsigned int nv = s_valid;
            s_valid = 0u;
            if (!redo && nv) atomicAdd(key_valid + slot * CSTRIDE, nv);
        }
        __syncthreads();
        const unsigned int base = s_base + (unsigned int)before_me;
        int mine = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) if (w == wave) mine = held[w];
        for (int i = lane; i < mine; i += 64) dst[base + i] = queue[i];
#pragma unroll
        for (int w = 0; w < NWV; ++w) held[w] = 0;
    };

    const long long first_chunk = (long long)blockIdx.x * SUBS;
    for (int sub = 0; sub < SUBS; ++sub) {
        const long long e0 = (first_chunk + sub) * CHUNK;
        if (e0 >= E) break;                                       // uniform
        const bool vec_ok = (((uintptr_t)(src + e0)) & 15) == 0;
        const long long rem = E - e0;
        const int len = rem < CHUNK ? (int)rem : CHUNK;
        vec4f v0, v1, v2, v3, v4, v5, v6, v7;
        if (vec_ok && len == CHUNK) {
            const vec4f* vp = reinterpret_cast<const vec4f*>(src + e0 + tid * 4);
            v0 = __builtin_nontemporal_load(vp + 0 * ST); v1 = __builtin_nontemporal_load(vp + 1 * ST);
            v2 = __builtin_nontemporal_load(vp + 2 * ST); v3 = __builtin_nontemporal_load(vp + 3 * ST);
            v4 = __builtin_nontemporal_load(vp + 4 * ST); v5 = __builtin_nontemporal_load(vp + 5 * ST);
            v6 = __builtin_nontemporal_load(vp + 6 * ST); v7 = __builtin_nontemporal_load(vp + 7 * ST);
        } else {
            auto ld4 = [&](int it) {
                vec4f r;
                const int off = it * ST * 4 + tid * 4;
                r.x = (off + 0 < len) ? src[e0 + off + 0] : -__builtin_inff();
                r.y = (off + 1 < len) ? src[e0 + off + 1] : -__builtin_inff();
                r.z = (off + 2 < len) ? src[e0 + off + 2] : -__builtin_inff();
                r.w = (off + 3 < len) ? src[e0 + off + 3] : -__builtin_inff();
                return r;
            };
            v0 = ld4(0); v1 = ld4(1); v2 = ld4(2); v3 = ld4(3); v4 = ld4(4); v5 = ld4(5); v6 = ld4(6); v7 = ld4(7);
        }
        // candidates per element of the whole wave: one compare and a ballot each (see filter_chunk); counted first,
        // because the workgroup has to know whether the queues can take them, then written behind what the queue holds
        const bool every = !prefilter;
        const bool partial = len < CHUNK;
        auto passes_bound = [&](auto fast, float x, int e) -> bool {
            if constexpr (decltype(fast)::value) {
                return !(x <= pre);
            } else {
                bool c = every || !(x <= pre);
                if (partial) c = c && ((e >> 2) * (ST * 4) + tid * 4 + (e & 3) < len);
                return c;
            }
        };
        auto count_all = [&](auto fast) -> int {
            int t = 0;
            auto c4 = [&](const vec4f& v, int it) {
                t += __popcll(__ballot(passes_bound(fast, v.x, it * 4 + 0))) + __popcll(__ballot(passes_bound(fast, v.y, it * 4 + 1))) +
                     __popcll(__ballot(passes_bound(fast, v.z, it * 4 + 2))) + __popcll(__ballot(passes_bound(fast, v.w, it * 4 + 3)));
            };
            c4(v0, 0); c4(v1, 1); c4(v2, 2); c4(v3, 3); c4(v4, 4); c4(v5, 5); c4(v6, 6); c4(v7, 7);
            return t;
        };
        const bool usual = !every && !partial;
        const int my_total = usual ? count_all(std::true_type{}) : count_all(std::false_type{});
        if (lane == 0) s_tot[sub & 1][wave] = my_total;
        __syncthreads();                           // (two buffers: the next chunk writes the other one)
        int tot[NWV];
        bool fits = true, dense = false;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            tot[w] = s_tot[sub & 1][w];
            fits = fits && (held[w] + tot[w] <= QCAP);
            dense = dense || (tot[w] > QCAP);
        }
        if (!fits) flush();                        // uniform: every wave sees the same numbers
        // appends the (logit, position in the chunk) entries of all 32 elements from queue position `from` on
        auto fill_all = [&](auto fast, int from) {
            int pos = from;
            auto e1 = [&](float x, int e) {
                const bool c = passes_bound(fast, x, e);
                const unsigned long long m = __ballot(c);
                if (m != 0ull) {
                    const unsigned int off = (unsigned int)((e >> 2) * (ST * 4) + tid * 4 + (e & 3));
                    if (c) queue[pos + mbcnt(m)] = ((unsigned long long)f2u(x) << 32) | off;
                    pos += __popcll(m);
                }
            };
            auto e4 = [&](const vec4f& v, int it) {
                e1(v.x, it * 4 + 0); e1(v.y, it * 4 + 1); e1(v.z, it * 4 + 2); e1(v.w, it * 4 + 3);
            };
            e4(v0, 0); e4(v1, 1); e4(v2, 2); e4(v3, 3); e4(v4, 4); e4(v5, 5); e4(v6, 6); e4(v7, 7);
        };
        // the same for the eight elements of two registers (one pass over a dense chunk), from position 0; returns
        // their number.  The registers are named at the call: picking them by a run-time pass number would move all
        // eight to scratch memory.
        auto fill_group = [&](const vec4f& ga, const vec4f& gc, int it0) -> int {
            int pos = 0;
            auto e1 = [&](float x, int e) {
                const unsigned int off = (unsigned int)((e >> 2) * (ST * 4) + tid * 4 + (e & 3));
                bool c = every || !(x <= pre);
                if (partial) c = c && ((int)off < len);
                const unsigned long long m = __ballot(c);
                if (c) queue[pos + mbcnt(m)] = ((unsigned long long)f2u(x) << 32) | off;
                pos += __popcll(m);
            };
            e1(ga.x, it0 * 4 + 0); e1(ga.y, it0 * 4 + 1); e1(ga.z, it0 * 4 + 2); e1(ga.w, it0 * 4 + 3);
            e1(gc.x, it0 * 4 + 4); e1(gc.y, it0 * 4 + 5); e1(gc.z, it0 * 4 + 6); e1(gc.w, it0 * 4 + 7);
            return pos;
        };
        // scores the queue entries [from, from + n) in place: key, or 0 for a logit that fails the exact test
        auto score = [&](int from, int n) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the queue is read by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int nvalid = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                bool ok = false;
                if (i0 + lane < n) {
                    const unsigned long long ent = queue[from + i0 + lane];
                    const uint32_t flat = flat_of(p, l, (uint32_t)e0 + (uint32_t)ent);
                    const float sc = score_of(p, l, b, u2f((uint32_t)(ent >> 32)), flat);
                    ok = sc > p.thresh && sc >= cut;
                    queue[from + i0 + lane] = ok ? (((unsigned long long)f2u(sc) << 32) | (0xFFFFFFFFull - (unsigned long long)flat)) : 0ull;
                }
                nvalid += __popcll(__ballot(ok));
            }
            return nvalid;
        };
        if (!dense) {
            int mine = 0, had = 0;
#pragma unroll
            for (int w = 0; w < NWV; ++w) if (w == wave) { mine = tot[w]; had = held[w]; }
            if (usual) fill_all(std::true_type{}, had); else fill_all(std::false_type{}, had);
            nvalid_acc += score(had, mine);
#pragma unroll
            for (int w = 0; w < NWV; ++w) held[w] += tot[w];
        } else {
            // a chunk some wave cannot hold: nothing is deferred (the queues are empty here, see below); every wave
            // makes four passes of 8 element slots (at most 64 * 8 = QCAP entries) with a reservation of its own
            flush();                               // uniform; empties the queues if `fits` held above
            auto dense_pass = [&](const vec4f& ga, const vec4f& gc, int it0) {
                __builtin_amdgcn_wave_barrier();   // the previous pass has copied its keys out of the queue
                const int total = fill_group(ga, gc, it0);
                unsigned int base = 0;
                if (total > 0 && lane == 0) base = atomicAdd(counter, (unsigned int)total);
                const int nvalid = score(0, total);
                base = (unsigned int)__builtin_amdgcn_readlane((int)base, 0);
                if (lane == 0 && !redo && nvalid) atomicAdd(key_valid + slot * CSTRIDE, (unsigned int)nvalid);
                for (int i = lane; i < total; i += 64) dst[base + i] = queue[i];
            };
            dense_pass(v0, v1, 0); dense_pass(v2, v3, 2); dense_pass(v4, v5, 4); dense_pass(v6, v7, 6);
        }
    }
    flush();
}

// ---- block-wide helpers for the select kernel ----------------------------------------------------
// Finds, from the top, the histogram bin in which the running count reaches `need`.
// Returns (bin, keys in the bins above it, keys in it) through s_tmp[16..18] and found = 1 in s_tmp[19];
// if the histogram holds fewer than `need` keys, s_tmp[19] = 0 and s_tmp[17] = its total.
// All threads must call it.
__device__ __forceinline__ void find_cut_bin(const unsigned int* s_hist, int nb, int need, int* s_tmp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nb + SEL_T - 1) / SEL_T;     // 4 when nb = 4096, else 1 (nb <= 1024)
    int local = 0;
    for (int i = 0; i < per; ++i) { const int bin = tid * per + i; if (bin < nb) local += (int)s_hist[bin]; }
    int suf = local;                              // inclusive suffix sum over the lanes of the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(suf, o); if (lane + o < 64) suf += t; }
    if (lane == 0) s_tmp[wave] = suf;             // wave total
    __syncthreads();
    int higher = 0;                               // keys in waves above this one
    for (int i = wave + 1; i < SEL_T / 64; ++i) higher += s_tmp[i];
    const int above_me = higher + suf - local;    // keys in bins strictly above my bins
    if (tid == 0) {                               // thread 0 owns the lowest bins: above_me + local = all keys
        const int all = above_me + local;
        s_tmp[19] = all >= need ? 1 : 0;
        if (all < need) { s_tmp[16] = 0; s_tmp[17] = all; s_tmp[18] = 0; }
    }
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
}

// One workgroup per (image, level).  Levels with at least SAMPLE_MIN logits: S logits are read in runs of SAMPLE_RUN
// at hashed positions (one run per window of E / runs logits) and the r-th best of them, r = 3*K*S/E + 32, is located
// with one pass over a 4096-bin histogram.  About 3K + 32*E/S logits of the level are at least as good, so with
// overwhelming probability at least K are -- and if not, the filter's redo pass repairs it.
// Without centre-ness the score is a non-decreasing function of the logit (up to the last bits of the specified
// sigmoid), so the samples are ranked by their logits and only the located bound is put through the sigmoid; with
// centre-ness every sample is scored.  The bounds handed to the filter carry margins that cover those last bits and
// the rounding of logf: a bound that is too low only lets more keys through.
// Sample of a level: S logits in runs of SAMPLE_RUN, one run per window of E / runs logits at a hashed offset; the
// r-th best sample is the bound.  Returns false if the level gets no bound (small level, or the bound would remove
// little).
struct SampleGeometry { long long S; uint32_t window, spread; int r; };
__device__ __forceinline__ bool sample_geometry(long long E, int topk, SampleGeometry* g) {
    if (E < SAMPLE_MIN) return false;
    long long S = (E / 64) & ~16383ll;
    S = S < 16384 ? 16384 : (S > SAMPLE_MAX ? SAMPLE_MAX : S);
    const long long runs = S / SAMPLE_RUN;
    const long long window = E / runs;                         // >= 512 logits
    const long long r64 = (3ll * topk * S + E - 1) / E + 32;
    if (r64 > S / 4) return false;                             // the bound would remove little
    g->S = S; g->window = (uint32_t)window; g->spread = (uint32_t)window - (SAMPLE_RUN - 1); g->r = (int)r64;
    return true;
}
constexpr int SU = 16;                                         // samples per thread whose loads are in flight together
constexpr long long SAMPLE_IN_REGS = 2 * SEL_T * SU;           // larger samples are gathered by several workgroups
constexpr int SAMPLE_PART = SEL_T * SU;                        // samples per workgroup of sample_part_kernel
constexpr int SAMPLE_PARTS_MAX = (int)(SAMPLE_MAX / SAMPLE_PART);

// Histogram bins of a sample.  Without centre-ness the samples are ranked by their logits: bin = 128 * (logit -
// origin), origin = the prefilter bound (or -16 when every logit is scored), the last bin also takes everything
// above origin + 32; with centre-ness by their scores: 4096 bins over (threshold, 1].  The bound that comes out is the
// lower edge of a bin, i.e. it is located to 1/128 in the logit (0.2 % in the score) -- the margins applied to it
// afterwards are wider, and a bound that is a little low only lets a few more keys through.
constexpr int BIN_NONE = 0xFFFF;
struct SampleBins { float origin, scale; };
__device__ __forceinline__ SampleBins sample_bins(const FilterParams& p) {
    SampleBins sb;
    if (p.has_ctr) { sb.origin = p.thresh > 0.0f ? p.thresh : 0.0f; sb.scale = (float)HBINS / (1.0f - sb.origin); }
    else { sb.origin = p.prefilter ? p.pre_logit : -16.0f; sb.scale = 128.0f; }
    return sb;
}
__device__ __forceinline__ int bin_of(const SampleBins& sb, float v) {
    const float t = (v - sb.origin) * sb.scale;
    return t >= (float)(HBINS - 1) ? HBINS - 1 : (t > 0.0f ? (int)t : 0);      // NaN -> 0
}
__device__ __forceinline__ float bin_edge(const SampleBins& sb, int bin) { return sb.origin + (float)bin / sb.scale; }

// bins (BIN_NONE: cannot pass the threshold) of the samples base + u * SEL_T + tid, u < NU; all NU scattered loads of a
// thread are requested before the first is used
template <int NU>
__device__ __forceinline__ void sample_keys(const FilterParams& p, int l, int b, const float* __restrict__ src,
                                            const SampleGeometry& g, const SampleBins& sb, uint32_t base, int* bin) {
    const int tid = threadIdx.x;
    auto position = [&](int u) -> uint32_t {
        const uint32_t i = base + u * SEL_T + tid;
        const uint32_t run = i / SAMPLE_RUN;
        return run * g.window + __umulhi(run * 2654435761u, g.spread) + (i % SAMPLE_RUN);
    };
    float x[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) x[u] = src[position(u)];
    // two loops, not one with the test inside: unrolled, the centre-ness path (a dependent load and two sigmoids per
    // sample) is hundreds of instructions per sample that the plain path would have to jump over NU times
    if (p.has_ctr) {
#pragma unroll 1
        for (int u = 0; u < NU; ++u) {
            bin[u] = BIN_NONE;
            if (p.prefilter && x[u] <= p.pre_logit) continue;
            const float sc = score_of(p, l, b, x[u], flat_of(p, l, position(u)));
            if (sc > p.thresh) bin[u] = bin_of(sb, sc);
        }
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) bin[u] = (p.prefilter && x[u] <= p.pre_logit) ? BIN_NONE : bin_of(sb, x[u]);
    }
}

// Scratch of a large sample, in the level's own key region (free until the filter runs; parts * 16 KiB of its E * 8
// bytes): one 4096-bin histogram per part.
__device__ __forceinline__ uint32_t* sample_stash(const FilterParams& p, unsigned long long* keys, int l, int b) {
    return reinterpret_cast<uint32_t*>(keys + (long long)b * p.keys_per_image + p.key_off[l]);
}

// Gathering a large sample is bound by the latency of its scattered loads (one batch of 16 per thread takes ~5 us,
// and one workgroup needs S / 16384 batches): the parts of such a sample are gathered by a workgroup each, which
// leaves a partial histogram for sample_kernel.
__global__ __launch_bounds__(SEL_T) void sample_part_kernel(FilterParams p, unsigned long long* __restrict__ keys) {
    __shared__ unsigned int s_hist[HBINS];
    const int tid = threadIdx.x;
    const int part = blockIdx.x, l = blockIdx.y, b = blockIdx.z;
    SampleGeometry g;
    if (!sample_geometry(p.level_elems[l], p.topk, &g)) return;
    if (g.S <= SAMPLE_IN_REGS || (long long)part * SAMPLE_PART >= g.S) return;
    for (int i = tid; i < HBINS; i += SEL_T) s_hist[i] = 0;
    __syncthreads();
    const float* src = p.cls[l] + (long long)b * p.cls_stride[l];
    const SampleBins sb = sample_bins(p);
    int bin[SU];
    sample_keys<SU>(p, l, b, src, g, sb, (uint32_t)part * SAMPLE_PART, bin);
#pragma unroll
    for (int u = 0; u < SU; ++u)
        if (bin[u] != BIN_NONE) atomicAdd(&s_hist[bin[u]], 1u);
    __syncthreads();
    uint32_t* hist = sample_stash(p, keys, l, b) + (long long)part * HBINS;
    for (int i = tid; i < HBINS; i += SEL_T) hist[i] = s_hist[i];
}

__global__ __launch_bounds__(SEL_T) void sample_kernel(FilterParams p, CutInfo* __restrict__ cuts,
                                                       unsigned int* __restrict__ counters, long long counter_words,
                                                       unsigned long long* __restrict__ keys) {
    __shared__ unsigned int s_hist[HBINS];
    __shared__ int s_tmp[32];
    const int tid = threadIdx.x;
    const int l = blockIdx.x, b = blockIdx.y;
    const long long E = p.level_elems[l];
    // this (image, level)'s three key counters start at zero (the filter launches that follow add to them)
    if (tid < 3) counters[tid * counter_words + ((long long)b * p.num_levels + l) * CSTRIDE] = 0u;
    CutInfo out;
    out.cut = 0.0f; out.pre = p.pre_logit; out.active = 0; out.pad = 0;
    SampleGeometry g;
    if (!sample_geometry(E, p.topk, &g)) {
        if (tid == 0) cuts[(long long)b * p.num_levels + l] = out;
        return;
    }
    const long long S = g.S;
    const float* src = p.cls[l] + (long long)b * p.cls_stride[l];
    constexpr int U = SU;
    const SampleBins sb = sample_bins(p);
    for (int i = tid; i < HBINS; i += SEL_T) s_hist[i] = 0;
    __syncthreads();
    if (S <= SAMPLE_IN_REGS) {                                 // up to 32 samples per thread, gathered here
        int bin[2 * U];
        if (S > SEL_T * U) {
            sample_keys<2 * U>(p, l, b, src, g, sb, 0u, bin);
        } else {
            sample_keys<U>(p, l, b, src, g, sb, 0u, bin);
#pragma unroll
            for (int u = 0; u < U; ++u) bin[U + u] = BIN_NONE;
        }
#pragma unroll
        for (int u = 0; u < 2 * U; ++u)
            if (bin[u] != BIN_NONE) atomicAdd(&s_hist[bin[u]], 1u);
    } else {                                                   // larger samples: the parts' histograms
        const uint32_t* stash = sample_stash(p, keys, l, b);
        const int parts = (int)(S / SAMPLE_PART);
        for (int i = tid; i < HBINS; i += SEL_T) {
            unsigned int c = 0;
            for (int q = 0; q < parts; ++q) c += stash[(long long)q * HBINS + i];
            s_hist[i] = c;
        }
    }
    __syncthreads();
    find_cut_bin(s_hist, HBINS, g.r, s_tmp);
    const int bin1 = s_tmp[16], found = s_tmp[19];
    if (tid == 0) {
        // bin 0 of the logit scale also holds everything below the origin when nothing is prefiltered: no bound then
        if (found && (bin1 > 0 || p.has_ctr || p.prefilter)) {
            const float edge = bin_edge(sb, bin1);             // lower edge of the located bin
            // 0.9999: below the score of every logit >= the bound even where the specified sigmoid is not monotone
            const float cut = p.has_ctr ? edge : spec_sigmoidf(edge) * 0.9999f;
            if (cut > p.thresh && cut < 1.0f) {
                float c = cut * 0.996f;                        // below every true score that rounds to >= cut
                if (p.has_ctr) c = c * c;                      // sqrt(sig(cls) * sig(ctr)) >= cut needs sig(cls) >= cut^2
                const float lg = logf(c / (1.0f - c));
                float pre = lg - 0.01f * (1.0f + fabsf(lg));
                if (p.prefilter && pre < p.pre_logit) pre = p.pre_logit;
                out.cut = cut; out.pre = pre; out.active = 1;
            }
        }
        cuts[(long long)b * p.num_levels + l] = out;
    }
}

// Streams keys[0, total) through f(key) with 16 independent 16-byte loads per thread in flight: one
// workgroup reads the whole array, so its rate is (bytes in flight) / (memory latency).
// `keys` is 16-byte aligned; slots past `total` are passed as 0 (no valid key is 0).
template <typename F>
__device__ __forceinline__ void stream_keys(const unsigned long long* __restrict__ keys, unsigned int total, F f) {
    constexpr int U = 16;
    const ulonglong2* kv = reinterpret_cast<const ulonglong2*>(keys);
    const unsigned int nvec = (total + 1) >> 1;
    for (unsigned int v0 = 0; v0 < nvec; v0 += SEL_T * U) {
        ulonglong2 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned int v = v0 + u * SEL_T + threadIdx.x;
            k[u] = v < nvec ? kv[v] : make_ulonglong2(0ull, 0ull);
            if (2 * v + 1 >= total) k[u].y = 0ull;             // odd tail: the slot after the last key is not ours
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { f(k[u].x); f(k[u].y); }
    }
}

// Appends the keys of keys[0, total) that satisfy `pred` to dst[*counter ...), capacity cap, in no particular order.
// A wave reserves the slots for all keys of a batch (32 per lane) with ONE returning atomic: one per key instruction,
// as lds_append makes, is 36 dependent round trips to a counter that sixteen waves contend for when a level leaves
// 36 k keys (10 us of a 19 us descent).
template <typename P>
__device__ __forceinline__ void stream_append(const unsigned long long* __restrict__ keys, unsigned int total, P pred,
                                              unsigned long long* dst, int* counter, int cap) {
    constexpr int U = 16;
    const int lane = threadIdx.x & 63;
    const ulonglong2* kv = reinterpret_cast<const ulonglong2*>(keys);
    const unsigned int nvec = (total + 1) >> 1;
    for (unsigned int v0 = 0; v0 < nvec; v0 += SEL_T * U) {
        ulonglong2 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned int v = v0 + u * SEL_T + threadIdx.x;
            k[u] = v < nvec ? kv[v] : make_ulonglong2(0ull, 0ull);
            if (2 * v + 1 >= total) k[u].y = 0ull;             // odd tail: the slot after the last key is not ours
        }
        int mine = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) mine += (pred(k[u].x) ? 1 : 0) + (pred(k[u].y) ? 1 : 0);
        const int incl = wave_inclusive_sum(mine);
        const int wave_total = __builtin_amdgcn_readlane(incl, 63);
        if (wave_total == 0) continue;
        int base = 0;
        if (lane == 63) base = atomicAdd(counter, wave_total);
        int pos = __builtin_amdgcn_readlane(base, 63) + incl - mine;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pred(k[u].x)) { if (pos < cap) dst[pos] = k[u].x; ++pos; }
            if (pred(k[u].y)) { if (pos < cap) dst[pos] = k[u].y; ++pos; }
        }
    }
}

// wave-aggregated append of `k` (where take) to dst[*counter ...), capacity cap
__device__ __forceinline__ void lds_append(bool take, unsigned long long k, unsigned long long* dst, int* counter, int cap) {
    const int lane = threadIdx.x & 63;
    const uint64_t bal = __ballot(take);
    if (bal) {
        int base = 0;
        const int first = __ffsll((unsigned long long)bal) - 1;
        if (lane == first) base = atomicAdd(counter, __popcll(bal));
        base = __builtin_amdgcn_readlane(base, first);
        const int pos = base + mbcnt(bal);
        if (take && pos < cap) dst[pos] = k;
    }
}

// What the gather of a level reads through, fetched from the kernel's arguments once, at its start: the first read of
// an argument line is a memory round trip, which is better not left for the moment the boxes are needed.
struct LevelInputs {
    const float* reg;
    const float* anchors;
    long long reg_stride, anchor_stride;
    unsigned int plane, apl;
};
__device__ __forceinline__ LevelInputs level_inputs(const SelectParams& p, int l) {
    LevelInputs li;
    li.reg = p.reg[l]; li.anchors = p.anchors[l];
    li.reg_stride = p.reg_stride[l]; li.anchor_stride = p.anchor_stride[l];
    li.plane = p.plane[l]; li.apl = p.apl[l];
    asm volatile("" :: "s"(li.reg), "s"(li.anchors), "s"(li.reg_stride), "s"(li.anchor_stride), "s"(li.plane), "s"(li.apl));
    return li;
}

// The two inputs of a candidate's box: its regression values and its anchor (key = score bits << 32 | ~flat index).
__device__ __forceinline__ void gather_candidate(const LevelInputs& li, int num_classes, int b, unsigned long long key,
                                                 float4& r, float4& an) {
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)key;
    const uint32_t a = flat / (uint32_t)num_classes;
    if (li.plane == 0) {
        r = *reinterpret_cast<const float4*>(li.reg + ((long long)b * li.reg_stride + a) * 4);
    } else {                                                     // channel (a_i*4 + k) of location hw
        const uint32_t hw = a / li.apl;
        const float* rp = li.reg + (long long)b * li.reg_stride * 4 + (long long)(a - hw * li.apl) * 4 * li.plane + hw;
        r = make_float4(rp[0], rp[li.plane], rp[2ll * li.plane], rp[3ll * li.plane]);
    }
    an = *reinterpret_cast<const float4*>(li.anchors + ((long long)b * li.anchor_stride + a) * 4);
}

// Decodes, clips and writes the candidate of rank `rank` from its gathered inputs.
__device__ __forceinline__ void decode_store(const SelectParams& p, int l, int b, unsigned long long key, int rank,
                                             const float4 r, const float4 an,
                                             const int* __restrict__ image_hw, int num_levels,
                                             float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
                                             int* __restrict__ cand_labels, long long* __restrict__ cand_index) {
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)key;
    const float score = u2f((uint32_t)(key >> 32));
    const uint32_t a = flat / (uint32_t)p.num_classes;
    const int label = (int)(flat - a * (uint32_t)p.num_classes);
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

// Histogram increment with the lanes that share the first active lane's bin counted by one atomic: scores that
// saturate (thousands of candidates at exactly 1.0) otherwise serialise 64 atomics per instruction on one counter.
__device__ __forceinline__ void hist_add(unsigned int* s_hist, bool valid, unsigned int bin) {
    const uint64_t act = __ballot(valid);
    if (act == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int fl = __ffsll((unsigned long long)act) - 1;
    const unsigned int fb = (unsigned int)__builtin_amdgcn_readlane((int)bin, fl);
    const uint64_t same = __ballot(valid && bin == fb);
    if (lane == fl) atomicAdd(&s_hist[fb], (unsigned int)__popcll(same));
    if (valid && bin != fb) atomicAdd(&s_hist[bin], 1u);
}

__global__ __launch_bounds__(SEL_T) void select_kernel(SelectParams p, int num_levels,
                                                       const unsigned long long* __restrict__ all_keys,
                                                       const unsigned int* __restrict__ key_count,
                                                       const unsigned int* __restrict__ key_valid,
                                                       const unsigned int* __restrict__ key_count_redo,
                                                       const CutInfo* __restrict__ cuts,
                                                       const int* __restrict__ image_hw,
                                                       float* __restrict__ cand_boxes,
                                                       float* __restrict__ cand_scores,
                                                       int* __restrict__ cand_labels,
                                                       long long* __restrict__ cand_index,
                                                       int* __restrict__ cand_count) {
    __shared__ unsigned long long s_keys[CAP];
    __shared__ unsigned int s_hist[HBINS];
    __shared__ unsigned long long s_red[2];
    __shared__ int s_tmp[32];
    const int tid = threadIdx.x;
    const int l = blockIdx.x, b = blockIdx.y;
    const int K = p.topk;
    const long long slot = (long long)b * num_levels + l;
    // four independent reads, requested together (a chain of conditional reads would be three round trips)
    const unsigned int n_first = key_count[slot * CSTRIDE], n_valid = key_valid[slot * CSTRIDE];
    const unsigned int n_redo = key_count_redo[slot * CSTRIDE];
    const int cut_active = cuts[slot].active;
    const unsigned int total = (cut_active && n_valid < (unsigned int)K) ? n_redo : n_first;
    const unsigned long long* keys = all_keys + (long long)b * p.keys_per_image + p.key_off[l];
    if (total == 0) {
        if (tid == 0) cand_count[b * num_levels + l] = 0;
        return;
    }
    const LevelInputs li = level_inputs(p, l);
#ifdef OD_SEL_STAMPS
    unsigned long long st[16];
    int st_n = 0;
#define SEL_STAMP() do { if (st_n < 15) st[st_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SEL_STAMP() do { } while (0)
#endif
    SEL_STAMP();
    if (tid == 0) { s_tmp[20] = 0; s_tmp[21] = 0; }
    __syncthreads();

    // ---- phase 1: bring the keys that can still be among the best K into LDS ---------------------------
    // invariant: the K-th best key lies in [lo, lo + 2^w); `need` of the wanted keys lie inside that
    // interval, the other K - need are above it
    unsigned long long lo = p.lo0;
    int w = p.w0;
    int need = K;
    int cnt;
    if (total <= (unsigned int)CAP) {
        for (int i = tid; i < (int)total; i += SEL_T) s_keys[i] = keys[i];
        cnt = (int)total;
        __syncthreads();
    } else {
        // radix descent over the dense global key array (12 bits per coalesced pass) until the keys at or
        // above the cut fit in LDS
        while (true) {
            const int d = w < HBITS ? w : HBITS;
            const int shift = w - d;
            const int nb = 1 << d;
            for (int i = tid; i < nb; i += SEL_T) s_hist[i] = 0;
            __syncthreads();
            const unsigned long long span = (w >= 64) ? ~0ull : ((1ull << w) - 1);
            stream_keys(keys, total, [&](unsigned long long k) {
                if (k != 0ull && k >= lo && (k - lo) <= span) atomicAdd(&s_hist[(unsigned int)((k - lo) >> shift)], 1u);
            });
            __syncthreads();
            SEL_STAMP();
            find_cut_bin(s_hist, nb, need, s_tmp);
            SEL_STAMP();
            const int bstar = s_tmp[16], above = s_tmp[17], h = s_tmp[18], found = s_tmp[19];
            __syncthreads();
            if (!found) break;                     // fewer valid keys than K: everything at or above lo is taken
            lo = lo + ((unsigned long long)bstar << shift);
            w = shift;
            const int n_ge = (K - need) + above + h;
            need -= above;
            if (n_ge <= CAP || w == 0) break;
        }
        stream_append(keys, total, [&](unsigned long long k) { return k >= lo && k != 0ull; }, s_keys, &s_tmp[20], CAP);
        __syncthreads();
        cnt = s_tmp[20];
        if (cnt > CAP) cnt = CAP;   // unreachable for unique keys; keeps LDS accesses in bounds
        __syncthreads();            // s_tmp[20] is reused below
    }

    SEL_STAMP();
    // ---- phase 2: radix refinement inside LDS until about K keys remain (the sort costs n log^2 n) ------
    int target = 64;
    while (target < K) target <<= 1;             // sorting fewer than this many keys gains nothing
    if (cnt > target && w > 0) {
        // The keys stay where they are (every thread holds its eight in registers) while the passes narrow the
        // interval; they are compacted once, when few enough lie at or above its lower end.
        unsigned long long k[CAP / SEL_T];
#pragma unroll
        for (int u = 0; u < CAP / SEL_T; ++u) {
            const int i = u * SEL_T + tid;
            k[u] = i < cnt ? s_keys[i] : 0ull;
        }
        int n_ge = cnt;
        for (int pass = 0; n_ge > target && w > 0; ++pass) {
            for (int i = tid; i < HBINS; i += SEL_T) s_hist[i] = 0;
            if (tid == 0) { s_tmp[20] = 0; s_red[0] = ~0ull; s_red[1] = 0ull; }
            __syncthreads();
            SEL_STAMP();
            if (pass > 0) {
                // A pass that left too many keys met keys that agree in the bits it looked at (equal scores: thousands
                // of saturated logits differ only in their index, 32 bits further down).  The bits that all keys of the
                // interval share are skipped: AND and OR of (key - lo) differ first at the highest bit that varies.
                const unsigned long long span = (w >= 64) ? ~0ull : ((1ull << w) - 1);
                unsigned long long all_and = ~0ull, all_or = 0ull;
#pragma unroll
                for (int u = 0; u < CAP / SEL_T; ++u) {
                    if (k[u] != 0ull && k[u] >= lo && (k[u] - lo) <= span) { all_and &= k[u] - lo; all_or |= k[u] - lo; }
                }
                all_and = ((unsigned long long)wave_and_all((uint32_t)(all_and >> 32)) << 32) | wave_and_all((uint32_t)all_and);
                all_or = ((unsigned long long)wave_or_all((uint32_t)(all_or >> 32)) << 32) | wave_or_all((uint32_t)all_or);
                if ((tid & 63) == 0 && all_or >= all_and) { atomicAnd(&s_red[0], all_and); atomicOr(&s_red[1], all_or); }
                __syncthreads();
                const unsigned long long a = s_red[0], o = s_red[1];
                if (o >= a) {                                // the interval holds keys
                    const unsigned long long diff = a ^ o;
                    const int wn = diff ? 64 - __clzll((long long)diff) : 0;
                    if (wn < w) { lo += (wn < 64) ? ((a >> wn) << wn) : 0ull; w = wn; }
                }
                if (w == 0) break;                           // one key value left in the interval
            }
            const int d = w < HBITS ? w : HBITS;
            const int shift = w - d;
            const int nb = 1 << d;
            const unsigned long long span = (w >= 64) ? ~0ull : ((1ull << w) - 1);
#pragma unroll
            for (int u = 0; u < CAP / SEL_T; ++u) {
                const bool in = k[u] != 0ull && k[u] >= lo && (k[u] - lo) <= span;
                hist_add(s_hist, in, in ? (unsigned int)((k[u] - lo) >> shift) : 0u);
            }
            __syncthreads();
            SEL_STAMP();
            find_cut_bin(s_hist, nb, need, s_tmp);
            SEL_STAMP();
            const int bstar = s_tmp[16], above = s_tmp[17], h = s_tmp[18], found = s_tmp[19];
            if (found) {
                lo = lo + ((unsigned long long)bstar << shift);
                w = shift;
                n_ge = (K - need) + above + h;
                need -= above;
            } else {                                     // fewer valid keys than K (zero keys inflated cnt): keep them all
                n_ge = (K - need) + above;               // `above` = all valid keys of the interval
                w = 0;
            }
        }
        // compact in place: every thread holds its keys in registers, so the buffer can be rewritten (nothing reads
        // it between the loads above and the barrier below)
#pragma unroll
        for (int u = 0; u < CAP / SEL_T; ++u) lds_append(k[u] >= lo && k[u] != 0ull, k[u], s_keys, &s_tmp[20], CAP);
        __syncthreads();
        cnt = n_ge;
    }
    unsigned long long* sorted = s_keys;
    SEL_STAMP();

    int n_pad = 64;
    while (n_pad < cnt) n_pad <<= 1;
    // At most one key per thread (top-k up to 1024): the set of candidates is final, only their order is not.  Every
    // thread requests the regression values and the anchor of its key now, the sort runs while those cold, scattered
    // reads are on their way, and the key's place in the sorted list (found by bisection) says where the box goes.
    const bool early = cnt <= SEL_T;
    unsigned long long mine = 0ull;
    float4 my_r = make_float4(0.f, 0.f, 0.f, 0.f), my_an = my_r;
    if (early) {
        if (tid < cnt) mine = sorted[tid];
        if (mine != 0ull) gather_candidate(li, p.num_classes, b, mine, my_r, my_an);
        // zero keys (slots the filter reserved for logits that then failed the exact threshold) do not count
        const uint64_t bal = __ballot(mine != 0ull);
        if ((tid & 63) == 0 && bal) atomicAdd(&s_tmp[21], __popcll(bal));
    }
    for (int i = cnt + tid; i < n_pad; i += SEL_T) sorted[i] = 0ull;
    __syncthreads();
    SEL_STAMP();
    block_bitonic_sort_desc<SEL_T>(sorted, n_pad);
    SEL_STAMP();

    if (early) {
        const int valid = s_tmp[21];
        const int n_out = valid < K ? valid : K;
        if (mine != 0ull) {
            int rank = 0;                                  // keys greater than mine (keys are unique, the list descends)
            for (int step = n_pad >> 1; step >= 1; step >>= 1) rank += (sorted[rank + step - 1] > mine) ? step : 0;
            if (rank < n_out)
                decode_store(p, l, b, mine, rank, my_r, my_an, image_hw, num_levels, cand_boxes, cand_scores, cand_labels,
                             cand_index);
        }
        if (tid == 0) cand_count[b * num_levels + l] = n_out;
#ifdef OD_SEL_STAMPS
        SEL_STAMP();
        __syncthreads();
        if (tid == 0 && cand_index)
            for (int i = 0; i < 16; ++i) cand_index[slot * K + K - 16 + i] = i == 15 ? (long long)total : (i < st_n ? (long long)st[i] : 0ll);
#endif
        return;
    }
    // more keys than threads: zero keys sort last
    __syncthreads();
    for (int i = tid; i < cnt; i += SEL_T)
        if (sorted[i] != 0ull && (i + 1 == n_pad || sorted[i + 1] == 0ull)) s_tmp[21] = i + 1;
    __syncthreads();
    const int valid = s_tmp[21];
    const int n_out = valid < K ? valid : K;
    for (int r = tid; r < n_out; r += SEL_T) {
        float4 rr, an;
        gather_candidate(li, p.num_classes, b, sorted[r], rr, an);
        decode_store(p, l, b, sorted[r], r, rr, an, image_hw, num_levels, cand_boxes, cand_scores, cand_labels, cand_index);
    }
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
    long long key_off[OD_MAX_LEVELS];
    long long keys_per_image;
    size_t keys_bytes, counts_bytes;               // counts: three counter arrays (reserved, valid, redo), then the cuts
    size_t counter_bytes;                          // one counter array
};

bool make_layout(const int64_t* level_size, int num_levels, int batch, int num_classes, Layout* lay) {
    long long chunks = 0, slots = 0;
    for (int l = 0; l < OD_MAX_LEVELS; ++l) lay->key_off[l] = 0;
    for (int l = 0; l < num_levels; ++l) {
        if (level_size[l] < 0) return false;
        long long e = level_size[l] * (long long)num_classes;
        if (e > 0xFFFFFFFFll) return false;
        lay->chunk_start[l] = (int)chunks;
        lay->key_off[l] = slots;
        chunks += (e + CHUNK - 1) / CHUNK;
        slots += (e + 31) & ~31ll;                 // every level starts on a 256-byte boundary
        if (chunks > 0x3fffffff) return false;
    }
    for (int l = num_levels; l <= OD_MAX_LEVELS; ++l) lay->chunk_start[l] = (int)chunks;
    lay->total_chunks = (int)chunks;
    lay->keys_per_image = slots;
    lay->keys_bytes = ((size_t)batch * (size_t)slots * 8 + 255) & ~(size_t)255;
    lay->counter_bytes = (size_t)batch * (size_t)num_levels * CSTRIDE * 4;
    lay->counts_bytes = 3 * lay->counter_bytes + (((size_t)batch * (size_t)num_levels * sizeof(CutInfo) + 255) & ~(size_t)255);
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

namespace {

int select_candidates_impl(const float* const* cls_logits, const int64_t* cls_batch_stride,
                           const float* const* bbox_ctrness, const int64_t* ctr_batch_stride,
                           const float* const* bbox_regression, const int64_t* reg_batch_stride,
                           const float* const* anchors, const int64_t* anchor_batch_stride,
                           const int64_t* level_size, const int64_t* plane_size, int32_t num_levels, int32_t batch,
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
    unsigned long long* keys = (unsigned long long*)workspace;
    unsigned int* key_count = (unsigned int*)((char*)workspace + lay.keys_bytes);
    unsigned int* key_valid = (unsigned int*)((char*)key_count + lay.counter_bytes);
    unsigned int* key_count_redo = (unsigned int*)((char*)key_valid + lay.counter_bytes);
    CutInfo* cuts = (CutInfo*)((char*)key_count_redo + lay.counter_bytes);

    FilterParams fp;
    SelectParams sp;
    fp.num_levels = num_levels;
    fp.num_classes = num_classes;
    fp.has_ctr = bbox_ctrness ? 1 : 0;
    fp.topk = topk;
    fp.total_chunks = lay.total_chunks;
    for (int l = 0; l <= OD_MAX_LEVELS; ++l) fp.chunk_start[l] = lay.chunk_start[l];
    for (int l = 0; l < OD_MAX_LEVELS; ++l) { fp.key_off[l] = lay.key_off[l]; sp.key_off[l] = lay.key_off[l]; }
    fp.keys_per_image = lay.keys_per_image;
    sp.keys_per_image = lay.keys_per_image;
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
        fp.plane[l] = fp.chans[l] = sp.plane[l] = sp.apl[l] = 0;
        fp.defer[l] = (on && level_size[l] * (long long)num_classes >= DEFER_MIN) ? 1 : 0;
        if (on && plane_size && level_size[l] > 0) {
            const long long plane = plane_size[l];
            if (plane < 1 || level_size[l] % plane != 0) return OD_ERR_ARG;
            const long long apl = level_size[l] / plane;
            fp.plane[l] = sp.plane[l] = (unsigned int)plane;     // level_size * num_classes < 2^32 was checked above
            fp.chans[l] = (unsigned int)(apl * num_classes);
            sp.apl[l] = (unsigned int)apl;
            if (cls_batch_stride[l] < level_size[l] * (long long)num_classes || reg_batch_stride[l] < level_size[l] * 4)
                return OD_ERR_ARG;
        }
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

    bool sampled = false;                          // does any level get a sampled cut?
    for (int l = 0; l < num_levels; ++l) sampled = sampled || fp.level_elems[l] >= SAMPLE_MIN;
    bool large_sample = false;                     // does any level's sample exceed what one workgroup keeps in registers?
    for (int l = 0; l < num_levels; ++l) large_sample = large_sample || ((fp.level_elems[l] / 64) & ~16383ll) > SAMPLE_IN_REGS;   // S of sample_geometry
    if (large_sample) {
        OD_LAUNCH_KERNEL(OD_KERNEL_SAMPLE, sample_part_kernel,
                         dim3((unsigned)SAMPLE_PARTS_MAX, (unsigned)num_levels, (unsigned)batch), dim3(SEL_T), 0, s, fp, keys);
    }
    OD_LAUNCH_KERNEL(OD_KERNEL_SAMPLE, sample_kernel, dim3((unsigned)num_levels, (unsigned)batch), dim3(SEL_T), 0, s,
                     fp, cuts, key_count, (long long)(lay.counter_bytes / 4), keys);
    if (lay.total_chunks > 0) {
        int lead = 0;                              // deferred levels at the head of the list (the finest ones, usually)
        while (lead < num_levels && fp.defer[lead]) ++lead;
        const int chunk_base = lay.chunk_start[lead];
        const dim3 grid((unsigned)(lay.total_chunks - chunk_base), (unsigned)batch);
        if (grid.x > 0)
            OD_LAUNCH_KERNEL(OD_KERNEL_FILTER, filter_kernel, grid, dim3(ST), 0, s, fp, keys, key_count, key_valid,
                             key_count_redo, (const CutInfo*)cuts, chunk_base);
        for (int l = 0; l < num_levels; ++l) {
            if (!fp.defer[l]) continue;
            const long long units = ((fp.level_elems[l] + CHUNK - 1) / CHUNK + SUBS - 1) / SUBS;
            OD_LAUNCH_KERNEL(OD_KERNEL_FILTER, filter_defer_kernel, dim3((unsigned)units, (unsigned)batch), dim3(ST), 0, s,
                             fp, l, keys, key_count, key_valid, key_count_redo, (const CutInfo*)cuts, 0);
        }
        if (sampled) {
            if (grid.x > 0)
                OD_LAUNCH_KERNEL(OD_KERNEL_FILTER_REDO, filter_redo_kernel, dim3((unsigned)REDO_WG, (unsigned)batch),
                                 dim3(ST), 0, s, fp, keys, key_count, key_valid, key_count_redo, (const CutInfo*)cuts,
                                 (int)lay.total_chunks);
            for (int l = 0; l < num_levels; ++l) {
                if (!fp.defer[l]) continue;
                const long long units = ((fp.level_elems[l] + CHUNK - 1) / CHUNK + SUBS - 1) / SUBS;
                OD_LAUNCH_KERNEL(OD_KERNEL_FILTER_REDO, filter_defer_kernel, dim3((unsigned)units, (unsigned)batch),
                                 dim3(ST), 0, s, fp, l, keys, key_count, key_valid, key_count_redo, (const CutInfo*)cuts, 1);
            }
        }
    }
    OD_LAUNCH_KERNEL(OD_KERNEL_SELECT, select_kernel, dim3((unsigned)num_levels, (unsigned)batch), dim3(SEL_T), 0, s,
                     sp, (int)num_levels, (const unsigned long long*)keys, (const unsigned int*)key_count,
                     (const unsigned int*)key_valid, (const unsigned int*)key_count_redo, (const CutInfo*)cuts,
                     image_hw, cand_boxes, cand_scores, cand_labels, (long long*)cand_index, cand_count);
    return OD_OK;
}

}  // namespace

extern "C" int od_select_candidates(const float* const* cls_logits, const int64_t* cls_batch_stride,
                                    const float* const* bbox_ctrness, const int64_t* ctr_batch_stride,
                                    const float* const* bbox_regression, const int64_t* reg_batch_stride,
                                    const float* const* anchors, const int64_t* anchor_batch_stride,
                                    const int64_t* level_size, int32_t num_levels, int32_t batch,
                                    int32_t num_classes, const int32_t* image_hw, float score_thresh,
                                    int32_t topk, int32_t coder, float bbox_xform_clip, float* cand_boxes,
                                    float* cand_scores, int32_t* cand_labels, int64_t* cand_index,
                                    int32_t* cand_count, void* workspace, size_t workspace_bytes, void* stream) {
    return select_candidates_impl(cls_logits, cls_batch_stride, bbox_ctrness, ctr_batch_stride, bbox_regression,
                                  reg_batch_stride, anchors, anchor_batch_stride, level_size, nullptr, num_levels, batch,
                                  num_classes, image_hw, score_thresh, topk, coder, bbox_xform_clip, cand_boxes,
                                  cand_scores, cand_labels, cand_index, cand_count, workspace, workspace_bytes, stream);
}

extern "C" int od_select_candidates_planar(const float* const* cls_logits, const int64_t* cls_batch_stride,
                                           const float* const* bbox_ctrness, const int64_t* ctr_batch_stride,
                                           const float* const* bbox_regression, const int64_t* reg_batch_stride,
                                           const float* const* anchors, const int64_t* anchor_batch_stride,
                                           const int64_t* level_size, const int64_t* plane_size, int32_t num_levels,
                                           int32_t batch, int32_t num_classes, const int32_t* image_hw,
                                           float score_thresh, int32_t topk, int32_t coder, float bbox_xform_clip,
                                           float* cand_boxes, float* cand_scores, int32_t* cand_labels,
                                           int64_t* cand_index, int32_t* cand_count, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    if (!plane_size) return OD_ERR_ARG;
    return select_candidates_impl(cls_logits, cls_batch_stride, bbox_ctrness, ctr_batch_stride, bbox_regression,
                                  reg_batch_stride, anchors, anchor_batch_stride, level_size, plane_size, num_levels,
                                  batch, num_classes, image_hw, score_thresh, topk, coder, bbox_xform_clip, cand_boxes,
                                  cand_scores, cand_labels, cand_index, cand_count, workspace, workspace_bytes, stream);
}

extern "C" int od_decode_boxes(const float* rel_codes, const float* boxes, int64_t n, int32_t coder,
                               float bbox_xform_clip, float clip_h, float clip_w, float* out, void* stream) {
    if (n < 0) return OD_ERR_ARG;
    if (n == 0) return OD_OK;
    if (!rel_codes || !boxes || !out) return OD_ERR_ARG;
    if (((uintptr_t)rel_codes & 15) || ((uintptr_t)boxes & 15) || ((uintptr_t)out & 15)) return OD_ERR_ARG;
    if (coder != OD_CODER_BOXCODER && coder != OD_CODER_LINEAR) return OD_ERR_ARG;
    OD_LAUNCH_KERNEL(OD_KERNEL_DECODE, decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)rel_codes, (const float4*)boxes, (long long)n, (int)coder,
                     bbox_xform_clip, clip_h, clip_w, (float4*)out);
    return OD_OK;
}
