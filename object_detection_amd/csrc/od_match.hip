// Anchor <-> ground-truth matching kernels.
//
// od_match_retinanet: fused torchvision box_iou + Matcher(high, low, allow_low_quality_matches)
// (the per-image loop body of RetinaNet.compute_loss; reference trigger object_detection/train.py:62,
// thresholds object_detection/configs/config_all.py:10-11).  The [M,N] IoU matrix is never stored.
//
// The kernels are VALU-issue bound (a non-packed fp32 VALU instruction occupies a SIMD for 4 cycles on
// gfx950), so the design minimises instructions per (anchor, GT) pair, not bytes:
//   * work unit = one WAVE owning 256 consecutive anchors (4 per lane, in registers).  The wave keeps
//     the bounding box of its anchors and skips every GT that cannot overlap it: such a GT has IoU
//     exactly +0 with all 256 anchors, which can neither win a strict '>' argmax nor raise a per-GT
//     maximum.  Small units make tight boxes: ~2 % of the GT survive on the finest pyramid level.
//   * GT boxes are read 64 at a time (one per lane), the survivors are visited in ascending order
//     through a ballot mask and broadcast with v_readlane: no LDS, no barriers, scalar loop control,
//     and "first maximal index" tie-breaking is unchanged.
//   * a wave tile that meets more than HEAVY GT (coarse pyramid levels) would be a long serial loop on
//     the critical path; it is queued and processed by 4 cooperating waves instead (heavy kernel).
//   * which 256 anchors share a tile is a free choice: consecutive ones by default, 4 grid rows x 64 anchors where
//     the caller describes the anchor list as grids (TileLayout), which makes the tiles' boxes tighter.
//
// assign / heavy kernels: per-anchor max / argmax / thresholds -> `matched`, the raw argmax -> workspace,
//   per-GT maximum over the wave's anchors (DPP max) -> global atomic max, and one 8-byte entry per
//   (tile, GT with positive IoU): the GT and the tile's maximum.
// restore kernel (only with allow_low_quality_matches): torchvision restores anchor a to its own argmax
//   when iou[g,a] equals the maximum of row g.  Such an anchor attains its tile's maximum, so the kernel
//   compares each entry's value with the global maximum of its GT; for the few entries that match (the
//   tiles that tie for a GT's maximum) a wave recomputes that (tile, GT) row segment with the same IoU code (hence
//   bit-identical values) and patches `matched` from the stored argmax for every anchor attaining it.
//   If some GT has maximum 0 (it overlaps no anchor) every anchor ties at 0 for that row and torchvision
//   restores ALL anchors; a few blocks of the restore kernel look for such rows and do that.
#include "od_common.h"
#include "../../include/od_boxes.h"

#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace {

using namespace od;

constexpr int MT = 256;          // threads per workgroup = 4 independent waves
#ifndef OD_ASSIGN_LDS
#define OD_ASSIGN_LDS 0          // experiment knob (tools/exp_build.sh): unused LDS per workgroup caps the kernel's
#endif                           // waves per CU, which leaves room for another stream's kernel beside it
constexpr int ASSIGN_LDS = OD_ASSIGN_LDS;
#ifndef OD_R
#define OD_R 4
#endif
constexpr int R = OD_R;          // anchors per lane
constexpr int WTILE = 64 * R;    // anchors per wave
constexpr int HEAVY = 24;        // a wave tile meeting more GT than this goes to the cooperative kernel ...
constexpr int HEAVY_DENSE = 48;  // ... in an image with more than DENSE_GT boxes: more than this.  With dense targets every
constexpr int DENSE_GT = 256;    // tile meets tens of GT and the cooperative kernel's per-tile cost (four waves that load
                                 // the tile and walk the GT list) is better spent on the tiles that meet hundreds
                                 // (2000 GT per image: 69 -> 57 us per image with 48 instead of 24; 100 GT per image: the
                                 // assign kernel 49.7 -> 53.6 us)
constexpr int LCAP = HEAVY_DENSE;   // entry slots of a light wave tile: one per GT with positive IoU
__device__ __forceinline__ int heavy_threshold(int M) { return M > DENSE_GT ? HEAVY_DENSE : HEAVY; }
static_assert(LCAP % HEAVY == 0, "the restore kernel walks the entry slots in steps of HEAVY");
#ifndef OD_HT
#define OD_HT 256
#endif
constexpr int HT = OD_HT;        // threads of the cooperative (heavy tile) workgroup
#ifndef OD_GSTRIDE
#define OD_GSTRIDE 32
#endif
constexpr int GSTRIDE = OD_GSTRIDE;   // words between two GT maxima: atomics on one 128-byte line serialise
constexpr int HCAP_PER_GT = 160;  // heavy-entry buffer capacity = HCAP_PER_GT * m_total + 4096
constexpr int COUNT_ALL = 512;    // images with more GT stop counting a tile's GT as soon as the tile is known to be heavy
constexpr int GMAX_LDS = 4096;    // GT per image up to which the cooperative kernel accumulates the row maxima in LDS
constexpr int ECAP = 128;         // entries a wave of the cooperative kernel stages in LDS between two flushes

struct MatchWs {                 // workspace carve-up (all device pointers)
    uint32_t* gt_max_key;        // [m_total * GSTRIDE] ordered-key maximum IoU of every GT row, one per GSTRIDE words
    void* argmax;                // [batch, n]         per-anchor argmax before thresholding: uint16 when every GT
                                 //                    index fits (halves this write stream), else int32
    int argmax16;
    int* wave_cnt;               // [batch * wtiles]   entries of a light tile, -2 = heavy tile
    uint2* wave_list;            // [LCAP, batch * wtiles] entry-major: x = GT (index into the batch's GT list), y = value key (tile maximum)
    unsigned long long* heavy_count;   // [batch * 16]   per image (128 B apart): heavy tiles << 32 | heavy entries
    int4* heavy_queue;           // [batch, wtiles]    (tile of the image, -, GT met (a lower bound beyond COUNT_ALL GT), -) for the cooperative kernel
    uint4* heavy_entries;        // [heavy_cap]        x = wave tile, y = GT (as above), z = value key; image b owns the
                                 //                    slots [entry_base(b), entry_base(b) + entry_cap(b))
    long long list_stride;       // batch * wtiles
    uint2* coop_entries;         // [batch, coop_tiles, CCAP]  one-launch path: entries of the cooperative tiles (x = GT, y = value key)
    uint2* tile_head;            // [batch * wtiles, 8]        one-launch path: a 64-byte line per tile: (entry count, -), then its first 7 entries
    uint2* tile_extra;           // [batch * wtiles, FEXTRA]   its further entries (both live in the memory of wave_list)
};

constexpr uint32_t kZeroKey = 0x80000000u;           // ordered_key(+0.0f), the initial per-GT maximum

__device__ __forceinline__ void put_argmax(const MatchWs& ws, long long i, int g) {
    if (ws.argmax16) reinterpret_cast<unsigned short*>(ws.argmax)[i] = (unsigned short)g;
    else reinterpret_cast<int*>(ws.argmax)[i] = g;
}
__device__ __forceinline__ int get_argmax(const MatchWs& ws, long long i) {
    return ws.argmax16 ? (int)reinterpret_cast<const unsigned short*>(ws.argmax)[i] : reinterpret_cast<const int*>(ws.argmax)[i];
}

// heavy-entry slots of an image: proportional to its own GT count, so that images cannot starve each other
__device__ __forceinline__ long long entry_base(int b, int m0) { return (long long)HCAP_PER_GT * m0 + 256ll * b; }
__device__ __forceinline__ long long entry_cap(int M) { return (long long)HCAP_PER_GT * M + 256; }

// Which 256 anchors form wave tile `wt`.  The anchor list is cut into segments; a plain segment is tiled by 256
// consecutive anchors.  Where the caller has said that a range of the list is a grid (`rows` rows of `row_len`
// anchors, neighbouring rows spatially adjacent: a pyramid level in (y, x, a) order) with at least R rows of at least 64
// anchors, a tile is R rows x 64 anchors instead: R runs of 64 consecutive anchors, one below the other.  Its bounding
// box is a fraction as wide, so it meets fewer GT (finest level of a 1024 x 1024 image: 2.2 -> 1.5 GT per tile), and
// the loads and stores stay 1 KiB / 512 B runs.  A row length that is not a multiple of 64 (a row count that is not a
// multiple of R) makes the last column (band) of tiles start 64 anchors (R rows) before the end of the row (grid), so
// it overlaps its neighbour; likewise the last tile of a plain segment may reach into the next segment.  An anchor that
// belongs to two tiles is simply processed twice: every tile computes the same values for it (a GT a tile skips has
// IoU 0 with all its anchors), every write is of the same value, the row maxima are idempotent, and a duplicated
// entry restores the same anchors again.  Any assignment of anchors to tiles gives the same results.
constexpr int MAX_GRIDS = 8;
constexpr int MAX_SEGMENTS = MAX_GRIDS + 1;        // the described grids (tiled as grids or not) and what follows them
struct TileLayout {                  // four words per segment: they travel in scalar registers from the kernel's start
    int segments;
    int tile_end[MAX_SEGMENTS];      // one past the last wave tile of segment s (it starts where segment s - 1 ends)
    int first[MAX_SEGMENTS];         // its first anchor (layouts are only used for lists of fewer than 2^31 anchors)
    int row_len[MAX_SEGMENTS];       // grid: anchors per row; plain segment: 0
    int rows[MAX_SEGMENTS];
};

// first anchor of each of the R runs of wave tile wt (the same for all lanes of a wave)
__device__ __forceinline__ void tile_runs(const TileLayout& L, int wt, long long (&first)[R]) {
    const int wtu = __builtin_amdgcn_readfirstlane(wt);               // scalar arithmetic from here on
    int sbegin = 0, sfirst = 0, srow = 0, srows = 0;                   // no layout: one plain segment from anchor 0
#pragma unroll
    for (int k = 0; k < MAX_SEGMENTS; ++k) {
        const int begin = k == 0 ? 0 : L.tile_end[k - 1];
        if (k < L.segments && wtu >= begin && wtu < L.tile_end[k]) {
            sbegin = begin; sfirst = L.first[k]; srow = L.row_len[k]; srows = L.rows[k];
        }
    }
    const int tl = wtu - sbegin;
    if (srow == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) first[r] = (long long)sfirst + ((long long)tl * WTILE + r * 64);
    } else {
        const int cols = (srow + 63) >> 6;                             // tiles per band
        int bi = (int)((float)tl * __builtin_amdgcn_rcpf((float)cols));   // tl / cols, corrected below (tl < 2^24)
        if (bi * cols > tl) --bi;
        if ((bi + 1) * cols <= tl) ++bi;
        const int bj = tl - bi * cols;
        int row0 = R * bi, col0 = 64 * bj;
        row0 = row0 < srows - R ? row0 : srows - R;                    // last band / column: flush with the end
        col0 = col0 < srow - 64 ? col0 : srow - 64;
        const int base = sfirst + row0 * srow + col0;                  // < 2^31: layouts are used below that only
#pragma unroll
        for (int r = 0; r < R; ++r) first[r] = (long long)(base + r * srow);
    }
}

struct WaveTile {
    float ax1[R], ay1[R], ax2[R], ay2[R], aarea[R];
    float bx1, by1, bx2, by2;    // bounding box of the wave's anchors (wave-uniform)
};

__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ void load_wave_tile(WaveTile& t, const float4* __restrict__ A, const long long (&first)[R],
                                               long long n) {
    const int lane = threadIdx.x & 63;
    float bx1 = __builtin_inff(), by1 = __builtin_inff(), bx2 = -__builtin_inff(), by2 = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long long idx = first[r] + lane;
        if (idx > n - 1) idx = n - 1;               // out-of-range lanes replay the last anchor
        const float4 a = A[idx];
        t.ax1[r] = a.x; t.ay1[r] = a.y; t.ax2[r] = a.z; t.ay2[r] = a.w;
        t.aarea[r] = box_area(t.ax1[r], t.ay1[r], t.ax2[r], t.ay2[r]);
        bx1 = fminf(bx1, a.x); by1 = fminf(by1, a.y);
        bx2 = fmaxf(bx2, a.z); by2 = fmaxf(by2, a.w);
    }
    // wave-uniform: kept in scalar registers
    t.bx1 = uniform_f(wave_fmin_all(bx1)); t.by1 = uniform_f(wave_fmin_all(by1));
    t.bx2 = uniform_f(wave_fmax_all(bx2)); t.by2 = uniform_f(wave_fmax_all(by2));
}

// Visit, in ascending GT order, every GT of [g_begin, g_end) whose box can overlap the wave's bounding
// box.  Of those, only the ones whose running rank r satisfies r % nshare == share are handed to
// f(g, j, rank, gx1, gy1, gx2, gy2, garea) (wave-uniform arguments, j = lane that holds the GT): nshare
// waves that scan the same GT thereby split the survivors evenly, each still in ascending order.
// chunk_done(g0, cur) runs after each 64-GT chunk, AFTER the next chunk's loads have been issued, so the
// memory operations it issues (atomics) are younger than those loads and never delay them.  `cur` is the
// value of gmax[m0 + g0 + lane] read one chunk earlier (a lower bound of the row maximum, which only
// grows): a wave whose own maximum does not exceed it can skip its atomic.
__device__ __forceinline__ bool overlaps_tile(const WaveTile& t, const float4& gb) {
    return !(gb.z <= t.bx1 || gb.x >= t.bx2 || gb.w <= t.by1 || gb.y >= t.by2);
}

// Hands the GT marked in `mask` (lane j holds GT g0 + j in gb) to f in ascending order; see below.
template <typename F>
__device__ __forceinline__ void visit_survivors(unsigned long long mask, const float4& gb, int g0, int& rank, int share,
                                                int nshare, F f) {
    const float ga = box_area(gb.x, gb.y, gb.z, gb.w);
    while (mask) {
        const int j = __ffsll(mask) - 1;
        mask &= mask - 1;
        const bool mine = (nshare == 1) || (rank % nshare == share);
        if (mine)
            f(g0 + j, j, rank, bcast_lane(gb.x, j), bcast_lane(gb.y, j), bcast_lane(gb.z, j), bcast_lane(gb.w, j),
              bcast_lane(ga, j));
        ++rank;
    }
}

// The first chunk of a GT range (lane j: box g_begin + j and, with gmax, its current row maximum): requested by
// for_each_overlapping_gt itself, or earlier by its caller so that the request travels together with the tile's anchors.
struct GtChunk { float4 box; uint32_t max_key; };
__device__ __forceinline__ GtChunk first_gt_chunk(const float4* __restrict__ gt, int m0, const uint32_t* gmax, int g_begin,
                                                  int g_end) {
    const int lane = threadIdx.x & 63;
    GtChunk c;
    c.box = make_float4(0.f, 0.f, 0.f, 0.f);
    c.max_key = 0u;
    if (g_begin + lane < g_end) {
        c.box = gt[m0 + g_begin + lane];
        if (gmax) c.max_key = gmax[(long long)(m0 + g_begin + lane) * GSTRIDE];
    }
    return c;
}

template <typename F, typename D>
__device__ __forceinline__ void for_each_overlapping_gt(const WaveTile& t, const float4* __restrict__ gt, int m0,
                                                        const uint32_t* gmax, int g_begin, int g_end, int share,
                                                        int nshare, F f, D chunk_done, int chunk_step, const GtChunk& head) {
    // chunk_step > 1: only every chunk_step-th 64-GT chunk, starting with the one at g_begin (waves that split a long
    // GT list by chunks instead of by rank do not repeat each other's culling)
    const int lane = threadIdx.x & 63;
    float4 nxt = head.box;
    uint32_t nxt_max = head.max_key;
    int rank = 0;
    for (int g0 = g_begin; g0 < g_end; g0 += 64 * chunk_step) {
        const float4 gb = nxt;
        const uint32_t cur = nxt_max;
        const int gn = g0 + 64 * chunk_step + lane;
        if (gn < g_end) {                              // prefetch the next chunk
            nxt = gt[m0 + gn];
            if (gmax) nxt_max = gmax[(long long)(m0 + gn) * GSTRIDE];
        }
        const bool ok = (g0 + lane < g_end) && overlaps_tile(t, gb);
        visit_survivors(__ballot(ok), gb, g0, rank, share, nshare, f);
        chunk_done(g0, cur);
    }
}

// IoU of the wave's anchors with one GT; updates (best, bidx).  With want_max it also returns the ordered key of
// the maximum over the wave's 256 anchors (kZeroKey: no anchor of the wave overlaps this GT), else 0.
__device__ __forceinline__ uint32_t score_gt(const WaveTile& t, int g, float gx1, float gy1, float gx2, float gy2,
                                             float ga, float (&best)[R], int (&bidx)[R], bool want_max) {
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        v[r] = iou_pair_uniform(gx1, gy1, gx2, gy2, ga, t.ax1[r], t.ay1[r], t.ax2[r], t.ay2[r], t.aarea[r]);
        if (v[r] > best[r]) { best[r] = v[r]; bidx[r] = g; }
    }
    if (!want_max) return 0u;
    float m = v[0];
#pragma unroll
    for (int r = 1; r < R; ++r) m = fmaxf(m, v[r]);
    return wave_umax_all(ordered_key(m));
}

// Seven waves per SIMD: 72 registers (three values spilled) instead of the 74 the kernel takes by itself, which would
// leave room for six -- measured 49.7 -> 47.7 us alone and 55 -> 52 us inside the overlapped step.
__global__ __launch_bounds__(MT) __attribute__((amdgpu_waves_per_eu(7))) void match_assign_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n, int wtiles, TileLayout layout,
    int image_fastest, const float4* __restrict__ gt, const int* __restrict__ gt_offsets, MatchWs ws, int want_restore,
    float high, float low, long long* __restrict__ matched) {
    const int lane = threadIdx.x & 63;
    // Workgroups are handed out x-fastest.  x = image, y = tile block: the same tiles of all images start together,
    // and the coarse (GT-heavy) pyramid levels, which sit at the end of the anchor list, start first -- with the image
    // as the slow index the last image's long tiles started when the kernel was nearly done and ran on as its tail.
    // (Anchor lists with more tile blocks than y can hold fall back to x = tile block, y = image.)
    const int b = image_fastest ? blockIdx.x : blockIdx.y;
    const int tb = image_fastest ? blockIdx.y : blockIdx.x;
    const int wt = wtiles - 1 - (int)(tb * (MT / 64) + (threadIdx.x >> 6));
    if (wt < 0) return;
    long long first[R];
    tile_runs(layout, wt, first);
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    long long* out = matched + (long long)b * n;
    const long long wslot = (long long)b * wtiles + wt;

    if (M <= 0) {                                   // empty-target guard of compute_loss
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long idx = first[r] + lane;
            if (idx < n) out[idx] = -1;
        }
        if (want_restore && lane == 0) ws.wave_cnt[wslot] = 0;
        return;
    }

    // up to two GT chunks are requested together with the anchors and stay in registers: one memory round
    // trip for the whole tile instead of one per pass and chunk
    constexpr int RESIDENT = 128;
    float4 gA = make_float4(0.f, 0.f, 0.f, 0.f), gB = gA;
    if (M <= RESIDENT) {
        if (lane < M) gA = gt[m0 + lane];
        if (64 + lane < M) gB = gt[m0 + 64 + lane];
    }
    WaveTile t;
    load_wave_tile(t, anchors + (long long)b * anchor_batch_stride, first, n);
    unsigned long long maskA = 0ull, maskB = 0ull;
    if (M <= RESIDENT) {
        maskA = __ballot(lane < M && overlaps_tile(t, gA));
        maskB = __ballot(64 + lane < M && overlaps_tile(t, gB));
    }
    const int heavy_thr = heavy_threshold(M);
    if (M > heavy_thr) {                             // count the GT this tile meets; queue it if heavy
        int cnt = __popcll(maskA) + __popcll(maskB);
        if (M > RESIDENT) {
            for (int g0 = 0; g0 < M; g0 += 64) {
                const int g = g0 + lane;
                bool ok = g < M;
                if (ok) ok = overlaps_tile(t, gt[m0 + g]);
                cnt += __popcll(__ballot(ok));
                if (M > COUNT_ALL && cnt > heavy_thr) break;  // wave-uniform; dense images: most tiles are heavy
            }
        }
        if (cnt > heavy_thr) {
            if (lane == 0) {                       // queue slot (the cooperative kernel reserves its entries itself)
                const unsigned long long old = atomicAdd(ws.heavy_count + b * 16, 1ull << 32);
                ws.heavy_queue[(long long)b * wtiles + (int)(old >> 32)] = make_int4(wt, 0, cnt, 0);
                if (want_restore) ws.wave_cnt[wslot] = -2;
            }
            return;
        }
    }
    float best[R];
    int bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0.0f; bidx[r] = 0; }
    int lcount = 0;                                  // wave-uniform
    uint32_t kreg = 0u;                              // lane j: wave maximum of the IoU with GT g0 + j

    auto eval = [&](int g, int j, int, float gx1, float gy1, float gx2, float gy2, float ga) {
        const uint32_t k = score_gt(t, g, gx1, gy1, gx2, gy2, ga, best, bidx, want_restore != 0);
        if (want_restore && k > kZeroKey) {            // wave-uniform: this GT overlaps an anchor of the wave
            kreg = (lane == j) ? k : kreg;
            if (lane == 0) ws.wave_list[(long long)lcount * ws.list_stride + wslot] = make_uint2((uint32_t)(m0 + g), k);
            ++lcount;                                  // <= heavy_thr <= LCAP for a light tile
        }
    };
    auto chunk_done = [&](int g0, uint32_t cur) {
        if (!want_restore) return;
        if (kreg > cur) atomicMax(&ws.gt_max_key[(long long)(m0 + g0 + lane) * GSTRIDE], kreg);
        kreg = 0u;
    };
    if (M <= RESIDENT) {
        int rank = 0;
        visit_survivors(maskA, gA, 0, rank, 0, 1, eval);
        chunk_done(0, 0u);
        if (maskB) {
            visit_survivors(maskB, gB, 64, rank, 0, 1, eval);
            chunk_done(64, 0u);
        }
    } else {
        for_each_overlapping_gt(t, gt, m0, nullptr, 0, M, 0, 1, eval, chunk_done, 1, first_gt_chunk(gt, m0, nullptr, 0, M));
    }

    int lane_again = lane;                           // opaque: the load indices do not stay in registers until here
    asm volatile("" : "+v"(lane_again));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long idx = first[r] + lane_again;
        if (idx < n) {
            out[idx] = (best[r] < low) ? -1 : ((best[r] < high) ? -2 : (long long)bidx[r]);
            if (want_restore) put_argmax(ws, (long long)b * n + idx, bidx[r]);
        }
    }
    if (want_restore && lane == 0) ws.wave_cnt[wslot] = lcount;
}

// Cooperative kernel for the queued heavy tiles: one workgroup of NW waves per tile.  The waves share
// the tile's 256 anchors, every wave scans all GT but scores only every NW-th survivor; per-anchor
// (max, first argmax) are merged through LDS (equal maxima -> smaller GT index, which is the first maximal
// index of the full row).
// Nothing inside the GT loop goes to global memory: a store or an atomic issued between two chunks would have
// to complete before the next chunk's prefetched boxes can be used (the number of younger operations is not
// known at compile time, so the wait is for all of them), one memory round trip per 64 GT -- with 2000 GT per
// image that was 2.4 times the kernel's arithmetic.  Entries (GT, tile maximum) are staged per wave in LDS and
// written ECAP at a time to slots reserved with one atomic per flush; row maxima are accumulated with LDS
// atomics and merged into the global maxima once per tile (images with more than GMAX_LDS GT keep the
// per-chunk global atomics).
__global__ __launch_bounds__(HT) __attribute__((amdgpu_waves_per_eu(5))) void match_heavy_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n, int wtiles, TileLayout layout,
    const float4* __restrict__ gt, const int* __restrict__ gt_offsets, MatchWs ws, int want_restore,
    float high, float low, long long* __restrict__ matched) {
    constexpr int NW = HT / 64;
    __shared__ float s_best[NW][WTILE];
    __shared__ int s_bidx[NW][WTILE];
    __shared__ int s_cost[HT];
    __shared__ int s_tile[HT];
    __shared__ int s_pick;
    __shared__ uint32_t s_gmax[GMAX_LDS];          // row maxima of the current tile (ordered keys; 0 = no overlap)
    __shared__ uint2 s_ent[NW][ECAP];              // staged entries of each wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
#ifdef OD_HEAVY_STAMPS
    unsigned long long hst[8];
    hst[0] = __builtin_amdgcn_s_memrealtime();
#define HSTAMP(i) hst[i] = __builtin_amdgcn_s_memrealtime()
#else
#define HSTAMP(i) do { } while (0)
#endif
    const unsigned long long hc = ws.heavy_count[b * 16];
    const int nheavy = (int)(hc >> 32);
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    const bool lds_max = want_restore && M <= GMAX_LDS;
    const long long ebase = entry_base(b, m0), ecap = entry_cap(M);
    if (lds_max) {
        for (int g = threadIdx.x; g < M; g += HT) s_gmax[g] = 0u;
        __syncthreads();
    }
    // Blocks are numbered image-fastest, so the hardware hands out "the r-th tile of every image" together.
    // With r = rank by cost (GT met, descending) every CU receives a mix of expensive and cheap tiles
    // instead of whatever the arrival order of the queue gives; done when the queue fits one pass.
    // Short GT lists: every wave culls all GT and takes every NW-th survivor (even split whatever the list looks
    // like).  Long lists: every wave takes every NW-th chunk of 64 GT, culling included -- with 2000 GT the
    // culling of all chunks by all waves was 40 % of the kernel's instructions.
    const bool by_chunk = M > 64 * NW * 2;
    const bool by_cost = nheavy <= HT;
    if (by_cost) {                                   // the queue in LDS: picking a tile costs no further memory round trip
        int4 q = make_int4(0, 0, -1, 0);
        if ((int)threadIdx.x < nheavy) q = ws.heavy_queue[(long long)b * wtiles + threadIdx.x];
        s_cost[threadIdx.x] = q.z;
        s_tile[threadIdx.x] = q.x;
        __syncthreads();
    }
    for (int item = blockIdx.y; item < nheavy; item += gridDim.y) {
        int pick = item;
        if (by_cost) {                               // the queue entry whose (cost desc, position asc) rank is `item`
            if ((int)threadIdx.x < nheavy) {
                const int mine = s_cost[threadIdx.x];
                int rank = 0;
                for (int i = 0; i < nheavy; ++i) {
                    const int c = s_cost[i];
                    rank += (c > mine || (c == mine && i < (int)threadIdx.x)) ? 1 : 0;
                }
                if (rank == item) s_pick = (int)threadIdx.x;
            }
            __syncthreads();
            pick = s_pick;
        }
        HSTAMP(1);
        const int wt = by_cost ? s_tile[pick] : ws.heavy_queue[(long long)b * wtiles + pick].x;
        const long long wslot = (long long)b * wtiles + wt;
        long long first[R];
        tile_runs(layout, wt, first);
        long long* out = matched + (long long)b * n;
        // the wave's first GT chunk does not depend on the tile: requested before the anchors, whose bounding box
        // (a wave reduction) would otherwise have to be complete before the request is made
        const uint32_t* gmax_global = (want_restore && !lds_max) ? ws.gt_max_key : nullptr;
        const int g_first = by_chunk ? 64 * wave : 0;
        const GtChunk head = first_gt_chunk(gt, m0, gmax_global, g_first, M);
        WaveTile t;
        load_wave_tile(t, anchors + (long long)b * anchor_batch_stride, first, n);
        HSTAMP(2);
        float best[R];
        int bidx[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { best[r] = 0.0f; bidx[r] = 0; }
        uint32_t kreg = 0u;
        int staged = 0;                               // wave-uniform: entries of this wave waiting in s_ent[wave]
        // slots of the image's entry region for the staged entries (one returning atomic per flush); if the region
        // is full nothing is written and the total, which still counts them, tells the restore kernel so
        auto flush_entries = [&]() {
            if (staged == 0) return;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // lane 0 wrote the entries, all lanes read them
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned long long old = 0ull;
            if (lane == 0) old = atomicAdd(ws.heavy_count + b * 16, (unsigned long long)staged);
            const long long at = (long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old);
            if (at + staged <= ecap) {
                for (int i = lane; i < staged; i += 64) {
                    const uint2 e = s_ent[wave][i];
                    ws.heavy_entries[ebase + at + i] = make_uint4((uint32_t)wslot, e.x, e.y, 0u);
                }
            }
            staged = 0;
        };
        for_each_overlapping_gt(t, gt, m0, gmax_global, g_first, M, by_chunk ? 0 : wave, by_chunk ? 1 : NW,
            [&](int g, int j, int, float gx1, float gy1, float gx2, float gy2, float ga) {
                const uint32_t k = score_gt(t, g, gx1, gy1, gx2, gy2, ga, best, bidx, want_restore != 0);
                if (want_restore && k > kZeroKey) {    // wave-uniform: this GT overlaps an anchor of the tile
                    kreg = (lane == j) ? k : kreg;
                    if (lane == 0) s_ent[wave][staged] = make_uint2((uint32_t)(m0 + g), k);
                    if (++staged == ECAP) flush_entries();
                }
            },
            [&](int g0, uint32_t cur) {
                if (!want_restore) return;
                if (lds_max) {
                    if (kreg > kZeroKey) atomicMax(&s_gmax[g0 + lane], kreg);
                } else if (kreg > cur) {
                    atomicMax(&ws.gt_max_key[(long long)(m0 + g0 + lane) * GSTRIDE], kreg);
                }
                kreg = 0u;
            }, by_chunk ? NW : 1, head);
        HSTAMP(3);
        flush_entries();
        HSTAMP(4);
#pragma unroll
        for (int r = 0; r < R; ++r) { s_best[wave][r * 64 + lane] = best[r]; s_bidx[wave][r * 64 + lane] = bidx[r]; }
        __syncthreads();
        HSTAMP(5);
        if (lds_max) {                                // merge the tile's row maxima into the global ones
            for (int g = threadIdx.x; g < M; g += HT) {
                const uint32_t v = s_gmax[g];
                if (v > kZeroKey) {
                    s_gmax[g] = 0u;                   // ready for the next tile
                    uint32_t* gm = &ws.gt_max_key[(long long)(m0 + g) * GSTRIDE];
                    if (v > *gm) atomicMax(gm, v);
                }
            }
        }
        for (int e = threadIdx.x; e < WTILE; e += HT) {  // anchor e & 63 of run e >> 6
            float bb = s_best[0][e];
            int bi = s_bidx[0][e];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float ob = s_best[w][e];
                const int oi = s_bidx[w][e];
                if (ob > bb || (ob == bb && oi < bi)) { bb = ob; bi = oi; }
            }
            long long run = first[0];
#pragma unroll
            for (int r = 1; r < R; ++r) run = ((e >> 6) == r) ? first[r] : run;
            const long long idx = run + (e & 63);
            if (idx < n) {
                out[idx] = (bb < low) ? -1 : ((bb < high) ? -2 : (long long)bi);
                if (want_restore) put_argmax(ws, (long long)b * n + idx, bi);
            }
        }
        __syncthreads();
#ifdef OD_HEAVY_STAMPS
        HSTAMP(6);
        if (threadIdx.x == 0 && first[0] + 8 <= n)
            for (int i = 0; i < 7; ++i) out[first[0] + i] = (long long)(hst[i] - hst[0]) + (1ll << 40) * (i + 1);
#endif
    }
}

// Start values of one call: every GT row maximum = key(+0), the per-image heavy-tile counters = 0.  One small launch
// instead of two memsets (a memset is a launch as well, and writes the whole strided maxima array).
__global__ __launch_bounds__(256) void match_init_kernel(MatchWs ws, long long m_total, int batch, int want_restore) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (want_restore && i < m_total) ws.gt_max_key[i * GSTRIDE] = kZeroKey;
    if (i < batch) ws.heavy_count[i * 16] = 0ull;
}

// Wave-cooperative re-examination of one tile against the GT range [g_begin, g_end) of its image: the IoUs
// are recomputed with the same code as in the assign kernel (bit-identical values) and every anchor that
// attains a GT's global maximum is restored.  All 64 lanes must call it with wave-uniform arguments.
__device__ __forceinline__ void recheck_segment(const float4* __restrict__ anchors, long long anchor_batch_stride,
                                                long long n, int wtiles, const TileLayout& layout,
                                                const float4* __restrict__ gt,
                                                const int* __restrict__ gt_offsets, const MatchWs& ws, int wslot,
                                                int g_begin, int g_count, long long* __restrict__ matched) {
    const int lane = threadIdx.x & 63;
    const int b = wslot / wtiles, wt = wslot - b * wtiles;
    long long first[R];
    tile_runs(layout, wt, first);
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    if (g_begin >= M) return;
    WaveTile t;
    load_wave_tile(t, anchors + (long long)b * anchor_batch_stride, first, n);
    bool restore[R];
#pragma unroll
    for (int r = 0; r < R; ++r) restore[r] = false;
    const int g_end = (g_begin + g_count < M) ? g_begin + g_count : M;
    for_each_overlapping_gt(t, gt, m0, nullptr, g_begin, g_end, 0, 1,
        [&](int g, int j, int, float gx1, float gy1, float gx2, float gy2, float ga) {
            const float gmax = ordered_key_inv(ws.gt_max_key[(long long)(m0 + g) * GSTRIDE]);
            if (!(gmax > 0.0f)) return;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float v = iou_pair(gx1, gy1, gx2, gy2, ga, t.ax1[r], t.ay1[r], t.ax2[r], t.ay2[r], t.aarea[r]);
                restore[r] = restore[r] || (v == gmax);
            }
        },
        [&](int, uint32_t) {}, 1, first_gt_chunk(gt, m0, nullptr, g_begin, g_end));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long idx = first[r] + lane;
        if (idx < n && restore[r]) matched[(long long)b * n + idx] = (long long)get_argmax(ws, (long long)b * n + idx);
    }
}

// Up to RB (tile, GT) row segments whose maximum equals the GT's global maximum: the wave recomputes their 256 IoUs
// with the code of the assign kernels (bit-identical values) and restores every anchor that attains the maximum.
// The loads of all RB segments are issued before the first IoU, so a wave with several hits (neighbouring tiles
// often tie for the same GT) pays the memory latency once.  Wave-uniform arguments; wslot[q] < 0 = unused.
// RB = 1 measured best (restore 19.5 / 18.2 / 17.1 us for RB = 4 / 2 / 1): the registers of further segments (138 / 82 /
// 60 VGPRs) cost more occupancy in this latency-bound kernel than the shared round trip saves.
#ifndef OD_RB
#define OD_RB 1
#endif
constexpr int RB = OD_RB;
__device__ __forceinline__ void restore_row_segments(const float4* __restrict__ anchors, long long anchor_batch_stride,
                                                     long long n, int wtiles, const TileLayout& layout,
                                                     const float4* __restrict__ gt,
                                                     const int* __restrict__ gt_offsets, const MatchWs& ws,
                                                     const int (&wslot)[RB], const int (&g)[RB], const uint32_t (&key)[RB],
                                                     long long* __restrict__ matched) {
    const int lane = threadIdx.x & 63;
    float4 a[RB][R], gb[RB];
    int am[RB][R];                                         // the anchors' argmax, requested with the boxes
    long long first[RB][R];                                // first anchor of each run of the segment's tile
    long long image0[RB];                                  // flat index of the image's first anchor in `matched`
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        if (wslot[q] < 0) continue;
        const int b = wslot[q] / wtiles, wt = wslot[q] - b * wtiles;
        tile_runs(layout, wt, first[q]);
        const float4* A = anchors + (long long)b * anchor_batch_stride;
        // g[q] indexes the batch's GT list; a vector load, so that the wave does not stop for a scalar one before
        // the anchor loads of the next segment are issued
        int gi = g[q];
        asm volatile("" : "+v"(gi));
        gb[q] = gt[gi];
        image0[q] = (long long)b * n;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long idx = first[q][r] + lane;
            const long long ic = idx < n ? idx : n - 1;
            a[q][r] = A[ic];
            am[q][r] = get_argmax(ws, (long long)b * n + ic);
        }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        if (wslot[q] < 0) continue;
        const float ga = box_area(gb[q].x, gb[q].y, gb[q].z, gb[q].w);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 v4 = a[q][r];
            const float v = iou_pair(gb[q].x, gb[q].y, gb[q].z, gb[q].w, ga, v4.x, v4.y, v4.z, v4.w,
                                     box_area(v4.x, v4.y, v4.z, v4.w));
            const long long idx = first[q][r] + lane;
            if (idx < n && ordered_key(v) == key[q]) matched[image0[q] + idx] = (long long)am[q][r];
        }
    }
}

// Restore from the entries.  Blocks [0, light_blocks) take the light tiles, one thread per (entry slot,
// tile); the next heavy_blocks blocks take the heavy tiles' entries, one thread per entry.  For an entry whose
// value equals its GT's global maximum the wave re-examines that row segment (restore_row_segments).  If the
// heavy-entry buffer was too small (pathological inputs) the heavy blocks re-examine every heavy tile against all
// GT of its image instead.  The last ZB blocks per image look for a GT row whose maximum IoU is exactly 0: every
// anchor attains that maximum, so the image restores every anchor (torchvision's behaviour when a GT overlaps no
// anchor); the values written are the ones the entry blocks write, so the order between the blocks is irrelevant.
constexpr int ZB = 8;
#ifndef OD_RT
#define OD_RT 256
#endif
constexpr int RT = OD_RT;      // threads of a restore block
#ifndef OD_LSTEP
#define OD_LSTEP 8
#endif
constexpr int LSTEP = OD_LSTEP;  // entry slots that get blocks of their own; a block goes on to slot e + LSTEP while its tiles have that many entries
static_assert(LCAP % LSTEP == 0, "the restore kernel walks the entry slots in steps of LSTEP");
__global__ __launch_bounds__(RT) void match_restore_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n, int wtiles, TileLayout layout,
    int batch, int light_blocks, int heavy_blocks, int heavy_first, const float4* __restrict__ gt,
    const int* __restrict__ gt_offsets, MatchWs ws, long long* __restrict__ matched) {
    const long long total = (long long)batch * wtiles;
    const int lane = threadIdx.x & 63;
    __shared__ int s_hn;                                 // hits of the block in the current round: (tile, GT, value key)
    __shared__ int s_hs[RT], s_hg[RT];
    __shared__ uint32_t s_hk[RT];
    // Block order = hand-out order.  With dense targets nearly every tile is a heavy one and the heavy-entry blocks are
    // the long ones: they (and the zero-maximum blocks) go first (2000 GT per image: 150 -> 118 us for 32 images);
    // otherwise they have little to do and the light blocks go first (100 GT per image: 17.5 against 18.3 us).
    const int special = heavy_blocks + batch * ZB;
    const bool light = heavy_first ? (int)blockIdx.x >= special : (int)blockIdx.x < light_blocks;
    const int lb = heavy_first ? (int)blockIdx.x - special : (int)blockIdx.x;        // light block number
    // light blocks share one index space; heavy blocks are dealt round-robin to the images
    const int hb = heavy_first ? (int)blockIdx.x : (int)blockIdx.x - light_blocks;
    if (!light && hb >= heavy_blocks) {                                  // zero-maximum blocks
        const int zb = hb - heavy_blocks;
        const int b = zb / ZB, part = zb - b * ZB;
        if (b >= batch) return;
        const int m0 = gt_offsets[b], M = gt_offsets[b + 1] - m0;
        int any0 = 0;
        for (int g = threadIdx.x; g < M; g += RT)
            any0 |= (ordered_key_inv(ws.gt_max_key[(long long)(m0 + g) * GSTRIDE]) == 0.0f) ? 1 : 0;
        if (!__syncthreads_or(any0)) return;
        for (long long idx = (long long)part * RT + threadIdx.x; idx < n; idx += ZB * RT)
            matched[(long long)b * n + idx] = (long long)get_argmax(ws, (long long)b * n + idx);
        return;
    }
    const int hblocks = heavy_blocks / batch;                            // heavy blocks per image
    const int hb_image = light ? 0 : hb % batch;
    const long long nthreads = (long long)(light ? light_blocks : hblocks) * RT;
    const long long first = (long long)(light ? lb : hb / batch) * RT + (threadIdx.x & ~63);
    long long limit = total * LCAP, ebase = 0;
    bool overflow = false;
    int nheavy = 0;
    if (!light) {
        const unsigned long long hc = ws.heavy_count[hb_image * 16];
        const int m0 = gt_offsets[hb_image], M = gt_offsets[hb_image + 1] - m0;
        nheavy = (int)(hc >> 32);
        overflow = (long long)(uint32_t)hc > entry_cap(M);
        limit = (overflow || hb / batch >= hblocks) ? 0 : (long long)(uint32_t)hc;
        ebase = entry_base(hb_image, m0);
    }
    // Heavy entries: consecutive entries belong to one tile and often hit together, so they are dealt to different
    // waves (entry = lane * waves + wave); light entries are slot-major already.
    // A light block takes 256 neighbouring tiles at one entry slot e < LSTEP (block -> slot and tile range by a scalar
    // division, nothing per thread): items = LSTEP * tile_blocks, dealt to the light blocks in turn.  Most tiles have
    // fewer than LSTEP entries; a block goes on to slot e + LSTEP (and so on, up to LCAP) while one of its tiles has
    // that many.  (With a block for every slot below HEAVY two thirds of the blocks found nothing to do: the kernel
    // took the same time alone, but its waves stood in the way of the other streams' kernels.)
    const long long nwaves = nthreads >> 6, wave_id = first >> 6;
    const long long tile_blocks = (total + RT - 1) / RT;
    const long long light_items = tile_blocks * LSTEP;
    for (long long k0 = 0; light ? (k0 + lb < light_items) : (k0 < limit); k0 += light ? (long long)light_blocks : nthreads) {
        int pass = 0, more = 0;
        do {
            uint32_t word = 0u, key = 0u;
            int wslot = 0, my_count = 0;
            bool have = false;
            if (light) {                                     // slot-major: neighbouring threads, neighbouring tiles
                const uint32_t item = (uint32_t)(k0 + lb);                  // < HEAVY * 2^23: 32-bit division
                const int e = (int)(item / (uint32_t)tile_blocks) + pass * LSTEP;
                const long long tile = (long long)(item % (uint32_t)tile_blocks) * RT + threadIdx.x;
                if (tile < total) {
                    wslot = (int)tile;
                    // the slot is read together with the tile's entry count (one memory round trip instead of two); a
                    // slot beyond the count holds stale data and is ignored
                    const uint2 ent = ws.wave_list[(long long)e * total + tile];
                    my_count = ws.wave_cnt[wslot];
                    if (e < my_count) {                      // also skips heavy tiles (-2)
                        word = ent.x; key = ent.y; have = true;
                    }
                }
                ++pass;
                more = (pass < LCAP / LSTEP) ? __syncthreads_or(my_count > e + LSTEP ? 1 : 0) : 0;
            } else {
                const long long i = k0 + (long long)lane * nwaves + wave_id;
                if (i < limit) {
                    const uint4 ent = ws.heavy_entries[ebase + i];
                    wslot = (int)ent.x; word = ent.y; key = ent.z; have = true;
                }
            }
            bool hit = false;
            if (have && key > kZeroKey) {
                hit = key == ws.gt_max_key[(long long)word * GSTRIDE];
            }
            // Hits cluster: neighbouring tiles tie for the same GT and sit in neighbouring lanes, and every hit costs the
            // wave a memory round trip.  The block's hits go to a list in LDS and are dealt to its four waves in turn
            // (restore 17.7 -> 13.6 us at 16 images x 100 GT).  The loop bounds are uniform over the block.
            if (threadIdx.x == 0) s_hn = 0;
            __syncthreads();
            if (hit) {
                const int q = atomicAdd(&s_hn, 1);
                s_hs[q] = wslot; s_hg[q] = (int)word; s_hk[q] = key;
            }
            __syncthreads();
            const int hn = s_hn;
            for (int h0 = (int)(threadIdx.x >> 6) * RB; h0 < hn; h0 += (RT / 64) * RB) {
                int hs[RB], hg[RB];
                uint32_t hk[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const bool in = h0 + q < hn;
                    const int hq = in ? h0 + q : h0;
                    hs[q] = in ? __builtin_amdgcn_readfirstlane(s_hs[hq]) : -1;
                    hg[q] = __builtin_amdgcn_readfirstlane(s_hg[hq]);
                    hk[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_hk[hq]);
                }
                restore_row_segments(anchors, anchor_batch_stride, n, wtiles, layout, gt, gt_offsets, ws, hs, hg, hk, matched);
            }
            __syncthreads();                                 // the list is reused by the next round
        } while (more);
    }
    if (overflow && hb / batch < hblocks) {   // one wave per (heavy tile, 64-GT chunk)
        const int chunks = (gt_offsets[hb_image + 1] - gt_offsets[hb_image] + 63) >> 6;
        const long long items = (long long)nheavy * chunks;
        for (long long q = first >> 6; q < items; q += nthreads >> 6) {
            const long long h = q / chunks;
            const int wslot = hb_image * wtiles + ws.heavy_queue[(long long)hb_image * wtiles + h].x;
            recheck_segment(anchors, anchor_batch_stride, n, wtiles, layout, gt, gt_offsets, ws, wslot, (int)(q - h * chunks) * 64,
                            64, matched);
        }
    }
}

// ================================================================================================================
// One-launch matching for ordinary target counts (on average at most FUSED_AVG_GT boxes per image, at most
// FUSED_MAX_GT in the batch).
//
// The four-launch chain above (init, assign, cooperative, restore) spends a third of its time in launch gaps and in
// the tail of the cooperative kernel, which cannot start before the assign kernel has classified the last tile.
// Here the tiles that are known to be GT-heavy before any box is read -- the coarse pyramid levels at the end of the
// anchor list -- are given to cooperative workgroups of the SAME launch (dispatched first), every other tile is a
// light tile whatever it meets, and nothing is queued.  No global atomics and nothing to initialise: a tile only
// leaves its entries (GT, tile maximum); the row maxima are formed in LDS by the restore kernel.
//
//   match_fused_kernel        grid (image, y): y < coop_tiles -> workgroup = one cooperative tile (the last tiles
//                             of the list), else 4 light tiles, one per wave (coarse end of the list first).
//   match_restore_lds_kernel  grid (G, image): workgroup j of an image owns the GT rows g with g % G == j.  It reads ALL
//                             entries of the image (about 11 000 at 786 240 anchors x 100 GT: 200 KB of entry lines out
//                             of the L2) and keeps the maxima of its rows in LDS, reads them again (now cached),
//                             lists the entries of its rows that equal the row's maximum and restores those row
//                             segments (restore_row_segments, as the chain does).
// Whole-chip time is what a step made of overlapping streams pays for: the maxima through global atomics in the fused
// kernel (the chain's way) made that kernel 6 us longer on every CU, and the step slower, although the restore
// kernel then was 8 us shorter -- it is a narrow kernel (G workgroups per image) that other streams' kernels run beside.
// Rows, not tiles, are dealt to the workgroups: a workgroup that forms the maxima of all rows makes an LDS
// look-up and an atomic for every entry of the image (11 000 x 2 LDS operations with scattered addresses: 4 us of its
// one CU); with every G-th row it makes an eighth of them, and every entry is skipped or taken by a register test.
// The entries of a light tile are tile-major: one 64-byte line holds the tile's entry count and its first 7 entries
// (a tile of the fine levels meets 1.5 GT on average), read as a flat array of 16-byte words, consecutive lanes
// consecutive words; the few tiles with more entries continue in a second array.
// A tile with more entries than its slots hold (FCAP for a light tile, CCAP for a cooperative one) is marked -3:
// the restore kernel recomputes that tile against the image's whole GT list, once for the maxima and once for the
// restore.  Results are those of the chain (and of the checker) bit for bit: same tiles, same IoU code.
constexpr int CCAP = 256;            // entry slots of a cooperative tile
constexpr int FUSED_AVG_GT = 256;    // path used when m_total <= FUSED_AVG_GT * batch ...
constexpr int FUSED_MAX_GT = 16384;  // ... and m_total <= this (the row maxima of an image live in LDS: 4 bytes per GT)
constexpr int TILE_OVERFLOW = -3;
constexpr int FHEAD = 7;             // entries of a light tile that share the 64-byte line of its entry count
constexpr int FEXTRA = LCAP - FHEAD - 1;   // further entry slots of a light tile; capacity FHEAD + FEXTRA
constexpr int FCAP = FHEAD + FEXTRA;

#ifndef OD_FUSED_WAVES
#define OD_FUSED_WAVES 7
#endif
__global__ __launch_bounds__(MT) __attribute__((amdgpu_waves_per_eu(OD_FUSED_WAVES))) void match_fused_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n, int wtiles, int coop_tiles,
    TileLayout layout, const float4* __restrict__ gt, const int* __restrict__ gt_offsets, MatchWs ws, int want_restore,
    float high, float low, long long* __restrict__ matched) {
    constexpr int NW = MT / 64;
    __shared__ float s_best[NW][WTILE];
    __shared__ int s_bidx[NW][WTILE];
    __shared__ uint2 s_ent[NW][ECAP];
    __shared__ int s_total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    long long* out = matched + (long long)b * n;
    const float4* A = anchors + (long long)b * anchor_batch_stride;

    if ((int)blockIdx.y < coop_tiles) {
        // ---- cooperative tile: the four waves share 256 anchors and split the GT list ------------------------
        const int ct = (int)blockIdx.y;
        const int wt = wtiles - 1 - ct;
        const long long wslot = (long long)b * wtiles + wt;
        long long first[R];
        tile_runs(layout, wt, first);
        if (M <= 0) {
            for (int e = threadIdx.x; e < WTILE; e += MT) {
                long long run = first[0];
#pragma unroll
                for (int r = 1; r < R; ++r) run = ((e >> 6) == r) ? first[r] : run;
                const long long idx = run + (e & 63);
                if (idx < n) out[idx] = -1;
            }
            if (want_restore && threadIdx.x == 0) ws.tile_head[wslot * 8] = make_uint2(0u, 0u);
            return;
        }
        if (threadIdx.x == 0) s_total = 0;
        const bool by_chunk = M > 64 * NW * 2;      // long lists: every wave culls and scores every NW-th chunk of 64
        const int g_first = by_chunk ? 64 * wave : 0;
        const GtChunk head = first_gt_chunk(gt, m0, nullptr, g_first, M);
        WaveTile t;
        load_wave_tile(t, A, first, n);
        __syncthreads();                            // s_total = 0 is visible
        float best[R];
        int bidx[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { best[r] = 0.0f; bidx[r] = 0; }
        int staged = 0;                             // wave-uniform: entries waiting in s_ent[wave]
        uint2* region = ws.coop_entries + ((long long)b * coop_tiles + ct) * CCAP;
        auto flush_entries = [&]() {
            if (staged == 0) return;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // lane 0 wrote the entries, all lanes read them
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int at = 0;
            if (lane == 0) at = atomicAdd(&s_total, staged);
            at = __builtin_amdgcn_readfirstlane(at);
            if (at + staged <= CCAP)
                for (int i = lane; i < staged; i += 64) region[at + i] = s_ent[wave][i];
            staged = 0;
        };
        for_each_overlapping_gt(t, gt, m0, nullptr, g_first, M, by_chunk ? 0 : wave, by_chunk ? 1 : NW,
            [&](int g, int, int, float gx1, float gy1, float gx2, float gy2, float ga) {
                const uint32_t k = score_gt(t, g, gx1, gy1, gx2, gy2, ga, best, bidx, want_restore != 0);
                if (want_restore && k > kZeroKey) {
                    if (lane == 0) s_ent[wave][staged] = make_uint2((uint32_t)(m0 + g), k);
                    if (++staged == ECAP) flush_entries();
                }
            },
            [&](int, uint32_t) {}, by_chunk ? NW : 1, head);
        flush_entries();
#pragma unroll
        for (int r = 0; r < R; ++r) { s_best[wave][r * 64 + lane] = best[r]; s_bidx[wave][r * 64 + lane] = bidx[r]; }
        __syncthreads();
        for (int e = threadIdx.x; e < WTILE; e += MT) {
            float bb = s_best[0][e];
            int bi = s_bidx[0][e];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float ob = s_best[w][e];
                const int oi = s_bidx[w][e];
                if (ob > bb || (ob == bb && oi < bi)) { bb = ob; bi = oi; }
            }
            long long run = first[0];
#pragma unroll
            for (int r = 1; r < R; ++r) run = ((e >> 6) == r) ? first[r] : run;
            const long long idx = run + (e & 63);
            if (idx < n) {
                out[idx] = (bb < low) ? -1 : ((bb < high) ? -2 : (long long)bi);
                if (want_restore) put_argmax(ws, (long long)b * n + idx, bi);
            }
        }
        if (want_restore && threadIdx.x == 0)
            ws.tile_head[wslot * 8] = make_uint2((uint32_t)(s_total <= CCAP ? s_total : TILE_OVERFLOW), 0u);
        return;
    }

    // ---- light tiles: one wave each, whatever the tile meets ----------------------------------------------------
    const int light_tiles = wtiles - coop_tiles;
    const int wt = light_tiles - 1 - (int)(((int)blockIdx.y - coop_tiles) * NW + wave);
    if (wt < 0) return;
    long long first[R];
    tile_runs(layout, wt, first);
    const long long wslot = (long long)b * wtiles + wt;
    if (M <= 0) {                                   // empty-target guard of compute_loss
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long idx = first[r] + lane;
            if (idx < n) out[idx] = -1;
        }
        if (want_restore && lane == 0) ws.tile_head[wslot * 8] = make_uint2(0u, 0u);
        return;
    }
    // The first two GT chunks are requested together with the anchors: one memory round trip for the whole tile at up to
    // 128 GT.  Longer lists go on in steps of 128 through the same code (one more round trip per step, not hidden by a
    // prefetch: a second instance of the scoring code, or two more chunks in flight, would take the registers that
    // keep seven waves per SIMD free of spills -- and spilled registers of 51 000 waves went to HBM, 43 MiB per launch).
    float4 gA = make_float4(0.f, 0.f, 0.f, 0.f), gB = gA;
    if (lane < M) gA = gt[m0 + lane];
    if (64 + lane < M) gB = gt[m0 + 64 + lane];
    WaveTile t;
    load_wave_tile(t, A, first, n);
    float best[R];
    int bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0.0f; bidx[r] = 0; }
    int lcount = 0;                                  // wave-uniform
    uint2* head = ws.tile_head + wslot * 8;
    uint2* extra = ws.tile_extra + wslot * FEXTRA;
    auto eval = [&](int g, int, int, float gx1, float gy1, float gx2, float gy2, float ga) {
        const uint32_t k = score_gt(t, g, gx1, gy1, gx2, gy2, ga, best, bidx, want_restore != 0);
        if (want_restore && k > kZeroKey) {          // wave-uniform: this GT overlaps an anchor of the wave
            if (lane == 0 && lcount < FCAP)
                (lcount < FHEAD ? head + 1 + lcount : extra + (lcount - FHEAD))[0] = make_uint2((uint32_t)(m0 + g), k);
            ++lcount;
        }
    };
    int rank = 0;
    for (int g0 = 0; g0 < M; g0 += 128) {
        if (g0 > 0) {
            gA = gB = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g0 + lane < M) gA = gt[m0 + g0 + lane];
            if (g0 + 64 + lane < M) gB = gt[m0 + g0 + 64 + lane];
        }
        const unsigned long long maskA = __ballot(g0 + lane < M && overlaps_tile(t, gA));
        const unsigned long long maskB = __ballot(g0 + 64 + lane < M && overlaps_tile(t, gB));
        visit_survivors(maskA, gA, g0, rank, 0, 1, eval);
        if (maskB) visit_survivors(maskB, gB, g0 + 64, rank, 0, 1, eval);
    }
    // The lane number is made opaque here: otherwise the four 64-bit anchor indices formed for the tile's loads stay in
    // vector registers until these stores (8 registers: the difference between seven waves per SIMD with and without spills).
    int lane_again = lane;
    asm volatile("" : "+v"(lane_again));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long idx = first[r] + lane_again;
        if (idx < n) {
            out[idx] = (best[r] < low) ? -1 : ((best[r] < high) ? -2 : (long long)bidx[r]);
            if (want_restore) put_argmax(ws, (long long)b * n + idx, bidx[r]);
        }
    }
    if (want_restore && lane == 0) head[0] = make_uint2((uint32_t)(lcount <= FCAP ? lcount : TILE_OVERFLOW), 0u);
}

// Low-quality restore of the one-launch path.  See the section comment above.
#ifndef OD_RL_T
#define OD_RL_T 1024
#endif
constexpr int RL_T = OD_RL_T;        // threads of a workgroup
constexpr int RL_W = RL_T / 64;
constexpr int RL_WPT = 12;           // 16-byte words of the tiles' lines a thread requests together
constexpr int HIT_CAP = 1024;        // hits a workgroup lists; beyond that it re-examines every tile for its rows
constexpr int XT_CAP = 2048;         // light tiles with further entries that a workgroup lists (beyond: read one by one)
static_assert(FEXTRA % 8 == 0 && FEXTRA < 64, "eight lanes share a tile's further entries; their number is packed into 6 bits");
constexpr int OVER_CAP = 256;        // overflowed tiles of an image that are listed; beyond that every tile's count is scanned
constexpr int CAND_CAP = 6144;       // entries of its rows that a workgroup lists (about 1400 at 786 240 anchors x 100 GT); beyond: as above

// One tile against the rows g of its image with g % G == j.  With `matched` null: the tile's maximum of every such row is
// merged into the LDS maxima (a tile whose entry slots overflowed).  Otherwise: every anchor that attains such a row's
// (positive) maximum is restored.
__device__ __forceinline__ void tile_rows_from_lds(const float4* __restrict__ A, long long n, const TileLayout& layout,
                                                   int wt, const float4* __restrict__ gt, int m0, int M, int G, int j,
                                                   uint32_t* s_max, const MatchWs& ws, long long image0,
                                                   long long* __restrict__ matched) {
    const int lane = threadIdx.x & 63;
    long long first[R];
    tile_runs(layout, wt, first);
    WaveTile t;
    load_wave_tile(t, A, first, n);
    bool restore[R];
#pragma unroll
    for (int r = 0; r < R; ++r) restore[r] = false;
    for_each_overlapping_gt(t, gt, m0, nullptr, 0, M, 0, 1,
        [&](int g, int, int, float gx1, float gy1, float gx2, float gy2, float ga) {
            if (g % G != j) return;
            float v[R];
#pragma unroll
            for (int r = 0; r < R; ++r)
                v[r] = iou_pair_uniform(gx1, gy1, gx2, gy2, ga, t.ax1[r], t.ay1[r], t.ax2[r], t.ay2[r], t.aarea[r]);
            if (matched == nullptr) {
                float m = v[0];
#pragma unroll
                for (int r = 1; r < R; ++r) m = fmaxf(m, v[r]);
                const uint32_t k = wave_umax_all(ordered_key(m));
                if (k > kZeroKey && lane == 0) atomicMax(&s_max[g], k);
            } else {
                const uint32_t key = s_max[g];
                if (!(key > kZeroKey)) return;
#pragma unroll
                for (int r = 0; r < R; ++r) restore[r] = restore[r] || (ordered_key(v[r]) == key);
            }
        },
        [&](int, uint32_t) {}, 1, first_gt_chunk(gt, m0, nullptr, 0, M));
    if (matched != nullptr) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long idx = first[r] + lane;
            if (idx < n && restore[r]) matched[image0 + idx] = (long long)get_argmax(ws, image0 + idx);
        }
    }
}

__global__ __launch_bounds__(RL_T) void match_restore_lds_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n, int wtiles, int coop_tiles,
    TileLayout layout, const float4* __restrict__ gt, const int* __restrict__ gt_offsets, MatchWs ws,
    long long* __restrict__ matched, int batch, int by_xcd) {
    extern __shared__ uint32_t s_max[];                  // [M] row maxima (ordered keys; 0 = nothing overlaps); only this workgroup's rows are kept
    __shared__ int s_hn, s_on, s_xn, s_cn;
    __shared__ int s_hs[HIT_CAP], s_hg[HIT_CAP];
    __shared__ uint32_t s_hk[HIT_CAP];
    __shared__ int s_over[OVER_CAP];
    __shared__ int s_xt[XT_CAP];                         // light tiles with more entries than their line holds: tile << 6 | further entries
    __shared__ int s_cw[CAND_CAP];                       // entries of this workgroup's rows: tile, GT, value key
    __shared__ uint32_t s_cg[CAND_CAP], s_ck[CAND_CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid (8 * G, ceil(batch / 8)): the G workgroups of an image share an XCD, hence its L2 -- the entry lines of the image
    // are fetched into it once (workgroups go to the XCDs in turn by their linear number); grid (G, batch): plain
    const int G = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int b = by_xcd ? (int)((blockIdx.x & 7u) + 8u * blockIdx.y) : (int)blockIdx.y;
    const int j = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (b >= batch) return;
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    if (M <= 0) return;
    const float4* A = anchors + (long long)b * anchor_batch_stride;
    const long long image0 = (long long)b * n;
    const long long slot0 = (long long)b * wtiles;
    const int light_tiles = wtiles - coop_tiles;
    // is GT `g` (index into the batch's list, as the entries hold it) one of this workgroup's rows?  With a power of
    // two one subtraction and one AND; otherwise the quotient through a multiplication by ceil(2^32 / G), which is exact
    // for row numbers below 2^32 / G (they are below 2^23) -- a division per entry was a quarter of the kernel at G = 3
    const uint32_t gmask = (uint32_t)G - 1u;
    const bool pow2 = (G & (G - 1)) == 0;
    const uint32_t gmagic = (uint32_t)((0x100000000ull + (unsigned long long)G - 1ull) / (unsigned long long)G);
    auto mine = [&](uint32_t g) {
        const uint32_t l = g - (uint32_t)m0;
        return pow2 ? (l & gmask) == (uint32_t)j : (l - __umulhi(l, gmagic) * (uint32_t)G) == (uint32_t)j;
    };
#ifdef OD_RL_STAMPS
    unsigned long long rst[8];
#define RLSTAMP(i) rst[i] = __builtin_amdgcn_s_memrealtime()
#else
#define RLSTAMP(i) do { } while (0)
#endif
    RLSTAMP(0);
    for (int g = tid; g < M; g += RL_T) s_max[g] = 0u;
    if (tid == 0) { s_hn = 0; s_on = 0; s_xn = 0; s_cn = 0; }
    __syncthreads();
    // An entry first looks (a plain read) whether it can raise its row's maximum at all: the maxima only grow, so a
    // stale look costs an atomic that changes nothing, never a missed one.
    auto raise_max = [&](uint32_t g, uint32_t key) {
        const int l = (int)g - m0;
        if (key > s_max[l]) atomicMax(&s_max[l], key);
    };
    auto push_hit = [&](int wt, uint32_t g, uint32_t key) {
        const int q = atomicAdd(&s_hn, 1);
        if (q < HIT_CAP) { s_hs[q] = (int)(slot0 + wt); s_hg[q] = (int)g; s_hk[q] = key; }
    };
    const int part = lane & 3;                           // which quarter of a tile's line this lane reads
    const uint4* heads = reinterpret_cast<const uint4*>(ws.tile_head + slot0 * 8);
    const int words = light_tiles * 4;
    // The lines of the light tiles (count + first 7 entries, 64 bytes) are read as one flat array of 16-byte words,
    // consecutive lanes consecutive words (one tile per lane would make a wave request touch 64 cache lines instead of
    // 8).  A lane holds two entry slots of a tile; the tile's count sits in the first lane of its quad and is fetched
    // with a quad broadcast.  A slot beyond the count holds whatever the memory held: it is not looked at.
    // What bounds the kernel is the number of dependent memory round trips (1-2.5 us each for this one workgroup per
    // CU, cached or not), so requests that do not depend on each other are issued together -- (entry lines, counts
    // of the cooperative tiles), then (entries of the cooperative tiles, further entries of the listed tiles) -- and
    // nothing is read twice: every entry of one of this workgroup's rows raises the row's maximum and goes to a list in
    // LDS (about 1400 of the image's 11 000), and once the maxima are final the list is compared with them.
    // N possible entries per lane at once (called by whole waves): one reservation in the list per wave and call
    auto take = [&](auto n_tag, const bool* valid, const int* wt, const uint32_t* g, const uint32_t* key) {
        constexpr int N = decltype(n_tag)::value;
        int mine_n = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) mine_n += valid[i] ? 1 : 0;
        const int incl = (int)wave_scan_incl((uint32_t)mine_n);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) return;
        int base = 0;
        if (lane == 63) base = atomicAdd(&s_cn, total);
        int q = __builtin_amdgcn_readlane(base, 63) + incl - mine_n;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (valid[i]) {
                raise_max(g[i], key[i]);
                if (q < CAND_CAP) { s_cw[q] = wt[i]; s_cg[q] = g[i]; s_ck[q] = key[i]; }
                ++q;
            }
        }
    };
    uint4 word[RL_WPT];
    auto load_words = [&](int q0) {
#pragma unroll
        for (int u = 0; u < RL_WPT; ++u) {
            const int q = q0 + u * RL_T + tid;
            word[u] = q < words ? heads[q] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto process_words = [&](int q0) {
        bool valid[2 * RL_WPT];
        int wts[2 * RL_WPT];
        uint32_t gs[2 * RL_WPT], keys[2 * RL_WPT];
#pragma unroll
        for (int u = 0; u < RL_WPT; ++u) {
            const int wt = (q0 + u * RL_T + tid) >> 2;
            const int cnt = __builtin_amdgcn_update_dpp(0, (int)word[u].x, 0x00, 0xf, 0xf, false);   // quad_perm [0,0,0,0]
            valid[2 * u] = part > 0 && cnt > 2 * part - 1 && mine(word[u].x);
            valid[2 * u + 1] = cnt > 2 * part && mine(word[u].z);
            wts[2 * u] = wts[2 * u + 1] = wt;
            gs[2 * u] = word[u].x; keys[2 * u] = word[u].y;
            gs[2 * u + 1] = word[u].z; keys[2 * u + 1] = word[u].w;
            if (part == 0 && cnt == TILE_OVERFLOW) {
                const int q = atomicAdd(&s_on, 1);
                if (q < OVER_CAP) s_over[q] = wt;
            }
            if (part == 0 && cnt > FHEAD) {              // the few tiles that meet more GT: listed
                const int q = atomicAdd(&s_xn, 1);
                if (q < XT_CAP) s_xt[q] = (wt << 6) | (cnt - FHEAD);
            }
        }
        take(std::integral_constant<int, 2 * RL_WPT>{}, valid, wts, gs, keys);
    };
    // cooperative tiles: a wave takes every RL_W-th one, four at a time: their counts are requested together, then all
    // their entries (at most CCAP / 64 per lane and tile)
    int ccnt[4];
    uint2 cent[4][CCAP / 64];
    auto load_coop_counts = [&](int c0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ct = c0 + k * RL_W;
            ccnt[k] = ct < coop_tiles ? (int)ws.tile_head[(slot0 + wtiles - 1 - ct) * 8].x : 0;
        }
    };
    auto load_coop_entries = [&](int c0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint2* region = ws.coop_entries + ((long long)b * coop_tiles + c0 + k * RL_W) * CCAP;
#pragma unroll
            for (int i = 0; i < CCAP / 64; ++i)
                if (lane + 64 * i < ccnt[k]) cent[k][i] = region[lane + 64 * i];
        }
    };
    auto process_coop = [&](int c0) {
        constexpr int N = 4 * (CCAP / 64);
        bool valid[N];
        int wts[N];
        uint32_t gs[N], keys[N];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int wt = wtiles - 1 - (c0 + k * RL_W);
            if (ccnt[k] == TILE_OVERFLOW && lane == 0) {
                const int q = atomicAdd(&s_on, 1);
                if (q < OVER_CAP) s_over[q] = wt;
            }
#pragma unroll
            for (int i = 0; i < CCAP / 64; ++i) {
                const int e = k * (CCAP / 64) + i;
                valid[e] = lane + 64 * i < ccnt[k] && mine(cent[k][i].x);
                wts[e] = wt; gs[e] = cent[k][i].x; keys[e] = cent[k][i].y;
            }
        }
        take(std::integral_constant<int, N>{}, valid, wts, gs, keys);
    };
    // further entries of the listed tiles: eight lanes per tile, every lane requests all its entries (at most
    // FEXTRA / 8) at once
    constexpr int XL = 8, XPL = FEXTRA / XL;
    const int xpart = lane & (XL - 1);
    int x_wt = 0, x_ne = 0;
    uint2 xent[XPL];
    auto load_further = [&](int it, int listed_x) {
        x_ne = 0;
        if (it < listed_x) {
            const int item = s_xt[it];
            x_wt = item >> 6;
            x_ne = item & 63;
            const uint2* extra = ws.tile_extra + (slot0 + x_wt) * FEXTRA;
#pragma unroll
            for (int i = 0; i < XPL; ++i)
                if (xpart + XL * i < x_ne) xent[i] = extra[xpart + XL * i];
        }
    };
    auto process_further = [&]() {
        bool valid[XPL];
        int wts[XPL];
        uint32_t gs[XPL], keys[XPL];
#pragma unroll
        for (int i = 0; i < XPL; ++i) {
            valid[i] = xpart + XL * i < x_ne && mine(xent[i].x);
            wts[i] = x_wt; gs[i] = xent[i].x; keys[i] = xent[i].y;
        }
        take(std::integral_constant<int, XPL>{}, valid, wts, gs, keys);
    };

    // ---- row maxima and the list of this workgroup's entries --------------------------------------------------------
    load_words(0);
    load_coop_counts(wave);
    process_words(0);
    for (int q0 = RL_T * RL_WPT; q0 < words; q0 += RL_T * RL_WPT) {
        load_words(q0);
        process_words(q0);
    }
    RLSTAMP(1);
    __syncthreads();                                     // the lists of tiles with further entries / overflowed tiles are complete
    const int listed_x = s_xn < XT_CAP ? s_xn : XT_CAP;
    load_coop_entries(wave);
    load_further(tid / XL, listed_x);
    process_coop(wave);
    process_further();
    for (int c0 = wave + 4 * RL_W; c0 < coop_tiles; c0 += 4 * RL_W) {
        load_coop_counts(c0);
        load_coop_entries(c0);
        process_coop(c0);
    }
    for (int it0 = RL_T / XL; it0 < listed_x; it0 += RL_T / XL) {      // (whole waves go round: `take` uses ballots)
        load_further(it0 + tid / XL, listed_x);
        process_further();
    }
    bool rescan = s_xn > XT_CAP;                         // block-uniform: a list was too short; every tile is re-examined below
    {   // tiles whose entries did not fit: their maxima (of this workgroup's rows) are recomputed
        const int over_total = s_on;
        const int over_listed = over_total < OVER_CAP ? over_total : OVER_CAP;
        for (int q = wave; q < over_listed; q += RL_W)
            tile_rows_from_lds(A, n, layout, __builtin_amdgcn_readfirstlane(s_over[q]), gt, m0, M, G, j, s_max, ws, image0, nullptr);
        if (over_total > OVER_CAP || rescan)
            for (int wt = wave; wt < wtiles; wt += RL_W) {
                const int cnt = (int)ws.tile_head[(slot0 + wt) * 8].x;
                if (cnt == TILE_OVERFLOW || (rescan && cnt > FHEAD && wt < light_tiles))
                    tile_rows_from_lds(A, n, layout, wt, gt, m0, M, G, j, s_max, ws, image0, nullptr);
            }
    }
    RLSTAMP(2);
    __syncthreads();                                     // the maxima of this workgroup's rows are final, its list is complete
    RLSTAMP(3);
    {   // A GT that overlaps no anchor: every anchor ties at 0 for its row and is restored (torchvision's behaviour);
        // the workgroup that owns such a row rewrites the whole image from the stored argmax, which covers
        // everything else it would do.
        int any0 = 0;
        for (int g = j + G * tid; g < M; g += G * RL_T) any0 |= (s_max[g] == 0u) ? 1 : 0;
        if (__syncthreads_or(any0)) {
            for (long long idx = tid; idx < n; idx += RL_T)
                matched[image0 + idx] = (long long)get_argmax(ws, image0 + idx);
            return;
        }
    }
    // ---- the listed entries that equal their row's maximum ---------------------------------------------------------
    const int cands = s_cn < CAND_CAP ? s_cn : CAND_CAP;
    rescan = rescan || s_cn > CAND_CAP;
    for (int i = tid; i < cands; i += RL_T)
        if (s_ck[i] == s_max[(int)s_cg[i] - m0]) push_hit(s_cw[i], s_cg[i], s_ck[i]);
    RLSTAMP(4);
    __syncthreads();
    RLSTAMP(5);
    const int hits = s_hn;
    const int listed = hits < HIT_CAP ? hits : HIT_CAP;
    for (int h0 = wave * RB; h0 < listed; h0 += RL_W * RB) {
        int hs[RB], hg[RB];
        uint32_t hk[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const bool in = h0 + q < listed;
            const int hq = in ? h0 + q : h0;
            hs[q] = in ? __builtin_amdgcn_readfirstlane(s_hs[hq]) : -1;
            hg[q] = __builtin_amdgcn_readfirstlane(s_hg[hq]);
            hk[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_hk[hq]);
        }
        restore_row_segments(anchors, anchor_batch_stride, n, wtiles, layout, gt, gt_offsets, ws, hs, hg, hk, matched);
    }
    RLSTAMP(6);
#ifdef OD_RL_STAMPS
    if (lane == 0 && b == 0 && j < 2 && (wave == 0 || wave == 15))
        printf("restore stamps (10 ns) wg %d wave %d: lines %llu rest+barrier %llu zero-check+list(%d) %llu barrier %llu hits(%d) %llu\n",
               j, wave, rst[1] - rst[0], rst[3] - rst[1], s_cn, rst[4] - rst[3], rst[5] - rst[4], hits, rst[6] - rst[5]);
#endif
    // what did not fit: with more entries or ties than a list holds every tile that has entries is re-examined for this
    // workgroup's rows; tiles whose entries overflowed always are
    const int over_total = s_on;
    if (rescan || hits > HIT_CAP || over_total > OVER_CAP) {
        for (int wt = wave; wt < wtiles; wt += RL_W) {
            const int cnt = (int)ws.tile_head[(slot0 + wt) * 8].x;
            if ((rescan || hits > HIT_CAP) ? cnt != 0 : cnt == TILE_OVERFLOW)
                tile_rows_from_lds(A, n, layout, wt, gt, m0, M, G, j, s_max, ws, image0, matched);
        }
    } else {
        for (int q = wave; q < over_total; q += RL_W)
            tile_rows_from_lds(A, n, layout, __builtin_amdgcn_readfirstlane(s_over[q]), gt, m0, M, G, j, s_max, ws, image0, matched);
    }
}

// ---- FCOS centre sampling: 64 anchors per workgroup, GT staged in LDS ---------------------------------
// A GT can match an anchor only if the anchor's centre lies within radius * size of the GT's centre.  The four
// waves of a workgroup share 64 consecutive anchors (a piece of a row of one pyramid level, mostly; one per lane)
// and split the GT list in interleaved runs of 64.  A wave keeps the bounding box of the anchors' centres and
// their largest sampling radius, tests 64 GT at a time against that box (one per lane, with a margin that covers
// the rounding of the two formulations) and evaluates only the survivors, in ascending order, with all lanes;
// the four partial (maximum, first index) pairs are merged through LDS.  On the coarse levels, where every GT
// survives, the split shortens the serial loop that bounds the kernel.
// A GT that is skipped has value 0 * (1e8 - area) = +0 for every anchor and can never win the strict '>' of the
// argmax, and the split keeps "first index of the maximum" -- provided every value is >= +0, i.e. every GT area is
// finite and below 1e8.  An image whose GT list holds any other box is evaluated by one wave, every GT in list
// order, exactly as the sequential formulation (GT 0 initialises the running maximum, as torch's argmax does).
constexpr int FT = 256;
constexpr int FW = FT / 64;      // waves per workgroup = ways the GT list is split
constexpr int FGCH = 1024;

__global__ __launch_bounds__(FT) void match_fcos_kernel(
    const float4* __restrict__ anchors, long long anchor_batch_stride, long long n, long long n_first,
    long long n_last, const float4* __restrict__ gt, const int* __restrict__ gt_offsets, float radius,
    long long* __restrict__ matched, int image_fastest) {
    __shared__ float4 s_g[FGCH];
    __shared__ float s_best[FW][64];
    __shared__ int s_bi[FW][64];
    // workgroups are handed out x-fastest: x = image, y = anchor block, the last blocks (coarse levels: every GT
    // survives the culling there, the longest loops) first; lists with more blocks than y holds: x = block, y = image
    const int b = image_fastest ? blockIdx.x : blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long blk = image_fastest ? (long long)(gridDim.y - 1 - blockIdx.y) : (long long)blockIdx.x;
    const long long idx = blk * 64 + lane;
    const int m0 = gt_offsets[b];
    const int M = gt_offsets[b + 1] - m0;
    const bool valid = idx < n;
    // lanes past the end replay the last anchor (their result is not written)
    const float4 a = anchors[(long long)b * anchor_batch_stride + (valid ? idx : n - 1)];
    // anchor_centers = (a[:, :2] + a[:, 2:]) / 2 ; anchor_sizes = a[:, 2] - a[:, 0]
    const float cx = (a.x + a.z) / 2.0f, cy = (a.y + a.w) / 2.0f;
    const float size = a.z - a.x;
    const float lower = (idx < n_first) ? 0.0f : size * 4.0f;
    const float upper = (idx >= n - n_last) ? __builtin_inff() : size * 8.0f;
    const float rs = radius * size;
    // the wave's box of centres, widened by the largest radius and a rounding margin
    const float cx_lo = wave_fmin_all(cx), cx_hi = wave_fmax_all(cx);
    const float cy_lo = wave_fmin_all(cy), cy_hi = wave_fmax_all(cy);
    const float rs_max = wave_fmax_all(rs);
    const float pad = rs_max + 1e-3f * (rs_max + fmaxf(fabsf(cx_lo), fabsf(cx_hi)) + fmaxf(fabsf(cy_lo), fabsf(cy_hi)) + 1.0f);
    const float x_lo = cx_lo - pad, x_hi = cx_hi + pad, y_lo = cy_lo - pad, y_hi = cy_hi + pad;

    // one pass over the image's GT list: is every value >= +0 ?  The first FGCH boxes are staged on the way.
    int plain_here = 1;
    for (int g = threadIdx.x; g < M; g += FT) {
        const float4 G = gt[m0 + g];
        if (g < FGCH) s_g[g] = G;
        const float garea = (G.z - G.x) * (G.w - G.y);
        plain_here &= (garea < 1e8f && garea > -__builtin_inff()) ? 1 : 0;         // false for NaN as well
    }
    const bool plain = __syncthreads_and(plain_here) != 0;

    float best = 0.0f;
    int bi = plain ? 0x7fffffff : 0;
    auto evaluate = [&](const float4 G) {
        float gcx = (G.x + G.z) / 2.0f, gcy = (G.y + G.w) / 2.0f;
        float dmax = fmaxf(fabsf(cx - gcx), fabsf(cy - gcy));
        float l = cx - G.x, t = cy - G.y, r = G.z - cx, bt = G.w - cy;
        float mn = fminf(fminf(l, t), fminf(r, bt));
        float mx = fmaxf(fmaxf(l, t), fmaxf(r, bt));
        bool match = (dmax < rs) && (mn > 0.0f) && (mx > lower) && (mx < upper);
        float garea = (G.z - G.x) * (G.w - G.y);
        return (match ? 1.0f : 0.0f) * (1e8f - garea);
    };
    for (int c0 = 0; c0 < M; c0 += FGCH) {
        if (c0 > 0) {                                      // the first chunk is staged already
            __syncthreads();
            for (int g = threadIdx.x; g < FGCH && c0 + g < M; g += FT) s_g[g] = gt[m0 + c0 + g];
            __syncthreads();
        }
        const int cnt = min(FGCH, M - c0);
        if (!plain) {
            if (wave == 0) {                               // sequential formulation, every GT in list order
                for (int j = 0; j < cnt; ++j) {
                    const float v = evaluate(s_g[j]);
                    if ((c0 + j == 0) || v > best) { best = v; bi = c0 + j; }
                }
            }
            continue;
        }
        for (int j0 = wave * 64; j0 < cnt; j0 += FW * 64) {
            const int jl = j0 + lane;
            bool pass = false;
            if (jl < cnt) {
                const float4 G = s_g[jl];
                const float gcx = (G.x + G.z) / 2.0f, gcy = (G.y + G.w) / 2.0f;
                // written so that a NaN on either side (an anchor or a GT without a finite centre) keeps the GT
                pass = !(gcx <= x_lo) && !(gcx >= x_hi) && !(gcy <= y_lo) && !(gcy >= y_hi);
            }
            unsigned long long mask = __ballot(pass);
            while (mask) {                                 // wave-uniform loop, ascending GT index
                const int j = j0 + __ffsll(mask) - 1;
                mask &= mask - 1;
                const float v = evaluate(s_g[j]);
                if (v > best) { best = v; bi = c0 + j; }
            }
        }
    }
    s_best[wave][lane] = best;
    s_bi[wave][lane] = bi;
    __syncthreads();
    if (wave != 0 || !valid) return;
    if (plain) {
#pragma unroll
        for (int w = 1; w < FW; ++w) {
            const float ob = s_best[w][lane];
            const int oi = s_bi[w][lane];
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
    }
    matched[(long long)b * n + idx] = (M <= 0 || best < 1e-5f) ? -1 : (long long)bi;
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

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Measurement knobs, read once from the environment: OD_MATCH_LEGACY=1 keeps the four-launch chain for every input
// (A/B timing of the two paths), OD_MATCH_RESTORE_GROUPS sets the restore kernel's workgroups per image.
static bool legacy_chain() {
    static const bool v = [] { const char* e = getenv("OD_MATCH_LEGACY"); return e && e[0] == '1'; }();
    return v;
}
// Workgroups per image of the fused path's restore kernel: od_match_restore_groups() if set, else the environment's
// OD_MATCH_RESTORE_GROUPS, else 16 (shortest kernel with the GPU to itself).  Below 2 one workgroup would own every GT
// row and list every entry of its image: more than its lists hold at ordinary sizes.
static std::atomic<int> g_restore_groups{0};
static int restore_groups() {
    const int set = g_restore_groups.load(std::memory_order_relaxed);
    if (set >= 2) return set;
    static const int v = [] {
        const char* e = getenv("OD_MATCH_RESTORE_GROUPS");
        const int g = e ? atoi(e) : 0;
        return (g >= 2 && g <= 64) ? g : 16;
    }();
    return v;
}

extern "C" int od_match_restore_groups(int32_t groups) {
    if (groups != 0 && (groups < 2 || groups > 64)) return OD_ERR_ARG;
    return g_restore_groups.exchange(groups, std::memory_order_relaxed);
}

// OD_MATCH_XCD=0: plain grid of the fused path's restore kernel (default: its XCD-aware grid)
static int xcd_mapping() {
    static const int v = [] {
        const char* e = getenv("OD_MATCH_XCD");
        return e ? atoi(e) : 1;
    }();
    return v;
}

static size_t heavy_capacity(int64_t m_total, int32_t batch) {   // = entry_base(batch, m_total)
    return (size_t)HCAP_PER_GT * (size_t)(m_total > 0 ? m_total : 0) + 256 * (size_t)batch;
}

// Most wave tiles a layout can have: grids are tiled as grids only if that costs at most a quarter more tiles than plain
// tiling (+1 each), and every plain segment rounds up once.
static size_t tiles_cap(int64_t n) { return (size_t)((n + WTILE - 1) / WTILE) * 5 / 4 + 8 * MAX_SEGMENTS; }

// Most cooperative tiles the one-launch path designates (their entry regions are part of the workspace).
static size_t coop_cap(int64_t n) { return tiles_cap(n) / 16 + 8; }

static void match_ws_layout(int64_t m_total, int32_t batch, int64_t n, size_t off[10]) {
    const size_t wtiles = tiles_cap(n);
    const size_t tiles = (size_t)batch * wtiles;
    off[0] = 0;                                                                     // gt_max_key
    off[1] = off[0] + align256((size_t)(m_total > 0 ? m_total : 0) * GSTRIDE * sizeof(uint32_t));   // argmax
    off[2] = off[1] + align256((size_t)batch * (size_t)n * sizeof(int));            // wave_cnt
    off[3] = off[2] + align256(tiles * sizeof(int));                                // wave_list
    off[4] = off[3] + align256(tiles * LCAP * sizeof(uint2));                       // (reserved)
    off[5] = off[4] + align256((size_t)batch * sizeof(int));                        // 256 B reserved, then heavy_count
    off[6] = off[5] + 256 + (size_t)batch * 128;                                    // heavy_queue
    off[7] = off[6] + align256(tiles * sizeof(int4));                               // heavy_entries
    off[8] = off[7] + align256(heavy_capacity(m_total, batch) * sizeof(uint4));     // coop_entries
    off[9] = off[8] + align256((size_t)batch * coop_cap(n) * CCAP * sizeof(uint2)); // end
}

extern "C" size_t od_match_retinanet_workspace_bytes(int64_t m_total, int32_t batch, int64_t n) {
    if (m_total < 0 || batch < 0 || n < 0) return 0;
    size_t off[10];
    match_ws_layout(m_total, batch, n, off);
    return off[9];
}

extern "C" int od_match_retinanet(const float* anchors, int64_t anchor_batch_stride, int64_t n,
                                  const float* gt_boxes, const int32_t* gt_offsets, int64_t m_total,
                                  int32_t batch, float high, float low, int32_t allow_low_quality,
                                  int64_t* matched, void* workspace, size_t workspace_bytes, void* stream) {
    return od_match_retinanet_grids(anchors, anchor_batch_stride, n, nullptr, 0, gt_boxes, gt_offsets, m_total, batch, high,
                                    low, allow_low_quality, matched, workspace, workspace_bytes, stream);
}

extern "C" int od_match_retinanet_grids(const float* anchors, int64_t anchor_batch_stride, int64_t n,
                                        const int64_t* anchor_grids, int32_t num_grids,
                                        const float* gt_boxes, const int32_t* gt_offsets, int64_t m_total,
                                        int32_t batch, float high, float low, int32_t allow_low_quality,
                                        int64_t* matched, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || batch < 0 || m_total < 0 || num_grids < 0 || (num_grids > 0 && !anchor_grids)) return OD_ERR_ARG;
    // Segments of the anchor list (see TileLayout): the grids that can be tiled as grids, plain stretches elsewhere.
    // Grids must lie inside [0, n), ascending and disjoint, else the argument is wrong.
    TileLayout layout;
    memset(&layout, 0, sizeof(layout));
    long long wtiles_ll = 0;
    {
        bool fits = true;                        // false: more segments than the layout holds -> one plain segment
        auto add_segment = [&](int64_t first, int64_t len, int64_t row_len, int64_t rows) {
            if (len <= 0) return;
            if (layout.segments == MAX_SEGMENTS) { fits = false; return; }
            const int k = layout.segments++;
            layout.first[k] = (int)first;
            layout.row_len[k] = (int)row_len;
            layout.rows[k] = (int)rows;
            wtiles_ll += row_len > 0 ? ((row_len + 63) / 64) * ((rows + R - 1) / R) : (len + WTILE - 1) / WTILE;
            layout.tile_end[k] = (int)(wtiles_ll < 0x7fffffff ? wtiles_ll : 0x7fffffff);
        };
        int64_t prev_end = 0, plain_from = 0;
        for (int32_t i = 0; i < num_grids; ++i) {
            const int64_t first = anchor_grids[3 * i], row_len = anchor_grids[3 * i + 1], rows = anchor_grids[3 * i + 2];
            if (first < prev_end || row_len <= 0 || rows <= 0 || row_len > 0x7fffffffll || rows > 0x7fffffffll) return OD_ERR_ARG;
            if (row_len * rows > n - first) return OD_ERR_ARG;
            prev_end = first + row_len * rows;
            // worth tiling as a grid: whole 64-anchor runs and R rows exist, the overlapping last column / band does not
            // inflate the number of tiles by more than a quarter, and the grid is a large one (at least 64 tiles).  The
            // tiles of a small grid -- a coarse pyramid level, large anchors -- meet most GT whatever their shape;
            // tighter boxes there only move tiles from the cooperative kernel to single waves with long serial loops
            // (measured on the 32 x 32 level of the 1024 x 1024 pyramid: assign kernel 49 -> 54 us).
            const int64_t tiles = ((row_len + 63) / 64) * ((rows + R - 1) / R);
            const int64_t plain_tiles = (row_len * rows + WTILE - 1) / WTILE;
            if (row_len < 64 || rows < R || plain_tiles < 64 || tiles * 4 > plain_tiles * 5 + 4) continue;
            add_segment(plain_from, first - plain_from, 0, 0);
            add_segment(first, row_len * rows, row_len, rows);
            plain_from = prev_end;
        }
        add_segment(plain_from, n - plain_from, 0, 0);
        if (!fits || n > 0x7fffffffll) {
            memset(&layout, 0, sizeof(layout));
            wtiles_ll = 0;
            fits = true;
            add_segment(0, n, 0, 0);
        }
    }
    if (n == 0 || batch == 0) return OD_OK;
    if (!anchors || !gt_offsets || !matched || (m_total > 0 && !gt_boxes)) return OD_ERR_ARG;
    if (anchor_batch_stride != 0 && anchor_batch_stride < n) return OD_ERR_ARG;
    if (((uintptr_t)anchors & 15) || ((uintptr_t)gt_boxes & 15)) return OD_ERR_ARG;
    if (m_total >= (1 << 23) || batch > 65535 || n > 0x7fffffffll * 64) return OD_ERR_LIMIT;   // GT index packs into 23 bits
    if (workspace_bytes < od_match_retinanet_workspace_bytes(m_total, batch, n)) return OD_ERR_WORKSPACE;
    if (!workspace || ((uintptr_t)workspace & 255)) return OD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    size_t off[10];
    match_ws_layout(m_total, batch, n, off);
    if (wtiles_ll > (long long)tiles_cap(n)) return OD_ERR_WORKSPACE;         // cannot happen: see tiles_cap
    const int wtiles = (int)wtiles_ll;
    const long long total_tiles = (long long)batch * wtiles;
    if (total_tiles > 0x7fffffff || (long long)wtiles * m_total > 0x7fffffffll || wtiles >= (1 << 24)) return OD_ERR_LIMIT;   // int counters
    MatchWs ws;
    char* w = (char*)workspace;
    ws.gt_max_key = (uint32_t*)(w + off[0]);
    ws.argmax = (void*)(w + off[1]);
    ws.argmax16 = m_total <= 65536 ? 1 : 0;
    ws.wave_cnt = (int*)(w + off[2]);
    ws.wave_list = (uint2*)(w + off[3]);
    ws.heavy_count = (unsigned long long*)(w + off[5] + 256);
    ws.heavy_queue = (int4*)(w + off[6]);
    ws.heavy_entries = (uint4*)(w + off[7]);
    ws.list_stride = total_tiles;
    ws.coop_entries = (uint2*)(w + off[8]);
    ws.tile_head = ws.wave_list;                                   // the one-launch path lays the same memory out tile-major
    ws.tile_extra = ws.wave_list + (size_t)total_tiles * 8;
    const int restore = (allow_low_quality && m_total > 0) ? 1 : 0;
    const long long stride = (long long)anchor_batch_stride;
    // One launch (+ the restore launch) for ordinary target counts.  Cooperative tiles = the tiles after the last
    // grid-tiled segment (the coarse pyramid levels: they meet most GT whatever the targets are); an anchor list
    // nobody described: its last 1/32.
    if (m_total <= (int64_t)FUSED_AVG_GT * batch && m_total <= FUSED_MAX_GT && !legacy_chain()) {
        int coop = 0;
        bool any_grid = false;
        for (int k = 0; k < layout.segments; ++k) any_grid = any_grid || layout.row_len[k] > 0;
        if (any_grid) {
            int k = layout.segments - 1;
            if (layout.row_len[k] == 0) coop = wtiles - (k > 0 ? layout.tile_end[k - 1] : 0);
        } else {
            coop = (wtiles + 31) / 32;
        }
        if (m_total == 0) coop = 0;
        if (coop > (int)coop_cap(n)) coop = (int)coop_cap(n);
        if (coop > wtiles) coop = wtiles;
        // (Dealing the blocks of tiles to the XCDs so that the shared anchor table crosses the fabric once instead of
        // eight times cut the kernel's fetched bytes and made it 2 us slower: it is bound by vector instructions.)
        const long long light_blocks = ((long long)(wtiles - coop) + MT / 64 - 1) / (MT / 64);
        if (coop + light_blocks <= 65535) {
            OD_LAUNCH_KERNEL(OD_KERNEL_MATCH_ASSIGN, match_fused_kernel, dim3((unsigned)batch, (unsigned)(coop + light_blocks)),
                             dim3(MT), 0, s, (const float4*)anchors, stride, (long long)n, wtiles, coop, layout,
                             (const float4*)gt_boxes, gt_offsets, ws, restore, high, low, (long long*)matched);
            if (restore) {
                const size_t lds = (size_t)(m_total > 0 ? m_total : 1) * sizeof(uint32_t);
                const unsigned G = (unsigned)restore_groups();
                const int r_xcd = xcd_mapping() ? 1 : 0;
                const dim3 rgrid = r_xcd ? dim3(8u * G, (unsigned)((batch + 7) / 8)) : dim3(G, (unsigned)batch);
                OD_LAUNCH_KERNEL(OD_KERNEL_MATCH_RESTORE, match_restore_lds_kernel, rgrid,
                                 dim3(RL_T), lds, s, (const float4*)anchors, stride, (long long)n, wtiles, coop, layout,
                                 (const float4*)gt_boxes, gt_offsets, ws, (long long*)matched, (int)batch, r_xcd);
            }
            return OD_OK;
        }
    }
    const unsigned tile_blocks = (unsigned)((wtiles + MT / 64 - 1) / (MT / 64));
    const int image_fastest = tile_blocks <= 65535u ? 1 : 0;
    dim3 grid = image_fastest ? dim3((unsigned)batch, tile_blocks) : dim3(tile_blocks, (unsigned)batch);
    {
        const long long init_threads = m_total > batch ? m_total : batch;
        OD_LAUNCH_KERNEL(OD_KERNEL_MATCH_INIT, match_init_kernel, dim3((unsigned)((init_threads + 255) / 256)), dim3(256),
                         0, s, ws, (long long)m_total, (int)batch, restore);
    }
    OD_LAUNCH_KERNEL(OD_KERNEL_MATCH_ASSIGN, match_assign_kernel, grid, dim3(MT), ASSIGN_LDS, s, (const float4*)anchors, stride,
                     (long long)n, wtiles, layout, image_fastest, (const float4*)gt_boxes, gt_offsets, ws, restore, high, low,
                     (long long*)matched);
    if (m_total > HEAVY) {
        int heavy_y = 4096 / batch > 64 ? 4096 / batch : 64;
        if (heavy_y > wtiles) heavy_y = wtiles;
        OD_LAUNCH_KERNEL(OD_KERNEL_MATCH_HEAVY, match_heavy_kernel, dim3((unsigned)batch, (unsigned)heavy_y),
                         dim3(HT), 0, s, (const float4*)anchors, stride, (long long)n, wtiles, layout,
                         (const float4*)gt_boxes, gt_offsets, ws, restore, high, low, (long long*)matched);
    }
    if (restore) {
        const long long light_items = ((total_tiles + RT - 1) / RT) * LSTEP;      // (entry slot below LSTEP, 256 tiles)
        const int light_blocks = (int)(light_items < 16384 ? light_items : 16384);
        const int heavy_entry_blocks = m_total > HEAVY ? batch * (512 / batch > 1 ? 512 / batch : 1) : 0;
        OD_LAUNCH_KERNEL(OD_KERNEL_MATCH_RESTORE, match_restore_kernel,
                         dim3((unsigned)(light_blocks + heavy_entry_blocks + batch * ZB)), dim3(RT), 0, s,
                         (const float4*)anchors, stride, (long long)n, wtiles, layout, (int)batch, light_blocks,
                         heavy_entry_blocks, (m_total > (long long)DENSE_GT * batch) ? 1 : 0, (const float4*)gt_boxes,
                         gt_offsets, ws, (long long*)matched);
    }
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
    if (batch > 65535 || (n + 63) / 64 > 0x7fffffffll) return OD_ERR_LIMIT;
    const long long blocks = (n + 63) / 64;
    const int image_fastest = blocks <= 65535 ? 1 : 0;
    dim3 grid = image_fastest ? dim3((unsigned)batch, (unsigned)blocks) : dim3((unsigned)blocks, (unsigned)batch);
    OD_LAUNCH_KERNEL(OD_KERNEL_MATCH_FCOS, match_fcos_kernel, grid, dim3(FT), 0, (hipStream_t)stream,
                     (const float4*)anchors, (long long)anchor_batch_stride, (long long)n, (long long)n_first_level,
                     (long long)n_last_level, (const float4*)gt_boxes, gt_offsets, center_sampling_radius,
                     (long long*)matched, image_fastest);
    return OD_OK;
}

extern "C" int od_box_iou(const float* boxes1, int64_t m, const float* boxes2, int64_t n, float* out, void* stream) {
    if (m < 0 || n < 0) return OD_ERR_ARG;
    if (m == 0 || n == 0) return OD_OK;
    if (!boxes1 || !boxes2 || !out) return OD_ERR_ARG;
    if (((uintptr_t)boxes1 & 15) || ((uintptr_t)boxes2 & 15)) return OD_ERR_ARG;
    if (m > 65535) return OD_ERR_LIMIT;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)m);
    OD_LAUNCH_KERNEL(OD_KERNEL_BOX_IOU, box_iou_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const
                     float4*)boxes1, (long long)m, (const float4*)boxes2, (long long)n, out);
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
    OD_LAUNCH_KERNEL(OD_KERNEL_MATCHER, matcher_rowmax_kernel, dim3((unsigned)m), dim3(256), 0, s, quality, (long
                     long)n, rowmax);
    OD_LAUNCH_KERNEL(OD_KERNEL_MATCHER, matcher_col_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     quality, (long long)m, (long long)n, rowmax, high, low, allow_low_quality ? 1 : 0, (long
                     long*)matches);
    return OD_OK;
}
