// Shared device helpers for the box-pipeline kernels (gfx950 / CDNA4, wave64).
// Build flags that the numerics below rely on: -ffp-contract=off (no FMA fusion), default
// correctly rounded fp32 division and sqrt, fp32 denormals preserved (hipcc defaults).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <tuple>
#include <utility>

#define OD_OK 0
#define OD_ERR_ARG (-1)        // null pointer / negative size / inconsistent shapes
#define OD_ERR_LIMIT (-2)      // size beyond a documented limit of this library
#define OD_ERR_WORKSPACE (-3)  // workspace too small

#define OD_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

#define OD_HIP(call)                              \
    do {                                          \
        hipError_t e__ = (call);                  \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

namespace od {

// Per-kernel timing (od_profile.hip): true when launches of `kernel_id` are being timed; *start / *stop are then
// the events that receive the kernel's own begin and end time stamps.
bool profile_events(int kernel_id, hipEvent_t* start, hipEvent_t* stop);

// hipExtLaunchKernel with the arguments converted to the kernel's parameter types: the start / stop events take
// the time stamps of the dispatch itself (what a kernel trace shows), not those of markers around it.
template <typename... K, typename... A, size_t... I>
inline hipError_t launch_timed_impl(void (*kernel)(K...), dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                                    hipEvent_t start, hipEvent_t stop, std::index_sequence<I...>, A&&... args) {
    std::tuple<K...> params(static_cast<K>(std::forward<A>(args))...);
    void* ptrs[] = {(void*)&std::get<I>(params)...};
    return hipExtLaunchKernel((const void*)kernel, grid, block, ptrs, shmem, s, start, stop, 0);
}
template <typename... K, typename... A>
inline hipError_t launch_timed(void (*kernel)(K...), dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                               hipEvent_t start, hipEvent_t stop, A&&... args) {
    static_assert(sizeof...(K) == sizeof...(A), "one argument per kernel parameter");
    return launch_timed_impl(kernel, grid, block, shmem, s, start, stop, std::index_sequence_for<K...>{},
                             std::forward<A>(args)...);
}

}  // namespace od

// Launches kernel `id`; when profiling selects it, through the timed form.
#define OD_LAUNCH_KERNEL(id, kernel, grid, block, shmem, stream, ...)                                       \
    do {                                                                                                    \
        hipEvent_t ea__ = nullptr, eb__ = nullptr;                                                          \
        if (od::profile_events((id), &ea__, &eb__)) {                                                       \
            OD_HIP(od::launch_timed(kernel, grid, block, shmem, stream, ea__, eb__, __VA_ARGS__));          \
        } else {                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                            \
        }                                                                                                   \
        OD_LAUNCH_CHECK();                                                                                  \
    } while (0)

namespace od {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// Order-preserving map float -> uint32 (total order, -0 < +0, NaN(+) above +inf).
__device__ __forceinline__ uint32_t ordered_key(float f) {
    uint32_t b = f2u(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_key_inv(uint32_t k) {
    return u2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- wave64 reductions over DPP (no LDS traffic). Result valid in lane 63. --------------------
// old = 0 with bound_ctrl off: lanes without a source keep 0, the identity of umax / add / or.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

__device__ __forceinline__ uint32_t wave_umax_lane63(uint32_t v) {
    v = max(v, dpp_u32<0x111>(v));        // row_shr:1
    v = max(v, dpp_u32<0x112>(v));        // row_shr:2
    v = max(v, dpp_u32<0x114>(v));        // row_shr:4
    v = max(v, dpp_u32<0x118>(v));        // row_shr:8  -> lane 15 of each row holds the row max
    v = max(v, dpp_u32<0x142, 0xa>(v));   // row_bcast:15 into rows 1 and 3
    v = max(v, dpp_u32<0x143, 0xc>(v));   // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t wave_umax_all(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_umax_lane63(v), 63);
}

__device__ __forceinline__ float wave_fmax_all(float v) {   // any floats (uses the ordered key)
    return ordered_key_inv(wave_umax_all(ordered_key(v)));
}
__device__ __forceinline__ float wave_fmin_all(float v) {
    return ordered_key_inv(~wave_umax_all(~ordered_key(v)));
}

__device__ __forceinline__ uint32_t wave_or_all(uint32_t v) {
    v |= dpp_u32<0x111>(v);
    v |= dpp_u32<0x112>(v);
    v |= dpp_u32<0x114>(v);
    v |= dpp_u32<0x118>(v);
    v |= dpp_u32<0x142, 0xa>(v);
    v |= dpp_u32<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_and_all(uint32_t v) { return ~wave_or_all(~v); }

__device__ __forceinline__ uint32_t wave_sum_all(uint32_t v) {
    v += dpp_u32<0x111>(v);
    v += dpp_u32<0x112>(v);
    v += dpp_u32<0x114, 0xf, 0xe>(v);     // rocPRIM-style masks keep the partial sums disjoint
    v += dpp_u32<0x118, 0xf, 0xc>(v);
    v += dpp_u32<0x142, 0xa>(v);
    v += dpp_u32<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// inclusive prefix sum over the lanes of a wave (lane i: v[0] + ... + v[i]); the same DPP steps as wave_sum_all
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += dpp_u32<0x111>(v);
    v += dpp_u32<0x112>(v);
    v += dpp_u32<0x114, 0xf, 0xe>(v);
    v += dpp_u32<0x118, 0xf, 0xc>(v);
    v += dpp_u32<0x142, 0xa>(v);
    v += dpp_u32<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(uint64_t mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__device__ __forceinline__ float bcast_lane(float v, int lane) {   // lane must be wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int bcast_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---- specified fp32 exp / sigmoid: one rounding per listed operation, no FMA -------------------
// k = rint(x*LOG2E); r = (x - k*LN2_HI) - k*LN2_LO; Horner degree-5 on r, times r*r, + r, + 1;
// result = (p * 2^k1) * 2^k2 with k1 = trunc(k/2), k2 = k - k1.  x > EXP_HI -> inf, x < EXP_LO -> 0.
__device__ __forceinline__ float spec_expf(float x) {
    const float LOG2E = 1.44269504088896341f, LN2_HI = 0.693359375f, LN2_LO = -2.12194440e-4f;
    const float EXP_HI = 88.72283935546875f, EXP_LO = -103.97208404541015625f;
    float xc = fminf(fmaxf(x, EXP_LO), EXP_HI);
    float k = __builtin_rintf(xc * LOG2E);
    float r = xc - k * LN2_HI;
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
    float t = p * u2f((uint32_t)(k1 + 127) << 23);
    t = t * u2f((uint32_t)(k2 + 127) << 23);
    t = (x > EXP_HI) ? __builtin_inff() : t;
    t = (x < EXP_LO) ? 0.0f : t;
    t = (x != x) ? x : t;
    return t;
}

__device__ __forceinline__ float spec_sigmoidf(float x) {
    float e = spec_expf(-x);
    float d = 1.0f + e;
    return 1.0f / d;
}

// ---- IoU exactly as torchvision.ops.boxes.box_iou computes one (gt, anchor) entry -------------
// inter = clamp(min(x2)-max(x1),0) * clamp(min(y2)-max(y1),0); union = (area_gt + area_a) - inter
// v_min_f32 / v_max_f32 issued directly.  fminf()/fmaxf() make the compiler quiet a possible signalling NaN
// of every operand it cannot trace to an arithmetic result (one extra v_max per operand and use, even for
// loop-invariant registers); for non-NaN inputs the bare instructions return the same value.
__device__ __forceinline__ float hw_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float hw_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float iou_pair(float gx1, float gy1, float gx2, float gy2, float garea,
                                          float ax1, float ay1, float ax2, float ay2, float aarea) {
    float w = fmaxf(hw_min(gx2, ax2) - hw_max(gx1, ax1), 0.0f);
    float h = fmaxf(hw_min(gy2, ay2) - hw_max(gy1, ay1), 0.0f);
    float inter = w * h;
    float uni = (garea + aarea) - inter;
    return inter / uni;
}

// The same with a wave-uniform first box (its coordinates sit in scalar registers, e.g. after v_readlane): the
// scalar operand goes into the instruction directly instead of through a v_mov per coordinate and call.
__device__ __forceinline__ float hw_min_su(float uniform, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(b));
    return r;
}
__device__ __forceinline__ float hw_max_su(float uniform, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(b));
    return r;
}
__device__ __forceinline__ float iou_pair_uniform(float gx1, float gy1, float gx2, float gy2, float garea,
                                                  float ax1, float ay1, float ax2, float ay2, float aarea) {
    float w = fmaxf(hw_min_su(gx2, ax2) - hw_max_su(gx1, ax1), 0.0f);
    float h = fmaxf(hw_min_su(gy2, ay2) - hw_max_su(gy1, ay1), 0.0f);
    float inter = w * h;
    float uni = (garea + aarea) - inter;
    return inter / uni;
}

__device__ __forceinline__ float box_area(float x1, float y1, float x2, float y2) {
    return (x2 - x1) * (y2 - y1);
}

// ---- block-wide bitonic sort of 64-bit keys in LDS, descending --------------------------------------
// For a workgroup of NT threads (NT = 1024 in the callers) and n_pad = power of two, 64 <= n_pad <= 8 * NT.
// Thread t keeps the E = max(1, n_pad / NT) consecutive elements [t*E, t*E + E) in registers.  A stage of
// stride j runs inside the registers (j < E), through lane shuffles (E <= j < 64 E, partners in the same
// wave) or through LDS (partners in another wave); only the last kind needs barriers: 10 LDS stages for
// n_pad = 1024 or 4096 instead of 55 or 78.  Keys must be in s[0, n_pad) and visible (barrier) on entry;
// on return they are sorted and visible to all threads.
// Value of lane (lane ^ LX) for LX = 1, 2, 4, 8 through DPP moves (the partner is in the lane's row of 16: no trip
// through the LDS crossbar that a general shuffle makes); other distances use the shuffle.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long old, unsigned long long v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)old, (int)(uint32_t)v, CTRL, 0xf, BANK_MASK, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(old >> 32), (int)(uint32_t)(v >> 32), CTRL, 0xf,
                                                              BANK_MASK, false);
    return ((unsigned long long)hi << 32) | lo;
}
template <int LX>
__device__ __forceinline__ unsigned long long xor_lane_u64(unsigned long long v) {
    if (LX == 1) return dpp_u64<0xB1, 0xf>(v, v);                       // quad_perm [1,0,3,2]
    if (LX == 2) return dpp_u64<0x4E, 0xf>(v, v);                       // quad_perm [2,3,0,1]
    if (LX == 4) return dpp_u64<0x114, 0xa>(dpp_u64<0x104, 0x5>(v, v), v);   // row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3
    if (LX == 8) return dpp_u64<0x128, 0xf>(v, v);                      // row_ror:8
    return __shfl_xor(v, LX);
}

// One compare-exchange stage of the bitonic network between lanes LX apart (element stride LX * E), sequence length k.
template <int LX, int E>
__device__ __forceinline__ void bitonic_lane_stage(unsigned long long (&v)[E], int base, int k) {
    constexpr int j = LX * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = base + e;
        const unsigned long long o = xor_lane_u64<LX>(v[e]);
        const bool keep_larger = ((i & j) == 0) == ((i & k) == 0);
        const bool o_larger = o > v[e];
        v[e] = (o_larger == keep_larger) ? o : v[e];
    }
}

template <int E, int NT>
__device__ __forceinline__ void bitonic_sort_desc_impl(unsigned long long* s, int n_pad) {
    const int tid = threadIdx.x;
    const int active = n_pad / E;
    const bool on = tid < active;                 // whole waves: n_pad >= 64
    const int base = tid * E;
    constexpr int WSPAN = 64 * E;                 // elements held by one wave
    unsigned long long v[E];
    bool in_regs = false;
    for (int k = 2; k <= n_pad; k <<= 1) {
        int j = k >> 1;
        if (j >= WSPAN) {
            if (in_regs && on) {
#pragma unroll
                for (int e = 0; e < E; ++e) s[base + e] = v[e];
            }
            in_regs = false;
            __syncthreads();
            for (; j >= WSPAN; j >>= 1) {
                const int pairs = n_pad >> 1;
                for (int pr = tid; pr < pairs; pr += NT) {
                    const int lo = ((pr & ~(j - 1)) << 1) | (pr & (j - 1));   // insert a 0 bit at log2(j)
                    const unsigned long long x = s[lo], y = s[lo | j];
                    const bool desc = (lo & k) == 0;
                    if ((x < y) == desc) { s[lo] = y; s[lo | j] = x; }
                }
                __syncthreads();
            }
        }
        if (!in_regs && on) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = s[base + e];
        }
        in_regs = true;
        if (on) {
            const int lx0 = j / E;                // partners in another lane of the wave: distances lx0, lx0 / 2, ... 1
            if (lx0 >= 32) bitonic_lane_stage<32, E>(v, base, k);
            if (lx0 >= 16) bitonic_lane_stage<16, E>(v, base, k);
            if (lx0 >= 8) bitonic_lane_stage<8, E>(v, base, k);
            if (lx0 >= 4) bitonic_lane_stage<4, E>(v, base, k);
            if (lx0 >= 2) bitonic_lane_stage<2, E>(v, base, k);
            if (lx0 >= 1) bitonic_lane_stage<1, E>(v, base, k);
#pragma unroll
            for (int jj = E >> 1; jj >= 1; jj >>= 1) {   // partners in this thread's registers
                if (jj <= (k >> 1)) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & jj) == 0) {
                            const bool desc = ((base + e) & k) == 0;
                            const unsigned long long x = v[e], y = v[e | jj];
                            if ((x < y) == desc) { v[e] = y; v[e | jj] = x; }
                        }
                    }
                }
            }
        }
    }
    if (on) {
#pragma unroll
        for (int e = 0; e < E; ++e) s[base + e] = v[e];
    }
    __syncthreads();
}

template <int NT>
__device__ __forceinline__ void block_bitonic_sort_desc(unsigned long long* s, int n_pad) {
    const int e = n_pad / NT;
    if (e <= 1) bitonic_sort_desc_impl<1, NT>(s, n_pad);
    else if (e == 2) bitonic_sort_desc_impl<2, NT>(s, n_pad);
    else if (e == 4) bitonic_sort_desc_impl<4, NT>(s, n_pad);
    else bitonic_sort_desc_impl<8, NT>(s, n_pad);
}

}  // namespace od
