// Anchor generation over the feature pyramid: one launch for all levels.
// Follows torchvision 0.16.2 AnchorGenerator.grid_anchors semantics: per level, int32 shifts
// (x*stride_w, y*stride_h) added to the rounded base anchors, output order (y, x, a), levels
// concatenated finest-first.  Reference trigger: object_detection/retinanet/retinanet_model.py:22-27,42-46.
#include "od_common.h"
#include "../../include/od_boxes.h"

namespace {

struct AnchorParams {
    int num_levels;
    int grid_w[OD_MAX_LEVELS];
    int stride_h[OD_MAX_LEVELS];
    int stride_w[OD_MAX_LEVELS];
    int num_base[OD_MAX_LEVELS];
    long long level_start[OD_MAX_LEVELS + 1];            // first anchor index of each level
    float base[OD_MAX_LEVELS][OD_MAX_BASE_ANCHORS][4];   // rounded cell anchors, host computed
};

__global__ __launch_bounds__(256) void anchor_kernel(AnchorParams p, float4* __restrict__ out, long long n_total) {
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_total) return;
    int l = 0;
#pragma unroll
    for (int i = 1; i < OD_MAX_LEVELS; ++i)
        if (i < p.num_levels && idx >= p.level_start[i]) l = i;
    long long local = idx - p.level_start[l];
    int A = p.num_base[l];
    int a = (int)(local % A);
    long long loc = local / A;
    int gw = p.grid_w[l];
    int x = (int)(loc % gw);
    int y = (int)(loc / gw);
    float fx = (float)(x * p.stride_w[l]);
    float fy = (float)(y * p.stride_h[l]);
    const float* b = p.base[l][a];
    out[idx] = make_float4(fx + b[0], fy + b[1], fx + b[2], fy + b[3]);
}

}  // namespace

extern "C" int od_version(void) { return 100; }

extern "C" int od_anchor_generate(const int32_t* grid_h, const int32_t* grid_w, const int32_t* stride_h,
                                  const int32_t* stride_w, const int32_t* num_base, const float* base_anchors,
                                  int32_t num_levels, float* out, int64_t n_total, void* stream) {
    if (!grid_h || !grid_w || !stride_h || !stride_w || !num_base || !base_anchors || !out) return OD_ERR_ARG;
    if (num_levels < 1 || num_levels > OD_MAX_LEVELS) return OD_ERR_LIMIT;
    AnchorParams p;
    p.num_levels = num_levels;
    long long start = 0;
    const float* b = base_anchors;
    for (int l = 0; l < OD_MAX_LEVELS; ++l) {
        if (l < num_levels) {
            if (num_base[l] < 1 || num_base[l] > OD_MAX_BASE_ANCHORS) return OD_ERR_LIMIT;
            if (grid_h[l] < 0 || grid_w[l] < 0) return OD_ERR_ARG;
            p.grid_w[l] = grid_w[l] > 0 ? grid_w[l] : 1;
            p.stride_h[l] = stride_h[l];
            p.stride_w[l] = stride_w[l];
            p.num_base[l] = num_base[l];
            p.level_start[l] = start;
            for (int a = 0; a < num_base[l]; ++a)
                for (int c = 0; c < 4; ++c) p.base[l][a][c] = b[4 * a + c];
            b += 4 * num_base[l];
            start += (long long)grid_h[l] * grid_w[l] * num_base[l];
        } else {
            p.grid_w[l] = 1; p.stride_h[l] = 0; p.stride_w[l] = 0; p.num_base[l] = 1;
            p.level_start[l] = start;
        }
    }
    p.level_start[OD_MAX_LEVELS] = start;
    if (start != n_total) return OD_ERR_ARG;
    if (n_total == 0) return OD_OK;
    if (((uintptr_t)out & 15) != 0) return OD_ERR_ARG;
    unsigned blocks = (unsigned)((n_total + 255) / 256);
    OD_LAUNCH_KERNEL(OD_KERNEL_ANCHOR, anchor_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p,
                     reinterpret_cast<float4*>(out), (long long)n_total);
    return OD_OK;
}
