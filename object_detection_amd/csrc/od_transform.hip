// Image side of torchvision GeneralizedRCNNTransform.forward in one pass: normalize, bilinear resize and the
// zero-padded batch tensor (normalize -> resize -> batch_images; the reference configures the transform through
// get_model(min_size=size, max_size=size), reference object_detection/train.py:142-144).
// Follows torch.nn.functional.interpolate(mode="bilinear", align_corners=False, recompute_scale_factor=True):
//   scale = float(in) / out;  src = fma(scale, dst + 0.5, -0.5), clamped at 0;  i0 = int(src), i1 = i0 + (i0 < in - 1);
//   out = (1-ly) * ((1-lx) * p00 + lx * p01) + ly * ((1-lx) * p10 + lx * p11)
// on the normalized taps (x - mean) / std, every operation rounded separately.  One HBM pass: the source images are
// read once (neighbouring outputs share taps through the caches) and the batch tensor, padding included, is
// written once; the torch sequence makes three passes (normalize, interpolate, copy into the batch).
#include "od_common.h"
#include "../../include/od_boxes.h"

namespace {

constexpr int PRE_MAX = 48;             // images per launch (kernel-argument block)

struct PrepImage {
    const float* src;                   // [channels, h, w]
    int h, w;                           // source size
    int oh, ow;                         // resized size
    float sh, sw;                       // float(h) / oh, float(w) / ow
};

struct PrepParams {
    PrepImage img[PRE_MAX];
    float mean[4], std[4];
    int channels;
    int out_h, out_w;                   // padded batch size
};

__device__ __forceinline__ float prep_pixel(const PrepImage& im, const float* plane, int y, int x, float mean, float sd) {
    // one rounding for scale * (dst + 0.5) - 0.5, as the torch kernels compile it (fused multiply-add); with two
    // roundings the interpolation weights of a few-hundred-pixel image move by about 1e-5
    float sy = fmaf(im.sh, (float)y + 0.5f, -0.5f);
    float sx = fmaf(im.sw, (float)x + 0.5f, -0.5f);
    sy = sy < 0.0f ? 0.0f : sy;
    sx = sx < 0.0f ? 0.0f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 < im.h - 1 ? y0 : im.h - 1;
    x0 = x0 < im.w - 1 ? x0 : im.w - 1;
    const int y1 = y0 + (y0 < im.h - 1 ? 1 : 0), x1 = x0 + (x0 < im.w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* r0 = plane + (long long)y0 * im.w;
    const float* r1 = plane + (long long)y1 * im.w;
    const float p00 = (r0[x0] - mean) / sd, p01 = (r0[x1] - mean) / sd;
    const float p10 = (r1[x0] - mean) / sd, p11 = (r1[x1] - mean) / sd;
    return hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
}

// block = 64 groups of four pixels x 4 rows; grid = (x tiles, row tiles, image * channel)
__global__ __launch_bounds__(256) void prep_kernel(PrepParams p, float* __restrict__ out, int vec_ok) {
    const int b = blockIdx.z / p.channels, c = blockIdx.z - b * p.channels;
    const PrepImage& im = p.img[b];
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= p.out_h || x >= p.out_w) return;
    const float* plane = im.src + (long long)c * im.h * im.w;
    const float mean = p.mean[c], sd = p.std[c];
    float v[4];
    if (im.h == im.oh && im.w == im.ow) {
        // No resize: scale is exactly 1, every source coordinate falls on a pixel and the other three taps have
        // weight 0, so the output is the normalized pixel itself (for finite pixels) -- a plain stream.
        const float* row = plane + (long long)y * im.w + x;
        if (y < im.oh && x + 3 < im.ow && ((uintptr_t)row & 15) == 0) {
            const float4 q = *reinterpret_cast<const float4*>(row);
            v[0] = (q.x - mean) / sd; v[1] = (q.y - mean) / sd; v[2] = (q.z - mean) / sd; v[3] = (q.w - mean) / sd;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (y < im.oh && x + k < im.ow) ? (row[k] - mean) / sd : 0.0f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = (y < im.oh && x + k < im.ow) ? prep_pixel(im, plane, y, x + k, mean, sd) : 0.0f;
    }
    float* dst = out + (((long long)b * p.channels + c) * p.out_h + y) * p.out_w + x;
    if (vec_ok) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x + k < p.out_w) dst[k] = v[k];
    }
}

}  // namespace

extern "C" int od_preprocess_images(const float* const* images, const int32_t* image_h, const int32_t* image_w,
                                    const int32_t* resized_h, const int32_t* resized_w, int32_t batch,
                                    int32_t channels, const float* mean, const float* std, float* out,
                                    int32_t out_h, int32_t out_w, void* stream) {
    if (batch < 0 || channels < 1 || channels > 4 || out_h < 1 || out_w < 1) return OD_ERR_ARG;
    if (batch == 0) return OD_OK;
    if (!images || !image_h || !image_w || !resized_h || !resized_w || !mean || !std || !out) return OD_ERR_ARG;
    if ((long long)out_h > 0x3fffc) return OD_ERR_LIMIT;          // grid rows
    for (int i = 0; i < batch; ++i) {
        if (!images[i] || ((uintptr_t)images[i] & 3)) return OD_ERR_ARG;
        if (image_h[i] < 1 || image_w[i] < 1 || resized_h[i] < 1 || resized_w[i] < 1) return OD_ERR_ARG;
        if (resized_h[i] > out_h || resized_w[i] > out_w) return OD_ERR_ARG;
    }
    const int vec_ok = ((out_w & 3) == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    for (int first = 0; first < batch; first += PRE_MAX) {
        const int count = (batch - first < PRE_MAX) ? batch - first : PRE_MAX;
        PrepParams p;
        for (int i = 0; i < PRE_MAX; ++i) {
            const int j = first + (i < count ? i : 0);
            p.img[i].src = images[j];
            p.img[i].h = image_h[j];
            p.img[i].w = image_w[j];
            p.img[i].oh = resized_h[j];
            p.img[i].ow = resized_w[j];
            p.img[i].sh = (float)image_h[j] / (float)resized_h[j];
            p.img[i].sw = (float)image_w[j] / (float)resized_w[j];
        }
        for (int k = 0; k < 4; ++k) {
            p.mean[k] = k < channels ? mean[k] : 0.0f;
            p.std[k] = k < channels ? std[k] : 1.0f;
        }
        p.channels = channels;
        p.out_h = out_h;
        p.out_w = out_w;
        float* dst = out + (long long)first * channels * out_h * out_w;
        const dim3 grid((unsigned)((out_w + 255) / 256), (unsigned)((out_h + 3) / 4), (unsigned)(count * channels));
        OD_LAUNCH_KERNEL(OD_KERNEL_PREPROCESS, prep_kernel, grid, dim3(256), 0, s, p, dst, vec_ok);
    }
    return OD_OK;
}
