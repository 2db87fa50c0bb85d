// image_ops.hip -- the data path's per-image work on the device: PIL-compatible resize of an 8-bit interleaved
// image, ToTensor, Normalize and the optional horizontal flip in two launches, for
//   datasets/example_dataset.py:76-100 (Image.open -> transform -> to_tensor -> normalize) and :106-131 (img.resize((new_w, new_h))),
//   datasets/target_dataset.py:23-71 (resize to exactly new_w x new_h).
// Image.resize is Pillow's two-pass separable convolution on 8-bit samples (Resample.c, not under /root/reference: a dependency
// of the reference; restated in scda_amd/device_image.py, which builds the coefficient tables, and pinned against PIL itself in
// tests/test_device_image*.py): per output coordinate a window [xmin, xmin + n) of the input and n fixed-point weights
// (22 fractional bits), accumulated in 32-bit integers from 2^21 and cut back to 8 bits -- horizontally first, INTO 8-BIT
// samples, then vertically.  Both passes here use exactly those integers, so the result is bit-identical for every filter the
// host builds tables for.  A pass Pillow skips (equal sizes) gets an identity table: (p << 22) + 2^21 >> 22 == p.
// HBM-bound byte work: one thread per output pixel, neighbouring threads read neighbouring windows (the cache lines are
// shared), the float planes are written coalesced.
#include <stdint.h>

#include "common.h"

namespace scda {

constexpr int IMG_PB = 32 - 8 - 2;     // Pillow's PRECISION_BITS for 8-bit samples

__device__ __forceinline__ int img_clip8(const int v) {
    const int s = v >> IMG_PB;           // arithmetic shift, as Pillow's table lookup on (in >> PRECISION_BITS)
    return s < 0 ? 0 : (s > 255 ? 255 : s);
}

// tmp[y][x][c] = clip8( 2^21 + sum_k src[row0 + y][xmin(x) + k][c] * kk[x][k] ),   y < rows, x < OW
template <int C>
__global__ __launch_bounds__(256) void image_resize_h_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ tmp,
                                                             const int *__restrict__ bounds, const int *__restrict__ kk,
                                                             const int ksize, const int W, const int OW, const int row0,
                                                             const long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)blockDim.x * gridDim.x) {
        const int x = (int)(i % OW);
        const long long y = i / OW;
        const int xmin = bounds[2 * x], n = bounds[2 * x + 1];
        const int *k = kk + (size_t)x * ksize;
        const uint8_t *p = src + ((size_t)(row0 + y) * W + xmin) * C;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << (IMG_PB - 1);
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)p[t * C + c] * w;
        }
        uint8_t *o = tmp + (size_t)i * C;
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = (uint8_t)img_clip8(acc[c]);
    }
}

// out[c][y][xo] = ((clip8( 2^21 + sum_k tmp[ymin(y) - row0 + k][x][c] * kk[y][k] ) / 255) - mean) / std,  xo = flip ? OW-1-x : x
template <int C>
__global__ __launch_bounds__(256) void image_resize_v_normalize_kernel(const uint8_t *__restrict__ tmp, float *__restrict__ out,
                                                                       const int *__restrict__ bounds, const int *__restrict__ kk,
                                                                       const int ksize, const int OW, const int OH, const int row0,
                                                                       const float mean, const float stdv, const int flip,
                                                                       const int normalize, const long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)blockDim.x * gridDim.x) {
        const int xo = (int)(i % OW);
        const int y = (int)(i / OW);
        const int x = flip ? OW - 1 - xo : xo;
        const int ymin = bounds[2 * y] - row0, n = bounds[2 * y + 1];
        const int *k = kk + (size_t)y * ksize;
        const uint8_t *p = tmp + ((size_t)ymin * OW + x) * C;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << (IMG_PB - 1);
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)p[(size_t)t * OW * C + c] * w;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float v = (float)img_clip8(acc[c]) / 255.0f;                  // ToTensor: .float().div(255)
            if (normalize) v = (v - mean) / stdv;                         // (t - mean) / std, one IEEE operation each
            out[((size_t)c * OH + y) * OW + xo] = v;
        }
    }
}

}  // namespace scda

using namespace scda;

SCDA_API size_t scda_image_resize_tmp_bytes(int rows, int out_w, int C) {
    return rows > 0 && out_w > 0 && C > 0 ? (size_t)rows * out_w * C : 0;
}

SCDA_API int scda_image_resize_normalize_hip(const unsigned char *src, int H, int W, int C, const int *bounds_h, const int *kk_h,
                                             int ksize_h, int out_w, const int *bounds_v, const int *kk_v, int ksize_v, int out_h,
                                             int row0, int rows, unsigned char *tmp, size_t tmp_bytes, int normalize, float mean,
                                             float stdv, int flip, float *out, void *stream) {
    if (!src || !bounds_h || !kk_h || !bounds_v || !kk_v || !tmp || !out || H <= 0 || W <= 0 || out_w <= 0 || out_h <= 0 ||
        ksize_h <= 0 || ksize_v <= 0) { set_error("scda_image_resize_normalize_hip: bad arguments"); return SCDA_EINVAL; }
    // (modes with an alpha channel are not plain per-channel work in PIL: Image.resize pre-multiplies them first)
    if (C != 1 && C != 3) { set_error("scda_image_resize_normalize_hip: %d channels (supported: 1 = L, 3 = RGB)", C); return SCDA_EINVAL; }
    if (row0 < 0 || rows <= 0 || row0 + rows > H) { set_error("scda_image_resize_normalize_hip: rows [%d, %d) outside the %d-row image", row0, row0 + rows, H); return SCDA_EINVAL; }
    if (tmp_bytes < scda_image_resize_tmp_bytes(rows, out_w, C)) { set_error("scda_image_resize_normalize_hip: intermediate buffer too small"); return SCDA_EINVAL; }
    if (normalize && stdv == 0.f) { set_error("scda_image_resize_normalize_hip: std = 0"); return SCDA_EINVAL; }
    hipStream_t st = as_stream(stream);
    const long long th = (long long)rows * out_w, tv = (long long)out_h * out_w;
#define IMG_LAUNCH(C_)                                                                                                              \
    do {                                                                                                                            \
        hipLaunchKernelGGL(image_resize_h_kernel<C_>, dim3(ew_grid(th)), dim3(256), 0, st, src, tmp, bounds_h, kk_h, ksize_h, W,    \
                           out_w, row0, th);                                                                                        \
        hipLaunchKernelGGL(image_resize_v_normalize_kernel<C_>, dim3(ew_grid(tv)), dim3(256), 0, st, tmp, out, bounds_v, kk_v,      \
                           ksize_v, out_w, out_h, row0, mean, stdv, flip, normalize, tv);                                           \
    } while (0)
    if (C == 1) IMG_LAUNCH(1);
    else IMG_LAUNCH(3);
#undef IMG_LAUNCH
    return launch_status("image_resize_normalize");
}
