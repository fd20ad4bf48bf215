// Image-side helpers of the reconstruction entry points, resident on the device:
//   * pifu_index            - `index(feat, uv)` of the reference as a standalone op (BasePIFuNet.py:11-23): the vertex
//                             colour lookup of gen_mesh_imgColor (reconstruction.py:112-115) and the sampler-in-isolation
//                             measurement;
//   * pifu_preprocess_image - the image half of readDataset.get_item (readData.py:15-29, 83-104): rect crop with a zero
//                             border, cv2.resize (INTER_LINEAR, 8-bit fixed-point path), BGR -> RGB, ToTensor, Normalize.
// Compiled with -ffp-contract=off: the resize coordinates follow OpenCV's double -> float rounding sequence exactly.
#include "pifu_internal.h"

namespace {

// bilinear grid_sample, align_corners = True, zero padding; one thread per point, channels looped (NCHW planes)
__global__ void k_index(const float *feat, int C, int H, int W, const float *uv, int64_t n, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = uv[i], v = uv[n + i];
    const float ix = ((u + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((v + 1.f) / 2.f) * (float)(H - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix;
    const float wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
    // NaN / huge coordinates sample nothing (every tap out of bounds), like grid_sample's bounds test
    const bool finite = fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
    const int x0 = finite ? (int)fx0 : -2, y0 = finite ? (int)fy0 : -2;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
    const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    const size_t plane = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        const float *p = feat + c * plane;
        float acc = 0.f;
        if (vx0 && vy0) acc += p[(size_t)y0 * W + x0] * nw;
        if (vx1 && vy0) acc += p[(size_t)y0 * W + x0 + 1] * ne;
        if (vx0 && vy1) acc += p[(size_t)(y0 + 1) * W + x0] * sw;
        if (vx1 && vy1) acc += p[(size_t)(y0 + 1) * W + x0 + 1] * se;
        out[(size_t)c * n + i] = acc;
    }
}

struct ResizeArgs {
    const uint8_t *src;   // [H][W][3] BGR
    int H, W;
    int rx, ry, rw, rh;   // crop rectangle in source coordinates (may leave the image: zeros outside)
    int size;             // output is size x size
    double scale_x, scale_y;
    int area2;            // exact 2x decimation: OpenCV switches INTER_LINEAR to the INTER_AREA fast path
    float *out;           // [3][size][size] RGB, (v / 255 - 0.5) / 0.5
};

__device__ __forceinline__ int crop_px(const ResizeArgs &a, int cx, int cy, int ch)
{
    const int x = a.rx + cx, y = a.ry + cy;
    if (x < 0 || y < 0 || x >= a.W || y >= a.H) return 0;            // cv2.copyMakeBorder(..., BORDER_CONSTANT, 0)
    return a.src[((size_t)y * a.W + x) * 3 + ch];
}

// one linear-resize coordinate: source index, clamped, and the two 11-bit fixed-point weights (resize.cpp,
// INTER_RESIZE_COEF_BITS = 11); `clamp_weights` is the x-axis rule (fx = 0 at the borders), rows are clamped instead
__device__ __forceinline__ void lin_coord(int d, double scale, int n, bool clamp_weights, int *s0, int *s1, int *w0, int *w1)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_weights) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n - 1) { f = 0.f; s = n - 1; }
        *s0 = s;
        *s1 = s + 1 < n ? s + 1 : s;
    } else {
        *s0 = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
        *s1 = s + 1 < 0 ? 0 : (s + 1 > n - 1 ? n - 1 : s + 1);
    }
    *w0 = __float2int_rn((1.f - f) * 2048.f);
    *w1 = __float2int_rn(f * 2048.f);
}

__global__ void k_crop_resize_normalize(const ResizeArgs a)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= a.size) return;
    int val[3];
    if (a.area2) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            val[c] = (crop_px(a, 2 * dx, 2 * dy, c) + crop_px(a, 2 * dx + 1, 2 * dy, c) + crop_px(a, 2 * dx, 2 * dy + 1, c) +
                      crop_px(a, 2 * dx + 1, 2 * dy + 1, c) + 2) >> 2;
    } else {
        int x0, x1, a0, a1, y0, y1, b0, b1;
        lin_coord(dx, a.scale_x, a.rw, true, &x0, &x1, &a0, &a1);
        lin_coord(dy, a.scale_y, a.rh, false, &y0, &y1, &b0, &b1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int S0 = crop_px(a, x0, y0, c) * a0 + crop_px(a, x1, y0, c) * a1;
            const int S1 = crop_px(a, x0, y1, c) * a0 + crop_px(a, x1, y1, c) * a1;
            val[c] = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        }
    }
    const size_t plane = (size_t)a.size * a.size, o = (size_t)dy * a.size + dx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v = val[2 - c] < 0 ? 0 : (val[2 - c] > 255 ? 255 : val[2 - c]);      // BGR -> RGB
        const float t = (float)v / 255.f;                                             // ToTensor
        a.out[c * plane + o] = (t - 0.5f) / 0.5f;                                     // Normalize(0.5, 0.5)
    }
}

}  // namespace

extern "C" int pifu_index(pifu_ctx *ctx, const float *feat_nchw, int C, int H, int W, const float *uv, int64_t n,
                          float *out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && feat_nchw && uv && out, PIFU_ERR_INVALID, "pifu_index: null argument");
    PIFU_REQUIRE(C > 0 && H > 0 && W > 0 && n >= 0, PIFU_ERR_INVALID, "pifu_index: bad shape C=%d H=%d W=%d n=%lld", C, H, W, (long long)n);
    PIFU_USE(ctx, stream);
    if (n == 0) return PIFU_OK;
    PIFU_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_index, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat_nchw, C, H, W, uv, n, out);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

extern "C" int pifu_preprocess_image(pifu_ctx *ctx, const uint8_t *bgr, int H, int W, const int32_t rect[4], int size,
                                     float *out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && bgr && rect && out, PIFU_ERR_INVALID, "pifu_preprocess_image: null argument");
    PIFU_REQUIRE(H > 0 && W > 0 && size > 0 && rect[2] > 0 && rect[3] > 0, PIFU_ERR_INVALID,
                 "pifu_preprocess_image: bad shape H=%d W=%d size=%d rect=(%d,%d,%d,%d)", H, W, size, rect[0], rect[1], rect[2], rect[3]);
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    ResizeArgs a;
    a.src = bgr; a.H = H; a.W = W;
    a.rx = rect[0]; a.ry = rect[1]; a.rw = rect[2]; a.rh = rect[3];
    a.size = size;
    // cv::resize: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
    a.scale_x = 1.0 / ((double)size / (double)a.rw);
    a.scale_y = 1.0 / ((double)size / (double)a.rh);
    a.area2 = (a.rw == 2 * size && a.rh == 2 * size) ? 1 : 0;
    a.out = out;
    hipLaunchKernelGGL(k_crop_resize_normalize, dim3((size + 255) / 256, size), dim3(256), 0, (hipStream_t)stream, a);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}
