// Supervised-forward loss of the reference (`CustomBCELoss.forward`, net_util.py:45-66) as a device reduction.
// Forward only: the HIP path has no autograd.  Deterministic: per-block partial sums in a fixed order (double), then a
// single-block fixed-order sum.
#include "pifu_internal.h"

namespace {

constexpr int LOSS_BLOCKS = 1024;

struct BceArgs {
    const float *pred, *gt;      // [B][n]
    const float *gamma;          // [B] or null (fixed_gamma)
    const float *w;              // null, [B] (w_full = 0) or [B][n] (w_full = 1)
    int64_t n;                   // elements per batch entry
    int64_t total;               // B * n
    float fixed_gamma;
    int brock, w_full;
    double *partial;             // [LOSS_BLOCKS]
    float *out;
};

// one term of the weighted cross entropy, in fp32 like the reference's tensor expression:
//   p = clamp(pred, 1e-5, 1 - 1e-5);  t = gt (brock: 3 gt - 1)
//   -(g t log p + (1 - g)(1 - t) log(1 - p)) * w
__device__ __forceinline__ float bce_term(float pred, float gt, float g, float w, int brock)
{
    const float p = fminf(fmaxf(pred, 1e-5f), 1.0f - 1e-5f);
    const float t = brock ? 3.0f * gt - 1.0f : gt;
    const float pos = g * t * logf(p);
    const float neg = (1.0f - g) * (1.0f - t) * logf(1.0f - p);
    return -(pos + neg) * w;
}

__global__ void k_bce_partial(const BceArgs a)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / a.n;
        const float g = a.gamma ? a.gamma[b] : a.fixed_gamma;
        const float w = a.w ? (a.w_full ? a.w[i] : a.w[b]) : 1.0f;
        acc += (double)bce_term(a.pred[i], a.gt[i], g, w, a.brock);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[blockIdx.x] = red[0];
}

__global__ void k_bce_final(const double *partial, int nblocks, int64_t total, float *out)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(red[0] / (double)total);
}

}  // namespace

extern "C" int pifu_bce_loss(pifu_ctx *ctx, const float *pred, const float *gt, int batch, int64_t n, const float *gamma,
                             float fixed_gamma, const float *w, int w_per_element, int brock, float *out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && pred && gt && out, PIFU_ERR_INVALID, "pifu_bce_loss: null argument");
    PIFU_REQUIRE(batch > 0 && n > 0, PIFU_ERR_INVALID, "pifu_bce_loss: empty input (the reference's mean() would be nan)");
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    int rc = pifu_ensure(ctx->misc, LOSS_BLOCKS * sizeof(double));
    if (rc) return rc;
    BceArgs a{};
    a.pred = pred; a.gt = gt; a.gamma = gamma; a.w = w; a.n = n; a.total = (int64_t)batch * n;
    a.fixed_gamma = fixed_gamma; a.brock = brock != 0; a.w_full = w_per_element != 0;
    a.partial = reinterpret_cast<double *>(ctx->misc.p);      // stream-ordered scratch
    a.out = out;
    const int nblocks = (int)((a.total + 1023) / 1024 < LOSS_BLOCKS ? (a.total + 1023) / 1024 : LOSS_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bce_partial, dim3(nblocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_bce_final, dim3(1), dim3(256), 0, st, (const double *)a.partial, nblocks, a.total, out);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}
