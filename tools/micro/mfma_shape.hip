// Micro-benchmark: does the fp32 MFMA SHAPE change the clock the chip holds?  (MI355X_MICROARCH.md, DVFS give-back item 7:
// for bf16 the 16x16 shape sustains ~1.13x the FLOP/s of the 32x32 shape at equal cycles.)  Same 64 x 64 output tile per
// wave, operands re-read from LDS by ds_read_b128 (random data), one barrier per 32-wide K step, 4 waves per block,
// `BPC` blocks per CU.  Reports wall TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_shape tools/micro/mfma_shape.hip ; run: ./mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LD = 36;

template <int SHAPE>     // 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_16x16x4_f32
__global__ __launch_bounds__(256) void k_shape(float *out, int iters, const float *src, unsigned long long *clk)
{
    __shared__ float lds[2 * 128 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 128 * LD; i += 256) lds[i] = src[(i * 7 + blockIdx.x * 13) & 0xFFFFF];
    __syncthreads();
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x16 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0};
        const float *A = lds + (wm * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
        const float *B = lds + 128 * LD + (wn * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 x0 = *reinterpret_cast<const float4 *>(A + s * 8);
                const float4 x1 = *reinterpret_cast<const float4 *>(A + 32 * LD + s * 8);
                const float4 y0 = *reinterpret_cast<const float4 *>(B + s * 8);
                const float4 y1 = *reinterpret_cast<const float4 *>(B + 32 * LD + s * 8);
                const float av0[4] = {x0.x, x0.y, x0.z, x0.w}, av1[4] = {x1.x, x1.y, x1.z, x1.w};
                const float bv0[4] = {y0.x, y0.y, y0.z, y0.w}, bv1[4] = {y1.x, y1.y, y1.z, y1.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv0[t], a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv1[t], a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv0[t], a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv1[t], a11, 0, 0, 0);
                }
            }
            __syncthreads();
        }
        for (int i = 0; i < 16; ++i) sum += a00[i] + a01[i] + a10[i] + a11[i];
    } else {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // lane (r = lane & 15, q = lane >> 4) holds k = 4 q .. 4 q + 3 of a 16-wide k chunk: MFMA step t uses element t
        const float *A = lds + (wm * 64 + (lane & 15)) * LD + (lane >> 4) * 4;
        const float *B = lds + 128 * LD + (wn * 64 + (lane & 15)) * LD + (lane >> 4) * 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {           // two 16-wide k chunks per 32-wide step
                float4 x[4], y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x[i] = *reinterpret_cast<const float4 *>(A + i * 16 * LD + s * 16);
                    y[i] = *reinterpret_cast<const float4 *>(B + i * 16 * LD + s * 16);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float av = t == 0 ? x[i].x : t == 1 ? x[i].y : t == 2 ? x[i].z : x[i].w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float bv = t == 0 ? y[j].x : t == 1 ? y[j].y : t == 2 ? y[j].z : y[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
    out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <int SHAPE>
static void run(const char *name, int bpc, float *out, const float *src, unsigned long long *clk, int cus)
{
    const int blocks = cus * bpc, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // settle the clock: ~2 s of back-to-back launches, then time the last ones
    float ms = 0.f;
    int reps = 0;
    for (int phase = 0; phase < 2; ++phase) {
        hipEventRecord(e0);
        const int n = phase == 0 ? 12 : 6;
        for (int r = 0; r < n; ++r) hipLaunchKernelGGL(k_shape<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, src, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        reps = n;
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int b = 0; b < blocks; ++b) mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double flop = (double)blocks * 4 /*waves*/ * iters * (64.0 * 64 * 32 * 2) * reps;
    printf("%-22s %d blocks/CU: %7.1f TFLOP/s wall, in-kernel clock %6.0f MHz (median), %.3f FLOP/cycle/CU\n", name, bpc,
           flop / (ms * 1e-3) / 1e12, mhz[mhz.size() / 2], flop / (ms * 1e-3) / (mhz[mhz.size() / 2] * 1e6) / cus);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out, *src;
    unsigned long long *clk;
    hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    hipMalloc(&clk, (size_t)cus * 4 * 2 * 8);
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMalloc(&src, h.size() * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<0>("v_mfma_f32_32x32x2", bpc, out, src, clk, cus);
        run<1>("v_mfma_f32_16x16x4", bpc, out, src, clk, cus);
    }
    return 0;
}
