// Micro-benchmark: what fp32 MFMA rate can a CU sustain?  (a) MFMAs only, (b) MFMAs + the fragment reads of the GEMM
// (16 ds_read_b128 per 64 MFMAs), (c) + a barrier per 64 MFMAs, with 1 / 2 / 3 blocks of 4 waves per CU.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_peak tools/micro/mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0 MFMA only; 1 + fragment reads; 2 + barrier per step; 3 + 8 ds_write_b128 per step (staging writes);
// 4 + the 32-op norm / LeakyReLU transform on the staged registers; 5 + 10 global b128 loads per step, one step ahead, from a
// small (L2-resident) buffer; 6 same, the A half streamed from a large buffer (HBM); 7 + an epilogue every 8 steps (a K = 256
// tile): bias add, 64 dword stores per thread to a streamed output, per-column statistics (shuffle + LDS + barrier);
// 8 + the column-init gather of the next tile (64 dword gathers from an L2-resident table) after the epilogue
template <int MODE, int PRIO = 0>
__global__ __launch_bounds__(256, 2) void k_peak(float *out, int iters, const float *src, size_t big_floats)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4 * 128 * 36; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    const int kq = tid & 7, sr = tid >> 3;
    float4 ra[4], rw[4], rsc = {1.f, 1.f, 1.f, 1.f}, rsh = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < 4; ++p) { ra[p] = make_float4(0.1f * tid, 0.2f, 0.3f, 0.4f); rw[p] = ra[p]; }
    // A: 128 rows x 256 floats per tile (8 steps of 32), a different tile every 8 steps; W: one 128 x 256 tile for everybody
    const size_t a_tiles = MODE >= 6 ? ((size_t)1 << 29) / (128 * 256) : 64;      // A region: [2^20, 2^20 + 2^29) floats
    size_t a_tile = (size_t)blockIdx.x % a_tiles;
    const float *Wb = src + (size_t)sr * 256 + kq * 4;
    auto gload = [&](int step) {
        const float *Ab = src + (MODE >= 6 ? (size_t)(1 << 20) : (size_t)(1 << 16)) + a_tile * (128 * 256) + (size_t)sr * 256 + kq * 4 + (step & 7) * 32;
        for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const float4 *>(Ab + (size_t)p * 32 * 256);
        rsc = *reinterpret_cast<const float4 *>(src + kq * 4 + (step & 7) * 32);
        rsh = *reinterpret_cast<const float4 *>(src + 4096 + kq * 4 + (step & 7) * 32);
        for (int p = 0; p < 4; ++p) rw[p] = *reinterpret_cast<const float4 *>(Wb + (size_t)p * 32 * 256 + (step & 7) * 32);
        if ((step & 7) == 7) a_tile = (a_tile + gridDim.x) % a_tiles;
    };
    if (MODE >= 5) gload(0);
    f32x16 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0};
    const float *A = lds + (lane & 31) * 36 + (lane >> 5) * 4;
    const float *B = lds + 128 * 36 + (lane & 31) * 36 + (lane >> 5) * 4;
    float4 x0 = {1.f, 2.f, 3.f, 4.f}, x1 = x0, y0 = x0, y1 = x0;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        float4 sa[4], sw[4];
        if (MODE >= 5) {        // the registers of the step fetched last iteration are staged now; the next fetch is issued first
            for (int p = 0; p < 4; ++p) { sa[p] = ra[p]; sw[p] = rw[p]; }
        }
        const float4 ssc = rsc, ssh = rsh;
        if (MODE >= 5) gload(it + 1);
        if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE >= 1) {
                x0 = *reinterpret_cast<const float4 *>(A + s * 8);
                x1 = *reinterpret_cast<const float4 *>(A + 32 * 36 + s * 8);
                y0 = *reinterpret_cast<const float4 *>(B + s * 8);
                y1 = *reinterpret_cast<const float4 *>(B + 32 * 36 + s * 8);
            }
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
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (MODE >= 3) {
            if (MODE < 5) for (int p = 0; p < 4; ++p) { sa[p] = ra[p]; sw[p] = rw[p]; }
            if (MODE >= 4) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float4 v = sa[p];
                    v.x = fmaf(v.x, ssc.x, ssh.x); v.y = fmaf(v.y, ssc.y, ssh.y); v.z = fmaf(v.z, ssc.z, ssh.z); v.w = fmaf(v.w, ssc.w, ssh.w);
                    v.x = fmaxf(v.x, 0.01f * v.x); v.y = fmaxf(v.y, 0.01f * v.y); v.z = fmaxf(v.z, 0.01f * v.z); v.w = fmaxf(v.w, 0.01f * v.w);
                    sa[p] = v;
                }
            }
            float *la = lds + (2 + (buf ^ 1)) * 128 * 36 + sr * 36 + kq * 4;     // staging region (not the one the fragments are read from)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                *reinterpret_cast<float4 *>(la + p * 32 * 36 - (buf ^ 1) * 0) = sa[p];
                *reinterpret_cast<float4 *>(la + p * 32 * 36 + 2 - 2) = sw[p];
            }
        }
        if (MODE >= 2) __syncthreads();
        if (MODE >= 7 && (it & 7) == 7) {
            const int h = lane >> 5, j = lane & 31, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
            float *Y = const_cast<float *>(src) + ((size_t)1 << 29) + ((size_t)1 << 21) + (((size_t)(it >> 3) * gridDim.x + blockIdx.x) % 4096) * (128 * 512)   /* Y region: 2^28 floats, ends below 2^30 */
                       + (size_t)(wm * 64 + 4 * h) * 512 + wn * 64 + j;
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                float v;
                v = a00[r] + 0.5f; Y[(size_t)rr * 512] = v; s0 += v; q0 = fmaf(v, v, q0);
                v = a10[r] + 0.5f; Y[(size_t)(rr + 32) * 512] = v; s0 += v; q0 = fmaf(v, v, q0);
                v = a01[r] + 0.25f; Y[(size_t)rr * 512 + 32] = v; s1 += v; q1 = fmaf(v, v, q1);
                v = a11[r] + 0.25f; Y[(size_t)(rr + 32) * 512 + 32] = v; s1 += v; q1 = fmaf(v, v, q1);
            }
            s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32); s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
            float *red = lds + 4 * 128 * 36;
            if (lane < 32) { red[(0 * 2 + wm) * 128 + wn * 64 + j] = s0; red[(1 * 2 + wm) * 128 + wn * 64 + j] = q0;
                             red[(0 * 2 + wm) * 128 + wn * 64 + 32 + j] = s1; red[(1 * 2 + wm) * 128 + wn * 64 + 32 + j] = q1; }
            __syncthreads();
            if (tid < 128) Y[(size_t)127 * 512 + 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
#pragma unroll
            for (int r = 0; r < 16; ++r) { a00[r] = 0.f; a01[r] = 0.f; a10[r] = 0.f; a11[r] = 0.f; }
            if (MODE >= 8) {
                const float *Q = src + (1 << 18) + wn * 64 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned c = (unsigned)((blockIdx.x * 7 + it + r + 4 * h) % 512) * 1536u;
                    a00[r] = Q[c]; a01[r] = Q[c + 32]; a10[r] = Q[c + 1536 * 3]; a11[r] = Q[c + 1536 * 3 + 32];
                }
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += a00[i] + a01[i] + a10[i] + a11[i];
    out[blockIdx.x * 256 + tid] = r + lds[tid];
}

template <int MODE, int PRIO = 0>
static void run(const char *name, int blocks_per_cu, int ncu, float *out, const float *src, size_t big_floats, int tiles_per_block = 250)
{
    const int iters = 8 * tiles_per_block;
    // dynamic LDS sized so that exactly `blocks_per_cu` blocks fit a CU (160 KB)
    const size_t lds = blocks_per_cu == 1 ? 120 * 1024 : 76 * 1024;
    hipFuncSetAttribute((const void *)(k_peak<MODE, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = ncu * blocks_per_cu * (250 / tiles_per_block);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((k_peak<MODE, PRIO>), dim3(grid), dim3(256), lds, 0, out, iters, src, big_floats);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 10.0 * grid * 4.0 * iters * 64.0 * 4096.0;
    (void)tiles_per_block;
    printf("%-34s %d block(s)/CU: %7.1f TFLOP/s\n", name, blocks_per_cu, flops / (ms * 1e-3) / 1e12);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *out;
    hipMalloc(&out, (size_t)ncu * 2 * 250 * 256 * sizeof(float));
    printf("CUs %d\n", ncu);
    const size_t big = (size_t)1 << 30;       // 4 GB of floats
    float *src;
    hipMalloc(&src, big * sizeof(float));
    hipMemset(src, 0, big * sizeof(float));
    for (int b = 1; b <= 2; ++b) run<0>("MFMA only", b, ncu, out, src, big);
    for (int b = 1; b <= 2; ++b) run<2>("+ fragment reads + barrier", b, ncu, out, src, big);
    for (int b = 1; b <= 2; ++b) run<3>("+ 8 ds_write_b128", b, ncu, out, src, big);
    for (int b = 1; b <= 2; ++b) run<4>("+ norm / LeakyReLU transform", b, ncu, out, src, big);
    for (int b = 1; b <= 2; ++b) run<5>("+ 10 global loads (L2 resident)", b, ncu, out, src, big);
    for (int b = 1; b <= 2; ++b) run<6>("+ A streamed from HBM", b, ncu, out, src, big);
    for (int b = 1; b <= 2; ++b) run<7>("+ epilogue every 8 steps (K 256)", b, ncu, out, src, big);
    for (int b = 1; b <= 2; ++b) run<8>("+ column-init gather", b, ncu, out, src, big);
    run<8, 1>("same, MFMA phase at s_setprio 3", 2, ncu, out, src, big);
    run<8>("same, ONE tile per block (no prio)", 2, ncu, out, src, big, 1);
    run<8, 1>("same, one tile per block, prio 3", 2, ncu, out, src, big, 1);
    run<7>("epilogue, no gather, one tile/block", 2, ncu, out, src, big, 1);
    return 0;
}
