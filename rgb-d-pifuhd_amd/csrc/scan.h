// Small deterministic scan helpers shared by the octree compaction and the marching-cubes numbering.
// wave = 64 lanes; blocks are 256 threads unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Exclusive prefix of `v` over the 256 threads of a block in thread order; *total = block sum.
// `sh` must hold at least 4 uint32.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *sh, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(inc, o);
        if (lane >= o) inc += n;
    }
    __syncthreads();            // protect sh against a previous use
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t s = sh[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// Single-block exclusive scan of `n` uint64 elements, in place; total written to *total.
// Launch with 1024 threads.  Deterministic; n up to a few million is fine (one pass per thread segment).
static __global__ void k_scan_u64(uint64_t *data, int64_t n, uint64_t *total)
{
    __shared__ uint64_t sh[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = t * per, hi = (lo + per < n) ? lo + per : n;
    uint64_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += data[i];
    sh[t] = s;
    __syncthreads();
    // Hillis-Steele over 1024 partial sums
    for (int o = 1; o < 1024; o <<= 1) {
        const uint64_t add = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    uint64_t run = sh[t] - s;   // exclusive prefix of this thread's segment
    for (int64_t i = lo; i < hi; ++i) {
        const uint64_t v = data[i];
        data[i] = run;
        run += v;
    }
    if (t == 1023) *total = sh[1023];
}
