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

// ---- exclusive scan of `n` uint64 elements over many blocks (deterministic: integer sums) -------------------------------------
// tiles of SCAN_TILE elements: (1) per-tile sums, (2) single-block scan of the tile sums (k_scan_u64), (3) per-tile exclusive
// scan + tile offset.  `tile_sums` is scratch of scan_tiles(n) + 1 uint64.  Replaces one 1024-thread block walking the whole
// array with a stride of n / 1024 elements between neighbouring threads (320 us for the 524 288 block counts of a 512^3
// marching cubes, 2.5 ms at 1024^3).
constexpr int SCAN_TILE = 4096;      // 256 threads x 16 elements
static inline int64_t scan_tiles(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

static __global__ __launch_bounds__(256) void k_scan_tile_sums(const uint64_t *data, int64_t n, uint64_t *tile_sums)
{
    __shared__ uint64_t sh[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_TILE / 256; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;          // coalesced
        if (i < n) s += data[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

static __global__ __launch_bounds__(256) void k_scan_tile_apply(uint64_t *data, int64_t n, const uint64_t *tile_offsets)
{
    __shared__ uint64_t sh[256];
    constexpr int PER = SCAN_TILE / 256;
    const int64_t lo = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * PER;      // 128 contiguous bytes per thread
    uint64_t v[PER], s = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = lo + k < n ? data[lo + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint64_t add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint64_t run = tile_offsets[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (lo + k < n) data[lo + k] = run;
        run += v[k];
    }
}

// in-place exclusive scan of data[0, n) on `st`; *total (device) = sum.  tile_sums: device scratch, scan_tiles(n) + 1 entries.
static inline void scan_u64(uint64_t *data, int64_t n, uint64_t *total, uint64_t *tile_sums, hipStream_t st)
{
    const int64_t nt = scan_tiles(n);
    if (nt <= 1) {      // one tile: the single-block kernel is as good
        hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, data, n, total);
        return;
    }
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)nt), dim3(256), 0, st, (const uint64_t *)data, n, tile_sums);
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, st, tile_sums, nt, total);
    hipLaunchKernelGGL(k_scan_tile_apply, dim3((unsigned)nt), dim3(256), 0, st, data, n, (const uint64_t *)tile_sums);
}
