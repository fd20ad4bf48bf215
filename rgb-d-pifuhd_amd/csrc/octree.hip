// Grid evaluation bookkeeping on the device: dense eval_grid and the coarse-to-fine eval_grid_octree of
// the reference (mesh_util.py:98-187), in the order-free per-voxel formulation of SURVEY.md App. G, which
// is bit-exact with the reference's sequential C-order block fills.  State is float64 (sdf) + uint8
// (notprocessed), exactly the reference's numpy state, resident in HBM for the whole reconstruction.
#include "pifu_internal.h"
#include "scan.h"

namespace {

constexpr int SEL_TILE = 2048;   // lattice points per block of the ordered compaction (8 passes x 256 threads)

__global__ void k_oct_begin(double *sdf, uint8_t *notproc, int R)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)R * R * R;
    if (i >= n) return;
    const int k = (int)(i % R), j = (int)((i / R) % R), ii = (int)(i / ((int64_t)R * R));
    sdf[i] = 0.0;
    notproc[i] = (ii < R - 1 && j < R - 1 && k < R - 1) ? 1 : 0;    // mesh_util.py:134-135
}

// lattice point l (C-order over the nl^3 lattice of stride reso) -> flat grid index
__device__ __forceinline__ int64_t lattice_to_flat(int64_t l, int nl, int reso, int R)
{
    const int c = (int)(l % nl), b = (int)((l / nl) % nl), a = (int)(l / ((int64_t)nl * nl));
    return ((int64_t)a * reso * R + (int64_t)b * reso) * R + (int64_t)c * reso;
}

__global__ void k_oct_count(const uint8_t *notproc, int R, int reso, int nl, int64_t nlat, uint64_t *block_counts)
{
    __shared__ uint32_t sh[4];
    const int64_t base = (int64_t)blockIdx.x * SEL_TILE;
    uint32_t cnt = 0;
#pragma unroll
    for (int p = 0; p < SEL_TILE / 256; ++p) {
        const int64_t l = base + p * 256 + threadIdx.x;
        if (l < nlat && notproc[lattice_to_flat(l, nl, reso, R)]) ++cnt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (uint64_t)sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void k_oct_scatter_index(const uint8_t *notproc, int R, int reso, int nl, int64_t nlat,
                                    const uint64_t *block_offsets, int64_t *index_out, int64_t cap)
{
    __shared__ uint32_t sh[4];
    const int64_t base = (int64_t)blockIdx.x * SEL_TILE;
    uint64_t run = block_offsets[blockIdx.x];
    for (int p = 0; p < SEL_TILE / 256; ++p) {
        const int64_t l = base + p * 256 + threadIdx.x;
        int64_t flat = -1;
        uint32_t flag = 0;
        if (l < nlat) {
            flat = lattice_to_flat(l, nl, reso, R);
            flag = notproc[flat] ? 1u : 0u;
        }
        uint32_t tot;
        const uint32_t pre = block_exclusive_scan_256(flag, sh, &tot);
        if (flag && (int64_t)(run + pre) < cap) index_out[run + pre] = flat;      // never past the caller's buffer
        run += tot;
    }
}

__global__ void k_oct_scatter_values(const int64_t *index, const float *vals, int64_t n, double *sdf, uint8_t *notproc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t f = index[i];
    sdf[f] = (double)vals[i];        // sdf[test_mask] = batch_eval(...)  (float32 -> float64 store)
    notproc[f] = 0;
}

// per cell of the stride-`reso` lattice: 8-corner min / max, mean and the skip decision (mesh_util.py:154-179)
__global__ void k_oct_cells(const double *sdf, const uint8_t *notproc, int R, int reso, int nc, double threshold,
                            double *cell_mean, uint8_t *cell_skip)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)nc * nc * nc;
    if (i >= n) return;
    const int cz = (int)(i % nc), cy = (int)((i / nc) % nc), cx = (int)(i / ((int64_t)nc * nc));
    const int64_t RR = (int64_t)R * R;
    const int64_t o = (int64_t)cx * reso * RR + (int64_t)cy * reso * R + (int64_t)cz * reso;
    double vmin = sdf[o], vmax = vmin;
#pragma unroll
    for (int c = 1; c < 8; ++c) {
        const double v = sdf[o + (int64_t)((c >> 2) & 1) * reso * RR + (int64_t)((c >> 1) & 1) * reso * R + (int64_t)(c & 1) * reso];
        vmin = v < vmin ? v : vmin;
        vmax = v > vmax ? v : vmax;
    }
    const int h = reso / 2;
    const bool np = notproc[o + (int64_t)h * RR + (int64_t)h * R + h] != 0;
    cell_mean[i] = 0.5 * (vmin + vmax);
    cell_skip[i] = ((vmax - vmin) < threshold && np) ? 1 : 0;
}

// per voxel: the lexicographically largest covering skip cell wins (= last writer of the reference's
// C-ordered python loop, mesh_util.py:181-184); blocks are inclusive [c*reso, c*reso + reso].
// A thread takes FILL_RUN consecutive voxels of a k row: they share the candidate cells along i and j, and - below the
// finest levels, where most rows lie in regions no skip cell of this level touches - one look at the row's candidate
// (i, j) cell rows decides for the whole run.  (One voxel per thread: 0.93 - 1.0 ms per level at 512^3, three levels per call.)
constexpr int FILL_RUN = 8;

__global__ void k_oct_fill(double *sdf, uint8_t *notproc, int R, int reso, int nc, const double *cell_mean,
                           const uint8_t *cell_skip)
{
    const int runs = (R + FILL_RUN - 1) / FILL_RUN;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)R * R * runs) return;
    const int k0 = (int)(t % runs) * FILL_RUN;
    const int j = (int)((t / runs) % R), i = (int)(t / ((int64_t)runs * R));
    // candidate cells along i and j, largest first
    int ci[2], cj[2], ni = 0, nj = 0;
    {
        const int q = i / reso;
        if (q <= nc - 1) ci[ni++] = q;
        if (i % reso == 0 && q - 1 >= 0 && q - 1 <= nc - 1) ci[ni++] = q - 1;
    }
    {
        const int q = j / reso;
        if (q <= nc - 1) cj[nj++] = q;
        if (j % reso == 0 && q - 1 >= 0 && q - 1 <= nc - 1) cj[nj++] = q - 1;
    }
    if (ni == 0 || nj == 0) return;
    int k1 = min(k0 + FILL_RUN, R);
    const int qlo = max(k0 / reso - 1, 0), qhi = min((k1 - 1) / reso, nc - 1);      // cells along k any voxel of the run can belong to
    if (qhi < qlo) return;
    bool any = false;
    for (int a = 0; a < ni && !any; ++a)
        for (int b = 0; b < nj && !any; ++b) {
            const uint8_t *row = cell_skip + ((int64_t)ci[a] * nc + cj[b]) * nc;
            for (int q = qlo; q <= qhi; ++q) any = any || row[q] != 0;
        }
    if (!any) return;
    const int64_t base = ((int64_t)i * R + j) * R;
    // most runs of a filled region lie strictly inside one cell row (i and j off the cell faces): after the run's first voxel,
    // which may sit on a face along k, the rest share the single cell k / reso - one look-up, then plain stores
    if (ni == 1 && nj == 1 && reso >= FILL_RUN && k1 - k0 == FILL_RUN && (k0 + 1) / reso == (k1 - 1) / reso && (k0 + 1) % reso != 0 &&
        (k0 + 1) / reso <= nc - 1) {
        const int64_t cell = ((int64_t)ci[0] * nc + cj[0]) * nc + (k0 + 1) / reso;
        if (cell_skip[cell]) {
            const double m = cell_mean[cell];
#pragma unroll
            for (int k = 1; k < FILL_RUN; ++k) { sdf[base + k0 + k] = m; notproc[base + k0 + k] = 0; }
        }
        k1 = k0 + 1;          // only the first voxel is left for the general path
    }
    for (int k = k0; k < k1; ++k) {
        int ck[2], nk = 0;
        const int q = k / reso;
        if (q <= nc - 1) ck[nk++] = q;
        if (k % reso == 0 && q - 1 >= 0 && q - 1 <= nc - 1) ck[nk++] = q - 1;
        bool done = false;
        for (int a = 0; a < ni && !done; ++a)
            for (int b = 0; b < nj && !done; ++b)
                for (int c = 0; c < nk && !done; ++c) {
                    const int64_t cell = ((int64_t)ci[a] * nc + cj[b]) * nc + ck[c];
                    if (cell_skip[cell]) {
                        sdf[base + k] = cell_mean[cell];
                        notproc[base + k] = 0;
                        done = true;
                    }
                }
    }
}

__global__ void k_f32_to_f64(const float *src, double *dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// One level's test points.  Both passes run the counting kernel + scan on the stream into the context's own `sel_counts`
// buffer (nothing else writes there), so the scatter always works from offsets computed from the mask as it is NOW; the
// scatter itself is bounded by `cap`.  What the counting pass keeps for the fill pass of the same level (the normal call
// pattern: count, allocate, fill) is only the TOTAL it already brought to the host - one host synchronisation per level
// instead of two.  The remembered total is dropped by every entry point that changes a mask (begin, commit).
int octree_select_impl(pifu_ctx *ctx, int R, int reso, const uint8_t *notproc, int64_t *index_out, int64_t cap,
                       int64_t *count_host, bool count_only, hipStream_t st)
{
    const int nl = (R + reso - 1) / reso;
    const int64_t nlat = (int64_t)nl * nl * nl;
    const int64_t nblocks = (nlat + SEL_TILE - 1) / SEL_TILE;
    pifu_ctx::SelState &sel = ctx->sel;
    const bool known = !count_only && sel.valid && sel.R == R && sel.reso == reso && sel.notproc == notproc;
    sel.valid = false;
    int rc = pifu_ensure(ctx->sel_counts, (size_t)(nblocks + 2 + scan_tiles(nblocks) + 1) * sizeof(uint64_t));
    if (rc) return rc;
    uint64_t *counts = (uint64_t *)ctx->sel_counts.p;
    uint64_t *total = counts + nblocks;
    hipLaunchKernelGGL(k_oct_count, dim3((unsigned)nblocks), dim3(256), 0, st, notproc, R, reso, nl, nlat, counts);
    scan_u64(counts, nblocks, total, counts + nblocks + 2, st);
    int64_t n = sel.n;
    if (!known) {
        PIFU_HIP(hipMemcpyAsync(ctx->pinned, total, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        PIFU_HIP(hipStreamSynchronize(st));
        n = (int64_t)(*(uint64_t *)ctx->pinned);
    }
    *count_host = n;
    if (count_only) {
        sel.n = n; sel.R = R; sel.reso = reso; sel.notproc = notproc;
        sel.valid = true;
        return PIFU_OK;
    }
    if (n == 0) return PIFU_OK;
    PIFU_REQUIRE(index_out && cap >= n, PIFU_ERR_INVALID, "pifu_octree_select: index buffer holds %lld entries, %lld needed",
                 (long long)cap, (long long)n);
    hipLaunchKernelGGL(k_oct_scatter_index, dim3((unsigned)nblocks), dim3(256), 0, st, notproc, R, reso, nl, nlat,
                       (const uint64_t *)counts, index_out, cap);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

}  // namespace

extern "C" int pifu_octree_begin(pifu_ctx *ctx, int R, double *sdf, uint8_t *notprocessed, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && sdf && notprocessed && R >= 2, PIFU_ERR_INVALID, "pifu_octree_begin: bad argument");
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    ctx->sel.valid = false;
    const int64_t n = (int64_t)R * R * R;
    hipLaunchKernelGGL(k_oct_begin, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, sdf, notprocessed, R);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

extern "C" int pifu_octree_select(pifu_ctx *ctx, int R, int reso, const uint8_t *notprocessed, int64_t *index_out,
                                  int64_t cap, int64_t *count_host, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && notprocessed && count_host && R >= 2 && reso >= 1, PIFU_ERR_INVALID, "pifu_octree_select: bad argument");
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    return octree_select_impl(ctx, R, reso, notprocessed, index_out, cap, count_host, index_out == nullptr, (hipStream_t)stream);
}

extern "C" int pifu_octree_commit(pifu_ctx *ctx, int R, int reso, double threshold, const int64_t *index,
                                  const float *values, int64_t count, double *sdf, uint8_t *notprocessed, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && sdf && notprocessed && R >= 2 && reso >= 1 && count >= 0, PIFU_ERR_INVALID, "pifu_octree_commit: bad argument");
    PIFU_REQUIRE(count == 0 || (index && values), PIFU_ERR_INVALID, "pifu_octree_commit: null index/values");
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    ctx->sel.valid = false;
    hipStream_t st = (hipStream_t)stream;
    if (count > 0)
        hipLaunchKernelGGL(k_oct_scatter_values, dim3(blocks_for(count, 256)), dim3(256), 0, st, index, values, count, sdf, notprocessed);
    if (reso > 1) {
        const int nl = (R + reso - 1) / reso;
        const int nc = nl - 1;
        if (nc >= 1) {
            const int64_t ncell = (int64_t)nc * nc * nc;
            const size_t mean_bytes = (size_t)round_up(ncell * 8, 256);
            int rc = pifu_ensure(ctx->oct_arena, mean_bytes + (size_t)ncell + 256);
            if (rc) return rc;
            double *cell_mean = (double *)ctx->oct_arena.p;
            uint8_t *cell_skip = (uint8_t *)ctx->oct_arena.p + mean_bytes;
            hipLaunchKernelGGL(k_oct_cells, dim3(blocks_for(ncell, 256)), dim3(256), 0, st, (const double *)sdf,
                               (const uint8_t *)notprocessed, R, reso, nc, threshold, cell_mean, cell_skip);
            const int64_t n = (int64_t)R * R * ((R + FILL_RUN - 1) / FILL_RUN);      // one thread per run of FILL_RUN voxels
            hipLaunchKernelGGL(k_oct_fill, dim3(blocks_for(n, 256)), dim3(256), 0, st, sdf, notprocessed, R, reso, nc,
                               (const double *)cell_mean, (const uint8_t *)cell_skip);
        }
    }
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

extern "C" int pifu_eval_grid(pifu_ctx *ctx, int R, const double *calib_inv, const float *calib, int64_t chunk,
                              int init_resolution, double threshold, int flags, double *sdf, uint8_t *notprocessed,
                              int64_t *n_evaluated, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && calib_inv && calib && sdf && R >= 2 && chunk > 0, PIFU_ERR_INVALID, "pifu_eval_grid: bad argument");
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int32_t res[3] = {R, R, R};
    const int64_t n = (int64_t)R * R * R;
    int64_t evaluated = 0;
    static thread_local DevBuf unused;
    (void)unused;
    if (init_resolution <= 0) {                                     // eval_grid (mesh_util.py:116-120)
        DevBuf &tmp = ctx->mc_arena;                                // scratch for the fp32 predictions
        int rc = pifu_ensure(tmp, (size_t)n * sizeof(float));
        if (rc) return rc;
        rc = pifu_eval_lattice(ctx, res, calib_inv, calib, nullptr, 0, n, chunk, flags, (float *)tmp.p, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const float *)tmp.p, sdf, n);
        evaluated = n;
    } else {                                                        // eval_grid_octree (mesh_util.py:124-187)
        PIFU_REQUIRE(notprocessed, PIFU_ERR_INVALID, "pifu_eval_grid: octree mode needs the notprocessed buffer");
        int rc = pifu_octree_begin(ctx, R, sdf, notprocessed, stream);
        if (rc) return rc;
        int reso = R / init_resolution;
        while (reso > 0) {
            int64_t cnt = 0;
            rc = octree_select_impl(ctx, R, reso, notprocessed, nullptr, 0, &cnt, true, st);
            if (rc) return rc;
            if (cnt > 0) {
                DevBuf &tmp = ctx->mc_arena;
                const size_t idx_bytes = (size_t)round_up(cnt * 8, 256);
                rc = pifu_ensure(tmp, idx_bytes + (size_t)cnt * sizeof(float));
                if (rc) return rc;
                int64_t *idx = (int64_t *)tmp.p;
                float *vals = (float *)((char *)tmp.p + idx_bytes);
                rc = octree_select_impl(ctx, R, reso, notprocessed, idx, cnt, &cnt, false, st);
                if (rc) return rc;
                rc = pifu_eval_lattice(ctx, res, calib_inv, calib, idx, 0, cnt, chunk, flags, vals, stream);
                if (rc) return rc;
                rc = pifu_octree_commit(ctx, R, reso, threshold, idx, vals, cnt, sdf, notprocessed, stream);
                if (rc) return rc;
                evaluated += cnt;
            } else {
                rc = pifu_octree_commit(ctx, R, reso, threshold, nullptr, nullptr, 0, sdf, notprocessed, stream);
                if (rc) return rc;
            }
            if (reso <= 1) break;
            reso /= 2;
        }
    }
    if (n_evaluated) *n_evaluated = evaluated;
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}
