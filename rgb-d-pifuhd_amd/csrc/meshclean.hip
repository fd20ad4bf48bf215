// Mesh clean-up of the image-coloured variant (`gen_mesh_imgColor_plus`, reconstruction.py:127-229,325-344) on gfx950:
//   pifu_face_components - connected components of a triangle list over SHARED EDGES (what the reference gets from
//                          trimesh's `mesh.split()`), as a lock-free union-find over faces driven by an edge hash table;
//   pifu_esti_color      - `esti_color`: every back-side vertex (z < 0) takes the mean colour of its <= 10 + 10 nearest
//                          silhouette vertices (0 <= z < 0.001) to the left and to the right, nearest in |dy|.
// Integer / index work: results are exact (bit-for-bit against the reference function for esti_color).
#include "pifu_internal.h"

#include <algorithm>
#include <vector>

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;

__device__ __forceinline__ unsigned long long edge_key(int a, int b)
{
    const unsigned lo = (unsigned)(a < b ? a : b), hi = (unsigned)(a < b ? b : a);
    return ((unsigned long long)lo << 32) | hi;
}

__device__ __forceinline__ unsigned edge_hash(unsigned long long k, unsigned mask)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k & mask;
}

__global__ void k_cc_init(int *parent, int64_t nfaces, unsigned long long *keys, int *cnt, int64_t cap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nfaces) parent[i] = (int)i;
    if (i < cap) { keys[i] = EMPTY_KEY; cnt[i] = 0; }
}

// every face registers its three edges: slot = open-addressing hash of the (lo, hi) vertex pair; the first two faces of an
// edge are remembered, the count keeps going (a count other than 2 = boundary or non-manifold edge)
__global__ void k_cc_edges(const int32_t *faces, int64_t nfaces, unsigned long long *keys, int *cnt, int *f0, int *f1, unsigned mask)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    const int v[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const unsigned long long key = edge_key(v[e], v[(e + 1) % 3]);
        unsigned h = edge_hash(key, mask);
        while (true) {
            const unsigned long long prev = atomicCAS(&keys[h], EMPTY_KEY, key);
            if (prev == EMPTY_KEY || prev == key) break;
            h = (h + 1) & mask;
        }
        const int c = atomicAdd(&cnt[h], 1);
        if (c == 0) f0[h] = (int)f;
        else if (c == 1) f1[h] = (int)f;
    }
}

__device__ __forceinline__ int cc_find(const int *parent, int x)
{
    int p = parent[x];
    while (p != x) { x = p; p = parent[x]; }
    return x;
}

// faces on the two sides of a manifold edge belong together: hook the larger root under the smaller (atomicMin keeps the
// forest acyclic and the final root = smallest face index of the component, independent of scheduling)
__global__ void k_cc_union(const int *cnt, const int *f0, const int *f1, int64_t cap, int *parent)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap || cnt[s] != 2) return;
    int a = f0[s], b = f1[s];
    while (true) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) break;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[a], b);
        if (old == a) break;
        a = old;
    }
}

__global__ void k_cc_label(const int32_t *faces, int64_t nfaces, const unsigned long long *keys, const int *cnt, unsigned mask,
                           const int *parent, int32_t *label, uint8_t *open)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    label[f] = cc_find(parent, (int)f);
    const int v[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
    uint8_t o = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const unsigned long long key = edge_key(v[e], v[(e + 1) % 3]);
        unsigned h = edge_hash(key, mask);
        while (keys[h] != key) h = (h + 1) & mask;
        const int c = cnt[h];
        if (c == 1) o |= 1;              // an edge only this face uses: the component is open whatever else it holds
        else if (c != 2) o |= 2;         // an edge three or more faces share: no adjacency through it; whether it pairs up
    }                                    // INSIDE a component is the caller's to decide (it knows the labels)
    open[f] = o;
}

// ---- esti_color ---------------------------------------------------------------------------------------------------------
// Silhouette vertices sorted by (y, x, index) with the distinct y values as "levels": a back vertex walks the levels outward
// from its own y, nearest |dy| first (both directions when they tie), and inside a level takes the candidates next to its x.
struct EstiArgs {
    const float *xyz;          // [3][n]
    int64_t n;
    const int *back;           // [nback] vertex indices with z < 0
    int nback;
    const float *sx;           // [ns] silhouette x, sorted by (y, x, index)
    const int *sidx;           // [ns] their vertex indices
    const float *level_y;      // [nl] distinct y ascending
    const int *level_start;    // [nl + 1]
    int nl;
    uint16_t *color;           // [n][4] in / out
    int *failed;               // set when a back vertex has no candidate on either side
};

struct Cand { float x; int idx; };

// up to `need` best LEFT candidates of one level (x_s < x; larger x first, equal x by ascending index), appended to buf
__device__ int take_left(const EstiArgs &a, int lv, float x, int need, Cand *buf, int nbuf, int cap)
{
    const int s0 = a.level_start[lv], s1 = a.level_start[lv + 1];
    int lo = s0, hi = s1;                  // first entry with x_s >= x
    while (lo < hi) { const int m = (lo + hi) >> 1; if (a.sx[m] < x) lo = m + 1; else hi = m; }
    int p = lo - 1, taken = 0;
    while (p >= s0 && nbuf < cap) {
        if (taken >= need && a.sx[p] != buf[nbuf - 1].x) break;       // complete a run of equal x before stopping
        buf[nbuf].x = a.sx[p]; buf[nbuf].idx = a.sidx[p];
        ++nbuf; ++taken; --p;
    }
    return nbuf;
}

__device__ int take_right(const EstiArgs &a, int lv, float x, int need, Cand *buf, int nbuf, int cap)
{
    const int s0 = a.level_start[lv], s1 = a.level_start[lv + 1];
    int lo = s0, hi = s1;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (a.sx[m] < x) lo = m + 1; else hi = m; }
    int p = lo, taken = 0;
    while (p < s1 && nbuf < cap) {
        if (taken >= need && a.sx[p] != buf[nbuf - 1].x) break;
        buf[nbuf].x = a.sx[p]; buf[nbuf].idx = a.sidx[p];
        ++nbuf; ++taken; ++p;
    }
    return nbuf;
}

// insertion sort of a handful of candidates: left wants (x descending, index ascending), right (x ascending, index ascending)
__device__ void sort_cands(Cand *b, int n, bool left)
{
    for (int i = 1; i < n; ++i) {
        const Cand c = b[i];
        int j = i - 1;
        while (j >= 0) {
            const bool after = left ? (b[j].x < c.x || (b[j].x == c.x && b[j].idx > c.idx))
                                    : (b[j].x > c.x || (b[j].x == c.x && b[j].idx > c.idx));
            if (!after) break;
            b[j + 1] = b[j];
            --j;
        }
        b[j + 1] = c;
    }
}

__global__ void k_esti_color(const EstiArgs a)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.nback) return;
    constexpr int K = 10, CAP = 48;
    const int v = a.back[t];
    const float x = a.xyz[v], y = a.xyz[a.n + v];
    int lo = 0, hi = a.nl;                 // first level with level_y >= y
    while (lo < hi) { const int m = (lo + hi) >> 1; if (a.level_y[m] < y) lo = m + 1; else hi = m; }
    int up = lo, dn = lo - 1;
    int left[K], right[K], nleft = 0, nright = 0;
    Cand buf[CAP];
    while ((nleft < K || nright < K) && (up < a.nl || dn >= 0)) {
        // |dy| in double, as the reference compares it (float32 coordinates promoted to Python floats)
        const double du = up < a.nl ? (double)a.level_y[up] - (double)y : 1e300;
        const double dd = dn >= 0 ? (double)y - (double)a.level_y[dn] : 1e300;
        const bool use_up = du <= dd, use_dn = dd <= du;
        if (nleft < K) {
            int nb = 0;
            if (use_dn) nb = take_left(a, dn, x, K - nleft, buf, nb, CAP);
            if (use_up) nb = take_left(a, up, x, K - nleft, buf, nb, CAP);
            sort_cands(buf, nb, true);
            for (int i = 0; i < nb && nleft < K; ++i) left[nleft++] = buf[i].idx;
        }
        if (nright < K) {
            int nb = 0;
            if (use_dn) nb = take_right(a, dn, x, K - nright, buf, nb, CAP);
            if (use_up) nb = take_right(a, up, x, K - nright, buf, nb, CAP);
            sort_cands(buf, nb, false);
            for (int i = 0; i < nb && nright < K; ++i) right[nright++] = buf[i].idx;
        }
        if (use_up) ++up;
        if (use_dn) --dn;
    }
    if (nleft + nright == 0) { *a.failed = 1; return; }
    unsigned sum[4] = {0, 0, 0, 0};
    for (int i = 0; i < nleft; ++i)
        for (int c = 0; c < 4; ++c) sum[c] += a.color[(size_t)left[i] * 4 + c];
    for (int i = 0; i < nright; ++i)
        for (int c = 0; c < 4; ++c) sum[c] += a.color[(size_t)right[i] * 4 + c];
    // uint16 sums / count in float64, stored back into a uint16 array: truncation
    for (int c = 0; c < 4; ++c) a.color[(size_t)v * 4 + c] = (uint16_t)((double)(sum[c] & 0xffffu) / (double)(nleft + nright));
}

}  // namespace

extern "C" int pifu_face_components(pifu_ctx *ctx, const int32_t *faces, int64_t nfaces, int32_t *label, uint8_t *open_edge,
                                    pifu_stream stream)
{
    PIFU_REQUIRE(ctx && faces && label && open_edge, PIFU_ERR_INVALID, "pifu_face_components: null argument");
    PIFU_REQUIRE(nfaces > 0 && nfaces < ((int64_t)1 << 29), PIFU_ERR_INVALID, "pifu_face_components: 0 < nfaces < 2^29");
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    int64_t cap = 1024;
    while (cap < 6 * nfaces) cap <<= 1;                      // 3 edges per face, load factor <= 0.5
    const size_t bytes = (size_t)cap * (8 + 4 + 4 + 4) + (size_t)nfaces * 4 + 1024;
    int rc = pifu_ensure(ctx->oct_arena, bytes);
    if (rc) return rc;
    char *base = (char *)ctx->oct_arena.p;
    unsigned long long *keys = (unsigned long long *)base;
    int *cnt = (int *)(base + (size_t)cap * 8);
    int *f0 = cnt + cap, *f1 = f0 + cap, *parent = f1 + cap;
    const unsigned mask = (unsigned)(cap - 1);
    const int64_t nmax = cap > nfaces ? cap : nfaces;
    hipLaunchKernelGGL(k_cc_init, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, st, parent, nfaces, keys, cnt, cap);
    hipLaunchKernelGGL(k_cc_edges, dim3((unsigned)((nfaces + 255) / 256)), dim3(256), 0, st, faces, nfaces, keys, cnt, f0, f1, mask);
    hipLaunchKernelGGL(k_cc_union, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, (const int *)cnt, (const int *)f0, (const int *)f1,
                       cap, parent);
    hipLaunchKernelGGL(k_cc_label, dim3((unsigned)((nfaces + 255) / 256)), dim3(256), 0, st, faces, nfaces,
                       (const unsigned long long *)keys, (const int *)cnt, mask, (const int *)parent, label, open_edge);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

extern "C" int pifu_esti_color(pifu_ctx *ctx, const float *xyz, int64_t n, uint16_t *color, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && xyz && color, PIFU_ERR_INVALID, "pifu_esti_color: null argument");
    PIFU_REQUIRE(n > 0 && n < ((int64_t)1 << 31), PIFU_ERR_INVALID, "pifu_esti_color: 0 < n < 2^31");
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    // the selection lists (back vertices, silhouette vertices sorted by (y, x, index)) are built on the host from one
    // copy of the projected coordinates: a few MB, once per mesh
    std::vector<float> h((size_t)3 * n);
    PIFU_HIP(hipMemcpyAsync(h.data(), xyz, h.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    PIFU_HIP(hipStreamSynchronize(st));
    const float *hx = h.data(), *hy = hx + n, *hz = hy + n;
    std::vector<int> back, sil;
    for (int64_t i = 0; i < n; ++i) {
        if (hz[i] < 0.f) back.push_back((int)i);                      // reconstruction.py:186
        else if (hz[i] < 0.001f) sil.push_back((int)i);               // :187 (0 <= z < 0.001, compared in float32)
    }
    if (back.empty()) return PIFU_OK;
    PIFU_REQUIRE(!sil.empty(), PIFU_ERR_STATE, "esti_color: no silhouette vertex (0 <= z < 0.001) to take colours from");
    std::sort(sil.begin(), sil.end(), [&](int a, int b) {
        if (hy[a] != hy[b]) return hy[a] < hy[b];
        if (hx[a] != hx[b]) return hx[a] < hx[b];
        return a < b;
    });
    std::vector<float> sx(sil.size()), ly;
    std::vector<int> ls;
    for (size_t i = 0; i < sil.size(); ++i) {
        sx[i] = hx[sil[i]];
        if (i == 0 || hy[sil[i]] != hy[sil[i - 1]]) { ly.push_back(hy[sil[i]]); ls.push_back((int)i); }
    }
    ls.push_back((int)sil.size());
    const size_t b_back = round_up(back.size() * 4, 256), b_sx = round_up(sx.size() * 4, 256), b_si = round_up(sil.size() * 4, 256),
                 b_ly = round_up(ly.size() * 4, 256), b_ls = round_up(ls.size() * 4, 256);
    int rc = pifu_ensure(ctx->oct_arena, b_back + b_sx + b_si + b_ly + b_ls + 256);
    if (rc) return rc;
    char *base = (char *)ctx->oct_arena.p;
    EstiArgs a{};
    a.xyz = xyz; a.n = n; a.color = color;
    a.back = (const int *)base; a.nback = (int)back.size();
    a.sx = (const float *)(base + b_back);
    a.sidx = (const int *)(base + b_back + b_sx);
    a.level_y = (const float *)(base + b_back + b_sx + b_si);
    a.level_start = (const int *)(base + b_back + b_sx + b_si + b_ly);
    a.nl = (int)ly.size();
    a.failed = (int *)(base + b_back + b_sx + b_si + b_ly + b_ls);
    PIFU_HIP(hipMemcpyAsync((void *)a.back, back.data(), back.size() * 4, hipMemcpyHostToDevice, st));
    PIFU_HIP(hipMemcpyAsync((void *)a.sx, sx.data(), sx.size() * 4, hipMemcpyHostToDevice, st));
    PIFU_HIP(hipMemcpyAsync((void *)a.sidx, sil.data(), sil.size() * 4, hipMemcpyHostToDevice, st));
    PIFU_HIP(hipMemcpyAsync((void *)a.level_y, ly.data(), ly.size() * 4, hipMemcpyHostToDevice, st));
    PIFU_HIP(hipMemcpyAsync((void *)a.level_start, ls.data(), ls.size() * 4, hipMemcpyHostToDevice, st));
    PIFU_HIP(hipMemsetAsync(a.failed, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_esti_color, dim3((unsigned)((back.size() + 63) / 64)), dim3(64), 0, st, a);
    int failed = 0;
    PIFU_HIP(hipMemcpyAsync(&failed, a.failed, sizeof(int), hipMemcpyDeviceToHost, st));
    PIFU_HIP(hipStreamSynchronize(st));          // the host vectors above must outlive the copies
    PIFU_REQUIRE(!failed, PIFU_ERR_STATE, "esti_color: a back-side vertex has no silhouette vertex on either side");
    return PIFU_OK;
}
