// Lewiner marching cubes on the device, bit-exact with scikit-image 0.18.3's serial implementation
// (`measure.marching_cubes_lewiner(sdf, thresh)` as called at mesh_util.py:84: step_size 1,
// gradient_direction 'descent', allow_degenerate True, use_classic False).  Algorithm: SURVEY.md App. B;
// the serial vertex / triangle numbering is reproduced with prefix sums:
//   * a cell OWNS an edge vertex iff it is the first cell in scan order (axis 0 outer, axis 2 inner) that
//     touches the edge, i.e. the smallest of the <= 4 cells sharing it; the centre vertex (id 12) is owned
//     by its cell; inside a cell vertices are numbered in first-reference order of its triangle row;
//   * pass 1 (k_mc_count): per-cell (owned vertices, triangles) -> per-block totals; a device scan gives
//     each block's base; pass 2 (k_mc_verts): positions + a per-grid-edge index volume; pass 3
//     (k_mc_faces): triangle indices through that volume.  Classification is recomputed per pass instead
//     of stored (active cells are a small fraction of the grid; the volume is read three times, mostly
//     from cache).  In-block ordering uses a wavefront shuffle scan (64 lanes) + a 4-entry LDS scan.
// Arithmetic follows scikit-image exactly: cube values v - level in double; its "FLT_EPSILON" is the
// DOUBLE epsilon; t = -b / (2a + eps) and t = c_p / (c_p - c_q + eps) in the interior test; vertex
// positions from strengths w = 1 / (eps + |v - level|).
#include "pifu_internal.h"
#include "scan.h"

#define MC_LUT_QUAL __device__ __constant__ const
#include "mc_luts.h"

namespace {

constexpr double MC_EPS = 2.220446049250313e-16;

__device__ __constant__ const signed char EDGE_A[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
__device__ __constant__ const signed char EDGE_B[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
// corner -> (dx,dy,dz)
__device__ __constant__ const signed char COFF[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// edge -> axis (0 = x, 1 = y, 2 = z), base-point offset (dx,dy,dz) of its lower endpoint, and the mask of
// cell coordinates (bit0 x, bit1 y, bit2 z) that must be 0 for this cell to be the first one touching it
__device__ __constant__ const signed char EDGE_AXIS[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
__device__ __constant__ const signed char EDGE_BASE[12][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 0}, {0, 0, 1}, {1, 0, 1},
                                                             {0, 1, 1}, {0, 0, 1}, {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}};
__device__ __constant__ const signed char EDGE_NEEDZERO[12] = {2 | 4, 4, 4, 1 | 4, 2, 0, 0, 1, 1 | 2, 2, 0, 1};
// interior-test table e:(p,q,(b0,b1),(c0,c1),(d0,d1))
__device__ __constant__ const signed char TI[12][8] = {{0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7},
                                                      {3, 0, 2, 1, 6, 5, 7, 4}, {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2},
                                                      {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0}, {0, 4, 3, 7, 2, 6, 1, 5},
                                                      {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};

__device__ bool test_face(const double *c, int f)
{
    int A, B, C, D;
    switch (f < 0 ? -f : f) {
    case 1: A = 0; B = 4; C = 5; D = 1; break;
    case 2: A = 1; B = 5; C = 6; D = 2; break;
    case 3: A = 2; B = 6; C = 7; D = 3; break;
    case 4: A = 3; B = 7; C = 4; D = 0; break;
    case 5: A = 0; B = 3; C = 2; D = 1; break;
    default: A = 4; B = 7; C = 6; D = 5; break;
    }
    const double d = __dsub_rn(__dmul_rn(c[A], c[C]), __dmul_rn(c[B], c[D]));
    if (d > -MC_EPS && d < MC_EPS) return f >= 0;
    return __dmul_rn(__dmul_rn((double)f, c[A]), d) >= 0;
}

__device__ bool test_interior(const double *c, int mc_case, int cfg, int subcfg, int s)
{
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0;
    if (mc_case == 4 || mc_case == 10) {
        const double d40 = c[4] - c[0], d62 = c[6] - c[2], d73 = c[7] - c[3], d51 = c[5] - c[1];
        const double a = __dsub_rn(__dmul_rn(d40, d62), __dmul_rn(d73, d51));
        double b = __dadd_rn(__dmul_rn(c[2], d40), __dmul_rn(c[0], d62));
        b = __dsub_rn(b, __dmul_rn(c[1], d73));
        b = __dsub_rn(b, __dmul_rn(c[3], d51));
        t = -b / __dadd_rn(__dadd_rn(a, a), MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = __dadd_rn(c[0], __dmul_rn(d40, t));
        Bt = __dadd_rn(c[3], __dmul_rn(d73, t));
        Ct = __dadd_rn(c[2], __dmul_rn(d62, t));
        Dt = __dadd_rn(c[1], __dmul_rn(d51, t));
    } else {
        int edge = -1;
        if (mc_case == 6) edge = MC_TEST6[cfg][2];
        else if (mc_case == 7) edge = MC_TEST7[cfg][4];
        else if (mc_case == 12) edge = MC_TEST12[cfg][3];
        else if (mc_case == 13) edge = MC_TILING13_5_1[cfg][subcfg][0];
        if (edge >= 0 && edge < 12) {
            const signed char *e = TI[edge];
            t = c[e[0]] / __dadd_rn(c[e[0]] - c[e[1]], MC_EPS);
            At = 0;
            Bt = __dadd_rn(c[e[2]], __dmul_rn(c[e[3]] - c[e[2]], t));
            Ct = __dadd_rn(c[e[4]], __dmul_rn(c[e[5]] - c[e[4]], t));
            Dt = __dadd_rn(c[e[6]], __dmul_rn(c[e[7]] - c[e[6]], t));
        }
    }
    int test = 0;
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    const double det = __dsub_rn(__dmul_rn(At, Ct), __dmul_rn(Bt, Dt));
    switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    case 7: case 11: case 13: case 14: case 15: return s < 0;
    case 5: return det < MC_EPS ? s > 0 : false;       // scikit-image falls through to False here
    case 10: return det >= MC_EPS ? s > 0 : false;
    }
    return false;
}

// triangle row of a cell; returns the number of triangles
__device__ int choose_tiling(const double *c, int index, const signed char **row)
{
    const int mc_case = MC_CASES[index][0], cfg = MC_CASES[index][1];
    int sc, s13, k;
    switch (mc_case) {
    case 1: *row = MC_TILING1[cfg]; return 1;
    case 2: *row = MC_TILING2[cfg]; return 2;
    case 3:
        if (test_face(c, MC_TEST3[cfg])) { *row = MC_TILING3_2[cfg]; return 4; }
        *row = MC_TILING3_1[cfg]; return 2;
    case 4:
        if (test_interior(c, 4, cfg, 0, MC_TEST4[cfg])) { *row = MC_TILING4_1[cfg]; return 2; }
        *row = MC_TILING4_2[cfg]; return 6;
    case 5: *row = MC_TILING5[cfg]; return 3;
    case 6:
        if (test_face(c, MC_TEST6[cfg][0])) { *row = MC_TILING6_2[cfg]; return 5; }
        if (test_interior(c, 6, cfg, 0, MC_TEST6[cfg][1])) { *row = MC_TILING6_1_1[cfg]; return 3; }
        *row = MC_TILING6_1_2[cfg]; return 9;
    case 7:
        sc = 0;
        for (k = 0; k < 3; ++k) if (test_face(c, MC_TEST7[cfg][k])) sc += 1 << k;
        switch (sc) {
        case 0: *row = MC_TILING7_1[cfg]; return 3;
        case 1: *row = MC_TILING7_2[cfg][0]; return 5;
        case 2: *row = MC_TILING7_2[cfg][1]; return 5;
        case 3: *row = MC_TILING7_3[cfg][0]; return 9;
        case 4: *row = MC_TILING7_2[cfg][2]; return 5;
        case 5: *row = MC_TILING7_3[cfg][1]; return 9;
        case 6: *row = MC_TILING7_3[cfg][2]; return 9;
        default:
            if (test_interior(c, 7, cfg, 0, MC_TEST7[cfg][3])) { *row = MC_TILING7_4_2[cfg]; return 9; }
            *row = MC_TILING7_4_1[cfg]; return 5;
        }
    case 8: *row = MC_TILING8[cfg]; return 2;
    case 9: *row = MC_TILING9[cfg]; return 4;
    case 10:
        if (test_face(c, MC_TEST10[cfg][0])) {
            if (test_face(c, MC_TEST10[cfg][1])) { *row = MC_TILING10_1_1_[cfg]; return 4; }
            *row = MC_TILING10_2[cfg]; return 8;
        }
        if (test_face(c, MC_TEST10[cfg][1])) { *row = MC_TILING10_2_[cfg]; return 8; }
        if (test_interior(c, 10, cfg, 0, MC_TEST10[cfg][2])) { *row = MC_TILING10_1_1[cfg]; return 4; }
        *row = MC_TILING10_1_2[cfg]; return 8;
    case 11: *row = MC_TILING11[cfg]; return 4;
    case 12:
        if (test_face(c, MC_TEST12[cfg][0])) {
            if (test_face(c, MC_TEST12[cfg][1])) { *row = MC_TILING12_1_1_[cfg]; return 4; }
            *row = MC_TILING12_2[cfg]; return 8;
        }
        if (test_face(c, MC_TEST12[cfg][1])) { *row = MC_TILING12_2_[cfg]; return 8; }
        if (test_interior(c, 12, cfg, 0, MC_TEST12[cfg][2])) { *row = MC_TILING12_1_1[cfg]; return 4; }
        *row = MC_TILING12_1_2[cfg]; return 8;
    case 13:
        sc = 0;
        for (k = 0; k < 6; ++k) if (test_face(c, MC_TEST13[cfg][k])) sc += 1 << k;
        s13 = MC_SUBCONFIG13[sc];
        if (s13 == 0) { *row = MC_TILING13_1[cfg]; return 4; }
        if (s13 <= 6) { *row = MC_TILING13_2[cfg][s13 - 1]; return 6; }
        if (s13 <= 18) { *row = MC_TILING13_3[cfg][s13 - 7]; return 10; }
        if (s13 <= 22) { *row = MC_TILING13_4[cfg][s13 - 19]; return 12; }
        if (s13 <= 26) {
            k = s13 - 23;
            if (test_interior(c, 13, cfg, k, MC_TEST13[cfg][6])) { *row = MC_TILING13_5_1[cfg][k]; return 6; }
            *row = MC_TILING13_5_2[cfg][k]; return 10;
        }
        if (s13 <= 38) { *row = MC_TILING13_3_[cfg][s13 - 27]; return 10; }
        if (s13 <= 44) { *row = MC_TILING13_2_[cfg][s13 - 39]; return 6; }
        *row = MC_TILING13_1_[cfg]; return 4;
    case 14: *row = MC_TILING14[cfg]; return 4;
    default: *row = nullptr; return 0;
    }
}

struct Vol {
    const void *p;
    int dtype;                 // 0 float32, 1 float64 (cast to float32 first, as scikit-image does)
    int n0, n1, n2;            // array axes; internal z,y,x = axes 0,1,2
    int c0, c1, c2;            // cells per axis
    int64_t ncells;
    double level;
    // 8 x 8 x 8-cell bricks whose 9 x 9 x 9 nodes are not all on one side of the level (k_mc_bricks); null = classify every cell
    const uint8_t *brick;
    int b0, b1, b2;
    // 1 for every group of 256 consecutive cells of the serial scan that touches such a brick (k_mc_groups); the three
    // classification passes give a block MC_GPB groups and skip the others without touching their cells
    const uint8_t *gflag;
    // slab mode (pifu_mc_slab_*): the volume holds node planes [zoff, zoff + n0) of a taller one.  Vertex ownership along
    // axis 0 and the axis-0 coordinate of the vertices refer to the tall volume; 0 for a whole volume
    int zoff;
};

__device__ __forceinline__ double vol_at(const Vol &v, int z, int y, int x)
{
    const size_t i = ((size_t)z * v.n1 + y) * v.n2 + x;
    return v.dtype == 0 ? (double)((const float *)v.p)[i] : (double)(float)((const double *)v.p)[i];
}

struct CellInfo {
    const signed char *row;
    int nt;
    int nown;                  // vertices this cell creates
    int x, y, z;
    double c[8];
};

// classify one cell; fills c[], row, nt and counts owned vertices.  Returns false for an empty cell.
__device__ bool classify(const Vol &v, int64_t cell, CellInfo &ci)
{
    ci.x = (int)(cell % v.c2);
    ci.y = (int)((cell / v.c2) % v.c1);
    ci.z = (int)(cell / ((int64_t)v.c2 * v.c1));
    ci.nt = 0;
    ci.nown = 0;
    ci.row = nullptr;
    // a cell of a brick whose nodes all lie on one side of the level has index 0 or 255: nothing to load
    if (v.brick && !v.brick[((size_t)(ci.z >> 3) * v.b1 + (ci.y >> 3)) * v.b2 + (ci.x >> 3)]) return false;
    int index = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ci.c[i] = vol_at(v, ci.z + COFF[i][2], ci.y + COFF[i][1], ci.x + COFF[i][0]) - v.level;
        if (ci.c[i] > 0) index |= 1 << i;
    }
    ci.nt = 0;
    ci.nown = 0;
    ci.row = nullptr;
    if (index == 0 || index == 255) return false;
    ci.nt = choose_tiling(ci.c, index, &ci.row);
    const int zero_mask = (ci.x == 0 ? 1 : 0) | (ci.y == 0 ? 2 : 0) | (ci.z + v.zoff == 0 ? 4 : 0);
    unsigned seen = 0;
    for (int k = 0; k < 3 * ci.nt; ++k) {
        const int e = ci.row[k];
        if (seen & (1u << e)) continue;
        seen |= 1u << e;
        if (e == 12 || (EDGE_NEEDZERO[e] & ~zero_mask) == 0) ++ci.nown;
    }
    return true;
}

__global__ void k_mc_minmax(const Vol v, float *out /* [2*gridDim] */)
{
    __shared__ float smin[4], smax[4];
    const int64_t n = (int64_t)v.n0 * v.n1 * v.n2;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float f = v.dtype == 0 ? ((const float *)v.p)[i] : (float)((const double *)v.p)[i];
        lo = fminf(lo, f);
        hi = fmaxf(hi, f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        out[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

__global__ void k_mc_minmax_final(const float *part, int n, float *out)
{
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 64) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (threadIdx.x == 0) { out[0] = lo; out[1] = hi; }
}

// pass 0: one wave per 8 x 8 x 8-cell brick: does the sign of (value - level) - evaluated exactly as classify() does -
// change anywhere among its 9 x 9 x 9 nodes?  ~96 % of the cells of a body at 512^3 sit in bricks where it does not; the
// three classification passes then skip them without touching the volume.
__global__ void k_mc_bricks(const Vol v, uint8_t *brick)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t nb = (int64_t)v.b0 * v.b1 * v.b2;
    if (b >= nb) return;
    const int bx = (int)(b % v.b2), by = (int)((b / v.b2) % v.b1), bz = (int)(b / ((int64_t)v.b2 * v.b1));
    bool pos = false, neg = false;
    for (int i = lane; i < 729; i += 64) {
        const int dz = i / 81, dy = (i / 9) % 9, dx = i % 9;
        const int z = min(bz * 8 + dz, v.n0 - 1), y = min(by * 8 + dy, v.n1 - 1), x = min(bx * 8 + dx, v.n2 - 1);
        const bool p = (vol_at(v, z, y, x) - v.level) > 0;
        pos |= p;
        neg |= !p;
    }
    const bool any_pos = __builtin_amdgcn_ballot_w64(pos) != 0ull, any_neg = __builtin_amdgcn_ballot_w64(neg) != 0ull;
    if (lane == 0) brick[b] = (any_pos && any_neg) ? 1 : 0;
}

constexpr int MC_GPB = 16;      // groups of 256 cells per block of the classification passes

// which groups of 256 consecutive cells can hold a surface cell at all: one thread per group walks its cells brick by brick
__global__ void k_mc_groups(const Vol v, int64_t ngroups, uint8_t *gflag)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int64_t lo = g * 256, hi = lo + 256 < v.ncells ? lo + 256 : v.ncells;
    uint8_t any = 0;
    for (int64_t c = lo; c < hi && !any;) {
        const int x = (int)(c % v.c2), y = (int)((c / v.c2) % v.c1), z = (int)(c / ((int64_t)v.c2 * v.c1));
        any = v.brick[((size_t)(z >> 3) * v.b1 + (y >> 3)) * v.b2 + (x >> 3)];
        int step = 8 - (x & 7);                                  // to the next brick along x ...
        if (step > v.c2 - x) step = v.c2 - x;                    // ... or the next row
        c += step;
    }
    gflag[g] = any;
}

// pass 1: group g covers cells [256 g, 256 g + 256) of the serial scan; per-group (vertices, triangles) created
__global__ void k_mc_count(const Vol v, int64_t ngroups, uint64_t *blk_verts, uint64_t *blk_tris)
{
    __shared__ uint32_t sh[4];
    for (int gi = 0; gi < MC_GPB; ++gi) {
        const int64_t group = (int64_t)blockIdx.x * MC_GPB + gi;
        if (group >= ngroups) break;
        if (!v.gflag[group]) {                                   // block-uniform
            if (threadIdx.x == 0) { blk_verts[group] = 0; blk_tris[group] = 0; }
            continue;
        }
        const int64_t cell = group * 256 + threadIdx.x;
        uint32_t packed = 0;
        if (cell < v.ncells) {
            CellInfo ci;
            if (classify(v, cell, ci)) packed = (uint32_t)ci.nown | ((uint32_t)ci.nt << 16);
        }
        uint32_t tot;
        block_exclusive_scan_256(packed, sh, &tot);
        if (threadIdx.x == 0) {
            blk_verts[group] = tot & 0xffffu;
            blk_tris[group] = tot >> 16;
        }
    }
}

__device__ __forceinline__ size_t edge_slot(const Vol &v, int axis, int z, int y, int x)
{
    return (size_t)axis * v.n0 * v.n1 * v.n2 + ((size_t)z * v.n1 + y) * v.n2 + x;
}

// pass 2: vertex positions + per-grid-edge vertex index volume
__global__ void k_mc_verts(const Vol v, int64_t ngroups, const uint64_t *blk_verts, int64_t vert_base, float *verts,
                           int32_t *edge_index)
{
    __shared__ uint32_t sh[4];
  for (int gi = 0; gi < MC_GPB; ++gi) {
    const int64_t group = (int64_t)blockIdx.x * MC_GPB + gi;
    if (group >= ngroups) break;
    if (!v.gflag[group]) continue;                               // block-uniform
    const int64_t cell = group * 256 + threadIdx.x;
    CellInfo ci;
    bool active = false;
    if (cell < v.ncells) active = classify(v, cell, ci);
    uint32_t tot;
    const uint32_t pre = block_exclusive_scan_256(active ? (uint32_t)ci.nown : 0u, sh, &tot);
    if (!active || ci.nown == 0) continue;
    // `verts` is this volume's own array (local numbering); the index volume holds numbers of the whole mesh
    int64_t vid = (int64_t)blk_verts[group] + pre;
    const int zero_mask = (ci.x == 0 ? 1 : 0) | (ci.y == 0 ? 2 : 0) | (ci.z + v.zoff == 0 ? 4 : 0);
    unsigned seen = 0;
    for (int k = 0; k < 3 * ci.nt; ++k) {
        const int e = ci.row[k];
        if (seen & (1u << e)) continue;
        seen |= 1u << e;
        if (e == 12) {
            double sw = 0, sx = 0, sy = 0, sz = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double w = 1.0 / __dadd_rn(MC_EPS, fabs(ci.c[i]));
                sx = __dadd_rn(sx, __dmul_rn(w, (double)COFF[i][0]));
                sy = __dadd_rn(sy, __dmul_rn(w, (double)COFF[i][1]));
                sz = __dadd_rn(sz, __dmul_rn(w, (double)COFF[i][2]));
                sw = __dadd_rn(sw, w);
            }
            verts[3 * vid + 0] = (float)__dadd_rn((double)(ci.z + v.zoff), sz / sw);
            verts[3 * vid + 1] = (float)__dadd_rn((double)ci.y, sy / sw);
            verts[3 * vid + 2] = (float)__dadd_rn((double)ci.x, sx / sw);
            ++vid;
        } else if ((EDGE_NEEDZERO[e] & ~zero_mask) == 0) {
            const int axis = EDGE_AXIS[e];
            const int bx = ci.x + EDGE_BASE[e][0], by = ci.y + EDGE_BASE[e][1], bz = ci.z + EDGE_BASE[e][2];
            // lower endpoint = the corner at the base point, upper = the other one
            const int ca = EDGE_A[e], cb = EDGE_B[e];
            const bool a_is_lo = COFF[ca][axis] == 0;
            const double clo = a_is_lo ? ci.c[ca] : ci.c[cb], chi = a_is_lo ? ci.c[cb] : ci.c[ca];
            const double wlo = 1.0 / __dadd_rn(MC_EPS, fabs(clo)), whi = 1.0 / __dadd_rn(MC_EPS, fabs(chi));
            const double t = whi / __dadd_rn(wlo, whi);
            const int gz = bz + v.zoff;
            verts[3 * vid + 0] = (float)(axis == 2 ? __dadd_rn((double)gz, t) : (double)gz);
            verts[3 * vid + 1] = (float)(axis == 1 ? __dadd_rn((double)by, t) : (double)by);
            verts[3 * vid + 2] = (float)(axis == 0 ? __dadd_rn((double)bx, t) : (double)bx);
            edge_index[edge_slot(v, axis, bz, by, bx)] = (int32_t)(vert_base + vid);
            ++vid;
        }
    }
  }
}

// pass 3: triangle indices ('descent' => the three indices of every triangle are reversed)
__global__ void k_mc_faces(const Vol v, int64_t ngroups, const uint64_t *blk_verts, const uint64_t *blk_tris, int64_t vert_base,
                           const int32_t *edge_index, int32_t *faces)
{
    __shared__ uint32_t sh[4];
  for (int gi = 0; gi < MC_GPB; ++gi) {
    const int64_t group = (int64_t)blockIdx.x * MC_GPB + gi;
    if (group >= ngroups) break;
    if (!v.gflag[group]) continue;                               // block-uniform
    const int64_t cell = group * 256 + threadIdx.x;
    CellInfo ci;
    bool active = false;
    if (cell < v.ncells) active = classify(v, cell, ci);
    uint32_t tot;
    const uint32_t pre = block_exclusive_scan_256(active ? ((uint32_t)ci.nown | ((uint32_t)ci.nt << 16)) : 0u, sh, &tot);
    if (!active) continue;
    const int64_t vbase = vert_base + (int64_t)blk_verts[group] + (pre & 0xffffu);
    int64_t f = (int64_t)blk_tris[group] + (pre >> 16);
    // index of the centre vertex = vbase + number of owned vertices first referenced before it
    const int zero_mask = (ci.x == 0 ? 1 : 0) | (ci.y == 0 ? 2 : 0) | (ci.z + v.zoff == 0 ? 4 : 0);
    int32_t centre = -1;
    {
        unsigned seen = 0;
        int rank = 0;
        for (int k = 0; k < 3 * ci.nt; ++k) {
            const int e = ci.row[k];
            if (seen & (1u << e)) continue;
            seen |= 1u << e;
            if (e == 12) { centre = (int32_t)(vbase + rank); break; }
            if ((EDGE_NEEDZERO[e] & ~zero_mask) == 0) ++rank;
        }
    }
    for (int t = 0; t < ci.nt; ++t) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = ci.row[3 * t + k];
            int32_t vid;
            if (e == 12) {
                vid = centre;
            } else {
                const int axis = EDGE_AXIS[e];
                vid = edge_index[edge_slot(v, axis, ci.z + EDGE_BASE[e][2], ci.y + EDGE_BASE[e][1], ci.x + EDGE_BASE[e][0])];
            }
            faces[3 * (f + t) + (2 - k)] = vid;
        }
    }
  }
}

struct McArena {
    uint64_t *blk_verts, *blk_tris, *totals, *tile_sums;
    float *mm;
    uint8_t *brick, *gflag;
    int32_t *edge_index;
    int64_t nblocks, nbricks;      // nblocks = groups of 256 cells
    size_t bytes;
};

constexpr int MM_BLOCKS = 1024;

// arena: [blk_verts nblocks+1][blk_tris nblocks+1][totals 2][minmax partials][scan scratch][brick map][group flags][edge_index 3*N int32]
McArena carve(const Vol &v, char *base)
{
    McArena a;
    a.nblocks = (v.ncells + 255) / 256;
    a.nbricks = (int64_t)v.b0 * v.b1 * v.b2;
    const size_t scan_bytes = (size_t)round_up((a.nblocks + 1) * 8, 256);
    const size_t mm_bytes = (size_t)round_up((2 * MM_BLOCKS + 2) * 4, 256);
    const size_t tile_bytes = (size_t)round_up((scan_tiles(a.nblocks) + 1) * 8, 256);
    const size_t brick_bytes = (size_t)round_up(a.nbricks, 256);
    const size_t edge_bytes = (size_t)3 * v.n0 * v.n1 * v.n2 * sizeof(int32_t);
    a.blk_verts = (uint64_t *)base;
    a.blk_tris = (uint64_t *)(base + scan_bytes);
    a.totals = (uint64_t *)(base + 2 * scan_bytes);
    a.mm = (float *)(base + 2 * scan_bytes + 256);
    a.tile_sums = (uint64_t *)(base + 2 * scan_bytes + 256 + mm_bytes);
    const size_t gflag_bytes = (size_t)round_up(a.nblocks, 256);
    a.brick = (uint8_t *)(base + 2 * scan_bytes + 256 + mm_bytes + tile_bytes);
    a.gflag = (uint8_t *)(base + 2 * scan_bytes + 256 + mm_bytes + tile_bytes + brick_bytes);
    a.edge_index = (int32_t *)(base + 2 * scan_bytes + 256 + mm_bytes + tile_bytes + brick_bytes + gflag_bytes);
    a.bytes = 2 * scan_bytes + 256 + mm_bytes + tile_bytes + brick_bytes + gflag_bytes + edge_bytes;
    return a;
}

Vol make_vol(const void *volume, int dtype, int n0, int n1, int n2, double level, int zoff)
{
    return Vol{volume, dtype, n0, n1, n2, n0 - 1, n1 - 1, n2 - 1, (int64_t)(n0 - 1) * (n1 - 1) * (n2 - 1), level, nullptr,
               (n0 - 1 + 7) / 8, (n1 - 1 + 7) / 8, (n2 - 1 + 7) / 8, nullptr, zoff};
}

// counting half, shared by the whole-volume and the slab entry points: totals and the value range on the host
int mc_count_impl(pifu_ctx *ctx, const void *volume, int dtype, int n0, int n1, int n2, double level, int zoff,
                  uint64_t *nv, uint64_t *nf, float *lo, float *hi, hipStream_t st)
{
    PIFU_HIP(hipSetDevice(ctx->device));
    ctx->mc.valid = false;
    Vol v = make_vol(volume, dtype, n0, n1, n2, level, zoff);
    McArena a = carve(v, nullptr);
    int rc = pifu_ensure(ctx->mc_arena, a.bytes);
    if (rc) return rc;
    a = carve(v, (char *)ctx->mc_arena.p);
    hipEvent_t prof_stop;
    rc = pifu_prof_begin(ctx, PIFU_PROF_MC, st, &prof_stop);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mc_bricks, dim3((unsigned)((a.nbricks + 3) / 4)), dim3(256), 0, st, v, a.brick);
    v.brick = a.brick;
    hipLaunchKernelGGL(k_mc_minmax, dim3(MM_BLOCKS), dim3(256), 0, st, v, a.mm + 2);
    hipLaunchKernelGGL(k_mc_minmax_final, dim3(1), dim3(64), 0, st, (const float *)(a.mm + 2), MM_BLOCKS, a.mm);
    hipLaunchKernelGGL(k_mc_groups, dim3((unsigned)((a.nblocks + 255) / 256)), dim3(256), 0, st, v, a.nblocks, a.gflag);
    v.gflag = a.gflag;
    hipLaunchKernelGGL(k_mc_count, dim3((unsigned)((a.nblocks + MC_GPB - 1) / MC_GPB)), dim3(256), 0, st, v, a.nblocks, a.blk_verts,
                       a.blk_tris);
    scan_u64(a.blk_verts, a.nblocks, a.totals, a.tile_sums, st);
    scan_u64(a.blk_tris, a.nblocks, a.totals + 1, a.tile_sums, st);      // stream order: the scratch is free again
    // timed: the counting half; work = one pass over the volume (the floor of any marching cubes: every sample is read once)
    rc = pifu_prof_end(ctx, PIFU_PROF_MC, prof_stop, (double)n0 * n1 * n2 * (dtype ? 8.0 : 4.0), st);
    if (rc) return rc;
    char *pin = (char *)ctx->pinned;
    PIFU_HIP(hipMemcpyAsync(pin, a.totals, 16, hipMemcpyDeviceToHost, st));
    PIFU_HIP(hipMemcpyAsync(pin + 16, a.mm, 8, hipMemcpyDeviceToHost, st));
    PIFU_HIP(hipStreamSynchronize(st));
    *nv = ((uint64_t *)pin)[0];
    *nf = ((uint64_t *)pin)[1];
    *lo = ((float *)(pin + 16))[0];
    *hi = ((float *)(pin + 16))[1];
    McState &m = ctx->mc;
    m.n0 = n0; m.n1 = n1; m.n2 = n2;
    m.level = level; m.dtype = dtype; m.volume = volume; m.zoff = zoff;
    m.nverts = (int64_t)*nv; m.nfaces = (int64_t)*nf;
    m.verts_done = false;
    m.vert_base = 0;
    return PIFU_OK;
}

int mc_verts_impl(pifu_ctx *ctx, float *verts, int64_t vert_base, hipStream_t st)
{
    McState &m = ctx->mc;
    Vol v = make_vol(m.volume, m.dtype, m.n0, m.n1, m.n2, m.level, m.zoff);
    const McArena a = carve(v, (char *)ctx->mc_arena.p);
    v.brick = a.brick;                                                        // built by the counting half for this volume
    v.gflag = a.gflag;
    hipLaunchKernelGGL(k_mc_verts, dim3((unsigned)((a.nblocks + MC_GPB - 1) / MC_GPB)), dim3(256), 0, st, v, a.nblocks,
                       (const uint64_t *)a.blk_verts, vert_base, verts, a.edge_index);
    m.verts_done = true;
    m.vert_base = vert_base;
    return PIFU_OK;
}

int mc_faces_impl(pifu_ctx *ctx, int32_t *faces, hipStream_t st)
{
    McState &m = ctx->mc;
    Vol v = make_vol(m.volume, m.dtype, m.n0, m.n1, m.n2, m.level, m.zoff);
    const McArena a = carve(v, (char *)ctx->mc_arena.p);
    v.brick = a.brick;
    v.gflag = a.gflag;
    hipLaunchKernelGGL(k_mc_faces, dim3((unsigned)((a.nblocks + MC_GPB - 1) / MC_GPB)), dim3(256), 0, st, v, a.nblocks,
                       (const uint64_t *)a.blk_verts, (const uint64_t *)a.blk_tris, m.vert_base, (const int32_t *)a.edge_index, faces);
    return PIFU_OK;
}

// the x- and y-edge index planes of node plane z of the current volume: [2][n1][n2] int32
int32_t *edge_plane(pifu_ctx *ctx, int axis, int z)
{
    const McState &m = ctx->mc;
    Vol v = make_vol(m.volume, m.dtype, m.n0, m.n1, m.n2, m.level, m.zoff);
    const McArena a = carve(v, (char *)ctx->mc_arena.p);
    return a.edge_index + (size_t)axis * m.n0 * m.n1 * m.n2 + (size_t)z * m.n1 * m.n2;
}

}  // namespace

extern "C" int pifu_mc_count(pifu_ctx *ctx, const void *volume, int dtype, int n0, int n1, int n2, double level,
                             int64_t *nverts, int64_t *nfaces, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && volume && nverts && nfaces, PIFU_ERR_INVALID, "pifu_mc_count: null argument");
    PIFU_REQUIRE(dtype == 0 || dtype == 1, PIFU_ERR_INVALID, "pifu_mc_count: dtype must be 0 (f32) or 1 (f64)");
    PIFU_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, PIFU_ERR_INVALID, "Input array must be at least 2x2x2.");
    uint64_t nv, nf;
    float lo, hi;
    int rc = mc_count_impl(ctx, volume, dtype, n0, n1, n2, level, 0, &nv, &nf, &lo, &hi, (hipStream_t)stream);
    if (rc) return rc;
    if (level < (double)lo || level > (double)hi) {
        pifu_set_error("Surface level must be within volume data range.");
        return PIFU_ERR_MC_LEVEL;
    }
    if (nv == 0) {
        pifu_set_error("No surface found at the given iso value.");
        return PIFU_ERR_MC_EMPTY;
    }
    PIFU_REQUIRE(nv < ((uint64_t)1 << 31), PIFU_ERR_INVALID, "pifu_mc_count: %llu vertices do not fit int32 face indices", (unsigned long long)nv);
    ctx->mc.valid = true;
    ctx->mc.slab = false;
    *nverts = (int64_t)nv;
    *nfaces = (int64_t)nf;
    return PIFU_OK;
}

extern "C" int pifu_mc_emit(pifu_ctx *ctx, float *verts, int32_t *faces, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && verts && faces, PIFU_ERR_INVALID, "pifu_mc_emit: null argument");
    PIFU_REQUIRE(ctx->mc.valid && !ctx->mc.slab, PIFU_ERR_STATE, "pifu_mc_emit: call pifu_mc_count first");
    PIFU_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const McState &m = ctx->mc;
    hipEvent_t prof_stop;
    int rc = pifu_prof_begin(ctx, PIFU_PROF_MC, st, &prof_stop);
    if (rc) return rc;
    mc_verts_impl(ctx, verts, 0, st);
    mc_faces_impl(ctx, faces, st);
    // timed: the emit half; work = the vertex and face streams written (12 B each)
    rc = pifu_prof_end(ctx, PIFU_PROF_MC, prof_stop, 12.0 * ((double)m.nverts + (double)m.nfaces), st);
    if (rc) return rc;
    PIFU_HIP(hipGetLastError());
    ctx->mc.valid = false;
    return PIFU_OK;
}

// ---- slab-sharded marching cubes (axis-0 slabs with a one-plane halo + exclusive prefix of the counts) ----------
extern "C" int pifu_mc_slab_count(pifu_ctx *ctx, const void *slab, int dtype, int n0, int n1, int n2, int z_first, double level,
                                  int64_t *nverts, int64_t *nfaces, float *vmin, float *vmax, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && slab && nverts && nfaces && vmin && vmax, PIFU_ERR_INVALID, "pifu_mc_slab_count: null argument");
    PIFU_REQUIRE(dtype == 0 || dtype == 1, PIFU_ERR_INVALID, "pifu_mc_slab_count: dtype must be 0 (f32) or 1 (f64)");
    PIFU_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2 && z_first >= 0, PIFU_ERR_INVALID,
                 "pifu_mc_slab_count: a slab needs at least 2 node planes of at least 2x2 nodes (got %d x %d x %d at plane %d)", n0, n1,
                 n2, z_first);
    uint64_t nv, nf;
    int rc = mc_count_impl(ctx, slab, dtype, n0, n1, n2, level, z_first, &nv, &nf, vmin, vmax, (hipStream_t)stream);
    if (rc) return rc;
    ctx->mc.valid = true;
    ctx->mc.slab = true;
    *nverts = (int64_t)nv;
    *nfaces = (int64_t)nf;
    return PIFU_OK;
}

extern "C" int pifu_mc_slab_verts(pifu_ctx *ctx, float *verts, int64_t vert_base, int32_t *halo_out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx, PIFU_ERR_INVALID, "pifu_mc_slab_verts: null context");
    PIFU_REQUIRE(ctx->mc.valid && ctx->mc.slab, PIFU_ERR_STATE, "pifu_mc_slab_verts: call pifu_mc_slab_count first");
    const McState &m = ctx->mc;
    PIFU_REQUIRE(verts || m.nverts == 0, PIFU_ERR_INVALID, "pifu_mc_slab_verts: null vertex array for %lld vertices", (long long)m.nverts);
    PIFU_REQUIRE(vert_base >= 0 && vert_base + m.nverts < ((int64_t)1 << 31), PIFU_ERR_INVALID,
                 "pifu_mc_slab_verts: vertex numbers %lld + %lld do not fit int32 face indices", (long long)vert_base, (long long)m.nverts);
    PIFU_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    mc_verts_impl(ctx, verts, vert_base, st);
    if (halo_out) {
        // the vertices on the x / y edges of the slab's top node plane, by their numbers in the whole mesh: the slab above
        // references them from its bottom cells.  Slots of edges without a vertex hold stale values no cell ever reads.
        const size_t plane = (size_t)m.n1 * m.n2 * sizeof(int32_t);
        PIFU_HIP(hipMemcpyAsync(halo_out, edge_plane(ctx, 0, m.n0 - 1), plane, hipMemcpyDeviceToDevice, st));
        PIFU_HIP(hipMemcpyAsync((char *)halo_out + plane, edge_plane(ctx, 1, m.n0 - 1), plane, hipMemcpyDeviceToDevice, st));
    }
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

extern "C" int pifu_mc_slab_faces(pifu_ctx *ctx, int32_t *faces, const int32_t *halo_in, pifu_stream stream)
{
    PIFU_REQUIRE(ctx, PIFU_ERR_INVALID, "pifu_mc_slab_faces: null context");
    PIFU_REQUIRE(ctx->mc.valid && ctx->mc.slab && ctx->mc.verts_done, PIFU_ERR_STATE,
                 "pifu_mc_slab_faces: call pifu_mc_slab_count and pifu_mc_slab_verts first");
    const McState &m = ctx->mc;
    PIFU_REQUIRE(faces || m.nfaces == 0, PIFU_ERR_INVALID, "pifu_mc_slab_faces: null face array for %lld faces", (long long)m.nfaces);
    PIFU_REQUIRE((m.zoff == 0) == (halo_in == nullptr), PIFU_ERR_INVALID,
                 "pifu_mc_slab_faces: the slab starting at node plane %d %s a halo", m.zoff, m.zoff ? "needs" : "must not be given");
    PIFU_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    if (halo_in) {
        const size_t plane = (size_t)m.n1 * m.n2 * sizeof(int32_t);
        PIFU_HIP(hipMemcpyAsync(edge_plane(ctx, 0, 0), halo_in, plane, hipMemcpyDeviceToDevice, st));
        PIFU_HIP(hipMemcpyAsync(edge_plane(ctx, 1, 0), (const char *)halo_in + plane, plane, hipMemcpyDeviceToDevice, st));
    }
    mc_faces_impl(ctx, faces, st);
    PIFU_HIP(hipGetLastError());
    ctx->mc.valid = false;
    return PIFU_OK;
}

// ---- vertex back-transform (mesh_util.py:87-89) ------------------------------------------------------------------------------
namespace {
struct Mat34 { double m[12]; };
__global__ void k_transform_verts(const float *verts, int64_t n, const Mat34 t, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = (double)verts[3 * i], y = (double)verts[3 * i + 1], z = (double)verts[3 * i + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double *m = t.m + 4 * r;
        out[3 * i + r] = __dadd_rn(__fma_rn(m[2], z, __fma_rn(m[1], y, __dmul_rn(m[0], x))), m[3]);
    }
}
}  // namespace

extern "C" int pifu_transform_verts(pifu_ctx *ctx, const float *verts, int64_t n, const double *mat, double *out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && mat && (n == 0 || (verts && out)) && n >= 0, PIFU_ERR_INVALID, "pifu_transform_verts: bad argument");
    if (n == 0) return PIFU_OK;
    PIFU_HIP(hipSetDevice(ctx->device));
    Mat34 t;
    for (int k = 0; k < 12; ++k) t.m[k] = mat[k];
    hipLaunchKernelGGL(k_transform_verts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, n, t, out);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}
