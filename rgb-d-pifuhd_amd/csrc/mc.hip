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

// corner -> (dx,dy,dz)
__device__ __constant__ const signed char COFF[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// edge e joins corners {0,1,2,3,4,5,6,7,0,1,2,3}[e] and {1,2,3,0,5,6,7,4,4,5,6,7}[e]; its axis (0 = x, 1 = y, 2 = z) and the
// offset of its lower endpoint are the immediates EDGE_AXIS_BITS / EDGE_BASE_BITS further down.  EDGE_NEEDZERO: the mask of
// cell coordinates (bit0 x, bit1 y, bit2 z) that must be 0 for this cell to be the first one touching the edge
constexpr uint64_t EDGE_NEEDZERO_BITS = (6ull << 0) | (4ull << 3) | (4ull << 6) | (5ull << 9) | (2ull << 12) | (0ull << 15) | (0ull << 18) |
                                        (1ull << 21) | (3ull << 24) | (2ull << 27) | (0ull << 30) | (1ull << 33);      // {2|4, 4, 4, 1|4, 2, 0, 0, 1, 1|2, 2, 0, 1}
// interior-test table e:(p,q,(b0,b1),(c0,c1),(d0,d1))
__device__ __constant__ const signed char TI[12][8] = {{0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7},
                                                      {3, 0, 2, 1, 6, 5, 7, 4}, {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2},
                                                      {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0}, {0, 4, 3, 7, 2, 6, 1, 5},
                                                      {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};

// the 8 values (v - level) of the cell being classified.  The ambiguity tests index them by table entries, i.e. dynamically:
// held in registers that costs a scratch array per lane (measured: 64 B written and re-read per cell of every active
// brick, 340 MB per 512^3 call); they live in LDS instead, corner i of thread t at [i * 256 + t] (conflict-free), as the
// float32 samples (8 KB per block: the counting pass is latency-bound and lives on its occupancy)
struct Corners {
    const float *p;            // the float32 samples; v - level is formed in double on every read, as classify() forms it
    double level;
    __device__ __forceinline__ double operator[](int i) const { return (double)p[i * 256] - level; }
};

__device__ bool test_face(const Corners c, int f)
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

__device__ bool test_interior(const Corners c, int mc_case, int cfg, int subcfg, int s)
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
__device__ int choose_tiling(const Corners c, int index, const signed char **row)
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
    // 8 x 8 x 8-cell bricks whose 9 x 9 x 9 nodes are not all on one side of the level (k_mc_nodebricks + k_mc_brick_combine); null = classify every cell
    const uint8_t *brick;
    int b0, b1, b2;
    // 1 for every group of 256 consecutive cells of the serial scan that touches such a brick (k_mc_groups); the three
    // counting pass gives a block a few groups (mc_groups_per_block) and skips the others without touching their cells
    const uint8_t *gflag;
    // slab mode (pifu_mc_slab_*): the volume holds node planes [zoff, zoff + n0) of a taller one.  Vertex ownership along
    // axis 0 and the axis-0 coordinate of the vertices refer to the tall volume; 0 for a whole volume
    int zoff;
};

__device__ __forceinline__ float vol_f32(const Vol &v, int z, int y, int x)
{
    const size_t i = ((size_t)z * v.n1 + y) * v.n2 + x;
    return v.dtype == 0 ? ((const float *)v.p)[i] : (float)((const double *)v.p)[i];
}

__device__ __forceinline__ double vol_at(const Vol &v, int z, int y, int x) { return (double)vol_f32(v, z, y, x); }

struct CellInfo {
    const signed char *row;
    int nt;
    int nown;                  // vertices this cell creates
    uint64_t own;              // their edge ids (12 = the centre vertex), 4 bits each, in the order the cell numbers them
};

// classify one cell (blocks of 256 threads; `corners`: 8 x 256 doubles of LDS); fills row, nt and the owned vertices.
// Returns false for an empty cell.
__device__ bool classify(const Vol &v, int64_t cell, CellInfo &ci, float *corners)
{
    const int x = (int)(cell % v.c2), y = (int)((cell / v.c2) % v.c1), z = (int)(cell / ((int64_t)v.c2 * v.c1));
    ci.nt = 0;
    ci.nown = 0;
    ci.own = 0;
    ci.row = nullptr;
    // a cell of a brick whose nodes all lie on one side of the level has index 0 or 255: nothing to load
    if (v.brick && !v.brick[((size_t)(z >> 3) * v.b1 + (y >> 3)) * v.b2 + (x >> 3)]) return false;
    int index = 0;
    float d[8];                                                  // constant indices only: registers
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d[i] = vol_f32(v, z + COFF[i][2], y + COFF[i][1], x + COFF[i][0]);
        if (((double)d[i] - v.level) > 0) index |= 1 << i;
    }
    if (index == 0 || index == 255) return false;
    float *mine = corners + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[i * 256] = d[i];
    ci.nt = choose_tiling(Corners{mine, v.level}, index, &ci.row);
    const int zero_mask = (x == 0 ? 1 : 0) | (y == 0 ? 2 : 0) | (z + v.zoff == 0 ? 4 : 0);
    // the triangle row (3 nt <= 36 edge ids): every byte is requested before the first one is looked at - one memory round
    // trip instead of a chain of 3 nt dependent ones - and walked from registers (all indices static after unrolling)
    const int nrow = 3 * ci.nt;
    signed char rb[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) rb[k] = k < nrow ? ci.row[k] : (signed char)0;
    unsigned seen = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        if (k < nrow) {
            const int e = rb[k];
            if (!(seen & (1u << e))) {
                seen |= 1u << e;
                // EDGE_NEEDZERO as an immediate, 3 bits per edge (12: the centre vertex, always the cell's own)
                const unsigned need = e == 12 ? 0u : (unsigned)((EDGE_NEEDZERO_BITS >> (3 * e)) & 7u);
                if ((need & ~(unsigned)zero_mask) == 0) {
                    ci.own |= (uint64_t)e << (4 * ci.nown);
                    ++ci.nown;
                }
            }
        }
    }
    return true;
}

// ---- pass 0: value range + which 8 x 8 x 8-cell bricks can hold a surface cell, from ONE coalesced pass over the nodes -------
// A cell brick b needs the nodes [8 b, 8 b + 8] of every axis: node brick b (8 x 8 x 8 nodes) plus the first node planes /
// lines / corner of its upper neighbours.  k_mc_nodebricks reads every node once - a block per (z brick, y brick) pencil,
// consecutive threads on consecutive x - and leaves per node brick, for each of the 8 sub-regions "nodes whose local
// coordinates in the axes of set r are 0", whether a positive and whether a non-positive value of (v - level) occurs,
// evaluated exactly as classify() does; k_mc_brick_combine ORs the 8 pieces a cell brick consists of.
// region sets: bit 0 = x, bit 1 = y, bit 2 = z.  A node whose zero coordinates are Z belongs to every region r subset of Z.
__device__ __forceinline__ unsigned regions_of(unsigned Z)
{
    // bit r set iff (r & ~Z) == 0, for Z = 0..7: 0x01, 0x03, 0x05, 0x0f, 0x11, 0x33, 0x55, 0xff
    return (unsigned)((0xff553311'0f050301ull >> (8 * Z)) & 0xffu);
}

// VEC = 2: a thread takes two consecutive x per row with one 16-byte (float64) / 8-byte (float32) load - needs an even n2;
// VEC = 1: one x per thread and row, any n2
template <int VEC>
__global__ __launch_bounds__(256) void k_mc_nodebricks(const Vol v, int nb1, int nb2, uint16_t *nbflags, float *mm /* [2*gridDim] */)
{
    __shared__ float smin[4], smax[4];
    const int by = blockIdx.x % nb1, bz = blockIdx.x / nb1;
    float lo = INFINITY, hi = -INFINITY;
    for (int x0 = 0; x0 < nb2 * 8; x0 += 256 * VEC) {
        const int x = x0 + (int)threadIdx.x * VEC;
        unsigned pos = 0, neg = 0;
        if (x < v.n2) {
            const unsigned zx = (x & 7) == 0 ? 1u : 0u;
            for (int dz = 0; dz < 8; ++dz) {
                const int z = bz * 8 + dz;
                if (z >= v.n0) break;                                        // block-uniform
#pragma unroll
                for (int dy = 0; dy < 8; ++dy) {
                    const int y = by * 8 + dy;
                    if (y >= v.n1) break;                                    // block-uniform
                    const size_t i = ((size_t)z * v.n1 + y) * v.n2 + x;
                    float f[VEC];
                    if (VEC == 1) {
                        f[0] = v.dtype == 0 ? ((const float *)v.p)[i] : (float)((const double *)v.p)[i];
                    } else if (v.dtype == 0) {
                        const float2 t = *reinterpret_cast<const float2 *>((const float *)v.p + i);
                        f[0] = t.x; f[VEC - 1] = t.y;
                    } else {
                        const double2 t = *reinterpret_cast<const double2 *>((const double *)v.p + i);
                        f[0] = (float)t.x; f[VEC - 1] = (float)t.y;
                    }
                    const unsigned zyz = (dy == 0 ? 2u : 0u) | (dz == 0 ? 4u : 0u);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        lo = fminf(lo, f[k]);
                        hi = fmaxf(hi, f[k]);
                        const unsigned reg = regions_of((k == 0 ? zx : 0u) | zyz);
                        if (((double)f[k] - v.level) > 0) pos |= reg; else neg |= reg;
                    }
                }
            }
        }
        unsigned both = pos | (neg << 8);
        both |= __shfl_xor(both, 1);
        both |= __shfl_xor(both, 2);
        if (VEC == 1) both |= __shfl_xor(both, 4);
        if ((threadIdx.x & (8 / VEC - 1)) == 0 && (x >> 3) < nb2) nbflags[((size_t)bz * nb1 + by) * nb2 + (x >> 3)] = (uint16_t)both;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        mm[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

__global__ __launch_bounds__(256) void k_mc_minmax_final(const float *part, int n, float *out)
{
    __shared__ float smin[4], smax[4];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        out[1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

__global__ void k_mc_brick_combine(const Vol v, int nb0, int nb1, int nb2, const uint16_t *nbflags, uint8_t *brick)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nb = (int64_t)v.b0 * v.b1 * v.b2;
    if (b >= nb) return;
    const int bx = (int)(b % v.b2), by = (int)((b / v.b2) % v.b1), bz = (int)(b / ((int64_t)v.b2 * v.b1));
    unsigned pos = 0, neg = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {                                        // the neighbour shifted along the axes of set s ...
        const int qx = bx + (s & 1), qy = by + ((s >> 1) & 1), qz = bz + ((s >> 2) & 1);
        if (qx >= nb2 || qy >= nb1 || qz >= nb0) continue;               // ... if the volume has nodes there
        const unsigned f = nbflags[((size_t)qz * nb1 + qy) * nb2 + qx];
        pos |= (f >> s) & 1u;                                            // ... contributes its region "zero along those axes"
        neg |= (f >> (8 + s)) & 1u;
    }
    brick[b] = (pos && neg) ? 1 : 0;
}

// groups of 256 cells per block of the counting pass: few, so that the surface regions spread over many blocks (512^3: 0.83 ms at
// 32, 0.58 at 8 and at 4), more on the largest volumes, where a block per 1 024 cells is a million blocks (1024^3: 5.2 ms at 8, 6.2 at 4)
inline int mc_groups_per_block(int64_t ngroups) { return ngroups > (int64_t)1 << 21 ? 8 : 4; }

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

// The surface cells, compacted.  One record per cell with at least one triangle, written by the counting pass - the only
// pass that classifies - and read by the two emit passes with one lane per RECORD (full waves, no second classification).
// Records of one group are contiguous; groups take their place in the list by an atomic counter, so the list order varies
// from run to run, the meshes do not: every record carries what fixes its output positions (its cell, hence its group, and
// its exclusive vertex / triangle prefixes inside the group).
struct SurfList {
    int64_t *cell;             // serial cell number
    const signed char **row;   // triangle row (edge ids, 3 per triangle) in the constant tables
    uint64_t *own;             // CellInfo::own
    uint32_t *meta;            // vertices before it in the group (12 bits) | triangles before it (12) | nown (4) | nt (4)
    int64_t cap;
};

// pass 1: group g covers cells [256 g, 256 g + 256) of the serial scan; per group (vertices << 32 | triangles) created,
// and the records of its surface cells.  A block gathers the records of its groups in LDS and takes its place in the
// list with ONE atomic per buffer-full, then copies them out with consecutive lanes on consecutive records.
constexpr int MC_REC_CAP = 256;       // records buffered per block (7 KB); a fuller buffer is flushed before the next group

__global__ __launch_bounds__(256) void k_mc_count(const Vol v, int64_t ngroups, int gpb, uint64_t *blk_vt, const SurfList L,
                                                  unsigned long long *nsurf)
{
    __shared__ uint32_t sh[4];
    __shared__ uint32_t wcnt[4];
    __shared__ unsigned long long base_sh;
    __shared__ float corners[8 * 256];
    __shared__ int64_t r_cell[MC_REC_CAP];
    __shared__ const signed char *r_row[MC_REC_CAP];
    __shared__ uint64_t r_own[MC_REC_CAP];
    __shared__ uint32_t r_meta[MC_REC_CAP];
    uint32_t nbuf = 0;                                           // block-uniform: every thread keeps the same count

    auto flush = [&]() {
        if (threadIdx.x == 0) base_sh = atomicAdd(nsurf, (unsigned long long)nbuf);
        __syncthreads();                                         // also: every record of the batch is in LDS
        const int64_t base = (int64_t)base_sh;
        for (uint32_t i = threadIdx.x; i < nbuf; i += 256) {
            const int64_t slot = base + i;
            if (slot < L.cap) {                                  // past the capacity: counted only, the host grows the list and reruns
                L.cell[slot] = r_cell[i];
                L.row[slot] = r_row[i];
                L.own[slot] = r_own[i];
                L.meta[slot] = r_meta[i];
            }
        }
        __syncthreads();                                         // the buffer may be refilled
        nbuf = 0;
    };

    for (int gi = 0; gi < gpb; ++gi) {
        const int64_t group = (int64_t)blockIdx.x * gpb + gi;
        if (group >= ngroups) break;
        if (!v.gflag[group]) {                                   // block-uniform
            if (threadIdx.x == 0) blk_vt[group] = 0;
            continue;
        }
        const int64_t cell = group * 256 + threadIdx.x;
        CellInfo ci;
        bool active = false;
        if (cell < v.ncells) active = classify(v, cell, ci, corners);
        const uint32_t packed = active ? ((uint32_t)ci.nown | ((uint32_t)ci.nt << 16)) : 0u;
        uint32_t tot;
        const uint32_t pre = block_exclusive_scan_256(packed, sh, &tot);
        // position among the group's surface cells
        const uint64_t bal = __builtin_amdgcn_ballot_w64(active);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) wcnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t spre = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)), stot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) spre += wcnt[w];
            stot += wcnt[w];
        }
        if (threadIdx.x == 0) blk_vt[group] = ((uint64_t)(tot & 0xffffu) << 32) | (uint64_t)(tot >> 16);
        if (nbuf + stot > MC_REC_CAP) flush();                   // block-uniform (stot <= 256 always fits an empty buffer)
        if (active) {
            const uint32_t i = nbuf + spre;
            r_cell[i] = cell;
            r_row[i] = ci.row;
            r_own[i] = ci.own;
            r_meta[i] = (pre & 0xfffu) | (((pre >> 16) & 0xfffu) << 12) | ((uint32_t)ci.nown << 24) | ((uint32_t)ci.nt << 28);
        }
        nbuf += stot;
    }
    if (nbuf) flush();                                           // block-uniform
}

__device__ __forceinline__ size_t edge_slot(const Vol &v, int axis, int z, int y, int x)
{
    return (size_t)axis * v.n0 * v.n1 * v.n2 + ((size_t)z * v.n1 + y) * v.n2 + x;
}

// edge -> axis (2 bits each) and base-point offset (dx | dy << 1 | dz << 2, 3 bits each) as immediates: EDGE_AXIS / EDGE_BASE
constexpr uint32_t EDGE_AXIS_BITS = 0u | (1u << 2) | (0u << 4) | (1u << 6) | (0u << 8) | (1u << 10) | (0u << 12) | (1u << 14) |
                                    (2u << 16) | (2u << 18) | (2u << 20) | (2u << 22);
constexpr uint64_t EDGE_BASE_BITS = 0ull | (1ull << 3) | (2ull << 6) | (0ull << 9) | (4ull << 12) | (5ull << 15) | (6ull << 18) |
                                    (4ull << 21) | (0ull << 24) | (1ull << 27) | (3ull << 30) | (2ull << 33);

// pass 2: one lane per surface cell: vertex positions + per-grid-edge vertex index volume
__global__ __launch_bounds__(256) void k_mc_verts(const Vol v, const SurfList L, int64_t nsurf, const uint64_t *blk_vt,
                                                  int64_t vert_base, float *verts, int32_t *edge_index)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsurf) return;
    const uint32_t meta = L.meta[i];
    const int nown = (meta >> 24) & 15;
    if (nown == 0) return;
    const int64_t cell = L.cell[i];
    uint64_t own = L.own[i];
    const int x = (int)(cell % v.c2), y = (int)((cell / v.c2) % v.c1), z = (int)(cell / ((int64_t)v.c2 * v.c1));
    // `verts` is this volume's own array (local numbering); the index volume holds numbers of the whole mesh
    int64_t vid = (int64_t)(blk_vt[cell >> 8] >> 32) + (meta & 0xfffu);
    for (int j = 0; j < nown; ++j, own >>= 4, ++vid) {
        const int e = (int)(own & 15u);
        if (e == 12) {
            double sw = 0, sx = 0, sy = 0, sz = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double val = vol_at(v, z + COFF[c][2], y + COFF[c][1], x + COFF[c][0]) - v.level;
                const double w = 1.0 / __dadd_rn(MC_EPS, fabs(val));
                sx = __dadd_rn(sx, __dmul_rn(w, (double)COFF[c][0]));
                sy = __dadd_rn(sy, __dmul_rn(w, (double)COFF[c][1]));
                sz = __dadd_rn(sz, __dmul_rn(w, (double)COFF[c][2]));
                sw = __dadd_rn(sw, w);
            }
            verts[3 * vid + 0] = (float)__dadd_rn((double)(z + v.zoff), sz / sw);
            verts[3 * vid + 1] = (float)__dadd_rn((double)y, sy / sw);
            verts[3 * vid + 2] = (float)__dadd_rn((double)x, sx / sw);
        } else {
            const int axis = (int)((EDGE_AXIS_BITS >> (2 * e)) & 3u);
            const unsigned bo = (unsigned)((EDGE_BASE_BITS >> (3 * e)) & 7u);
            const int bx = x + (int)(bo & 1u), by = y + (int)((bo >> 1) & 1u), bz = z + (int)(bo >> 2);
            // lower endpoint = the node at the base point, upper = its neighbour along the edge's axis
            const double clo = vol_at(v, bz, by, bx) - v.level;
            const double chi = vol_at(v, bz + (axis == 2), by + (axis == 1), bx + (axis == 0)) - v.level;
            const double wlo = 1.0 / __dadd_rn(MC_EPS, fabs(clo)), whi = 1.0 / __dadd_rn(MC_EPS, fabs(chi));
            const double t = whi / __dadd_rn(wlo, whi);
            const int gz = bz + v.zoff;
            verts[3 * vid + 0] = (float)(axis == 2 ? __dadd_rn((double)gz, t) : (double)gz);
            verts[3 * vid + 1] = (float)(axis == 1 ? __dadd_rn((double)by, t) : (double)by);
            verts[3 * vid + 2] = (float)(axis == 0 ? __dadd_rn((double)bx, t) : (double)bx);
            edge_index[edge_slot(v, axis, bz, by, bx)] = (int32_t)(vert_base + vid);
        }
    }
}

// pass 3: one lane per surface cell: triangle indices ('descent' => the three indices of every triangle are reversed)
__global__ __launch_bounds__(256) void k_mc_faces(const Vol v, const SurfList L, int64_t nsurf, const uint64_t *blk_vt,
                                                  int64_t vert_base, const int32_t *edge_index, int32_t *faces)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsurf) return;
    const uint32_t meta = L.meta[i];
    const int nt = (int)(meta >> 28), nown = (meta >> 24) & 15;
    const int64_t cell = L.cell[i];
    const signed char *row = L.row[i];
    const uint64_t vt = blk_vt[cell >> 8];
    const int x = (int)(cell % v.c2), y = (int)((cell / v.c2) % v.c1), z = (int)(cell / ((int64_t)v.c2 * v.c1));
    // the centre vertex is the cell's own: its first vertex number + the rank of id 12 among its owned vertices
    int32_t centre = -1;
    {
        uint64_t own = L.own[i];
        for (int j = 0; j < nown; ++j, own >>= 4)
            if ((own & 15u) == 12u) { centre = (int32_t)(vert_base + (int64_t)(vt >> 32) + (meta & 0xfffu) + j); break; }
    }
    int64_t f = (int64_t)(vt & 0xffffffffull) + ((meta >> 12) & 0xfffu);
    for (int t = 0; t < nt; ++t, ++f) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = row[3 * t + k];
            int32_t vid = centre;
            if (e != 12) {
                const int axis = (int)((EDGE_AXIS_BITS >> (2 * e)) & 3u);
                const unsigned bo = (unsigned)((EDGE_BASE_BITS >> (3 * e)) & 7u);
                vid = edge_index[edge_slot(v, axis, z + (int)(bo >> 2), y + (int)((bo >> 1) & 1u), x + (int)(bo & 1u))];
            }
            faces[3 * f + (2 - k)] = vid;
        }
    }
}

struct McArena {
    uint64_t *blk_vt, *totals, *tile_sums;
    float *mm;
    uint16_t *nbflags;
    uint8_t *brick, *gflag;
    SurfList list;
    int32_t *edge_index;
    int64_t nblocks, nbricks, nnb;      // nblocks = groups of 256 cells; nnb = node bricks
    int nb0, nb1, nb2;
    size_t bytes;
};

// arena: [blk_vt nblocks+1][totals][minmax partials][scan scratch][node-brick flags][brick map][group flags][surface list][edge_index 3*N int32]
McArena carve(const Vol &v, char *base, int64_t cap)
{
    McArena a;
    a.nblocks = (v.ncells + 255) / 256;
    a.nbricks = (int64_t)v.b0 * v.b1 * v.b2;
    a.nb0 = (v.n0 + 7) / 8; a.nb1 = (v.n1 + 7) / 8; a.nb2 = (v.n2 + 7) / 8;
    a.nnb = (int64_t)a.nb0 * a.nb1 * a.nb2;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base + off; off += (size_t)round_up((int64_t)bytes, 256); return p; };
    a.blk_vt = (uint64_t *)take((size_t)(a.nblocks + 1) * 8);
    a.totals = (uint64_t *)take(256);
    a.mm = (float *)take((size_t)(2 * (size_t)a.nb0 * a.nb1 + 2) * 4);
    a.tile_sums = (uint64_t *)take((size_t)(scan_tiles(a.nblocks) + 1) * 8);
    a.nbflags = (uint16_t *)take((size_t)a.nnb * 2);
    a.brick = (uint8_t *)take((size_t)a.nbricks);
    a.gflag = (uint8_t *)take((size_t)a.nblocks);
    a.list.cap = cap;
    a.list.cell = (int64_t *)take((size_t)cap * 8);
    a.list.row = (const signed char **)take((size_t)cap * 8);
    a.list.own = (uint64_t *)take((size_t)cap * 8);
    a.list.meta = (uint32_t *)take((size_t)cap * 4);
    a.edge_index = (int32_t *)take((size_t)3 * v.n0 * v.n1 * v.n2 * sizeof(int32_t));
    a.bytes = off;
    return a;
}

Vol make_vol(const void *volume, int dtype, int n0, int n1, int n2, double level, int zoff)
{
    return Vol{volume, dtype, n0, n1, n2, n0 - 1, n1 - 1, n2 - 1, (int64_t)(n0 - 1) * (n1 - 1) * (n2 - 1), level, nullptr,
               (n0 - 1 + 7) / 8, (n1 - 1 + 7) / 8, (n2 - 1 + 7) / 8, nullptr, zoff};
}

// first guess of the list capacity: one cell in 16 on the surface (a 512^3 body has ~1 in 500, the roughest synthetic field of
// the bench 1 in 30); a volume that needs more reruns the counting pass once with the exact number
int64_t initial_list_cap(const Vol &v) { return std::max<int64_t>(1 << 16, v.ncells / 16); }

// counting half, shared by the whole-volume and the slab entry points: totals and the value range on the host
int mc_count_impl(pifu_ctx *ctx, const void *volume, int dtype, int n0, int n1, int n2, double level, int zoff,
                  uint64_t *nv, uint64_t *nf, float *lo, float *hi, hipStream_t st)
{
    PIFU_HIP(hipSetDevice(ctx->device));
    ctx->mc.valid = false;
    Vol v = make_vol(volume, dtype, n0, n1, n2, level, zoff);
    // the list keeps the capacity an earlier, rougher volume needed (bounded by this volume's cell count)
    int64_t cap = std::min<int64_t>(std::max<int64_t>(initial_list_cap(v), ctx->mc.list_cap), v.ncells);
    hipEvent_t prof_stop;
    int rc = pifu_prof_begin(ctx, PIFU_PROF_MC, st, &prof_stop);
    if (rc) return rc;
    uint64_t nsurf = 0;
    for (int attempt = 0;; ++attempt) {
        McArena a = carve(v, nullptr, cap);
        rc = pifu_ensure(ctx->mc_arena, a.bytes);
        if (rc) return rc;
        a = carve(v, (char *)ctx->mc_arena.p, cap);
        const int npencils = a.nb0 * a.nb1;
        v.brick = nullptr; v.gflag = nullptr;
        // two x per thread when the rows (and with them every 16-byte load) are aligned
        const bool vec = n2 % 2 == 0 && ((uintptr_t)volume % 16) == 0;
        // timed on its own as well (PIFU_PROF_MC_NODES): the one pass of marching cubes that streams the volume, i.e. the part
        // the HBM roofline prices; everything after it works on the few per cent of the cells that hold surface
        hipEvent_t nodes_stop;
        rc = pifu_prof_begin(ctx, PIFU_PROF_MC_NODES, st, &nodes_stop);
        if (rc) return rc;
        if (vec) hipLaunchKernelGGL(k_mc_nodebricks<2>, dim3((unsigned)npencils), dim3(256), 0, st, v, a.nb1, a.nb2, a.nbflags, a.mm + 2);
        else hipLaunchKernelGGL(k_mc_nodebricks<1>, dim3((unsigned)npencils), dim3(256), 0, st, v, a.nb1, a.nb2, a.nbflags, a.mm + 2);
        rc = pifu_prof_end(ctx, PIFU_PROF_MC_NODES, nodes_stop, (double)n0 * n1 * n2 * (dtype ? 8.0 : 4.0), st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_mc_minmax_final, dim3(1), dim3(256), 0, st, (const float *)(a.mm + 2), npencils, a.mm);
        hipLaunchKernelGGL(k_mc_brick_combine, dim3((unsigned)((a.nbricks + 255) / 256)), dim3(256), 0, st, v, a.nb0, a.nb1, a.nb2,
                           (const uint16_t *)a.nbflags, a.brick);
        v.brick = a.brick;
        hipLaunchKernelGGL(k_mc_groups, dim3((unsigned)((a.nblocks + 255) / 256)), dim3(256), 0, st, v, a.nblocks, a.gflag);
        v.gflag = a.gflag;
        PIFU_HIP(hipMemsetAsync(a.totals, 0, 32, st));
        const int gpb = mc_groups_per_block(a.nblocks);
        hipLaunchKernelGGL(k_mc_count, dim3((unsigned)((a.nblocks + gpb - 1) / gpb)), dim3(256), 0, st, v, a.nblocks, gpb, a.blk_vt,
                           a.list, (unsigned long long *)(a.totals + 2));
        scan_u64(a.blk_vt, a.nblocks, a.totals, a.tile_sums, st);
        char *pin = (char *)ctx->pinned;
        PIFU_HIP(hipMemcpyAsync(pin, a.totals, 24, hipMemcpyDeviceToHost, st));
        PIFU_HIP(hipMemcpyAsync(pin + 24, a.mm, 8, hipMemcpyDeviceToHost, st));
        PIFU_HIP(hipStreamSynchronize(st));
        const uint64_t vt = ((uint64_t *)pin)[0];
        *nv = vt >> 32;
        *nf = vt & 0xffffffffull;
        nsurf = ((uint64_t *)pin)[2];
        *lo = ((float *)(pin + 24))[0];
        *hi = ((float *)(pin + 24))[1];
        // (vertices << 32 | triangles) sums carry no bit across the halves as long as the triangles stay below 2^32
        PIFU_REQUIRE(nsurf * 12ull < (1ull << 32), PIFU_ERR_INVALID, "marching cubes: %llu surface cells are more than the 32-bit "
                     "triangle numbering holds", (unsigned long long)nsurf);
        if ((int64_t)nsurf <= cap) break;
        // more surface cells than the list holds (every one was counted, the surplus not recorded): make room for exactly
        // that many and run the counting half again (the arena sections behind the list move with the capacity)
        PIFU_REQUIRE(attempt == 0, PIFU_ERR_STATE, "marching cubes: surface list overflow after resizing");
        cap = (int64_t)nsurf;
    }
    // timed: the counting half; work = one pass over the volume (the floor of any marching cubes: every sample is read once)
    rc = pifu_prof_end(ctx, PIFU_PROF_MC, prof_stop, (double)n0 * n1 * n2 * (dtype ? 8.0 : 4.0), st);
    if (rc) return rc;
    McState &m = ctx->mc;
    m.n0 = n0; m.n1 = n1; m.n2 = n2;
    m.level = level; m.dtype = dtype; m.volume = volume; m.zoff = zoff;
    m.nverts = (int64_t)*nv; m.nfaces = (int64_t)*nf;
    m.nsurf = (int64_t)nsurf; m.list_cap = cap;
    m.verts_done = false;
    m.vert_base = 0;
    return PIFU_OK;
}

int mc_verts_impl(pifu_ctx *ctx, float *verts, int64_t vert_base, hipStream_t st)
{
    McState &m = ctx->mc;
    Vol v = make_vol(m.volume, m.dtype, m.n0, m.n1, m.n2, m.level, m.zoff);
    const McArena a = carve(v, (char *)ctx->mc_arena.p, m.list_cap);
    if (m.nsurf > 0)
        hipLaunchKernelGGL(k_mc_verts, dim3((unsigned)((m.nsurf + 255) / 256)), dim3(256), 0, st, v, a.list, m.nsurf,
                           (const uint64_t *)a.blk_vt, vert_base, verts, a.edge_index);
    m.verts_done = true;
    m.vert_base = vert_base;
    return PIFU_OK;
}

int mc_faces_impl(pifu_ctx *ctx, int32_t *faces, hipStream_t st)
{
    McState &m = ctx->mc;
    Vol v = make_vol(m.volume, m.dtype, m.n0, m.n1, m.n2, m.level, m.zoff);
    const McArena a = carve(v, (char *)ctx->mc_arena.p, m.list_cap);
    if (m.nsurf > 0)
        hipLaunchKernelGGL(k_mc_faces, dim3((unsigned)((m.nsurf + 255) / 256)), dim3(256), 0, st, v, a.list, m.nsurf,
                           (const uint64_t *)a.blk_vt, m.vert_base, (const int32_t *)a.edge_index, faces);
    return PIFU_OK;
}

// the x- and y-edge index planes of node plane z of the current volume: [2][n1][n2] int32
int32_t *edge_plane(pifu_ctx *ctx, int axis, int z)
{
    const McState &m = ctx->mc;
    Vol v = make_vol(m.volume, m.dtype, m.n0, m.n1, m.n2, m.level, m.zoff);
    const McArena a = carve(v, (char *)ctx->mc_arena.p, m.list_cap);
    return a.edge_index + (size_t)axis * m.n0 * m.n1 * m.n2 + (size_t)z * m.n1 * m.n2;
}

}  // namespace

extern "C" int pifu_mc_count(pifu_ctx *ctx, const void *volume, int dtype, int n0, int n1, int n2, double level,
                             int64_t *nverts, int64_t *nfaces, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && volume && nverts && nfaces, PIFU_ERR_INVALID, "pifu_mc_count: null argument");
    PIFU_REQUIRE(dtype == 0 || dtype == 1, PIFU_ERR_INVALID, "pifu_mc_count: dtype must be 0 (f32) or 1 (f64)");
    PIFU_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, PIFU_ERR_INVALID, "Input array must be at least 2x2x2.");
    PIFU_USE(ctx, stream);
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
    PIFU_USE(ctx, stream);
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
    PIFU_USE(ctx, stream);
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
    PIFU_USE(ctx, stream);
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
    PIFU_USE(ctx, stream);
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
    PIFU_USE(ctx, stream);
    if (n == 0) return PIFU_OK;
    PIFU_HIP(hipSetDevice(ctx->device));
    Mat34 t;
    for (int k = 0; k < 12; ++k) t.m[k] = mat[k];
    hipLaunchKernelGGL(k_transform_verts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, n, t, out);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}
