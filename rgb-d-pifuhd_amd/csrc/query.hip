// Occupancy query for gfx950: projection + bilinear feature sampling + SurfaceClassifier MLPs.
//
// Replaces PIFuMRNet.query / PIFuNetwNML.query (PIFuMRNet.py:119-186, PIFuNetwNML.py:99-141),
// index()/orthogonal() (BasePIFuNet.py:11-43), DepthNormalizer (DepthNormalizer.py:17-24) and
// MLP.forward (MLP.py:43-75).
//
// Data layout (all fp32, point-major so that the K axis of every GEMM is contiguous):
//   featc [rows][K0c]   coarse MLP input  = [image features | z | 0-pad]      (K0c = roundup(257,32))
//   finef [rows][32]    fine-level sampled features (16) zero-padded to one K step
//   Y_l   [rows][Cout]  RAW (pre-norm) conv output of every MLP layer; GroupNorm + LeakyReLU are applied
//                       by the CONSUMER while it stages the tile (x*scale[chunk][c] + shift[chunk][c]),
//                       because the statistics span the whole chunk (SURVEY C-1) and are only known
//                       after the producing GEMM has finished.
//   rows: every reference-side query() call ("chunk") is padded to a multiple of 128 rows, so a GEMM
//   row tile never straddles two chunks; padded rows carry zero features and are masked out of the
//   statistics.
//
// The dominant kernel is k_mlp_gemm_tri (its predecessors k_mlp_gemm_buf / k_mlp_gemm live in tools/lab): exact-fp32
// MFMA (v_mfma_f32_32x32x2_f32), 128x128x32 block tile, 4 waves each owning a 64x64 output tile (2x2 MFMA tiles, 64
// accumulator VGPRs), register-staged LDS (one 36 KB buffer, three blocks per CU) with the norm+LeakyReLU prologue
// fused into the staging pass, per-column (sum, sum^2) partials for the next norm fused into the epilogue
// (deterministic: no atomics).
#include "pifu_internal.h"

#include <math.h>

#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = PIFU_BM, BN = PIFU_BN, BK = PIFU_BK;

struct Seg {
    const float *ptr;     // [rows][ld]
    const float *scale;   // [nchunks][sld]; modes 0 and 2 point at a constant ones / zeros row (sld 0)
    const float *shift;
    int ld;
    int width;            // multiple of BK
    int sld;              // stride between chunks of scale/shift (0 = shared by all chunks)
    int mode;             // 0 raw, 1 affine + LeakyReLU(0.01), 2 LeakyReLU only
};

struct ChunkGeom {
    int tpc;       // row tiles per chunk (P = tpc * BM rows per chunk)
    int chunk;     // points in a full chunk
    int last_len;  // points in the last chunk of this batch
    int nchunks;   // chunks in this batch
};

struct GemmArgs {
    Seg seg[3];
    int nseg;
    const float *W;   // [cout_pad][ldw]
    const __bf16 *Whi, *Wlo;   // split-precision planes of W (k_mlp_gemm_bf3), same shape
    int ldw;          // = padded K
    const float *bias;
    float *Y;         // [rows][ldy]
    int ldy;
    float2 *partial;  // [ntm][ldy] (sum, sum of squares) over the valid rows of the tile, or null
    int ntm, ntn;
    int ktot;         // K of this launch (a multiple of 32): the sum of the segment widths; <= ldw (set by run_gemm)
    ChunkGeom g;
#ifdef PIFU_LAB
    unsigned long long *stamps;   // diagnostic kernels only: [64 blocks][40 steps][4] s_memtime stamps
    int *tile_ctr;   // k_mlp_gemm_dyn: 8 work-queue counters (one per XCD label), zero at launch
    int dbg;   // EXPERIMENTS builds: bit0 skip Y stores, bit1 skip statistics, bit2 skip column init, bit3 skip MFMAs
#endif
    // column sharing (lattice sources): the accumulators start from the per-(i,j)-column product of the skip
    // input's image-feature part, Q[colidx[row]][qoff + c] + wz[c] * zf[row], instead of zero
    const float *colQ;    // [ncols][ldq] or null
    int ldq, qoff;
    const float *colwz;   // [cout_pad] weight of the depth feature, or null
    const float *zf;      // [rows] depth feature per point row
    const int *colidx;    // [rows] column of the point row
    int qrel;             // pixel tables: colQ is a per-ROW table [rows][ldq] (row r starts from colQ[r]); colidx is not read
    // k_mlp_gemm_tri<., ., 1> (an MLP without norm layers, the 128-wide layer in front of the Cout = 1 layer): the epilogue applies
    // LeakyReLU, dots each row with the last layer's weights and writes sigmoid(dot + bias) * mask - Y is neither stored nor read back
    const float *last_w;        // [128]
    float last_bias;
    const uint8_t *last_mask;   // per row (k_last_layer's), with last_mask_bit = 1, 2 or 0 (no mask)
    int last_mask_bit;
    float *last_out;            // [points of this batch], indexed chunk * g.chunk + local
    // the last layer's skip input under column sharing / pixel tables (coarse level): logit += last_colQ[col or row][last_qoff] + last_wz0 * zf
    const float *last_colQ;     // or null
    int last_ldq, last_qoff, last_qrel;
    float last_wz0;
    const float *last_zf;
    const int *last_colidx;
    // per-K-step operand descriptors (k_mlp_gemm_split): everything that depends on the segment is resolved on the
    // host, so the kernel fetches one descriptor per step with scalar loads instead of a ~60-instruction select chain
    struct StepDesc {
        const float *ptr;     // segment base + k offset inside the segment (row 0)
        const float *scale;   // + k offset (chunk 0)
        const float *shift;
        int ld4;              // row stride in BYTES
        int sld;              // chunk stride of scale/shift in elements
        int mode;             // 0 raw, 1 affine + LeakyReLU, 2 LeakyReLU
        int pad;
    } steps[40];
    int nsteps;
};

#if defined(PIFU_LAB) && defined(PIFU_GEMM_EXPERIMENTS)
#define PIFU_DBG(a_) ((a_).dbg)
#else
#define PIFU_DBG(a_) 0
#endif

__device__ __forceinline__ int chunk_len(const ChunkGeom &g, int c) { return c == g.nchunks - 1 ? g.last_len : g.chunk; }

__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }

// ------------------------------------------------------------------------------------------------
// Buffer addressing.  On gfx950 the fp32 MFMA shares the vector ALUs, so every VALU instruction of the staging path, the
// column-init gather and the epilogue is matrix time lost.  All global accesses of the GEMM are
// `buffer_load/store ... s[rsrc], soffset offen`: a uniform (SGPR) tile descriptor + a per-lane 32-bit byte offset + a
// scalar offset, so no 64-bit vector address arithmetic is left; the norm + LeakyReLU transform runs on packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32), and the operand segment of a K step comes from a cursor that is re-resolved only when it
// enters the next segment.
// ------------------------------------------------------------------------------------------------
#ifndef PIFU_STORE_AUX
#define PIFU_STORE_AUX 2
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    // aux 2 = nt: the layer outputs are streamed (GBs per launch, next read by another kernel), no point in L2 residency
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, PIFU_STORE_AUX);
}

// Staging / epilogue macros shared by k_mlp_gemm_tri and (lab builds) its predecessors.  They expect the locals of the
// kernel body: rW, voW, ldw4, kb, sr, kq, row0_ld, chunk_id, the segment cursor c_*, ra0-3 / rw0-3 / rsc / rsh / rslope.
#define BUF_GLOAD(step_)                                                                                   \
    {                                                                                                      \
        rslope = c_slope;                                                                                  \
        ra0 = buf_load4(c_rA, voA, c_ks4);                                                                 \
        ra1 = buf_load4(c_rA, voA, c_ks4 + 32u * c_ld4);                                                   \
        ra2 = buf_load4(c_rA, voA, c_ks4 + 64u * c_ld4);                                                   \
        ra3 = buf_load4(c_rA, voA, c_ks4 + 96u * c_ld4);                                                   \
        rsc = buf_load4(c_rsc, kb, c_ks4);                                                                 \
        rsh = buf_load4(c_rsh, kb, c_ks4);                                                                 \
        rw0 = buf_load4(rW, voW, c_kw4);                                                                   \
        rw1 = buf_load4(rW, voW, c_kw4 + 32u * ldw4);                                                      \
        rw2 = buf_load4(rW, voW, c_kw4 + 64u * ldw4);                                                      \
        rw3 = buf_load4(rW, voW, c_kw4 + 96u * ldw4);                                                      \
        c_ks4 += TBK * 4u;                                                                                 \
        c_kw4 += TBK * 4u;                                                                                 \
        if (--c_left == 0 && ++c_seg < a.nseg) {                                                           \
            const float *np_ = c_seg == 1 ? a.seg[1].ptr : a.seg[2].ptr;                                   \
            const float *nsc_ = c_seg == 1 ? a.seg[1].scale : a.seg[2].scale;                              \
            const float *nsh_ = c_seg == 1 ? a.seg[1].shift : a.seg[2].shift;                              \
            const int nld_ = c_seg == 1 ? a.seg[1].ld : a.seg[2].ld;                                       \
            const int nsld_ = c_seg == 1 ? a.seg[1].sld : a.seg[2].sld;                                    \
            const int nmd_ = c_seg == 1 ? a.seg[1].mode : a.seg[2].mode;                                   \
            c_left = (c_seg == 1 ? a.seg[1].width : a.seg[2].width) / TBK;                                 \
            c_ks4 = 0;                                                                                     \
            c_ld4 = (unsigned)nld_ * 4u;                                                                   \
            c_rA = buf_rsrc(np_ + row0_ld * (size_t)nld_);                                                 \
            c_rsc = buf_rsrc(nsc_ + (size_t)chunk_id * nsld_);                                             \
            c_rsh = buf_rsrc(nsh_ + (size_t)chunk_id * nsld_);                                             \
            c_slope = nmd_ == 0 ? 1.f : 0.01f;                                                             \
            voA = (unsigned)sr * c_ld4 + kb;                                                               \
        }                                                                                                  \
    }
    // norm affine + LeakyReLU on packed pairs: t = v * sc + sh;  v = max(t, slope * t)
#define BUF_XF(va_)                                                                                        \
    {                                                                                                      \
        f32x2 lo_ = {va_.x, va_.y}, hi_ = {va_.z, va_.w};                                                  \
        const f32x2 sl_ = {rsc.x, rsc.y}, sh_ = {rsc.z, rsc.w}, bl_ = {rsh.x, rsh.y}, bh_ = {rsh.z, rsh.w}; \
        const f32x2 k_ = {rslope, rslope};                                                                 \
        lo_ = __builtin_elementwise_fma(lo_, sl_, bl_);                                                    \
        hi_ = __builtin_elementwise_fma(hi_, sh_, bh_);                                                    \
        const f32x2 ml_ = lo_ * k_, mh_ = hi_ * k_;                                                        \
        va_.x = fmaxf(lo_.x, ml_.x); va_.y = fmaxf(lo_.y, ml_.y);                                          \
        va_.z = fmaxf(hi_.x, mh_.x); va_.w = fmaxf(hi_.y, mh_.y);                                          \
    }
#define BUF_LSTORE(buf_)                                                                                   \
    {                                                                                                      \
        float *la = lds + (buf_) * 2 * TILE + sr * LD + kq * 4;                                            \
        BUF_XF(ra0) BUF_XF(ra1) BUF_XF(ra2) BUF_XF(ra3)                                                    \
        *reinterpret_cast<float4 *>(la) = ra0;                                                             \
        *reinterpret_cast<float4 *>(la + 32 * LD) = ra1;                                                   \
        *reinterpret_cast<float4 *>(la + 64 * LD) = ra2;                                                   \
        *reinterpret_cast<float4 *>(la + 96 * LD) = ra3;                                                   \
        *reinterpret_cast<float4 *>(la + TILE) = rw0;                                                      \
        *reinterpret_cast<float4 *>(la + TILE + 32 * LD) = rw1;                                            \
        *reinterpret_cast<float4 *>(la + TILE + 64 * LD) = rw2;                                            \
        *reinterpret_cast<float4 *>(la + TILE + 96 * LD) = rw3;                                            \
    }
#define BUF_EPI(acc_, mt_, nt_, bias_, s_, q_, CHECK_)                                                     \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                       \
        const int rr = (mt_) * 32 + (r & 3) + 8 * (r >> 2);                                                \
        const float v = acc_[r] + bias_;                                                                   \
        if (!(PIFU_DBG(a) & 1)) buf_store1(v, rY, voY, (unsigned)rr * ldy4 + (nt_) * 128u);                \
        if (!(CHECK_) || wm * 64 + rr + 4 * h < nvalid) { s_ += v; q_ = fmaf(v, v, q_); }                  \
    }

// ------------------------------------------------------------------------------------------------
// k_mlp_gemm_tri: k_mlp_gemm_buf with a SINGLE LDS buffer (36 KB) and a register budget of 168, so that THREE blocks fit a
// CU (three waves per SIMD).  Stamps of the two-block kernels show that a second resident block only adds 8 % to what one
// block delivers alone: with two waves per SIMD the matrix pipe idles whenever both are outside their MFMA phase.  A
// K step costs one more barrier here (the tile is overwritten in place once every wave has read its fragments), which the
// two other resident blocks cover.  Same tiles, k order and statistics as k_mlp_gemm_buf: bit-identical results.
// ------------------------------------------------------------------------------------------------
template <int PRIO, int QREL = 0, int FUSE = 0>      // FUSE: 0 bias + store + statistics, 1 the Cout = 1 layer in the epilogue, 2 bias + store only
__global__ __launch_bounds__(256, 3) void k_mlp_gemm_tri(const GemmArgs a)
{
    constexpr int TBK = 32, LD = TBK + 4, TILE = BM * LD, NS = TBK / 8;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];   // [A|B], single buffer: 36 KB, three blocks per CU

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int q_tile = blockIdx.x;
    const int xcd = q_tile & 7, slot = q_tile >> 3;
    const int tile_n = slot % a.ntn;
    const int tile_m = (slot / a.ntn) * 8 + xcd;
    if (tile_m >= a.ntm) return;
    const int chunk_id = tile_m / a.g.tpc;
    const size_t row0 = (size_t)tile_m * BM, row0_ld = row0;
    const int col0 = tile_n * BN;

    const int kq = tid & 7, sr = tid >> 3;        // staging: float4 column kq of rows sr + 32 p
    const unsigned kb = kq * 16;
    const int nsteps = a.ktot / TBK;
    const unsigned ldw4 = (unsigned)a.ldw * 4u;
    const __amdgpu_buffer_rsrc_t rW = buf_rsrc(a.W + (size_t)col0 * a.ldw);
    const unsigned voW = (unsigned)sr * ldw4 + kb;

    float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3, rsc, rsh;
    float rslope = 1.f;

    // cursor of the next K step to fetch: every per-segment quantity is resolved when the cursor ENTERS a segment (a rare
    // uniform branch), so a step costs a handful of scalar adds instead of a ~70-instruction select chain
    int c_seg = 0, c_left = a.seg[0].width / TBK;
    unsigned c_ks4 = 0, c_kw4 = 0;
    unsigned c_ld4 = (unsigned)a.seg[0].ld * 4u;
    __amdgpu_buffer_rsrc_t c_rA = buf_rsrc(a.seg[0].ptr + row0 * (size_t)a.seg[0].ld);
    __amdgpu_buffer_rsrc_t c_rsc = buf_rsrc(a.seg[0].scale + (size_t)chunk_id * a.seg[0].sld);
    __amdgpu_buffer_rsrc_t c_rsh = buf_rsrc(a.seg[0].shift + (size_t)chunk_id * a.seg[0].sld);
    float c_slope = a.seg[0].mode == 0 ? 1.f : 0.01f;
    unsigned voA = (unsigned)sr * c_ld4 + kb;
    // norm affine + LeakyReLU on packed pairs: t = v * sc + sh;  v = max(t, slope * t)

    const int h = lane >> 5, j = lane & 31;
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    BUF_GLOAD(0)
    if (a.colQ && a.ldq == 0) {
        // per-point path: the depth feature is the one input column after the image features; it enters as the rank-1 start
        // value wz * z instead of a 32-wide K step of its own (the table is a single row of zeros: the same fmaf as below)
        const __amdgpu_buffer_rsrc_t rZ = buf_rsrc(a.zf + row0);
        const unsigned vrow = (unsigned)(wm * 64 + 4 * h) * 4u;
        const float wz0 = a.colwz[col0 + wn * 64 + j], wz1 = a.colwz[col0 + wn * 64 + 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned so = (unsigned)((r & 3) + 8 * (r >> 2)) * 4u;
            const float zA = buf_load1(rZ, vrow, so), zB = buf_load1(rZ, vrow, so + 128u);
            acc00[r] = fmaf(wz0, zA, 0.f);
            acc01[r] = fmaf(wz1, zA, 0.f);
            acc10[r] = fmaf(wz0, zB, 0.f);
            acc11[r] = fmaf(wz1, zB, 0.f);
        }
    } else if (QREL) {     // pixel tables: the start values of row r are row r of a per-row table (no index to fetch)
        const __amdgpu_buffer_rsrc_t rZ = buf_rsrc(a.zf + row0);
        const __amdgpu_buffer_rsrc_t rQ = buf_rsrc(a.colQ + row0 * (size_t)a.ldq + a.qoff + col0);
        const unsigned vrow = (unsigned)(wm * 64 + 4 * h) * 4u, ldq4 = (unsigned)a.ldq * 4u;
        const unsigned vq = (unsigned)(wm * 64 + 4 * h) * ldq4 + (unsigned)(wn * 64 + j) * 4u;
        const float wz0 = a.colwz ? a.colwz[col0 + wn * 64 + j] : 0.f, wz1 = a.colwz ? a.colwz[col0 + wn * 64 + 32 + j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned lr = (unsigned)((r & 3) + 8 * (r >> 2));
            const float zA = buf_load1(rZ, vrow, lr * 4u), zB = buf_load1(rZ, vrow, lr * 4u + 128u);
            acc00[r] = fmaf(wz0, zA, buf_load1(rQ, vq, lr * ldq4));
            acc01[r] = fmaf(wz1, zA, buf_load1(rQ, vq, lr * ldq4 + 128u));
            acc10[r] = fmaf(wz0, zB, buf_load1(rQ, vq, (lr + 32u) * ldq4));
            acc11[r] = fmaf(wz1, zB, buf_load1(rQ, vq, (lr + 32u) * ldq4 + 128u));
        }
    } else if (a.colQ) {   // column sharing: accumulators start from Q[col] + wz * z (gathered while the first tile is in flight)
        const __amdgpu_buffer_rsrc_t rI = buf_rsrc(a.colidx + row0), rZ = buf_rsrc(a.zf + row0);
        const __amdgpu_buffer_rsrc_t rQ = buf_rsrc(a.colQ + a.qoff + col0);     // descriptors stay block-uniform (SGPRs)
        const unsigned vrow = (unsigned)(wm * 64 + 4 * h) * 4u, ldq4 = (unsigned)a.ldq * 4u;
        const float wz0 = a.colwz ? a.colwz[col0 + wn * 64 + j] : 0.f, wz1 = a.colwz ? a.colwz[col0 + wn * 64 + 32 + j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned so = (unsigned)((r & 3) + 8 * (r >> 2)) * 4u;
            const unsigned ciA = __builtin_amdgcn_raw_buffer_load_b32(rI, vrow, so, 0);
            const unsigned ciB = __builtin_amdgcn_raw_buffer_load_b32(rI, vrow, so + 128u, 0);
            const float zA = buf_load1(rZ, vrow, so), zB = buf_load1(rZ, vrow, so + 128u);
            const unsigned qA = ciA * ldq4 + (unsigned)(wn * 64 + j) * 4u, qB = ciB * ldq4 + (unsigned)(wn * 64 + j) * 4u;
            acc00[r] = fmaf(wz0, zA, buf_load1(rQ, qA, 0));
            acc01[r] = fmaf(wz1, zA, buf_load1(rQ, qA, 128));
            acc10[r] = fmaf(wz0, zB, buf_load1(rQ, qB, 0));
            acc11[r] = fmaf(wz1, zB, buf_load1(rQ, qB, 128));
        }
    }
    BUF_LSTORE(0)
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const bool more = step + 1 < nsteps;
        if (more) BUF_GLOAD(step + 1)
        const float *A = lds + (wm * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
        const float *B = lds + TILE + (wn * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 a0 = *reinterpret_cast<const float4 *>(A + s * 8);
            const float4 a1 = *reinterpret_cast<const float4 *>(A + 32 * LD + s * 8);
            const float4 b0 = *reinterpret_cast<const float4 *>(B + s * 8);
            const float4 b1v = *reinterpret_cast<const float4 *>(B + 32 * LD + s * 8);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv0[t], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv1[t], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv0[t], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv1[t], acc11, 0, 0, 0);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __syncthreads();                 // every wave has read its fragments: the tile may be overwritten
        if (more) BUF_LSTORE(0)
        __syncthreads();
    }

    // ---- epilogue: bias, raw store, per-column statistics over the tile's valid rows
    const int local0 = (tile_m % a.g.tpc) * BM;
    int nvalid = chunk_len(a.g, chunk_id) - local0;
    nvalid = nvalid < 0 ? 0 : (nvalid > BM ? BM : nvalid);
    float *red = lds;   // [2 (sum|sq)][2 (wm)][BN]; safe: every wave passed the loop's final barrier
    if (FUSE == 1) {
        // the Cout = 1 layer in this layer's epilogue (pointwise norm: nothing but LeakyReLU lies between them, and the 128
        // columns of a row are all in this block): per lane the products of its two columns, per row 64 partial sums through
        // LDS ([128 rows][65]: conflict-free), summed in a fixed order by one thread per row
        constexpr int RS = 65;
        const float b0 = a.bias[wn * 64 + j], b1 = a.bias[wn * 64 + 32 + j];
        const float w0 = a.last_w[wn * 64 + j], w1 = a.last_w[wn * 64 + 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            red[(wm * 64 + rr) * RS + wn * 32 + j] = fmaf(leaky(acc00[r] + b0), w0, leaky(acc01[r] + b1) * w1);
            red[(wm * 64 + 32 + rr) * RS + wn * 32 + j] = fmaf(leaky(acc10[r] + b0), w0, leaky(acc11[r] + b1) * w1);
        }
        __syncthreads();
        if (tid < BM && tid < nvalid) {
            float dot = 0.f;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) dot += red[tid * RS + k];
            if (a.last_colQ) {
                const size_t qrow = a.last_qrel ? row0 + tid : (size_t)a.last_colidx[row0 + tid];
                dot += fmaf(a.last_wz0, a.last_zf[row0 + tid], a.last_colQ[qrow * a.last_ldq + a.last_qoff]);
            }
            float p = 1.f / (1.f + expf(-(dot + a.last_bias)));
            if (a.last_mask_bit && !(a.last_mask[row0 + tid] & a.last_mask_bit)) p = 0.f;
            a.last_out[(int64_t)chunk_id * a.g.chunk + local0 + tid] = p;
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rY = buf_rsrc(a.Y + row0 * (size_t)a.ldy + col0);
    const unsigned ldy4 = (unsigned)a.ldy * 4u;
    const unsigned voY = (unsigned)(wm * 64 + 4 * h) * ldy4 + (unsigned)(wn * 64 + j) * 4u;
    const float bias0 = a.bias[col0 + wn * 64 + j], bias1 = a.bias[col0 + wn * 64 + 32 + j];
    if (FUSE == 2) {      // no statistics wanted (pointwise norm, table / start-value products): bias + raw store, nothing else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ro = (unsigned)((r & 3) + 8 * (r >> 2)) * ldy4;
            buf_store1(acc00[r] + bias0, rY, voY, ro);
            buf_store1(acc10[r] + bias0, rY, voY, ro + 32u * ldy4);
            buf_store1(acc01[r] + bias1, rY, voY, ro + 128u);
            buf_store1(acc11[r] + bias1, rY, voY, ro + 32u * ldy4 + 128u);
        }
        return;
    }
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    if (nvalid >= BM) {
        BUF_EPI(acc00, 0, 0, bias0, s0, q0, false) BUF_EPI(acc10, 1, 0, bias0, s0, q0, false)
        BUF_EPI(acc01, 0, 1, bias1, s1, q1, false) BUF_EPI(acc11, 1, 1, bias1, s1, q1, false)
    } else {
        BUF_EPI(acc00, 0, 0, bias0, s0, q0, true) BUF_EPI(acc10, 1, 0, bias0, s0, q0, true)
        BUF_EPI(acc01, 0, 1, bias1, s1, q1, true) BUF_EPI(acc11, 1, 1, bias1, s1, q1, true)
    }
    if (a.partial) {
        s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32);
        s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
        if (lane < 32) {
            red[(0 * 2 + wm) * BN + wn * 64 + j] = s0;
            red[(1 * 2 + wm) * BN + wn * 64 + j] = q0;
            red[(0 * 2 + wm) * BN + wn * 64 + 32 + j] = s1;
            red[(1 * 2 + wm) * BN + wn * 64 + 32 + j] = q1;
        }
        __syncthreads();
        if (tid < BN)
            a.partial[(size_t)tile_m * a.ldy + col0 + tid] =
                make_float2(red[tid] + red[BN + tid], red[2 * BN + tid] + red[3 * BN + tid]);
    }
}

#ifdef PIFU_LAB
// lab builds only (tools/lab): the predecessors of k_mlp_gemm_tri (bit-identity A/B) and the rejected schedules
#include "gemm_predecessors.inc"
#include "gemm_dma.inc"
#include "gemm_wide32.inc"
#include "gemm_adma.inc"
#ifdef PIFU_GEMM_EXPERIMENTS
#include "gemm_experiments.inc"
#endif
#endif

// ------------------------------------------------------------------------------------------------
// k_mlp_gemm_split<0> (bf16 x 3) / <1> (fp16 x 3): opt-in split-precision versions of the MLP GEMM on the 16-bit matrix cores.
// Every fp32 operand x is split as x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in fp32), and
// the product is accumulated in fp32 as  A_hi*W_hi + A_lo*W_hi + A_hi*W_lo  (the lo*lo term, 2^-16 relative, is
// dropped).  3 x v_mfma_f32_32x32x16_bf16 (32 cycles each) replace 8 x v_mfma_f32_32x32x2_f32 (64 cycles each) per
// 32x32x16 block: 5.3x fewer matrix cycles.  Measured effect on the occupancy: ~1e-5 (SURVEY section 7 probe: 1.9e-6 ..
// 1.2e-5), inside the 1e-4 contract; the exact-fp32 kernel stays the default.  Same tiles, same fused prologue and
// epilogue, same K order per product.  W is split once per layer (k_split_bf16); A is split while it is staged.
// LDS (per buffer): A_hi, A_lo, W_hi, W_lo as [128 rows][32 k] bf16 with an 80-byte row stride (conflict-free
// ds_read_b128: lane (r, h) reads the 8 k values 8h..8h+7 of row r = one MFMA operand).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// F16 = 0: bf16 x 3 (above).  F16 = 1: fp16 x 3 - same three products on v_mfma_f32_32x32x16_f16, whose operands carry 11
// significant bits instead of 8: x = hi + lo is then good to 2^-22 and the dropped lo*lo term is 2^-22 relative, i.e. the
// result is within a few fp32 ulps of the exact-fp32 kernel's (measured ~1e-6 on the occupancy).  fp16 has 5 exponent bits,
// so the weight planes are stored scaled by 2^11 (hi' = fp16(2048 w), lo' = fp16(2048 w - hi'): both in the normal range for
// |w| < 32; checked when the planes are built), the activation planes unscaled (lo = fp16(a - hi) becomes subnormal for
// |a| < 2^-3 and is then quantised to 2^-25 absolute - below fp32's own rounding of O(1) data; |a| must stay below 65 504),
// all three products share the scale 2^11, the accumulators start from 2^11 x the column-init value and the epilogue
// multiplies by 2^-11 (exact) before the bias.
template <int F16> struct SplitT;
template <> struct SplitT<0> {
    typedef __bf16 T; typedef bf16x8 V8; typedef bf16x4 V4;
    static __device__ __forceinline__ f32x16 mfma(V8 x, V8 y, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); }
};
template <> struct SplitT<1> {
    typedef _Float16 T; typedef f16x8 V8; typedef f16x4 V4;
    static __device__ __forceinline__ f32x16 mfma(V8 x, V8 y, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); }
};

template <int F16>
__global__ void k_split_planes(const float *W, void *hi_, void *lo_, size_t n, int *bad)
{
    typedef typename SplitT<F16>::T T;
    T *hi = (T *)hi_, *lo = (T *)lo_;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = W[i] * (F16 ? 2048.f : 1.f);
    if (F16 && !(fabsf(x) < 65504.f)) *bad = 1;       // |w| >= 32 (or NaN): outside the scaled fp16 range
    const T h = (T)x;
    hi[i] = h;
    lo[i] = (T)(x - (float)h);
}

template <int F16, int QREL = 0>      // QREL: GemmArgs::qrel as a compile-time constant (a run-time select re-shaped the whole kernel: -10 %)
__global__ __launch_bounds__(256, 2) void k_mlp_gemm_split(const GemmArgs a)
{
    typedef typename SplitT<F16>::T T;
    typedef typename SplitT<F16>::V8 V8;
    typedef typename SplitT<F16>::V4 V4;
    constexpr float WS = F16 ? 2048.f : 1.f, WSI = F16 ? (1.f / 2048.f) : 1.f;
    constexpr int TBK = 32;
    constexpr int RS = 32;                       // row stride in bf16 elements (64 B); 16-byte chunks XOR-swizzled by (row >> 2) & 3
    constexpr int PLANE = BM * RS;               // one [128][32] bf16 plane
    __shared__ __attribute__((aligned(16))) T lds[2 * 4 * PLANE];   // [buf][A_hi | A_lo | W_hi | W_lo] = 65,536 B

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile_n = slot % a.ntn;
    const int tile_m = (slot / a.ntn) * 8 + xcd;
    if (tile_m >= a.ntm) return;
    const int chunk_id = tile_m / a.g.tpc;
    const size_t row0 = (size_t)tile_m * BM;
    const int col0 = tile_n * BN;
    const int kq = tid & 7, sr = tid >> 3;       // A staging: float4 column kq of rows sr + 32 p
    const int wq = tid & 3, wr = tid >> 2;       // W staging: 8 bf16 (16 B) column wq of rows wr + 64 p
    const int b1 = a.seg[0].width;
    const int b2 = b1 + (a.nseg > 1 ? a.seg[1].width : 0);
    const int nsteps = a.ktot / TBK;

    // Operands are prefetched TWO K steps ahead into two register sets (step j lives in set j & 1): with the short
    // bf16 MFMA phase a one-step-ahead prefetch would stall on the global-load latency at every staging pass.
    struct Stage { float4 a0, a1, a2, a3, sc, sh; V8 wh0, wh1, wl0, wl1; float slope; };
    Stage S0, S1;
    S0.slope = S1.slope = 1.f;

    // lean operand fetch: per-step descriptors come from the kernarg table (scalar loads, prefetched one call ahead),
    // addresses are uniform base + 32-bit lane offset
    const unsigned r0 = tile_m * BM + sr, r1 = r0 + 32, r2 = r0 + 64, r3 = r0 + 96;
    const unsigned kb4 = kq * 16;
    const unsigned ldw2 = a.ldw * 2;                                       // bytes per row of a bf16 plane
    const unsigned wo0 = (col0 + wr) * ldw2 + wq * 16, wo1 = wo0 + 64 * ldw2;
    const bool use_table = a.nsteps > 0;
    const float *d_ptr = a.steps[0].ptr, *d_scale = a.steps[0].scale, *d_shift = a.steps[0].shift;
    unsigned d_ld4 = a.steps[0].ld4;
    int d_sld = a.steps[0].sld, d_mode = a.steps[0].mode, d_next = 1;
#define BF3_GLOAD(S_, step_)                                                                               \
    {                                                                                                      \
    if (use_table) {                                                                                       \
        const char *ab = reinterpret_cast<const char *>(d_ptr);                                            \
        S_.slope = d_mode == 0 ? 1.f : 0.01f;                                                              \
        S_.a0 = *reinterpret_cast<const float4 *>(ab + (__umul24(r0, d_ld4) + kb4));                       \
        S_.a1 = *reinterpret_cast<const float4 *>(ab + (__umul24(r1, d_ld4) + kb4));                       \
        S_.a2 = *reinterpret_cast<const float4 *>(ab + (__umul24(r2, d_ld4) + kb4));                       \
        S_.a3 = *reinterpret_cast<const float4 *>(ab + (__umul24(r3, d_ld4) + kb4));                       \
        const char *sb = reinterpret_cast<const char *>(d_scale + (size_t)chunk_id * d_sld);               \
        const char *hb = reinterpret_cast<const char *>(d_shift + (size_t)chunk_id * d_sld);               \
        S_.sc = *reinterpret_cast<const float4 *>(sb + kb4);                                               \
        S_.sh = *reinterpret_cast<const float4 *>(hb + kb4);                                               \
        const char *whb = reinterpret_cast<const char *>((const T *)a.Whi + (step_) * TBK);                           \
        const char *wlb = reinterpret_cast<const char *>((const T *)a.Wlo + (step_) * TBK);                           \
        S_.wh0 = *reinterpret_cast<const V8 *>(whb + wo0);                                             \
        S_.wh1 = *reinterpret_cast<const V8 *>(whb + wo1);                                             \
        S_.wl0 = *reinterpret_cast<const V8 *>(wlb + wo0);                                             \
        S_.wl1 = *reinterpret_cast<const V8 *>(wlb + wo1);                                             \
        const int nx_ = d_next < nsteps ? d_next : nsteps - 1;                                             \
        const GemmArgs::StepDesc &dn = a.steps[nx_];                                                       \
        d_ptr = dn.ptr; d_scale = dn.scale; d_shift = dn.shift; d_ld4 = dn.ld4; d_sld = dn.sld; d_mode = dn.mode; \
        ++d_next;                                                                                          \
    } else {                                                                                               \
        const int kk = (step_) * TBK;                                                                      \
        const int sg = (kk >= b1) + (kk >= b2);                                                            \
        const float *ptr = sg == 0 ? a.seg[0].ptr : (sg == 1 ? a.seg[1].ptr : a.seg[2].ptr);               \
        const float *sc = sg == 0 ? a.seg[0].scale : (sg == 1 ? a.seg[1].scale : a.seg[2].scale);          \
        const float *sh = sg == 0 ? a.seg[0].shift : (sg == 1 ? a.seg[1].shift : a.seg[2].shift);          \
        const int ld = sg == 0 ? a.seg[0].ld : (sg == 1 ? a.seg[1].ld : a.seg[2].ld);                      \
        const int sld = sg == 0 ? a.seg[0].sld : (sg == 1 ? a.seg[1].sld : a.seg[2].sld);                  \
        const int md = sg == 0 ? a.seg[0].mode : (sg == 1 ? a.seg[1].mode : a.seg[2].mode);                \
        S_.slope = md == 0 ? 1.f : 0.01f;                                                                  \
        const int kl = kk - (sg == 0 ? 0 : (sg == 1 ? b1 : b2)) + kq * 4;                                  \
        const float *ap = ptr + (row0 + sr) * (size_t)ld + kl;                                             \
        S_.a0 = *reinterpret_cast<const float4 *>(ap);                                                     \
        S_.a1 = *reinterpret_cast<const float4 *>(ap + (size_t)32 * ld);                                   \
        S_.a2 = *reinterpret_cast<const float4 *>(ap + (size_t)64 * ld);                                   \
        S_.a3 = *reinterpret_cast<const float4 *>(ap + (size_t)96 * ld);                                   \
        S_.sc = *reinterpret_cast<const float4 *>(sc + (size_t)chunk_id * sld + kl);                       \
        S_.sh = *reinterpret_cast<const float4 *>(sh + (size_t)chunk_id * sld + kl);                       \
        const size_t wo = (size_t)(col0 + wr) * a.ldw + kk + wq * 8;                                       \
        S_.wh0 = *reinterpret_cast<const V8 *>((const T *)a.Whi + wo);                                            \
        S_.wh1 = *reinterpret_cast<const V8 *>((const T *)a.Whi + wo + (size_t)64 * a.ldw);                       \
        S_.wl0 = *reinterpret_cast<const V8 *>((const T *)a.Wlo + wo);                                            \
        S_.wl1 = *reinterpret_cast<const V8 *>((const T *)a.Wlo + wo + (size_t)64 * a.ldw);                       \
    }                                                                                                      \
    }
    // transform, split into hi / lo and write one float4 (4 k) of A
#define BF3_STA(S_, base_, p_, va_)                                                                        \
    {                                                                                                      \
        const float4 v_ = S_.va_;                                                                          \
        float t0_ = fmaf(v_.x, S_.sc.x, S_.sh.x), t1_ = fmaf(v_.y, S_.sc.y, S_.sh.y);                      \
        float t2_ = fmaf(v_.z, S_.sc.z, S_.sh.z), t3_ = fmaf(v_.w, S_.sc.w, S_.sh.w);                      \
        t0_ = fmaxf(t0_, S_.slope * t0_); t1_ = fmaxf(t1_, S_.slope * t1_);                                \
        t2_ = fmaxf(t2_, S_.slope * t2_); t3_ = fmaxf(t3_, S_.slope * t3_);                                \
        V4 h_, l_;                                                                                         \
        h_[0] = (T)t0_; h_[1] = (T)t1_; h_[2] = (T)t2_; h_[3] = (T)t3_;                            \
        l_[0] = (T)(t0_ - (float)h_[0]); l_[1] = (T)(t1_ - (float)h_[1]);                                  \
        l_[2] = (T)(t2_ - (float)h_[2]); l_[3] = (T)(t3_ - (float)h_[3]);                                  \
        const int row_ = sr + 32 * (p_);                                                                   \
        T *d_ = (base_) + row_ * RS + (((kq >> 1) ^ ((row_ >> 2) & 3)) << 3) + (kq & 1) * 4;           \
        *reinterpret_cast<V4 *>(d_) = h_;                                                              \
        *reinterpret_cast<V4 *>(d_ + PLANE) = l_;                                                      \
    }
#define BF3_LSTORE(S_, buf_)                                                                               \
    {                                                                                                      \
        T *base = lds + (buf_) * 4 * PLANE;                                                               \
        BF3_STA(S_, base, 0, a0) BF3_STA(S_, base, 1, a1) BF3_STA(S_, base, 2, a2) BF3_STA(S_, base, 3, a3) \
        /* rows wr and wr + 64 have the same (row >> 2) & 3, so one swizzled chunk serves both */          \
        T *wd = base + 2 * PLANE      + wr * RS + ((wq ^ ((wr >> 2) & 3)) << 3);                           \
        *reinterpret_cast<V8 *>(wd) = S_.wh0;                                                          \
        *reinterpret_cast<V8 *>(wd + 64 * RS) = S_.wh1;                                                \
        *reinterpret_cast<V8 *>(wd + PLANE) = S_.wl0;                                                  \
        *reinterpret_cast<V8 *>(wd + PLANE + 64 * RS) = S_.wl1;                                        \
    }

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    if (a.colQ) {
        const int hh = lane >> 5, jj = lane & 31;
        const int c0 = col0 + wn * 64 + jj, c1 = c0 + 32;
        const float wz0 = a.colwz ? a.colwz[c0] : 0.f, wz1 = a.colwz ? a.colwz[c1] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t rowA = row0 + wm * 64 + (r & 3) + 8 * (r >> 2) + 4 * hh, rowB = rowA + 32;
            const size_t ciA = QREL ? rowA : (size_t)a.colidx[rowA], ciB = QREL ? rowB : (size_t)a.colidx[rowB];
            const float zA = a.zf[rowA], zB = a.zf[rowB];
            const float *qA = a.colQ + ciA * a.ldq + a.qoff, *qB = a.colQ + ciB * a.ldq + a.qoff;
            acc00[r] = WS * fmaf(wz0, zA, qA[c0]);
            acc01[r] = WS * fmaf(wz1, zA, qA[c1]);
            acc10[r] = WS * fmaf(wz0, zB, qB[c0]);
            acc11[r] = WS * fmaf(wz1, zB, qB[c1]);
        }
    }

    // lane (r = lane & 31, h = lane >> 5): operand = 8 consecutive k (8h .. 8h+7) of row r inside a 16-wide k block
    const int frow = lane & 31, fsw = (frow >> 2) & 3, fh = lane >> 5;   // rows r and r + 32 / + 64 share the swizzle
    const int fo0 = frow * RS + (((0 + fh) ^ fsw) << 3), fo1 = frow * RS + (((2 + fh) ^ fsw) << 3);
#define BF3_COMPUTE(buf_)                                                                                  \
    {                                                                                                      \
        const T *Tb = lds + (buf_) * 4 * PLANE;                                                           \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                 \
            const int fo = kb == 0 ? fo0 : fo1;                                                            \
            const T *Ah = Tb + (wm * 64) * RS + fo, *Al = Ah + PLANE;                                      \
            const T *Bh = Tb + 2 * PLANE + (wn * 64) * RS + fo, *Bl = Bh + PLANE;                          \
            const V8 ah0 = *reinterpret_cast<const V8 *>(Ah), ah1 = *reinterpret_cast<const V8 *>(Ah + 32 * RS); \
            const V8 al0 = *reinterpret_cast<const V8 *>(Al), al1 = *reinterpret_cast<const V8 *>(Al + 32 * RS); \
            const V8 bh0 = *reinterpret_cast<const V8 *>(Bh), bh1 = *reinterpret_cast<const V8 *>(Bh + 32 * RS); \
            const V8 bl0 = *reinterpret_cast<const V8 *>(Bl), bl1 = *reinterpret_cast<const V8 *>(Bl + 32 * RS); \
            /* small terms first, the dominant hi*hi product last */                                       \
            acc00 = SplitT<F16>::mfma(al0, bh0, acc00);                     \
            acc01 = SplitT<F16>::mfma(al0, bh1, acc01);                     \
            acc10 = SplitT<F16>::mfma(al1, bh0, acc10);                     \
            acc11 = SplitT<F16>::mfma(al1, bh1, acc11);                     \
            acc00 = SplitT<F16>::mfma(ah0, bl0, acc00);                     \
            acc01 = SplitT<F16>::mfma(ah0, bl1, acc01);                     \
            acc10 = SplitT<F16>::mfma(ah1, bl0, acc10);                     \
            acc11 = SplitT<F16>::mfma(ah1, bl1, acc11);                     \
            acc00 = SplitT<F16>::mfma(ah0, bh0, acc00);                     \
            acc01 = SplitT<F16>::mfma(ah0, bh1, acc01);                     \
            acc10 = SplitT<F16>::mfma(ah1, bh0, acc10);                     \
            acc11 = SplitT<F16>::mfma(ah1, bh1, acc11);                     \
        }                                                                                                  \
    }
    // step t: issue the loads of step t+2 (into the set step t used), compute step t, stage step t+1 (other set)
#ifdef PIFU_LAB
#define BF3_STAMP(slot_)                                                                                   \
    if (a.stamps) {                                                                                        \
        unsigned long long ts_;                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (tid == 0 && blockIdx.x < 64 && t < 40) a.stamps[((size_t)blockIdx.x * 40 + t) * 4 + (slot_)] = ts_; \
    }
#else
#define BF3_STAMP(slot_)
#endif
#define BF3_STEP(t_, SC_, SN_)                                                                             \
    {                                                                                                      \
        const int t = (t_);                                                                                \
        BF3_STAMP(0)                                                                                       \
        if (t + 2 < nsteps) BF3_GLOAD(SC_, t + 2)                                                          \
        BF3_STAMP(1)                                                                                       \
        BF3_COMPUTE(t & 1)                                                                                 \
        BF3_STAMP(2)                                                                                       \
        if (t + 1 < nsteps) BF3_LSTORE(SN_, (t + 1) & 1)                                                   \
        BF3_STAMP(3)                                                                                       \
        __syncthreads();                                                                                   \
    }

    BF3_GLOAD(S0, 0)
    if (nsteps > 1) BF3_GLOAD(S1, 1)
    BF3_LSTORE(S0, 0)
    __syncthreads();
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        BF3_STEP(step, S0, S1)
        BF3_STEP(step + 1, S1, S0)
    }
    if (step < nsteps) BF3_STEP(step, S0, S1)

    const int h = lane >> 5, j = lane & 31;
    const int local0 = (tile_m % a.g.tpc) * BM;
    int nvalid = chunk_len(a.g, chunk_id) - local0;
    nvalid = nvalid < 0 ? 0 : (nvalid > BM ? BM : nvalid);
    float *red = reinterpret_cast<float *>(lds);
    (void)WSI;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = col0 + wn * 64 + nt * 32 + j;
        const float bias = a.bias[col];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x16 &acc = nt == 0 ? (mt == 0 ? acc00 : acc10) : (mt == 0 ? acc01 : acc11);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = F16 ? fmaf(acc[r], WSI, bias) : acc[r] + bias;
                a.Y[(row0 + row) * (size_t)a.ldy + col] = v;
                if (row < nvalid) { s += v; q = fmaf(v, v, q); }
            }
        }
        if (a.partial) {
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (lane < 32) {
                red[(0 * 2 + wm) * BN + wn * 64 + nt * 32 + j] = s;
                red[(1 * 2 + wm) * BN + wn * 64 + nt * 32 + j] = q;
            }
        }
    }
    if (a.partial) {
        __syncthreads();
        if (tid < BN)
            a.partial[(size_t)tile_m * a.ldy + col0 + tid] =
                make_float2(red[tid] + red[BN + tid], red[2 * BN + tid] + red[3 * BN + tid]);
    }
}

// ------------------------------------------------------------------------------------------------
// k_norm_finalize: (sum, sum^2) partials -> per-(chunk, channel) scale/shift of the following norm.
// One wave per (chunk, group); double accumulation in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
__global__ void k_norm_finalize(const float2 *partial, int ld, int cpg, int groups, const float *gamma,
                                const float *beta, float *scale, float *shift, ChunkGeom g, float eps, int halves,
                                const int *chunk_row0 = nullptr)
{
    // one 256-thread block per (chunk, group): a 200 000-point calc_normal chunk has 1 563 tiles x 32 channels per group,
    // too many partials for a single wave.  Fixed summation order (strided per thread, then a fixed LDS tree).
    __shared__ double shs[256], shq[256];
    const int wid = blockIdx.x;
    if (wid >= g.nchunks * groups) return;
    const int c = wid / groups, grp = wid % groups;
    const int len = chunk_len(g, c);
    // partial rows of this chunk: its 128-row tiles, or (depth walk) its column segments [chunk_row0[c], chunk_row0[c+1])
    const int prow0 = chunk_row0 ? chunk_row0[c] : c * g.tpc * halves;
    const int ntile = chunk_row0 ? chunk_row0[c + 1] - chunk_row0[c] : ((len + BM - 1) / BM) * halves;
    double s = 0.0, q = 0.0;
    const int total = ntile * cpg;       // (`halves` partial rows per 128-row tile: 2 for k_mlp_gemm_ws)
    {
        // four independent loads in flight per thread (a 200 000-point chunk has 50 000 partials per group, and the grid is
        // only chunks x groups blocks: one dependent load per iteration left the kernel waiting on latency), still a fixed order
        double s1 = 0.0, s2 = 0.0, s3 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
        const float2 *base = partial + (size_t)prow0 * ld + grp * cpg;
        int i = threadIdx.x;
        for (; i + 768 < total; i += 1024) {
            const float2 p0 = base[(size_t)(i / cpg) * ld + i % cpg];
            const float2 p1 = base[(size_t)((i + 256) / cpg) * ld + (i + 256) % cpg];
            const float2 p2 = base[(size_t)((i + 512) / cpg) * ld + (i + 512) % cpg];
            const float2 p3 = base[(size_t)((i + 768) / cpg) * ld + (i + 768) % cpg];
            s += (double)p0.x; q += (double)p0.y; s1 += (double)p1.x; q1 += (double)p1.y;
            s2 += (double)p2.x; q2 += (double)p2.y; s3 += (double)p3.x; q3 += (double)p3.y;
        }
        for (; i < total; i += 256) {
            const float2 p = base[(size_t)(i / cpg) * ld + i % cpg];
            s += (double)p.x;
            q += (double)p.y;
        }
        s = (s + s1) + (s2 + s3);
        q = (q + q1) + (q2 + q3);
    }
    shs[threadIdx.x] = s;
    shq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { shs[threadIdx.x] += shs[threadIdx.x + o]; shq[threadIdx.x] += shq[threadIdx.x + o]; }
        __syncthreads();
    }
    s = shs[0];
    q = shq[0];
    const double cnt = (double)len * cpg;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int i = threadIdx.x; i < cpg; i += 256) {
        const int ch = grp * cpg + i;
        const double gm = gamma[ch];
        scale[(size_t)c * ld + ch] = (float)(gm * rstd);
        shift[(size_t)c * ld + ch] = (float)((double)beta[ch] - mean * rstd * gm);
    }
}

// ------------------------------------------------------------------------------------------------
// k_last_layer: the Cout = 1 layer + sigmoid + bounding-box mask.  8 or 16 lanes per point row.
// ------------------------------------------------------------------------------------------------
struct LastArgs {
    Seg seg[3];
    int nseg;
    const float *w;          // [kpad]
    float bias;
    const uint8_t *mask;     // per row; bit0 = coarse in_bb (x,y,z), bit1 = fine in_bb (x,y)
    int mask_bit;            // 1, 2 or 0 (no mask)
    float *out;              // [points of this batch], indexed c * chunk + local
    int64_t rows;
    ChunkGeom g;
    const float *colQ;       // column sharing: logit += colQ[colidx[row]][qoff] + wz0 * zf[row]
    int ldq, qoff;
    float wz0;
    const float *zf;
    const int *colidx;
    int qrel;                // pixel tables: colQ is a per-row table, colidx is not read
};

// LPR lanes per point row: 8 for the usual 128-wide input (four independent 16-byte loads per lane in flight: the kernel
// only streams the rows once, so what it needs is bytes in flight), 16 otherwise.
template <int LPR>
__global__ __launch_bounds__(256) void k_last_layer(const LastArgs a)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const int sub = threadIdx.x & (LPR - 1);
    if (row >= a.rows) return;
    const int P = a.g.tpc * BM;
    const int c = (int)(row / P), local = (int)(row % P);
    const bool valid = local < chunk_len(a.g, c);
    float dot = 0.f;
    int kbase = 0;
    for (int sg = 0; sg < a.nseg; ++sg) {
        const Seg &S = a.seg[sg];
        const float *x = S.ptr + (size_t)row * S.ld;
#pragma unroll 4
        for (int k = sub * 4; k < S.width; k += LPR * 4) {
            float4 v = *reinterpret_cast<const float4 *>(x + k);
            if (S.mode == 1) {
                const float4 sc = *reinterpret_cast<const float4 *>(S.scale + (size_t)c * S.sld + k);
                const float4 sh = *reinterpret_cast<const float4 *>(S.shift + (size_t)c * S.sld + k);
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            }
            if (S.mode != 0) { v.x = leaky(v.x); v.y = leaky(v.y); v.z = leaky(v.z); v.w = leaky(v.w); }
            const float4 w = *reinterpret_cast<const float4 *>(a.w + kbase + k);
            dot = fmaf(v.x, w.x, dot); dot = fmaf(v.y, w.y, dot);
            dot = fmaf(v.z, w.z, dot); dot = fmaf(v.w, w.w, dot);
        }
        kbase += S.width;
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (sub == 0 && valid) {
        if (a.colQ) dot += fmaf(a.wz0, a.zf[row], a.colQ[(a.qrel ? (size_t)row : (size_t)a.colidx[row]) * a.ldq + a.qoff]);
        float p = 1.f / (1.f + expf(-(dot + a.bias)));
        if (a.mask_bit && !(a.mask[row] & a.mask_bit)) p = 0.f;     // pred * in_bb
        a.out[(int64_t)c * a.g.chunk + local] = p;
    }
}

static void launch_last_layer(const LastArgs &a, hipStream_t st)
{
    int widest = 0;
    for (int s = 0; s < a.nseg; ++s) widest = a.seg[s].width > widest ? a.seg[s].width : widest;
    if (widest <= 128) hipLaunchKernelGGL(k_last_layer<8>, dim3((unsigned)((a.rows * 8 + 255) / 256)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_last_layer<16>, dim3((unsigned)((a.rows * 16 + 255) / 256)), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------------------------------
// k_sample: project -> in-bbox masks -> bilinear taps (NHWC feature maps) -> MLP input rows.
// One wave per point row; one tap = one contiguous channel row, 16 B per lane.
// ------------------------------------------------------------------------------------------------
struct SrcDesc {
    int mode;                 // 0 explicit points [3][n]; 1 lattice (flat = first + i); 2 lattice (flat = index[first + i]);
                              // 3 calc_normal (points [3][n/4] perturbed by delta)
    const float *pts;
    int64_t pts_n;            // n of the [3][n] array
    int res[3];
    double cinv[12];          // rows of calib^-1 (3x4)
    const int64_t *index;
    int64_t first;            // first point of this batch
    float delta;
};

struct SampleArgs {
    SrcDesc src;
    float calib[12];          // 3x4 rows of the (local == global) calibration
    const float *fc; int Cc, Ccp, Hc, Wc;     // coarse NHWC map
    const float *ff; int Cf, Cfp, Hf, Wf;     // fine NHWC map (null if coarse only)
    float z_mul, z_div;
    float *featc; int ldc;
    float *finef; int ldf;
    uint8_t *mask;
    float *zf;                // per-point sampler: the depth feature once more as a column of its own (or null)
    int4 *pix_idx;            // pixel tables: the four coarse taps of the row (pixel index, 0 for a tap outside the map) and
    float4 *pix_wt;           // their weights (0 outside) INSTEAD of the gathered coarse features (featc is not written)
    int64_t rows;
    ChunkGeom g;
};

// One row of [g, 1] . calib_inv^T (mesh_util.py:64) in float64, as a fused chain in k order - what a BLAS dgemm
// micro-kernel evaluates for this 4-term dot product - and independent of the compiler's contraction choices.
__device__ __forceinline__ double lattice_dot(double gx, double gy, double gz, const double *m)
{
    return __fma_rn(1.0, m[3], __fma_rn(gz, m[2], __fma_rn(gy, m[1], __dmul_rn(gx, m[0]))));
}

// One row of orthogonal(): trans + rot @ p (`torch.baddbmm`, BasePIFuNet.py:41) in the order the reference's CPU BLAS
// evaluates it - products accumulated over k by FMA starting from the first product, translation added last.  The order
// only matters for points that re-project onto the box boundary (the in-box masks compare with -1 and 1): with this one
// the masks of the reference's goldens are reproduced for rotated calibrations too.
__device__ __forceinline__ float ortho_row(const float *k, float px, float py, float pz)
{
    return __fadd_rn(fmaf(k[2], pz, fmaf(k[1], py, __fmul_rn(k[0], px))), k[3]);
}

__device__ __forceinline__ void bilinear_setup(float u, float v, int W, int H, int (&off)[4], float (&wt)[4])
{
    // grid_sample(align_corners=True, padding zeros) as used by index() (BasePIFuNet.py:22)
    const float ix = ((u + 1.f) * 0.5f) * (float)(W - 1);
    const float iy = ((v + 1.f) * 0.5f) * (float)(H - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = (x0f + 1.f) - ix, wy0 = (y0f + 1.f) - iy;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wts[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};   // nw, ne, sw, se
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int x = x0 + (t & 1), y = y0 + (t >> 1);
        const bool ok = x >= 0 && x < W && y >= 0 && y < H;
        off[t] = ok ? (y * W + x) : -1;
        wt[t] = wts[t];
    }
}

// The point of row `row` of the batch (explicit list, calc_normal quad member, lattice point) projected by orthogonal():
// false for the padding rows of a chunk.
__device__ __forceinline__ bool projected_point(const SampleArgs &a, int64_t row, float &x, float &y, float &z)
{
    const int P = a.g.tpc * BM;
    const int c = (int)(row / P), local = (int)(row % P);
    if (local >= chunk_len(a.g, c)) return false;
    const int64_t i = a.src.first + (int64_t)c * a.g.chunk + local;   // point index within the call
    float px, py, pz;
    if (a.src.mode == 0) {
        px = a.src.pts[i]; py = a.src.pts[a.src.pts_n + i]; pz = a.src.pts[2 * a.src.pts_n + i];
    } else if (a.src.mode == 3) {
        const int64_t vtx = i >> 2;
        const int axis = (int)(i & 3) - 1;
        px = a.src.pts[vtx]; py = a.src.pts[a.src.pts_n + vtx]; pz = a.src.pts[2 * a.src.pts_n + vtx];
        if (axis == 0) px += a.src.delta;
        if (axis == 1) py += a.src.delta;
        if (axis == 2) pz += a.src.delta;
    } else {
        const int64_t flat = a.src.mode == 1 ? i : (a.src.mode == 4 ? i * a.src.res[2] : a.src.index[i]);
        const int64_t r12 = (int64_t)a.src.res[1] * a.src.res[2];
        const int gi = (int)(flat / r12), gj = (int)((flat / a.src.res[2]) % a.src.res[1]), gk = (int)(flat % a.src.res[2]);
        // create_grid (mesh_util.py:28-33): coords = idx * (2/res) + (-1), separate multiply and add in float64
        const double gx = __dadd_rn(__dmul_rn((double)gi, 2.0 / a.src.res[0]), -1.0);
        const double gy = __dadd_rn(__dmul_rn((double)gj, 2.0 / a.src.res[1]), -1.0);
        const double gz = __dadd_rn(__dmul_rn((double)gk, 2.0 / a.src.res[2]), -1.0);
        const double *m = a.src.cinv;   // [g,1] . calib_inv^T (mesh_util.py:64), then .float() (mesh_util.py:70)
        px = (float)lattice_dot(gx, gy, gz, m);
        py = (float)lattice_dot(gx, gy, gz, m + 4);
        pz = (float)lattice_dot(gx, gy, gz, m + 8);
    }
    const float *k = a.calib;      // orthogonal(): xyz = R p + t (BasePIFuNet.py:25-43)
    x = ortho_row(k, px, py, pz); y = ortho_row(k + 4, px, py, pz); z = ortho_row(k + 8, px, py, pz);
    return true;
}

__global__ void k_sample(const SampleArgs a)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= a.rows) return;
    float x, y, z;
    const bool valid = projected_point(a, row, x, y, z);
    float *fc_out = a.featc + (size_t)row * a.ldc;
    float *ff_out = a.finef ? a.finef + (size_t)row * a.ldf : nullptr;
    if (!valid) {   // padded row: zero features keep every downstream value finite; excluded from statistics
        for (int k = lane * 4; k < a.ldc; k += 256) *reinterpret_cast<float4 *>(fc_out + k) = make_float4(0, 0, 0, 0);
        if (ff_out && lane * 4 < a.ldf) *reinterpret_cast<float4 *>(ff_out + lane * 4) = make_float4(0, 0, 0, 0);
        if (lane == 0) { a.mask[row] = 0; if (a.zf) a.zf[row] = 0.f; }
        return;
    }
    const bool in_xy = x >= -1.f && x <= 1.f && y >= -1.f && y <= 1.f;
    const bool in_xyz = in_xy && z >= -1.f && z <= 1.f;
    if (lane == 0) a.mask[row] = (uint8_t)((in_xyz ? 1 : 0) | (in_xy ? 2 : 0));

    int off[4];
    float wt[4];
    bilinear_setup(x, y, a.Wc, a.Hc, off, wt);
    for (int ch = lane * 4; ch < a.Ccp; ch += 256) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (off[t] >= 0) {
                const float4 v = *reinterpret_cast<const float4 *>(a.fc + (size_t)off[t] * a.Ccp + ch);
                acc.x = fmaf(v.x, wt[t], acc.x); acc.y = fmaf(v.y, wt[t], acc.y);
                acc.z = fmaf(v.z, wt[t], acc.z); acc.w = fmaf(v.w, wt[t], acc.w);
            }
        }
        *reinterpret_cast<float4 *>(fc_out + ch) = acc;
    }
    // [.. image features (Cc) | z feature | zero pad]; Ccp == Cc is required (Cc % 4 == 0)
    for (int col = a.Cc + lane; col < a.ldc; col += 64)
        fc_out[col] = col == a.Cc ? (z * a.z_mul) / a.z_div : 0.f;
    if (a.zf && lane == 0) a.zf[row] = (z * a.z_mul) / a.z_div;

    if (ff_out) {
        bilinear_setup(x, y, a.Wf, a.Hf, off, wt);
        for (int ch = lane * 4; ch < a.ldf; ch += 256) {
            float4 acc = make_float4(0, 0, 0, 0);
            if (ch < a.Cfp) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (off[t] >= 0) {
                        const float4 v = *reinterpret_cast<const float4 *>(a.ff + (size_t)off[t] * a.Cfp + ch);
                        acc.x = fmaf(v.x, wt[t], acc.x); acc.y = fmaf(v.y, wt[t], acc.y);
                        acc.z = fmaf(v.z, wt[t], acc.z); acc.w = fmaf(v.w, wt[t], acc.w);
                    }
                }
            }
            *reinterpret_cast<float4 *>(ff_out + ch) = acc;
        }
    }
}

// Pixel tables: what k_sample leaves for a row, without the coarse gather - the four coarse taps (pixel index, weight; index
// 0 and weight 0 for a tap outside the map) for k_expand_pix, the depth feature, the in-box masks and the sampled fine
// features.  The coarse features are never read, so a wave per row would idle: four lanes per row, each gathering a
// quarter of the fine channels.  Same arithmetic as k_sample, value for value.
__global__ __launch_bounds__(256) void k_sample_pix(const SampleArgs a)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int sub = threadIdx.x & 3;
    if (row >= a.rows) return;
    float x = 0.f, y = 0.f, z = 0.f;
    const bool valid = projected_point(a, row, x, y, z);
    float *ff_out = a.finef ? a.finef + (size_t)row * a.ldf : nullptr;
    int off[4];
    float wt[4];
    if (sub == 0) {
        const bool in_xy = valid && x >= -1.f && x <= 1.f && y >= -1.f && y <= 1.f;
        const bool in_xyz = in_xy && z >= -1.f && z <= 1.f;
        a.mask[row] = (uint8_t)((in_xyz ? 1 : 0) | (in_xy ? 2 : 0));
        a.zf[row] = valid ? (z * a.z_mul) / a.z_div : 0.f;
        bilinear_setup(x, y, a.Wc, a.Hc, off, wt);
        a.pix_idx[row] = valid ? make_int4(off[0] < 0 ? 0 : off[0], off[1] < 0 ? 0 : off[1], off[2] < 0 ? 0 : off[2], off[3] < 0 ? 0 : off[3])
                               : make_int4(0, 0, 0, 0);
        a.pix_wt[row] = valid ? make_float4(off[0] < 0 ? 0.f : wt[0], off[1] < 0 ? 0.f : wt[1], off[2] < 0 ? 0.f : wt[2], off[3] < 0 ? 0.f : wt[3])
                              : make_float4(0, 0, 0, 0);
    }
    if (!ff_out) return;
    bilinear_setup(x, y, a.Wf, a.Hf, off, wt);
    for (int ch = sub * 4; ch < a.ldf; ch += 16) {
        float4 acc = make_float4(0, 0, 0, 0);
        if (valid && ch < a.Cfp) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (off[t] >= 0) {
                    const float4 v = *reinterpret_cast<const float4 *>(a.ff + (size_t)off[t] * a.Cfp + ch);
                    acc.x = fmaf(v.x, wt[t], acc.x); acc.y = fmaf(v.y, wt[t], acc.y);
                    acc.z = fmaf(v.z, wt[t], acc.z); acc.w = fmaf(v.w, wt[t], acc.w);
                }
            }
        }
        *reinterpret_cast<float4 *>(ff_out + ch) = acc;
    }
}

// Column sharing, per point row: depth feature, bounding-box masks and the (i, j) column of a lattice point.
// The image features are NOT gathered per point: they only depend on the column (k_sample over the columns).
__global__ void k_sample_col(const SampleArgs a, float *zf, int *colidx, int64_t col_base)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.rows) return;
    const int P = a.g.tpc * BM;
    const int c = (int)(row / P), local = (int)(row % P);
    if (local >= chunk_len(a.g, c)) { zf[row] = 0.f; colidx[row] = 0; a.mask[row] = 0; return; }
    const int64_t i = a.src.first + (int64_t)c * a.g.chunk + local;
    const int64_t flat = a.src.mode == 1 ? i : a.src.index[i];
    const int64_t r12 = (int64_t)a.src.res[1] * a.src.res[2];
    const int gi = (int)(flat / r12), gj = (int)((flat / a.src.res[2]) % a.src.res[1]), gk = (int)(flat % a.src.res[2]);
    const double gx = __dadd_rn(__dmul_rn((double)gi, 2.0 / a.src.res[0]), -1.0);
    const double gy = __dadd_rn(__dmul_rn((double)gj, 2.0 / a.src.res[1]), -1.0);
    const double gz = __dadd_rn(__dmul_rn((double)gk, 2.0 / a.src.res[2]), -1.0);
    const double *m = a.src.cinv;
    const float px = (float)lattice_dot(gx, gy, gz, m);
    const float py = (float)lattice_dot(gx, gy, gz, m + 4);
    const float pz = (float)lattice_dot(gx, gy, gz, m + 8);
    const float *k = a.calib;
    const float x = ortho_row(k, px, py, pz), y = ortho_row(k + 4, px, py, pz), z = ortho_row(k + 8, px, py, pz);
    const bool in_xy = x >= -1.f && x <= 1.f && y >= -1.f && y <= 1.f;
    const bool in_xyz = in_xy && z >= -1.f && z <= 1.f;
    a.mask[row] = (uint8_t)((in_xyz ? 1 : 0) | (in_xy ? 2 : 0));
    zf[row] = (z * a.z_mul) / a.z_div;
    colidx[row] = (int)((int64_t)gi * a.src.res[1] + gj - col_base);
}

// Column sharing, coarse layer 0: Y0[row][c] = Q[col[row]][c] + wz[c] * zf[row] + bias[c]  (the K = 257 GEMM is
// gone: its image-feature part is per column) + the per-tile (sum, sum^2) partials of the following norm.
// One block per (128-row tile, 128 channels): thread = 4 channels x 16 rows.
__global__ void k_expand(const float *Q, int ldq, const float *wz, const float *bias, const float *zf, const int *colidx,
                         float *Y, int ldy, float2 *partial, ChunkGeom g)
{
    __shared__ float red[2][8][128];
    const int tile_m = blockIdx.x, c0 = blockIdx.y * 128;
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int ch = c0 + cg * 4;
    const int chunk_id = tile_m / g.tpc;
    const int local0 = (tile_m % g.tpc) * BM;
    int nvalid = chunk_len(g, chunk_id) - local0;
    nvalid = nvalid < 0 ? 0 : (nvalid > BM ? BM : nvalid);
    const float4 w = *reinterpret_cast<const float4 *>(wz + ch), b = *reinterpret_cast<const float4 *>(bias + ch);
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int r = rg * 16; r < rg * 16 + 16; ++r) {
        const size_t row = (size_t)tile_m * BM + r;
        const float z = zf[row];
        const float4 v = *reinterpret_cast<const float4 *>(Q + (size_t)colidx[row] * ldq + ch);
        float4 o;
        o.x = fmaf(w.x, z, v.x) + b.x; o.y = fmaf(w.y, z, v.y) + b.y;
        o.z = fmaf(w.z, z, v.z) + b.z; o.w = fmaf(w.w, z, v.w) + b.w;
        if (Y) *reinterpret_cast<float4 *>(Y + row * ldy + ch) = o;      // null: statistics only (depth walk)
        if (r < nvalid) {
            s[0] += o.x; q[0] = fmaf(o.x, o.x, q[0]); s[1] += o.y; q[1] = fmaf(o.y, o.y, q[1]);
            s[2] += o.z; q[2] = fmaf(o.z, o.z, q[2]); s[3] += o.w; q[3] = fmaf(o.w, o.w, q[3]);
        }
    }
    if (!partial) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[0][rg][cg * 4 + k] = s[k]; red[1][rg][cg * 4 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) { ss += red[0][t][threadIdx.x]; qq += red[1][t][threadIdx.x]; }
        partial[(size_t)tile_m * ldy + c0 + threadIdx.x] = make_float2(ss, qq);
    }
}

// Pixel tables (explicit point lists).  index() is linear in the feature map, so W_feat * index(F, uv) = sum over the four
// taps of w_t * (W_feat * F[tap_t]): with G = W_feat * F computed once per feature map (one row per pixel), the coarse
// layer-0 output of a point is a bilinear blend of four rows of G (+ wz * z + bias) and the image-feature part of the
// skip connections is the same blend of other columns of G - no K = 256 product per point is left.  Blend in tap order
// (nw, ne, sw, se) by FMA, like k_sample blends the features themselves.
// One block per (128-row tile, 128 columns): thread = 4 columns x 16 rows.  Column blocks [0, n0) are layer 0 (output Y +
// statistics partials, like k_expand), the rest are the skip products (raw blend -> S, the per-row start-value table of
// the later GEMMs).  Consecutive rows with the same four taps (the p / p+dx / p+dy / p+dz rows of a calc_normal vertex,
// mostly) reuse the rows of G they hold in registers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float *p, const float4 &v)
{
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(p));
}

struct PixArgs {
    const float *G; int ldg;            // [H * W][ldg]
    int g0, g1;                         // first G column of layer 0 / of the skip blocks
    const int4 *idx; const float4 *wt;  // per row: taps and weights (k_sample in pixel mode)
    const float *wz, *bias, *zf;        // layer 0: depth weight, bias; depth feature per row
    float *Y; int ldy; float2 *partial; // layer-0 output [rows][ldy] + (sum, sum^2) partials per (tile, column), or null
    int n0;                             // 128-column blocks of layer 0
    float *S; int lds;                  // skip products [rows][lds]
    ChunkGeom g;
};

__global__ __launch_bounds__(256) void k_expand_pix(const PixArgs a)
{
    __shared__ float red[2][8][128];
    const int tile_m = blockIdx.x;
    const bool l0 = (int)blockIdx.y < a.n0;
    const int c0 = (l0 ? blockIdx.y : blockIdx.y - a.n0) * 128;
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int ch = c0 + cg * 4;
    const float *Gc = a.G + (l0 ? a.g0 : a.g1) + ch;
    const int chunk_id = tile_m / a.g.tpc;
    const int local0 = (tile_m % a.g.tpc) * BM;
    int nvalid = chunk_len(a.g, chunk_id) - local0;
    nvalid = nvalid < 0 ? 0 : (nvalid > BM ? BM : nvalid);
    float4 w = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
    if (l0) { w = *reinterpret_cast<const float4 *>(a.wz + ch); b = *reinterpret_cast<const float4 *>(a.bias + ch); }
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    int4 held = make_int4(-1, -1, -1, -1);
    float4 v0 = make_float4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
    for (int r = rg * 16; r < rg * 16 + 16; ++r) {
        const size_t row = (size_t)tile_m * BM + r;
        const int4 id = a.idx[row];
        const float4 wt = a.wt[row];
        if (id.x != held.x || id.y != held.y || id.z != held.z || id.w != held.w) {
            v0 = *reinterpret_cast<const float4 *>(Gc + (size_t)id.x * a.ldg);
            v1 = *reinterpret_cast<const float4 *>(Gc + (size_t)id.y * a.ldg);
            v2 = *reinterpret_cast<const float4 *>(Gc + (size_t)id.z * a.ldg);
            v3 = *reinterpret_cast<const float4 *>(Gc + (size_t)id.w * a.ldg);
            held = id;
        }
        float4 o;
        o.x = fmaf(v3.x, wt.w, fmaf(v2.x, wt.z, fmaf(v1.x, wt.y, fmaf(v0.x, wt.x, 0.f))));
        o.y = fmaf(v3.y, wt.w, fmaf(v2.y, wt.z, fmaf(v1.y, wt.y, fmaf(v0.y, wt.x, 0.f))));
        o.z = fmaf(v3.z, wt.w, fmaf(v2.z, wt.z, fmaf(v1.z, wt.y, fmaf(v0.z, wt.x, 0.f))));
        o.w = fmaf(v3.w, wt.w, fmaf(v2.w, wt.z, fmaf(v1.w, wt.y, fmaf(v0.w, wt.x, 0.f))));
        if (!l0) {       // streamed outputs (GBs per launch, next read by another kernel): keep them out of the caches G lives in
            nt_store4(a.S + row * a.lds + ch, o);
            continue;
        }
        const float z = a.zf[row];
        o.x = fmaf(w.x, z, o.x) + b.x; o.y = fmaf(w.y, z, o.y) + b.y;
        o.z = fmaf(w.z, z, o.z) + b.z; o.w = fmaf(w.w, z, o.w) + b.w;
        nt_store4(a.Y + row * a.ldy + ch, o);
        if (r < nvalid) {
            s[0] += o.x; q[0] = fmaf(o.x, o.x, q[0]); s[1] += o.y; q[1] = fmaf(o.y, o.y, q[1]);
            s[2] += o.z; q[2] = fmaf(o.z, o.z, q[2]); s[3] += o.w; q[3] = fmaf(o.w, o.w, q[3]);
        }
    }
    if (!l0 || !a.partial) return;      // block-uniform
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[0][rg][cg * 4 + k] = s[k]; red[1][rg][cg * 4 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) { ss += red[0][t][threadIdx.x]; qq += red[1][t][threadIdx.x]; }
        a.partial[(size_t)tile_m * a.ldy + c0 + threadIdx.x] = make_float2(ss, qq);
    }
}

// NCHW -> NHWC (channel-padded) repack of a feature map, once per image.
__global__ void k_nchw_to_nhwc(const float *src, float *dst, int C, int Cp, int HW)
{
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < Cp) dst[(size_t)p * Cp + c] = tile[tx][r];
    }
}

// phi export: [C][n] = leaky(Y[n][C] * scale + shift)  (netG.phi, PIFuNetwNML.py:137-138)
__global__ void k_phi_export(const float *Y, int ld, const float *scale, const float *shift, int mode, int C,
                             int64_t n, float *out)
{
    __shared__ float tile[32][33];
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int64_t p = p0 + r;
        const int c = c0 + tx;
        float v = 0.f;
        if (p < n && c < C) {
            v = Y[(size_t)p * ld + c];
            if (mode == 1) v = fmaf(v, scale[c], shift[c]);
            if (mode != 0) v = leaky(v);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const int64_t p = p0 + tx;
        if (c < C && p < n) out[(size_t)c * n + p] = tile[tx][r];
    }
}

// calc_normal tail: pred [4n] interleaved -> unit normals [3][n]  (PIFuMRNet.py:232-239)
__global__ void k_normals(const float *pred, int64_t n, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 f = *reinterpret_cast<const float4 *>(pred + 4 * i);
    const float nx = -(f.y - f.x), ny = -(f.z - f.x), nz = -(f.w - f.x);
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    const float d = fmaxf(len, 1e-8f);                       // F.normalize(dim=1, eps=1e-8)
    out[i] = nx / d; out[n + i] = ny / d; out[2 * n + i] = nz / d;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct LevelBufs {
    float *out[PIFU_MAX_LAYERS] = {nullptr};     // raw layer outputs [rows][cout_pad]
    float2 *partial[PIFU_MAX_LAYERS] = {nullptr};
    float *scale[PIFU_MAX_LAYERS] = {nullptr};   // [nchunks][cout_pad]
    float *shift[PIFU_MAX_LAYERS] = {nullptr};
};

struct Carver {
    char *base;
    size_t off = 0;
    template <typename T> T *take(size_t count)
    {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct WalkPlan {      // depth-walk workspace of one batch (walk.inc); all null when the walk is not available
    int *tile_cnt = nullptr, *tile_off = nullptr, *seg_row0 = nullptr, *chunk_seg0 = nullptr, *nseg_dev = nullptr, *nflips = nullptr;
    float *V = nullptr, *PR0 = nullptr;
    uint32_t *flip_key = nullptr;
    float2 *flip_d = nullptr, *partial_seg = nullptr;
    int64_t maxseg = 0;
    int ntiles = 0;
};

struct Plan {
    WalkPlan wk;
    float *featc; int ldc;
    float *finef; int ldf;
    uint8_t *mask;
    float *normal_pred;
    float *coarse_scratch;
    float *zf;        // column sharing: depth feature per row
    int *colidx;      // column sharing: column of the row
    int4 *pix_idx;    // pixel tables: taps and weights of the row, and the per-row skip products [rows][ldS]
    float4 *pix_wt;
    float *S; int ldS;
    LevelBufs lv[2];
    size_t bytes;
};

Plan carve(pifu_ctx *ctx, char *base, int64_t rows, int64_t max_chunks, bool fine, int64_t normal_pts, bool walk = false,
           int pix_ldS = -1 /* >= 0: pixel-table mode with that many skip-product columns */)
{
    Plan p;
    Carver c{base};
    if (walk) {
        const Mlp &m0 = ctx->mlp[0];
        const int C0 = m0.layer[0].cout_pad, N1 = m0.layer[1].cout_pad;
        WalkPlan &w = p.wk;
        w.maxseg = round_up(rows / 8 > 128 ? rows / 8 : 128, BM);
        w.ntiles = (int)((rows + 2047) / 2048);
        w.tile_cnt = c.take<int>(w.ntiles);
        w.tile_off = c.take<int>(w.ntiles + 1);
        w.seg_row0 = c.take<int>(w.maxseg + 1);
        w.chunk_seg0 = c.take<int>(max_chunks + 1);
        w.nseg_dev = c.take<int>(4);
        w.nflips = c.take<int>(w.maxseg);
        w.V = c.take<float>((size_t)2 * w.maxseg * C0);
        w.PR0 = c.take<float>((size_t)2 * w.maxseg * N1);
        w.flip_key = c.take<uint32_t>((size_t)w.maxseg * C0);
        w.flip_d = c.take<float2>((size_t)w.maxseg * C0);
        w.partial_seg = c.take<float2>((size_t)w.maxseg * N1);
    }
    const Mlp &mc = ctx->mlp[0];
    p.ldc = (int)round_up(mc.dims[0], BK);
    const bool pix = pix_ldS >= 0;
    p.featc = pix ? nullptr : c.take<float>((size_t)rows * p.ldc);      // pixel tables: the coarse features are never gathered
    p.pix_idx = pix ? c.take<int4>((size_t)rows) : nullptr;
    p.pix_wt = pix ? c.take<float4>((size_t)rows) : nullptr;
    p.ldS = pix ? pix_ldS : 0;
    p.S = pix && pix_ldS > 0 ? c.take<float>((size_t)rows * pix_ldS) : nullptr;
    p.ldf = fine ? (int)round_up(ctx->feat[1].C > 0 ? ctx->feat[1].C : 16, BK) : 0;
    p.finef = fine ? c.take<float>((size_t)rows * p.ldf) : nullptr;
    p.mask = c.take<uint8_t>((size_t)rows);
    p.normal_pred = normal_pts ? c.take<float>((size_t)normal_pts) : nullptr;
    p.coarse_scratch = c.take<float>((size_t)rows);
    p.zf = c.take<float>((size_t)rows);
    p.colidx = c.take<int>((size_t)rows);
    const int64_t ntm = rows / BM;
    for (int L = 0; L < (fine ? 2 : 1); ++L) {
        const Mlp &m = ctx->mlp[L];
        for (int i = 0; i < m.nlayers - 1; ++i) {
            const int cp = m.layer[i].cout_pad;
            p.lv[L].out[i] = c.take<float>((size_t)rows * cp);
            p.lv[L].partial[i] = c.take<float2>((size_t)ntm * cp * 2);     // x2: half-tile partials of k_mlp_gemm_ws
            p.lv[L].scale[i] = c.take<float>((size_t)max_chunks * cp);
            p.lv[L].shift[i] = c.take<float>((size_t)max_chunks * cp);
        }
    }
    p.bytes = c.off + 256;
    return p;
}

Seg make_prev_seg(const pifu_ctx *ctx, const Mlp &m, const LevelBufs &b, int layer_of_output)
{
    const MlpLayer &l = m.layer[layer_of_output];
    Seg s{};
    s.ptr = b.out[layer_of_output];
    s.ld = l.cout_pad;
    s.width = l.cout_pad;
    if (m.norm == PIFU_NORM_GROUP || m.norm == PIFU_NORM_BATCH_TRAIN) {
        s.scale = b.scale[layer_of_output];
        s.shift = b.shift[layer_of_output];
        s.sld = l.cout_pad;
        s.mode = 1;
    } else if (m.norm == PIFU_NORM_BATCH_EVAL) {
        s.scale = l.fixed_scale;
        s.shift = l.fixed_shift;
        s.sld = 0;
        s.mode = 1;
    } else {
        s.scale = ctx->ident_scale;
        s.shift = ctx->ident_shift;
        s.sld = 0;
        s.mode = 2;
    }
    return s;
}

}  // namespace

int pifu_set_features(pifu_ctx *ctx, int level, const float *feat_nchw, int C, int H, int W, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && feat_nchw, PIFU_ERR_INVALID, "pifu_set_features: null argument");
    PIFU_REQUIRE(level == 0 || level == 1, PIFU_ERR_INVALID, "pifu_set_features: level must be 0 or 1");
    PIFU_REQUIRE(C > 0 && C % 4 == 0 && H > 1 && W > 1, PIFU_ERR_INVALID,
                 "pifu_set_features: need C %% 4 == 0 and H, W > 1 (got C=%d H=%d W=%d)", C, H, W);
    PIFU_USE(ctx, stream);
    PIFU_HIP(hipSetDevice(ctx->device));
    {
        const int rc = pifu_enc_join(ctx, (hipStream_t)stream);      // a deferred coarse hourglass may still be writing its map
        if (rc) return rc;
    }
    FeatMap &f = ctx->feat[level];
    const size_t bytes = (size_t)H * W * C * sizeof(float);
    if (bytes > f.bytes) {
        if (f.nhwc) PIFU_HIP(hipFree(f.nhwc));
        f.nhwc = nullptr;
        PIFU_HIP(hipMalloc(&f.nhwc, bytes));
        f.bytes = bytes;
    }
    f.C = C; f.Cpad = C; f.H = H; f.W = W;
    if (level == 0) ctx->pix.valid = false;      // the pixel table belongs to the previous map
    ctx->col_cache.valid = false;
    dim3 grid((H * W + 31) / 32, (C + 31) / 32);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, dim3(256), 0, (hipStream_t)stream, feat_nchw, f.nhwc, C, C, H * W);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

static int split_planes(pifu_ctx *ctx, const float *W, size_t n, const __bf16 **hi, const __bf16 **lo, hipStream_t st);

#ifdef PIFU_LAB
#include "gemm_dispatch.inc"
#endif

// exact-fp32 GEMM launch: k_mlp_gemm_tri, three blocks per CU, no s_setprio (with three waves per SIMD raising the MFMA
// phase's priority costs 1.2 %; with two it gains 1-2 %)
static void launch_fp32_gemm(pifu_ctx *ctx, const GemmArgs &a, hipStream_t st)
{
#ifdef PIFU_LAB
    if (lab_launch_fp32_gemm(ctx, a, st)) return;
#endif
    (void)ctx;
    const dim3 grid((unsigned)round_up(a.ntm, 8) * a.ntn);
    if (a.last_out) hipLaunchKernelGGL((k_mlp_gemm_tri<0, 0, 1>), grid, dim3(256), 0, st, a);
    else if (!a.partial && a.qrel) hipLaunchKernelGGL((k_mlp_gemm_tri<0, 1, 2>), grid, dim3(256), 0, st, a);
    else if (!a.partial) hipLaunchKernelGGL((k_mlp_gemm_tri<0, 0, 2>), grid, dim3(256), 0, st, a);
    else if (a.qrel) hipLaunchKernelGGL((k_mlp_gemm_tri<0, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_mlp_gemm_tri<0, 0>), grid, dim3(256), 0, st, a);
}

static int run_gemm(pifu_ctx *ctx, GemmArgs &a, double flops, hipStream_t st);

// partial rows per 128-row tile that k_norm_finalize sums (lab builds: 2 for the half-tile partials of k_mlp_gemm_ws)
static inline int stat_halves(const pifu_ctx *ctx)
{
#ifdef PIFU_LAB
    return lab_stat_halves(ctx);
#else
    (void)ctx;
    return 1;
#endif
}

// An MLP without norm layers (MLP.py:37-41: the published pifuhd.pt): the layer in front of the Cout = 1 layer can take that layer
// into its epilogue when its 128 outputs are one column tile, the last layer reads nothing else (no skip input), and the exact-fp32
// kernel runs - the 128-wide activations (1 GB per 2 M-row batch) are then neither written nor read back by k_last_layer.
static bool fuse_last_layer(const pifu_ctx *ctx, const Mlp &m, int i, const float *pred_out, int upto_layer, bool skip_as_start_value = false)
{
#ifdef PIFU_LAB
    return false;
#else
    // (a last layer with a skip input is taken in only where that input arrives as a per-column / per-row start value:
    // launch_level_col; the per-point path would have to read the feature rows again).  The fused epilogue does not store the
    // layer's own output: not for the coarse layer whose output is phi, the fine level's input (MLP.py:61-62 with a merge
    // layer right in front of the last one - no published configuration, tests/test_gpu_pointwise.py has one)
    if (&m == &ctx->mlp[0] && i == m.merge_layer) return false;
    return m.norm == PIFU_NORM_NONE && i >= 0 && i == m.nlayers - 2 && upto_layer >= m.nlayers && pred_out != nullptr && ctx->precision == 0 &&
           m.layer[i].cout == BN && m.layer[i].cout_pad == BN && (m.layer[i + 1].cin_feat == 0 || skip_as_start_value) &&
           m.layer[i + 1].cout == 1;
#endif
}

static int launch_level(pifu_ctx *ctx, int L, const Plan &pl, const ChunkGeom &g, int64_t rows, const uint8_t *mask,
                        int mask_bit, float *pred_out, int upto_layer /* exclusive; nlayers for all */, hipStream_t st)
{
    const Mlp &m = ctx->mlp[L];
    const LevelBufs &b = pl.lv[L];
    const Mlp &mc = ctx->mlp[0];
    const int ntm = (int)(rows / BM);
    // the skip-concatenated MLP input ("feature" in MLP.forward)
    Seg feat[2];
    int nfeat = 0;
    // coarse level: the input is [image features | z].  When the image features fill whole K steps, z does not get a
    // (31/32 empty) step of its own: the GEMM covers the image features and starts from wz * z (GemmArgs::colQ with ldq = 0)
    const int cimg = m.dims[0] - 1;
    const bool zrank1 = L == 0 && cimg > 0 && cimg % BK == 0 && pl.ldc == cimg + BK && m.layer[0].wz != nullptr;
    if (L == 0) {
        feat[0] = Seg{pl.featc, ctx->ident_scale, ctx->ident_shift, pl.ldc, pl.ldc, 0, 0};
        nfeat = 1;
    } else {
        feat[0] = Seg{pl.finef, ctx->ident_scale, ctx->ident_shift, pl.ldf, pl.ldf, 0, 0};
        feat[1] = make_prev_seg(ctx, mc, pl.lv[0], mc.merge_layer);      // phi = act(norm(coarse layer merge))
        nfeat = 2;
    }
    bool fused_last = false;      // the Cout = 1 layer went into the previous layer's epilogue (fuse_last_layer)
    for (int i = 0; i < m.nlayers && i < upto_layer; ++i) {
        const MlpLayer &l = m.layer[i];
        Seg segs[3];
        int ns = 0;
        if (i > 0) segs[ns++] = make_prev_seg(ctx, m, b, i - 1);
        const bool has_feat = i == 0 || m.res[i];
        if (has_feat)
            for (int f = 0; f < nfeat; ++f) segs[ns++] = feat[f];
        int ksum = 0;
        for (int s = 0; s < ns; ++s) ksum += segs[s].width;
        PIFU_REQUIRE(ksum == l.kpad, PIFU_ERR_STATE, "internal: K mismatch at level %d layer %d (%d vs %d)", L, i, ksum, l.kpad);
        if (i < m.nlayers - 1) {
            GemmArgs a{};
            for (int s = 0; s < ns; ++s) a.seg[s] = segs[s];
            a.nseg = ns;
            if (zrank1 && has_feat) {       // the feature segment is the last one: its z step (and the weight columns of it) drop out
                a.seg[ns - 1].width = cimg;
                a.colQ = ctx->ident_shift; a.ldq = 0; a.qoff = 0; a.colwz = l.wz;
                a.zf = pl.zf; a.colidx = pl.colidx;
            }
            a.W = l.w; a.ldw = l.kpad; a.bias = l.bias;
            a.Y = b.out[i]; a.ldy = l.cout_pad;
            const bool stats = (m.norm == PIFU_NORM_GROUP || m.norm == PIFU_NORM_BATCH_TRAIN);
            a.partial = stats ? b.partial[i] : nullptr;
            a.ntm = ntm; a.ntn = l.cout_pad / BN; a.g = g;
            if (fuse_last_layer(ctx, m, i, pred_out, upto_layer)) {
                const MlpLayer &ll = m.layer[i + 1];
                a.last_w = ll.w; a.last_bias = ll.host_bias0; a.last_mask = mask; a.last_mask_bit = mask_bit; a.last_out = pred_out;
                fused_last = true;
            }
            const double valid = (double)(g.nchunks - 1) * g.chunk + g.last_len;
            int rc2 = run_gemm(ctx, a, 2.0 * valid * l.cout * (double)(l.cin_prev + l.cin_feat), st);
            if (rc2) return rc2;
            if (stats) {
                const int groups = m.norm == PIFU_NORM_GROUP ? 32 : l.cout;
                const int cpg = l.cout / groups;
                const int nw = g.nchunks * groups;
                hipLaunchKernelGGL(k_norm_finalize, dim3(nw), dim3(256), 0, st, (const float2 *)b.partial[i],
                                   l.cout_pad, cpg, groups, (const float *)l.gamma, (const float *)l.beta, b.scale[i],
                                   b.shift[i], g, 1e-5f, stat_halves(ctx)
                                   );
            }
        } else {
            if (!pred_out || fused_last) continue;      // (fused: the previous layer's epilogue wrote the prediction)
            LastArgs a{};
            for (int s = 0; s < ns; ++s) a.seg[s] = segs[s];
            a.nseg = ns;
            a.w = l.w; a.bias = l.host_bias0; a.mask = mask; a.mask_bit = mask_bit; a.out = pred_out; a.rows = rows; a.g = g;
            launch_last_layer(a, st);
        }
    }
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

#include "walk.inc"

// ------------------------------------------------------------------------------------------------
// Column sharing (lattice sources).  A lattice point (i, j, k) projects to image coordinates that depend on
// (i, j) only (reconstruction() feeds calib^-1 * grid back through calib), so the sampled image features - 256
// of the 257 coarse MLP inputs, 16 of the 272 fine ones - are per COLUMN.  Every product of a weight matrix
// with that part of the input (layer 0 and the skip connections, MLP.py:61-64) is therefore computed once per
// column, Q = W_feat * feat(col), and enters the per-point layers as the initial accumulator value
// Q[col] + w_z * z.  1 081 600 MAC per query instead of 1 458 049; same result up to fp32 summation order.
// ------------------------------------------------------------------------------------------------
static int upload_vec(float **dst, const std::vector<float> &host)
{
    if (*dst) { PIFU_HIP(hipFree(*dst)); *dst = nullptr; }
    PIFU_HIP(hipMalloc(dst, host.size() * sizeof(float)));
    PIFU_HIP(hipMemcpy(*dst, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return PIFU_OK;
}

static int build_column_weights(pifu_ctx *ctx, int L)
{
    Mlp &m = ctx->mlp[L];
    if (m.col_built) return PIFU_OK;
    const Mlp &mc = ctx->mlp[0];
    const int cphi = mc.dims[mc.merge_layer + 1];
    const int cfeat = L == 0 ? m.dims[0] - 1 : m.dims[0] - cphi;         // per-column input channels
    const int cphi_pad = L == 0 ? 0 : mc.layer[mc.merge_layer].cout_pad;
    m.col_k = (int)round_up(cfeat, BK);
    int off = 0;
    for (int i = 0; i < m.nlayers; ++i) {
        MlpLayer &l = m.layer[i];
        l.col_off = -1;
        if (l.cin_feat) { l.col_off = off; off += (int)round_up(l.cout, BN); }
    }
    m.col_rows = off;
    std::vector<float> cw((size_t)m.col_rows * m.col_k, 0.f);
    for (int i = 0; i < m.nlayers; ++i) {
        MlpLayer &l = m.layer[i];
        const int cin = l.cin_prev + l.cin_feat;
        const int prev_pad = i == 0 ? 0 : m.layer[i - 1].cout_pad;
        const int rows_pad = l.cout_pad > 0 ? l.cout_pad : 1;
        if (!l.cin_feat) {              // nothing per-column in this layer
            l.kpad_col = l.kpad; l.cin_col = cin;
            std::vector<float> w((size_t)rows_pad * l.kpad, 0.f);
            PIFU_HIP(hipMemcpy(w.data(), l.w, w.size() * sizeof(float), hipMemcpyDeviceToHost));
            int rc = upload_vec(&l.w_col, w);
            if (rc) return rc;
            continue;
        }
        l.kpad_col = prev_pad + cphi_pad;
        l.cin_col = l.cin_prev + (L == 0 ? 0 : cphi);
        std::vector<float> w((size_t)rows_pad * (l.kpad_col > 0 ? l.kpad_col : 1), 0.f), wz(rows_pad > BN ? rows_pad : BN, 0.f);
        for (int o = 0; o < l.cout; ++o) {
            const float *src = l.host_w.data() + (size_t)o * cin;
            for (int k = 0; k < l.cin_prev; ++k) w[(size_t)o * l.kpad_col + k] = src[k];
            for (int k = 0; k < cfeat; ++k) cw[(size_t)(l.col_off + o) * m.col_k + k] = src[l.cin_prev + k];
            if (L == 0) wz[o] = src[l.cin_prev + cfeat];
            else
                for (int k = 0; k < cphi; ++k) w[(size_t)o * l.kpad_col + prev_pad + k] = src[l.cin_prev + cfeat + k];
        }
        l.host_wz0 = wz[0];
        int rc = upload_vec(&l.w_col, w);
        if (rc) return rc;
        if ((rc = upload_vec(&l.wz, wz))) return rc;
    }
    int rc = upload_vec(&m.colW, cw);
    if (rc) return rc;
    if (L == 0 && m.nlayers >= 3 && !m.layer[1].cin_feat && m.layer[1].w_col) {      // depth walk: W1^T [C0][N1]
        MlpLayer &l1 = m.layer[1];
        if (l1.w_colT) { PIFU_HIP(hipFree(l1.w_colT)); l1.w_colT = nullptr; }
        PIFU_HIP(hipMalloc(&l1.w_colT, (size_t)l1.cout_pad * l1.kpad_col * sizeof(float)));
        hipLaunchKernelGGL(k_transpose, dim3((l1.kpad_col + 31) / 32, (l1.cout_pad + 31) / 32), dim3(256), 0, 0,
                           (const float *)l1.w_col, l1.cout_pad, l1.kpad_col, l1.w_colT);
        PIFU_HIP(hipDeviceSynchronize());
    }
    m.col_built = true;
    return PIFU_OK;
}

struct ColTables {
    float *featc = nullptr, *finef = nullptr;   // per-column sampled features [ncols_pad][ldc | ldf]
    uint8_t *mask = nullptr;
    float *Q[2] = {nullptr, nullptr};           // per-column products [ncols_pad][col_rows]
    int ldq[2] = {0, 0};
    int64_t col_base = 0, ncols = 0, ncols_pad = 0;
};

// hi / lo planes (bf16, or fp16 scaled by 2^11) of a packed fp32 weight matrix, built on the device the first time the
// matrix is used in the current precision mode (pifu_set_precision drops the planes of the other mode)
static int split_planes(pifu_ctx *ctx, const float *W, size_t n, const __bf16 **hi, const __bf16 **lo, hipStream_t st)
{
    auto it = ctx->bf16_planes.find((const void *)W);
    if (it == ctx->bf16_planes.end() || it->second.n != n) {
        if (it != ctx->bf16_planes.end()) { (void)hipFree(it->second.hi); ctx->bf16_planes.erase(it); }
        void *p = nullptr;
        PIFU_HIP(hipMalloc(&p, n * 2 * sizeof(__bf16) + 16));
        pifu_ctx::Planes pl{p, (char *)p + n * sizeof(__bf16), n};
        int *bad = (int *)((char *)p + n * 2 * sizeof(__bf16));
        PIFU_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
        if (ctx->precision == 2) {
            hipLaunchKernelGGL(k_split_planes<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, pl.hi, pl.lo, n, bad);
            int host_bad = 0;
            PIFU_HIP(hipMemcpyAsync(&host_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
            PIFU_HIP(hipStreamSynchronize(st));
            if (host_bad) {
                (void)hipFree(p);
                PIFU_REQUIRE(false, PIFU_ERR_INVALID, "fp16x3 precision needs finite weights with |w| < 32 (set precision 0 or 1)");
            }
        } else {
            hipLaunchKernelGGL(k_split_planes<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, pl.hi, pl.lo, n, bad);
        }
        it = ctx->bf16_planes.emplace((const void *)W, pl).first;
    }
    *hi = (const __bf16 *)it->second.hi;
    *lo = (const __bf16 *)it->second.lo;
    return PIFU_OK;
}

// per-K-step descriptor table (see GemmArgs::StepDesc); nsteps = 0 tells the kernels to use the segment selects instead
static void fill_step_table(GemmArgs &a, int64_t rows)
{
    a.nsteps = 0;
    int n = 0;
    for (int sgi = 0; sgi < a.nseg; ++sgi)
        for (int k0 = 0; k0 < a.seg[sgi].width; k0 += BK, ++n) {
            if (n >= 40 || (size_t)rows * a.seg[sgi].ld * 4 >= ((size_t)1 << 32) || a.seg[sgi].ld * 4 >= (1 << 24) || rows >= (1 << 24)) return;
            GemmArgs::StepDesc &d = a.steps[n];
            d.ptr = a.seg[sgi].ptr + k0; d.scale = a.seg[sgi].scale + k0; d.shift = a.seg[sgi].shift + k0;
            d.ld4 = a.seg[sgi].ld * 4; d.sld = a.seg[sgi].sld; d.mode = a.seg[sgi].mode;
        }
    if ((size_t)a.ntn * BN * a.ldw * 2 >= ((size_t)1 << 32)) return;
    a.nsteps = n;
}

static int run_gemm(pifu_ctx *ctx, GemmArgs &a, double flops, hipStream_t st)
{
    const int nblocks = (int)round_up(a.ntm, 8) * a.ntn;
    a.ktot = 0;
    for (int sgi = 0; sgi < a.nseg; ++sgi) a.ktot += a.seg[sgi].width;
    fill_step_table(a, (int64_t)a.ntm * BM);
    if (ctx->precision != 0) {
        int rc = split_planes(ctx, a.W, (size_t)a.ntn * BN * a.ldw, &a.Whi, &a.Wlo, st);
        if (rc) return rc;
        hipEvent_t e1b;
        rc = pifu_prof_begin(ctx, PIFU_PROF_GEMM, st, &e1b);
        if (rc) return rc;
        if (ctx->precision == 2 && a.qrel) hipLaunchKernelGGL((k_mlp_gemm_split<1, 1>), dim3(nblocks), dim3(256), 0, st, a);
        else if (ctx->precision == 2) hipLaunchKernelGGL((k_mlp_gemm_split<1, 0>), dim3(nblocks), dim3(256), 0, st, a);
        else if (a.qrel) hipLaunchKernelGGL((k_mlp_gemm_split<0, 1>), dim3(nblocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_mlp_gemm_split<0, 0>), dim3(nblocks), dim3(256), 0, st, a);
        return pifu_prof_end(ctx, PIFU_PROF_GEMM, e1b, flops, st);
    }
    hipEvent_t e1;
    int rc = pifu_prof_begin(ctx, PIFU_PROF_GEMM, st, &e1);
    if (rc) return rc;
    launch_fp32_gemm(ctx, a, st);
    return pifu_prof_end(ctx, PIFU_PROF_GEMM, e1, flops, st);
}

// per-column sampled features and products for columns [col_base, col_base + ncols)
static int build_column_tables(pifu_ctx *ctx, bool fine, const SampleArgs &proto, int ldc, int ldf, int64_t col_base,
                               int64_t ncols, ColTables &T, hipStream_t st)
{
    for (int L = 0; L < (fine ? 2 : 1); ++L) {
        int rc = build_column_weights(ctx, L);
        if (rc) return rc;
    }
    T.col_base = col_base; T.ncols = ncols; T.ncols_pad = round_up(ncols, BM);
    // the tables of an earlier call on the same inputs (the previous level of the same octree evaluation, typically)
    pifu_ctx::ColCache &cc = ctx->col_cache;
    bool hit = cc.valid && cc.fine == fine && cc.ldc == ldc && cc.ldf == ldf && cc.precision == ctx->precision &&
               cc.col_base == col_base && cc.ncols == ncols && ctx->col_arena.p != nullptr;
    for (int k = 0; hit && k < 3; ++k) hit = cc.res[k] == proto.src.res[k];
    for (int k = 0; hit && k < 12; ++k) hit = cc.calib[k] == proto.calib[k] && cc.cinv[k] == proto.src.cinv[k];
    cc.valid = false;
    Carver c{nullptr};
    for (int pass = 0; pass < 2; ++pass) {
        c = Carver{pass ? (char *)ctx->col_arena.p : nullptr};
        T.featc = c.take<float>((size_t)T.ncols_pad * ldc);
        T.finef = fine ? c.take<float>((size_t)T.ncols_pad * ldf) : nullptr;
        T.mask = c.take<uint8_t>((size_t)T.ncols_pad);
        for (int L = 0; L < (fine ? 2 : 1); ++L) {
            T.ldq[L] = ctx->mlp[L].col_rows;
            T.Q[L] = c.take<float>((size_t)T.ncols_pad * T.ldq[L]);
        }
        if (!pass) {
            if (c.off + 256 > ctx->col_arena.bytes) hit = false;      // (cannot happen for a hit: same sizes as when it was built)
            int rc = pifu_ensure(ctx->col_arena, c.off + 256);
            if (rc) return rc;
        }
    }
    if (hit) { cc.valid = true; return PIFU_OK; }
    ChunkGeom g;
    g.tpc = (int)(T.ncols_pad / BM); g.chunk = (int)ncols; g.last_len = (int)ncols; g.nchunks = 1;
    SampleArgs sa = proto;
    sa.src.mode = 4;                 // point = column representative (k = 0)
    sa.src.first = col_base;
    sa.featc = T.featc; sa.ldc = ldc; sa.finef = T.finef; sa.ldf = ldf; sa.mask = T.mask;
    sa.rows = T.ncols_pad; sa.g = g;
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((sa.rows + 3) / 4)), dim3(256), 0, st, sa);
    for (int L = 0; L < (fine ? 2 : 1); ++L) {
        const Mlp &m = ctx->mlp[L];
        GemmArgs a{};
        a.seg[0] = Seg{L == 0 ? T.featc : T.finef, ctx->ident_scale, ctx->ident_shift, L == 0 ? ldc : ldf, m.col_k, 0, 0};
        a.nseg = 1;
        a.W = m.colW; a.ldw = m.col_k; a.bias = ctx->ident_shift;      // zeros
        a.Y = T.Q[L]; a.ldy = T.ldq[L]; a.partial = nullptr;
        a.ntm = g.tpc; a.ntn = m.col_rows / BN; a.g = g;
        double real_rows = 0;
        for (int i = 0; i < m.nlayers; ++i) if (m.layer[i].cin_feat) real_rows += m.layer[i].cout;
        const int cfeat = L == 0 ? m.dims[0] - 1 : m.dims[0] - ctx->mlp[0].dims[ctx->mlp[0].merge_layer + 1];
        int rc = run_gemm(ctx, a, 2.0 * ncols * real_rows * cfeat, st);
        if (rc) return rc;
    }
    PIFU_HIP(hipGetLastError());
    cc.fine = fine; cc.ldc = ldc; cc.ldf = ldf; cc.precision = ctx->precision; cc.col_base = col_base; cc.ncols = ncols;
    for (int k = 0; k < 3; ++k) cc.res[k] = proto.src.res[k];
    for (int k = 0; k < 12; ++k) { cc.calib[k] = proto.calib[k]; cc.cinv[k] = proto.src.cinv[k]; }
    cc.valid = true;
    return PIFU_OK;
}

// Pixel tables (explicit point lists; see k_expand_pix): G[pixel][col_rows] = colW * F[pixel] over the pixels of the coarse
// map, the same stacked weights as the column tables.  Built on first use after the map, the coarse MLP or the precision
// mode changed (their setters drop `ctx->pix.valid`); 12.9 GFLOP for the 128^2 x 256 map.
struct PixMode {
    const float *G; int ldg;
    int g1;          // first G column of the skip blocks (= width of layer 0's block)
    int nS;          // 128-column skip blocks the call needs (those of the layers it runs)
};

static int build_pixel_table(pifu_ctx *ctx, hipStream_t st)
{
    int rc = build_column_weights(ctx, 0);
    if (rc) return rc;
    const Mlp &m = ctx->mlp[0];
    const FeatMap &f = ctx->feat[0];
    pifu_ctx::PixState &px = ctx->pix;
    // (a table another stream enqueued a moment ago, or one other streams are still blending from while it is rebuilt in place
    // for a new map: the calls themselves are ordered across streams, PIFU_USE)
    if (px.valid && px.feat == f.nhwc && px.H == f.H && px.W == f.W && px.C == f.C && px.precision == ctx->precision &&
        px.ldg == m.col_rows)
        return PIFU_OK;
    const int64_t hw = (int64_t)f.H * f.W, hw_pad = round_up(hw, BM);
    const size_t gbytes = ((size_t)hw_pad * m.col_rows * sizeof(float) + 255) & ~(size_t)255;
    const size_t fbytes = hw != hw_pad ? (size_t)hw_pad * f.Cpad * sizeof(float) : 0;     // row tiles must not read past the map
    rc = pifu_ensure(ctx->pix_arena, gbytes + fbytes + 256);
    if (rc) return rc;
    float *G = (float *)ctx->pix_arena.p;
    const float *A = f.nhwc;
    if (fbytes) {
        float *Fp = (float *)((char *)ctx->pix_arena.p + gbytes);
        PIFU_HIP(hipMemcpyAsync(Fp, f.nhwc, (size_t)hw * f.Cpad * sizeof(float), hipMemcpyDeviceToDevice, st));
        PIFU_HIP(hipMemsetAsync(Fp + (size_t)hw * f.Cpad, 0, (size_t)(hw_pad - hw) * f.Cpad * sizeof(float), st));
        A = Fp;
    }
    ChunkGeom g;
    g.tpc = (int)(hw_pad / BM); g.chunk = (int)hw; g.last_len = (int)hw; g.nchunks = 1;
    GemmArgs a{};
    a.seg[0] = Seg{A, ctx->ident_scale, ctx->ident_shift, f.Cpad, m.col_k, 0, 0};
    a.nseg = 1;
    a.W = m.colW; a.ldw = m.col_k; a.bias = ctx->ident_shift;      // zeros
    a.Y = G; a.ldy = m.col_rows; a.partial = nullptr;
    a.ntm = g.tpc; a.ntn = m.col_rows / BN; a.g = g;
    double real_rows = 0;
    for (int i = 0; i < m.nlayers; ++i) if (m.layer[i].cin_feat) real_rows += m.layer[i].cout;
    rc = run_gemm(ctx, a, 2.0 * hw * real_rows * (double)(m.dims[0] - 1), st);
    if (rc) return rc;
    px.valid = true; px.feat = f.nhwc; px.H = f.H; px.W = f.W; px.C = f.C; px.precision = ctx->precision; px.ldg = m.col_rows;
    return PIFU_OK;
}

static int launch_level_col(pifu_ctx *ctx, int L, const Plan &pl, const ColTables &T, const ChunkGeom &g, int64_t rows,
                            int mask_bit, float *pred_out, int upto_layer, hipStream_t st, int walk_nseg = 0,
                            const PixMode *px = nullptr /* coarse level from the pixel tables instead of column tables */)
{
    const bool walk = L == 0 && walk_nseg > 0;      // coarse layer 1 by the depth walk (walk.inc)
    const Mlp &m = ctx->mlp[L];
    const LevelBufs &b = pl.lv[L];
    const Mlp &mc = ctx->mlp[0];
    const int ntm = (int)(rows / BM);
    const bool stats = (m.norm == PIFU_NORM_GROUP || m.norm == PIFU_NORM_BATCH_TRAIN);
    const double valid = (double)(g.nchunks - 1) * g.chunk + g.last_len;
    bool fused_last = false;      // the Cout = 1 layer went into the previous layer's epilogue (fuse_last_layer)
    for (int i = 0; i < m.nlayers && i < upto_layer; ++i) {
        const MlpLayer &l = m.layer[i];
        Seg segs[3];
        int ns = 0;
        if (i > 0) segs[ns++] = make_prev_seg(ctx, m, b, i - 1);
        if (L == 1 && l.cin_feat) segs[ns++] = make_prev_seg(ctx, mc, pl.lv[0], mc.merge_layer);      // phi
        int ksum = 0;
        for (int s = 0; s < ns; ++s) ksum += segs[s].width;
        PIFU_REQUIRE(ksum == l.kpad_col, PIFU_ERR_STATE, "internal: column-mode K mismatch at level %d layer %d (%d vs %d)", L, i, ksum, l.kpad_col);
        if (i < m.nlayers - 1) {
            if (ns == 0 && walk && stats) {
                // coarse layer 0 under the depth walk: nobody reads its output, and its statistics have a closed form per
                // column segment (flip_d is free until k_seg_prepare, which runs after this layer's k_norm_finalize)
                const WalkPlan &w = pl.wk;
                hipLaunchKernelGGL(k_seg_stats0, dim3(walk_nseg), dim3(256), 0, st, (const float *)T.Q[L] + (l.col_off > 0 ? l.col_off : 0),
                                   T.ldq[L], (const float *)l.wz, (const float *)l.bias, (const float *)pl.zf, (const int *)pl.colidx,
                                   (const int *)w.seg_row0, walk_nseg, g, l.cout_pad, w.flip_d);
                const int groups = m.norm == PIFU_NORM_GROUP ? 32 : l.cout;
                hipLaunchKernelGGL(k_norm_finalize, dim3(g.nchunks * groups), dim3(256), 0, st, (const float2 *)w.flip_d,
                                   l.cout_pad, l.cout / groups, groups, (const float *)l.gamma, (const float *)l.beta,
                                   b.scale[i], b.shift[i], g, 1e-5f, 1, (const int *)w.chunk_seg0);
                continue;
            } else if (ns == 0 && px) {   // coarse layer 0 (+ the skip products of the later layers) blended from the pixel table
                PixArgs pa{};
                pa.G = px->G; pa.ldg = px->ldg; pa.g0 = l.col_off > 0 ? l.col_off : 0; pa.g1 = px->g1;
                pa.idx = pl.pix_idx; pa.wt = pl.pix_wt; pa.wz = l.wz; pa.bias = l.bias; pa.zf = pl.zf;
                pa.Y = b.out[i]; pa.ldy = l.cout_pad; pa.partial = stats ? b.partial[i] : nullptr;
                pa.n0 = l.cout_pad / 128; pa.S = pl.S; pa.lds = pl.ldS; pa.g = g;
                hipLaunchKernelGGL(k_expand_pix, dim3(ntm, pa.n0 + px->nS), dim3(256), 0, st, pa);
            } else if (ns == 0) {   // coarse layer 0: no per-point product left
                // with the depth walk nobody reads this layer's output: statistics only
                if (walk && !stats) continue;      // pointwise norm + depth walk: neither the output nor statistics are needed
                hipLaunchKernelGGL(k_expand, dim3(ntm, l.cout_pad / 128), dim3(256), 0, st, (const float *)T.Q[L], T.ldq[L],
                                   (const float *)l.wz, (const float *)l.bias, (const float *)pl.zf, (const int *)pl.colidx,
                                   walk ? (float *)nullptr : b.out[i], l.cout_pad, stats ? b.partial[i] : nullptr, g);
            } else if (walk && i == 1) {
                const MlpLayer &l0 = m.layer[0];
                const WalkPlan &w = pl.wk;
                const int C0 = l0.cout_pad, N1 = l.cout_pad;
                const int nseg = walk_nseg, nseg_pad = (int)round_up(nseg, BM);
                const Seg prev = make_prev_seg(ctx, m, b, 0);
                PrepArgs pa{};
                pa.Q = T.Q[0] + (l0.col_off > 0 ? l0.col_off : 0); pa.ldq = T.ldq[0];
                pa.wz = l0.wz; pa.bias0 = l0.bias; pa.scale = prev.scale; pa.shift = prev.shift; pa.sld = prev.sld;
                pa.zf = pl.zf; pa.colidx = pl.colidx; pa.seg_row0 = w.seg_row0; pa.nseg = nseg; pa.nseg_pad = nseg_pad;
                pa.g = g; pa.C0 = C0; pa.V = w.V; pa.flip_key = w.flip_key; pa.flip_d = w.flip_d; pa.nflips = w.nflips;
                hipLaunchKernelGGL(k_seg_prepare, dim3(nseg), dim3(256), 0, st, pa);
                GemmArgs a{};       // start values (P, R) of every segment: [m alpha ; m beta] * W1^T
                a.seg[0] = Seg{w.V, ctx->ident_scale, ctx->ident_shift, C0, C0, 0, 0};
                a.nseg = 1;
                a.W = l.w_col; a.ldw = l.kpad_col; a.bias = ctx->ident_shift;
                a.Y = w.PR0; a.ldy = N1; a.partial = nullptr;
                a.ntm = 2 * nseg_pad / BM; a.ntn = N1 / BN;
                a.g.tpc = a.ntm; a.g.chunk = 2 * nseg_pad; a.g.last_len = 2 * nseg_pad; a.g.nchunks = 1;
                int rc = run_gemm(ctx, a, 2.0 * 2.0 * nseg * (double)l.cout * (double)l.cin_col, st);
                if (rc) return rc;
                WalkArgs wa{};
                wa.PR0 = w.PR0; wa.nseg_pad = nseg_pad; wa.W1T = l.w_colT; wa.bias1 = l.bias;
                wa.flip_key = w.flip_key; wa.flip_d = w.flip_d; wa.nflips = w.nflips;
                wa.zf = pl.zf; wa.seg_row0 = w.seg_row0; wa.nseg = nseg; wa.g = g; wa.C0 = C0; wa.N1 = N1;
                wa.Y = b.out[i]; wa.ldy = l.cout_pad; wa.partial_seg = stats ? w.partial_seg : nullptr;
                // timed for bench.py (PIFU_PROF_WALK): work = the bytes it must move - y1 written (rows x N1 floats) + the depth
                // features read; the weight rows of the flips are L2 traffic
                hipEvent_t walk_stop;
                rc = pifu_prof_begin(ctx, PIFU_PROF_WALK, st, &walk_stop);
                if (rc) return rc;
#ifdef PIFU_LAB     // tools/walk_ab.sh
                const int wv = ctx->walk_variant;
                if (wv == 1) hipLaunchKernelGGL(k_walk_v1<1>, dim3(nseg, N1 / 256), dim3(256), 0, st, wa);
                else if (wv == 3) hipLaunchKernelGGL((k_walk<8, 2>), dim3(nseg, N1 / 512), dim3(256), 0, st, wa);
                else if (wv == 4) hipLaunchKernelGGL((k_walk<16, 1>), dim3(nseg, N1 / 256), dim3(256), 0, st, wa);
                else if (wv == 5) hipLaunchKernelGGL((k_walk<32, 1>), dim3(nseg, N1 / 256), dim3(256), 0, st, wa);
                else if (wv == 6) hipLaunchKernelGGL((k_walk_pair<16>), dim3(nseg, N1 / 512), dim3(256), 0, st, wa);
                else if (wv == 8) hipLaunchKernelGGL((k_walk<32, 2>), dim3(nseg, N1 / 512), dim3(256), 0, st, wa);
                else if (wv == 9 && N1 % 512 == 0) hipLaunchKernelGGL((k_walk<16, 4, 128>), dim3(nseg, N1 / 512), dim3(128), 0, st, wa);
                else if (wv == 10 && N1 % 512 == 0) hipLaunchKernelGGL((k_walk<8, 4, 128>), dim3(nseg, N1 / 512), dim3(128), 0, st, wa);
                else if (wv == 11 && N1 % 256 == 0) hipLaunchKernelGGL((k_walk<16, 2, 128>), dim3(nseg, N1 / 256), dim3(128), 0, st, wa);
                else
#endif
                if (N1 % 512 == 0) hipLaunchKernelGGL((k_walk<16, 2>), dim3(nseg, N1 / 512), dim3(256), 0, st, wa);
                else hipLaunchKernelGGL((k_walk<16, 1>), dim3(nseg, N1 / 256), dim3(256), 0, st, wa);
                rc = pifu_prof_end(ctx, PIFU_PROF_WALK, walk_stop, (double)rows * ((double)N1 * 4.0 + 4.0), st);
                if (rc) return rc;
                if (stats) {
                    const int groups = m.norm == PIFU_NORM_GROUP ? 32 : l.cout;
                    hipLaunchKernelGGL(k_norm_finalize, dim3(g.nchunks * groups), dim3(256), 0, st, (const float2 *)w.partial_seg,
                                       l.cout_pad, l.cout / groups, groups, (const float *)l.gamma, (const float *)l.beta,
                                       b.scale[i], b.shift[i], g, 1e-5f, 1, (const int *)w.chunk_seg0);
                }
                continue;
            } else {
                GemmArgs a{};
                for (int s = 0; s < ns; ++s) a.seg[s] = segs[s];
                a.nseg = ns;
                a.W = l.w_col; a.ldw = l.kpad_col; a.bias = l.bias;
                a.Y = b.out[i]; a.ldy = l.cout_pad;
                a.partial = stats ? b.partial[i] : nullptr;
                a.ntm = ntm; a.ntn = l.cout_pad / BN; a.g = g;
                if (l.cin_feat && px) {
                    a.colQ = pl.S; a.ldq = pl.ldS; a.qoff = l.col_off - px->g1; a.qrel = 1; a.colwz = L == 0 ? l.wz : nullptr;
                    a.zf = pl.zf;
                } else if (l.cin_feat) {
                    a.colQ = T.Q[L]; a.ldq = T.ldq[L]; a.qoff = l.col_off; a.colwz = L == 0 ? l.wz : nullptr;
                    a.zf = pl.zf; a.colidx = pl.colidx;
                }
                if (!a.qrel && fuse_last_layer(ctx, m, i, pred_out, upto_layer, true)) {
                    const MlpLayer &ll = m.layer[i + 1];
                    a.last_w = ll.w_col; a.last_bias = ll.host_bias0; a.last_mask = pl.mask; a.last_mask_bit = mask_bit; a.last_out = pred_out;
                    fused_last = true;
                    if (ll.cin_feat && px) {
                        a.last_colQ = pl.S; a.last_ldq = pl.ldS; a.last_qoff = ll.col_off - px->g1; a.last_qrel = 1;
                        a.last_wz0 = L == 0 ? ll.host_wz0 : 0.f; a.last_zf = pl.zf;
                    } else if (ll.cin_feat) {
                        a.last_colQ = T.Q[L]; a.last_ldq = T.ldq[L]; a.last_qoff = ll.col_off; a.last_qrel = 0;
                        a.last_wz0 = L == 0 ? ll.host_wz0 : 0.f; a.last_zf = pl.zf; a.last_colidx = pl.colidx;
                    }
                }
                int rc = run_gemm(ctx, a, 2.0 * valid * l.cout * (double)l.cin_col, st);
                if (rc) return rc;
            }
            if (stats) {
                const int groups = m.norm == PIFU_NORM_GROUP ? 32 : l.cout;
                const int cpg = l.cout / groups;
                const int nw = g.nchunks * groups;
                hipLaunchKernelGGL(k_norm_finalize, dim3(nw), dim3(256), 0, st, (const float2 *)b.partial[i],
                                   l.cout_pad, cpg, groups, (const float *)l.gamma, (const float *)l.beta, b.scale[i],
                                   b.shift[i], g, 1e-5f, 1);
            }
        } else {
            if (!pred_out) continue;
            if (fused_last) continue;      // fused: the previous layer's epilogue wrote the prediction
            LastArgs a{};
            for (int s = 0; s < ns; ++s) a.seg[s] = segs[s];
            a.nseg = ns;
            a.w = l.w_col; a.bias = l.host_bias0; a.mask = pl.mask; a.mask_bit = mask_bit; a.out = pred_out; a.rows = rows; a.g = g;
            if (l.cin_feat && px) {
                a.colQ = pl.S; a.ldq = pl.ldS; a.qoff = l.col_off - px->g1; a.qrel = 1; a.wz0 = L == 0 ? l.host_wz0 : 0.f;
                a.zf = pl.zf;
            } else if (l.cin_feat) {
                a.colQ = T.Q[L]; a.ldq = T.ldq[L]; a.qoff = l.col_off; a.wz0 = L == 0 ? l.host_wz0 : 0.f;
                a.zf = pl.zf; a.colidx = pl.colidx;
            }
            launch_last_layer(a, st);
        }
    }
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

// Pointwise MLP norms - none at all (MLP.py:37-41,65-69: what the published pifuhd.pt uses) or BatchNorm1d on running
// statistics - make a point's result independent of the query() call it is evaluated in: the reference's chunk size
// (num_samples) then only decides how the work is CUT, and the library cuts it its own way: whole 128-row tiles (no
// padding rows: 5 000 -> 5 120 costs 2.4 %), for dense lattices whole columns too (every depth-walk segment is then a
// complete column), no statistics epilogue / k_norm_finalize launches.  The grouping depends on (chunk, n, res[2]) only -
// never on the batch, the rank or the number of points - and the sharded evaluation deals out whole groups
// (pifu_effective_chunk), so N ranks reproduce one rank's field bit for bit.
static inline bool norm_pointwise(const Mlp &m) { return m.norm == PIFU_NORM_NONE || m.norm == PIFU_NORM_BATCH_EVAL; }

static int64_t effective_chunk(const pifu_ctx *ctx, int64_t chunk, int64_t n, int res2, bool fine, bool lattice)
{
    const bool pointwise = norm_pointwise(ctx->mlp[0]) && (!fine || norm_pointwise(ctx->mlp[1]));
    // lattice sources are always regrouped (the grouping must not depend on how many points a rank's share holds); explicit
    // point lists only when they span several chunks (a single-chunk pifu_query may export phi)
    if (!pointwise || (!lattice && n <= chunk)) return chunk;
    int64_t unit = BM;
    if (res2 > 0) {         // lcm(128, res2): whole tiles and whole columns
        int64_t a = BM, b = res2;
        while (b) { const int64_t t = a % b; a = b; b = t; }
        unit = (int64_t)BM / a * res2;
        // columns whose lcm with the tile is far beyond the group size (1 021 nodes: 130 688 rows) are cut like a point list:
        // whole columns are a convenience of the depth walk, whole tiles are what the grouping is for
        if (unit > 4 * (int64_t)PIFU_POINTWISE_CHUNK) unit = BM;
    }
    int64_t k = (PIFU_POINTWISE_CHUNK + unit / 2) / unit;
    return unit * (k < 1 ? 1 : k);
}

int64_t pifu_effective_chunk(pifu_ctx *ctx, int64_t chunk, int res2, int flags)
{
    if (!ctx || chunk <= 0) return chunk;
    return effective_chunk(ctx, chunk, 0, res2, !(flags & PIFU_Q_COARSE_ONLY), true);
}

int pifu_query_impl(pifu_ctx *ctx, int src_mode, const float *points, int64_t n, const int32_t *res,
                    const double *calib_inv, const int64_t *index, int64_t first, const float *calib,
                    int64_t chunk, int flags, float *pred_fine, float *pred_coarse, float *phi,
                    float delta, float **normal_pred_out, hipStream_t st)
{
    PIFU_REQUIRE(ctx, PIFU_ERR_INVALID, "null context");
    PIFU_REQUIRE(n >= 0 && chunk > 0, PIFU_ERR_INVALID, "query: need n >= 0 and chunk > 0");
    if (n == 0) return PIFU_OK;
    PIFU_USE(ctx, st);
    {
        const int rc = pifu_enc_join(ctx, st);      // a coarse hourglass left running by pifu_filter_coarse(PIFU_ENC_DEFER_JOIN)
        if (rc) return rc;
    }
    const bool coarse_only = flags & PIFU_Q_COARSE_ONLY;
    const bool fine = !coarse_only;
    PIFU_REQUIRE(ctx->mlp[0].defined && ctx->feat[0].nhwc, PIFU_ERR_STATE,
                 "query before the coarse MLP / feature map were set (filter must precede query)");
    PIFU_REQUIRE(!fine || (ctx->mlp[1].defined && ctx->feat[1].nhwc), PIFU_ERR_STATE,
                 "fine-level query before the fine MLP / feature map were set");
    for (int L = 0; L < (fine ? 2 : 1); ++L)
        for (int i = 0; i < ctx->mlp[L].nlayers; ++i)
            PIFU_REQUIRE(ctx->mlp[L].layer[i].set, PIFU_ERR_STATE, "MLP level %d layer %d has no weights", L, i);
    const Mlp &mc = ctx->mlp[0];
    PIFU_REQUIRE(ctx->feat[0].C + 1 == mc.dims[0], PIFU_ERR_INVALID,
                 "coarse MLP expects %d inputs but the feature map has %d channels (+1 depth)", mc.dims[0], ctx->feat[0].C);
    if (fine)
        PIFU_REQUIRE(ctx->feat[1].C + mc.dims[mc.merge_layer + 1] == ctx->mlp[1].dims[0], PIFU_ERR_INVALID,
                     "fine MLP expects %d inputs but fine features (%d) + phi (%d) differ", ctx->mlp[1].dims[0],
                     ctx->feat[1].C, mc.dims[mc.merge_layer + 1]);
    PIFU_HIP(hipSetDevice(ctx->device));

    // (the phi export belongs to single-chunk calls, which are never regrouped)
    chunk = effective_chunk(ctx, chunk, n, src_mode == 1 ? res[2] : 0, fine, src_mode == 1 || src_mode == 2);
    const int64_t eff_chunk = chunk < n ? chunk : n;
    const int64_t P = round_up(eff_chunk, BM);
    PIFU_REQUIRE(P <= (int64_t)1 << 30, PIFU_ERR_NOMEM, "chunk of %lld points is too large", (long long)eff_chunk);
    const int64_t total_chunks = (n + chunk - 1) / chunk;
    // rows per batch of launches: every batch pays ~20 small launches and the under-filled last round of each GEMM, so
    // bigger is better until ~2 M rows (measured 89 / 92 / 94 / 95 Mq/s at 0.25 / 0.5 / 1 / 2 M rows; ~12.5 KB of workspace
    // per row: 13 GB at the default, sized for 288 GB of HBM); the arena only ever grows to what a call needs
    int64_t target_rows = ctx->ws_rows > 0 ? ctx->ws_rows : 1048576;
    // calc_normal chunks are 200 000 rows: one per launch leaves the 128-column layers with 3.05 rounds of resident
    // blocks; four per launch fill the chip evenly (about 12.5 KB of workspace per row)
    if (src_mode == 3 && target_rows < 4 * P && 4 * P <= ((int64_t)1 << 20)) target_rows = 4 * P;
    int64_t batch_chunks = target_rows / P;
    if (batch_chunks < 1) batch_chunks = 1;
    if (batch_chunks > total_chunks) batch_chunks = total_chunks;
    int64_t rows_cap = batch_chunks * P;
    const int64_t normal_pts = src_mode == 3 ? n : 0;
    // column sharing applies when the projected (x, y) of a lattice point does not depend on k
    bool colmode = (src_mode == 1 || src_mode == 2) && ctx->col_enable && !(flags & PIFU_Q_NO_COLUMN_SHARING) &&
                   ctx->mlp[0].nlayers >= 2 && !ctx->mlp[0].res[0];
    if (colmode)
        colmode = calib[2] == 0.f && calib[6] == 0.f && calib_inv[2] == 0.0 && calib_inv[6] == 0.0 &&
                  (int64_t)res[0] * res[1] < ((int64_t)1 << 31);
    // depth walk of coarse layer 1 (walk.inc): layer 1 must take the activated layer-0 output only
    bool walk_ok = false;
    if (colmode && ctx->walk_enable && !(flags & PIFU_Q_NO_DEPTH_WALK)) {
        const Mlp &m0 = ctx->mlp[0];
        const int C0 = m0.layer[0].cout_pad, N1 = m0.nlayers >= 3 ? m0.layer[1].cout_pad : 0;
        walk_ok = m0.nlayers >= 3 && !m0.res[1] && m0.layer[0].has_norm && C0 <= 1024 && C0 % 256 == 0 &&
                  (N1 == 256 || N1 == 512 || N1 == 1024) && res[2] < 65536 && (mc.merge_layer >= 1 || !fine) &&
                  rows_cap * 4 < ((int64_t)1 << 31);
    }
    // pixel tables (explicit point lists): the coarse level's image-feature products blended from G = W_feat * F
    // PIFuNetwNML.calc_normal: the normals come from the COARSE net's own (unmasked) occupancies
    const bool coarse_normals = src_mode == 3 && coarse_only;
    const bool want_coarse = pred_coarse != nullptr || (flags & PIFU_Q_FULL_COARSE) || coarse_normals;
    const int upto_c = want_coarse ? mc.nlayers : mc.merge_layer + 1;
    // (also lattice sources whose calibration rules out column sharing: x or y depends on k)
    bool pixmode = !colmode && !(flags & PIFU_Q_NO_COLUMN_SHARING) && ctx->pix_mode != 0 &&
                   (ctx->pix_mode == 2 || (n >= PIFU_PIXEL_TABLE_MIN_POINTS && n >= (int64_t)ctx->feat[0].H * ctx->feat[0].W)) && mc.nlayers >= 2 && !mc.res[0] &&
                   ctx->feat[0].C % BK == 0 && ctx->feat[0].Cpad == ctx->feat[0].C;
#ifdef PIFU_LAB     // the lab kernels (tools/lab) do not know GemmArgs::qrel
    pixmode = false;
#endif
    PixMode pxm{};
    int rc = PIFU_OK;
    if (pixmode) {
        rc = build_pixel_table(ctx, st);
        if (rc != PIFU_OK) return rc;
        pxm.G = (const float *)ctx->pix_arena.p; pxm.ldg = ctx->pix.ldg;
        pxm.g1 = (int)round_up(mc.layer[0].cout, BN);
        int hi = pxm.g1;
        for (int i = 1; i < upto_c; ++i)
            if (mc.layer[i].cin_feat) hi = mc.layer[i].col_off + (int)round_up(mc.layer[i].cout, BN);
        pxm.nS = (hi - pxm.g1) / BN;
    }
    // The workspace of a batch of launches: when it exceeds pifu_set_workspace_limit, or the device cannot provide it, the
    // batch is halved (whole chunks) until it fits - more, smaller batches of launches, the same numbers (a chunk never
    // depends on which batch it runs in).  One chunk is the floor.
    Plan probe;
    for (;;) {
        probe = carve(ctx, nullptr, rows_cap, batch_chunks, fine, normal_pts, walk_ok, pixmode ? pxm.nS * BN : -1);
        // a bound is a bound: an arena that outgrew it (allocated before the bound was set: pifu_set_workspace_limit releases
        // those) is not reused, and the allocation carries no slack
        const bool over = ctx->ws_limit > 0 && probe.bytes > (size_t)ctx->ws_limit;
        rc = over ? PIFU_ERR_NOMEM : pifu_ensure(ctx->arena, probe.bytes, ctx->ws_limit > 0);
        if (rc != PIFU_ERR_NOMEM || batch_chunks <= 1) break;
        batch_chunks = (batch_chunks + 1) / 2;
        rows_cap = batch_chunks * P;
    }
    if (rc == PIFU_ERR_NOMEM && ctx->ws_limit > 0 && probe.bytes > (size_t)ctx->ws_limit)
        pifu_set_error("one chunk of %lld points needs %zu bytes of workspace, the limit is %lld (pifu_set_workspace_limit)",
                       (long long)eff_chunk, probe.bytes, (long long)ctx->ws_limit);
    if (rc != PIFU_OK) return rc;
    Plan pl = carve(ctx, (char *)ctx->arena.p, rows_cap, batch_chunks, fine, normal_pts, walk_ok, pixmode ? pxm.nS * BN : -1);

    float *fine_out = pred_fine;
    if (src_mode == 3) {   // calc_normal consumes the predictions itself
        fine_out = pl.normal_pred;
        if (coarse_normals) pred_coarse = pl.normal_pred;
        if (normal_pred_out) *normal_pred_out = pl.normal_pred;
    }
    const int coarse_mask_bit = (flags & PIFU_Q_NO_COARSE_MASK) ? 0 : 1;
    if ((flags & PIFU_Q_FULL_COARSE) && !pred_coarse) pred_coarse = pl.coarse_scratch;

    ColTables T;
    bool tables_ready = false;

    for (int64_t c0 = 0; c0 < total_chunks; c0 += batch_chunks) {
        const int64_t nc = (total_chunks - c0) < batch_chunks ? (total_chunks - c0) : batch_chunks;
        const int64_t pt0 = c0 * chunk;
        const int64_t npts = (pt0 + nc * chunk <= n) ? nc * chunk : (n - pt0);
        ChunkGeom g;
        g.tpc = (int)(P / BM);
        g.chunk = (int)chunk;
        g.nchunks = (int)nc;
        g.last_len = (int)(npts - (nc - 1) * chunk);
        const int64_t rows = nc * P;

        SampleArgs sa{};
        sa.src.mode = src_mode;
        sa.src.pts = points;
        sa.src.pts_n = src_mode == 3 ? n / 4 : n;
        if (res) { sa.src.res[0] = res[0]; sa.src.res[1] = res[1]; sa.src.res[2] = res[2]; }
        if (calib_inv) for (int k = 0; k < 12; ++k) sa.src.cinv[k] = calib_inv[k];
        sa.src.index = index;
        sa.src.first = first + pt0;
        sa.src.delta = delta;
        for (int k = 0; k < 12; ++k) sa.calib[k] = calib[k];
        sa.fc = ctx->feat[0].nhwc; sa.Cc = ctx->feat[0].C; sa.Ccp = ctx->feat[0].Cpad; sa.Hc = ctx->feat[0].H; sa.Wc = ctx->feat[0].W;
        if (fine) { sa.ff = ctx->feat[1].nhwc; sa.Cf = ctx->feat[1].C; sa.Cfp = ctx->feat[1].Cpad; sa.Hf = ctx->feat[1].H; sa.Wf = ctx->feat[1].W; }
        sa.z_mul = ctx->z_mul; sa.z_div = ctx->z_div;
        sa.featc = pl.featc; sa.ldc = pl.ldc; sa.finef = pl.finef; sa.ldf = pl.ldf; sa.mask = pl.mask;
        sa.rows = rows; sa.g = g;
        // coarse level: all layers when netG.preds is wanted, else only up to the merge layer (phi)
        const bool want_coarse_pred = pred_coarse != nullptr;
        const int upto = want_coarse_pred ? mc.nlayers : mc.merge_layer + 1;
        float *coarse_dst = want_coarse_pred ? (pred_coarse == pl.coarse_scratch ? pred_coarse : pred_coarse + pt0) : nullptr;
        if (colmode) {
            if (!tables_ready) {   // once per call: per-column features and products (inside the timed region)
                int64_t col_lo = 0, col_hi = (int64_t)res[0] * res[1] - 1;
                if (src_mode == 1) { col_lo = first / res[2]; col_hi = (first + n - 1) / res[2]; }
                rc = build_column_tables(ctx, fine, sa, pl.ldc, pl.ldf, col_lo, col_hi - col_lo + 1, T, st);
                if (rc != PIFU_OK) return rc;
                tables_ready = true;
            }
            hipLaunchKernelGGL(k_sample_col, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, sa, pl.zf, pl.colidx, T.col_base);
            int walk_nseg = 0;
            if (walk_ok && ctx->mlp[0].layer[1].w_colT && upto >= 2) {
                // column segments of this batch (ordered compaction of the segment starts)
                const WalkPlan &w = pl.wk;
                const int ntiles = (int)((rows + SEG_TILE - 1) / SEG_TILE);
                hipLaunchKernelGGL(k_seg_count, dim3(ntiles), dim3(256), 0, st, (const int *)pl.colidx, g, rows, w.tile_cnt);
                hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, st, (const int *)w.tile_cnt, ntiles, w.tile_off, w.nseg_dev,
                                   w.chunk_seg0, g.nchunks);
                hipLaunchKernelGGL(k_seg_scatter, dim3(ntiles), dim3(256), 0, st, (const int *)pl.colidx, g, rows,
                                   (const int *)w.tile_off, w.seg_row0, w.chunk_seg0, w.maxseg);
                int64_t nseg = 0;
                if (src_mode == 1) {        // dense range: the count is known on the host
                    for (int64_t c = 0; c < nc; ++c) {
                        const int64_t i0 = first + pt0 + c * chunk;
                        const int64_t len = c == nc - 1 ? g.last_len : chunk;
                        nseg += (i0 + len - 1) / res[2] - i0 / res[2] + 1;
                    }
                } else {
                    int host_n = 0;
                    PIFU_HIP(hipMemcpyAsync(&host_n, w.nseg_dev, sizeof(int), hipMemcpyDeviceToHost, st));
                    PIFU_HIP(hipStreamSynchronize(st));
                    nseg = host_n;
                }
                // worth it only for segments of some length (start values cost 2 C0 N1 MACs per segment)
                if (nseg > 0 && nseg <= w.maxseg && nseg * 8 <= npts) walk_nseg = (int)nseg;
            }
            rc = launch_level_col(ctx, 0, pl, T, g, rows, coarse_mask_bit, coarse_dst, upto, st, walk_nseg);
            if (rc != PIFU_OK) return rc;
            if (fine) {
                rc = launch_level_col(ctx, 1, pl, T, g, rows, (flags & PIFU_Q_NO_FINE_MASK) ? 0 : 2,
                                      fine_out ? fine_out + pt0 : nullptr, ctx->mlp[1].nlayers, st);
                if (rc != PIFU_OK) return rc;
            }
            continue;
        }
        // the chunk geometry indexes points relative to the batch start: src.first already points there
        rc = build_column_weights(ctx, 0);       // wz of the coarse layers (launch_level: the z step as a rank-1 start value)
        if (rc != PIFU_OK) return rc;
        {
            hipEvent_t es;
            rc = pifu_prof_begin(ctx, PIFU_PROF_SAMPLER, st, &es);
            if (rc != PIFU_OK) return rc;
            sa.zf = pl.zf;
            sa.pix_idx = pl.pix_idx; sa.pix_wt = pl.pix_wt;      // pixel tables: taps instead of gathered coarse features
            if (pixmode) hipLaunchKernelGGL(k_sample_pix, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, st, sa);
            else hipLaunchKernelGGL(k_sample, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, sa);
            // algorithmic bytes of the per-point sampler: 4 bilinear taps x (coarse + fine channels) gathered, one MLP
            // input row (coarse + fine) written, per valid point (pixel tables: fine channels + 32 B of taps only)
            const double per_point = pixmode ? 4.0 * (fine ? sa.Cfp : 0) * 4.0 + (double)pl.ldf * 4.0 + 32.0
                                             : 4.0 * (sa.Ccp + (fine ? sa.Cfp : 0)) * 4.0 + (double)(pl.ldc + pl.ldf) * 4.0;
            rc = pifu_prof_end(ctx, PIFU_PROF_SAMPLER, es, per_point * (double)npts, st);
            if (rc != PIFU_OK) return rc;
        }
        if (pixmode) rc = launch_level_col(ctx, 0, pl, T, g, rows, coarse_mask_bit, coarse_dst, upto, st, 0, &pxm);
        else rc = launch_level(ctx, 0, pl, g, rows, pl.mask, coarse_mask_bit, coarse_dst, upto, st);
        if (rc != PIFU_OK) return rc;
        if (phi && total_chunks == 1) {
            const MlpLayer &lm = mc.layer[mc.merge_layer];
            Seg s = make_prev_seg(ctx, mc, pl.lv[0], mc.merge_layer);
            dim3 grid((unsigned)((npts + 31) / 32), (lm.cout + 31) / 32);
            hipLaunchKernelGGL(k_phi_export, grid, dim3(256), 0, st, (const float *)pl.lv[0].out[mc.merge_layer], lm.cout_pad,
                               s.scale, s.shift, s.mode, lm.cout, npts, phi);
        }
        if (fine) {
            rc = launch_level(ctx, 1, pl, g, rows, pl.mask, (flags & PIFU_Q_NO_FINE_MASK) ? 0 : 2,
                              fine_out ? fine_out + pt0 : nullptr, ctx->mlp[1].nlayers, st);
            if (rc != PIFU_OK) return rc;
        }
    }
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

int pifu_reserve_auto(pifu_ctx *ctx, int64_t max_rows, double share, int64_t *rows_out)
{
    PIFU_REQUIRE(ctx && max_rows > 0 && share > 0.0 && share <= 1.0, PIFU_ERR_INVALID, "pifu_reserve_auto: bad argument");
    PIFU_REQUIRE(ctx->mlp[0].defined, PIFU_ERR_STATE, "pifu_reserve_auto: define the MLPs first (the workspace per row follows from them)");
    PIFU_HIP(hipSetDevice(ctx->device));
    size_t free_b = 0, total_b = 0;
    PIFU_HIP(hipMemGetInfo(&free_b, &total_b));
    const bool fine = ctx->mlp[1].defined;
    const Mlp &mc = ctx->mlp[0];
    const bool walk = mc.nlayers >= 3 && !mc.res[1] && mc.layer[0].cout_pad % 256 == 0;
    // what the arena already holds is reused, not added: it counts as free for this purpose
    const double budget = share * ((double)free_b + (double)ctx->arena.bytes);
    int64_t skip_cols = 0;
    for (int i = 1; i < mc.nlayers; ++i) if (mc.layer[i].cin_feat) skip_cols += round_up(mc.layer[i].cout, BN);
    const auto need = [&](int64_t rows) {
        const int64_t chunks = rows / BM;      // worst case: one chunk per tile
        const size_t a = carve(ctx, nullptr, rows, chunks, fine, 0, walk, -1).bytes;
        const size_t b = carve(ctx, nullptr, rows, chunks, fine, rows, false, (int)skip_cols).bytes;
        return (double)(a > b ? a : b);
    };
    int64_t lo = 1, hi = round_up(max_rows, BM) / BM;      // in tiles
    if (need(hi * BM) <= budget) lo = hi;
    else
        while (lo + 1 < hi) {
            const int64_t mid = (lo + hi) / 2;
            if (need(mid * BM) <= budget) lo = mid; else hi = mid;
        }
    ctx->ws_rows = lo * BM;
    if (rows_out) *rows_out = ctx->ws_rows;
    return PIFU_OK;
}

int pifu_query(pifu_ctx *ctx, const float *points, int64_t n, const float *calib, int64_t chunk, int flags,
               float *pred_fine, float *pred_coarse, float *phi, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && points && calib, PIFU_ERR_INVALID, "pifu_query: null argument");
    PIFU_REQUIRE(!phi || n <= chunk, PIFU_ERR_INVALID, "pifu_query: phi is only defined for a single chunk (n <= chunk)");
    return pifu_query_impl(ctx, 0, points, n, nullptr, nullptr, nullptr, 0, calib, chunk, flags, pred_fine, pred_coarse,
                           phi, 0.f, nullptr, (hipStream_t)stream);
}

int pifu_eval_lattice(pifu_ctx *ctx, const int32_t res[3], const double *calib_inv, const float *calib,
                      const int64_t *index, int64_t first, int64_t count, int64_t chunk, int flags, float *pred,
                      pifu_stream stream)
{
    PIFU_REQUIRE(ctx && res && calib_inv && calib && pred, PIFU_ERR_INVALID, "pifu_eval_lattice: null argument");
    PIFU_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, PIFU_ERR_INVALID, "pifu_eval_lattice: bad resolution");
    const bool coarse_only = flags & PIFU_Q_COARSE_ONLY;
    return pifu_query_impl(ctx, index ? 2 : 1, nullptr, count, res, calib_inv, index, first, calib, chunk, flags,
                           coarse_only ? nullptr : pred, coarse_only ? pred : nullptr, nullptr, 0.f, nullptr, (hipStream_t)stream);
}

// shared tail of the two calc_normal entry points: differences of the 4n occupancies -> unit normals; optional export of
// the occupancies themselves (what PIFuMRNet.calc_normal holds in `pred`, PIFuMRNet.py:223-225)
static int normals_from_pred4(const float *pred4, int64_t n, float *normals, float *pred4_out, hipStream_t st)
{
    if (pred4_out) PIFU_HIP(hipMemcpyAsync(pred4_out, pred4, (size_t)4 * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pred4, n, normals);
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

int pifu_calc_normal(pifu_ctx *ctx, const float *verts, int64_t n, const float *calib, float delta, float *normals,
                     float *pred4_out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && verts && calib && normals, PIFU_ERR_INVALID, "pifu_calc_normal: null argument");
    PIFU_REQUIRE(n > 0, PIFU_ERR_INVALID, "pifu_calc_normal: empty vertex batch (the reference raises here too)");
    float *pred4 = nullptr;
    int rc = pifu_query_impl(ctx, 3, verts, 4 * n, nullptr, nullptr, nullptr, 0, calib, 4 * n, PIFU_Q_NO_FINE_MASK,
                             nullptr, nullptr, nullptr, delta, &pred4, (hipStream_t)stream);
    if (rc != PIFU_OK) return rc;
    return normals_from_pred4(pred4, n, normals, pred4_out, (hipStream_t)stream);
}

// PIFuNetwNML.calc_normal (PIFuNetwNML.py:181-218): the same finite differences on the COARSE net's occupancies - one query of
// the 4n interleaved points through the coarse MLP alone, prediction not masked by the bounding box
int pifu_calc_normal_coarse(pifu_ctx *ctx, const float *verts, int64_t n, const float *calib, float delta, float *normals,
                            float *pred4_out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && verts && calib && normals, PIFU_ERR_INVALID, "pifu_calc_normal_coarse: null argument");
    PIFU_REQUIRE(n > 0, PIFU_ERR_INVALID, "pifu_calc_normal_coarse: empty vertex batch (the reference raises here too)");
    float *pred4 = nullptr;
    int rc = pifu_query_impl(ctx, 3, verts, 4 * n, nullptr, nullptr, nullptr, 0, calib, 4 * n, PIFU_Q_COARSE_ONLY | PIFU_Q_NO_COARSE_MASK,
                             nullptr, nullptr, nullptr, delta, &pred4, (hipStream_t)stream);
    if (rc != PIFU_OK) return rc;
    return normals_from_pred4(pred4, n, normals, pred4_out, (hipStream_t)stream);
}

int pifu_calc_normal_chunked(pifu_ctx *ctx, const float *verts, int64_t n, int64_t verts_per_call, const float *calib,
                             float delta, float *normals, float *pred4_out, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && verts && calib && normals, PIFU_ERR_INVALID, "pifu_calc_normal_chunked: null argument");
    PIFU_REQUIRE(n > 0 && verts_per_call > 0, PIFU_ERR_INVALID, "pifu_calc_normal_chunked: empty vertex batch");
    float *pred4 = nullptr;
    int rc = pifu_query_impl(ctx, 3, verts, 4 * n, nullptr, nullptr, nullptr, 0, calib, 4 * verts_per_call, PIFU_Q_NO_FINE_MASK,
                             nullptr, nullptr, nullptr, delta, &pred4, (hipStream_t)stream);
    if (rc != PIFU_OK) return rc;
    return normals_from_pred4(pred4, n, normals, pred4_out, (hipStream_t)stream);
}

namespace {
// one wave per block; the first wave to arrive on an XCD records that XCD's counters
__global__ void k_clock_probe(unsigned long long *out)
{
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg(63508) & 7u;       // HW_REG_XCC_ID
    const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    if (atomicCAS(out + 2 * xcc + 1, 0ull, r) == 0ull) out[2 * xcc] = t;
}
}  // namespace

extern "C" int pifu_clock_probe(pifu_ctx *ctx, uint64_t out[16], pifu_stream stream)
{
    PIFU_REQUIRE(ctx && out, PIFU_ERR_INVALID, "pifu_clock_probe: null argument");
    PIFU_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->clock_buf) PIFU_HIP(hipMalloc(&ctx->clock_buf, 16 * sizeof(unsigned long long)));
    PIFU_HIP(hipMemsetAsync(ctx->clock_buf, 0, 16 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_clock_probe, dim3(64), dim3(64), 0, st, (unsigned long long *)ctx->clock_buf);
    PIFU_HIP(hipGetLastError());
    PIFU_HIP(hipMemcpyAsync(out, ctx->clock_buf, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PIFU_HIP(hipStreamSynchronize(st));
    return PIFU_OK;
}
