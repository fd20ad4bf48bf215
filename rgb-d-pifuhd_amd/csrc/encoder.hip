// Image encoders of the hot path on gfx950: the stacked-hourglass HGFilter (Filter.py:23-228) and the
// pix2pixHD GlobalGenerator normal nets (networks.py:131-207), i.e. PIFuNetwNML.filter
// (PIFuNetwNML.py:73-97) and PIFuMRNet.filter_local (PIFuMRNet.py:83-118).
//
// Layout: every activation is NHWC fp32 ([pixel][channel], channel stride `ld` so that a tensor can be a
// channel slice of a wider buffer: the ConvBlock concat is written in place).  A convolution is an
// implicit GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): rows = output pixels, columns = output channels,
// K = kernel rows x (kernel columns x input channels).  The preceding norm (+ReLU) of the pre-activation ConvBlock
// (conv(relu(norm(x)))) is applied ONCE per input element by k_prep_act, which also materialises the zero / mirrored
// border (padding comes AFTER the activation); the GEMM kernel k_conv_dma then moves both operand tiles global -> LDS
// by LDS-DMA with no per-tap address or transform arithmetic left (see the section comment below).  Statistics
// for train-mode BatchNorm / InstanceNorm / GroupNorm come from a separate per-channel reduction pass
// (deterministic, double accumulation across blocks).
#include "pifu_internal.h"

#include <math.h>

#include <algorithm>

#include <map>
#include <string>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BK = 32;      // GEMM rows per block, K step (the N tile is a template parameter: 64 or 32)

// ======================================================================================================
// Convolution as an implicit GEMM whose BOTH operands reach LDS by LDS-DMA (k_conv_dma).
//
// What made k_conv_gemm slow (56 TFLOP/s on a 128 -> 128 3x3 layer with no padding waste at all) is its staging pass: per
// K step and thread ~200 vector instructions of tap addressing, bounds / reflection logic and the norm + ReLU transform,
// on a chip whose fp32 MFMA shares the vector ALU.  Here that work is done ONCE per input element instead of once per tap
// and output-channel tile: k_prep_act writes act(norm(x)) into a border-padded copy P (zeros or mirrored pixels), after
// which every A row of the implicit GEMM is a plain contiguous run of floats - for one kernel row dy the taps dx = 0..k-1
// of a pixel are k * C consecutive floats of P - and a K step is `base(row) + scalar offset`.  Both tiles then move by
// `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write, no transform), double-buffered with ONE barrier per K step.
// K order: kernel row dy, then the k * C floats of that row (dx-major, channel-minor), each row padded to the 32-float
// K step with zero WEIGHTS (the A side then reads the following pixels, which are finite).  7x7 convs on few channels pack
// densely this way (7 x 84 -> 7 x 96 for 12 channels instead of 49 x 32).  N tile = 64 or 32 output channels (three or four blocks per CU), so
// that the 32 / 64-channel layers of the hourglass ConvBlocks issue no padding MFMAs.
// LDS image of both tiles: unpadded [rows][32 floats] with the eight 16-byte chunks of a row XOR-swizzled by
// (row >> 1) & 7 - applied to the SOURCE address of the DMA (the LDS side of an LDS-DMA is lane-linear) and to the fragment
// read - which makes every ds_read_b128 of the MFMA operand fragments bank-conflict free.
// ======================================================================================================
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

// One LDS-DMA piece (1 KB per wave): lane i's 16 bytes land at lds_addr + 16 i.  Issued from asm so that the compiler
// does not order it in front of every later ds_read with a full vmcnt(0); the kernel waits for its DMAs itself.  `s_nop 4`:
// SGPR operands may come fresh from a VALU readfirstlane; `s_nop 0`: M0 write -> LDS-DMA.  M0 is saved and restored.
__device__ __forceinline__ void lds_dma1(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff)
{
    unsigned keep;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

struct PrepArgs {
    const float *X;         // source NHWC [H][W], channel stride ldx (>= Cp readable channels per pixel)
    int H, W, ldx;
    const float *scale;     // per-channel norm affine of the consumer's prologue (>= Cp entries)
    const float *shift;
    float slope;            // act(t) = max(t, slope * t): 0 ReLU, 1 identity
    int pad_lo, reflect;    // border in front of row / column 0; reflect: mirrored pixels instead of zeros
    float *P;               // destination [Hp][Wp][Cp] + 64 zero floats of slack
    int Hp, Wp, Cp;         // Cp % 4 == 0
};

__global__ void k_prep_act(const PrepArgs a)
{
    const int cg = a.Cp / 4;
    const size_t total = (size_t)a.Hp * a.Wp * cg;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) {
        if (i < total + 16) *reinterpret_cast<float4 *>(a.P + (size_t)a.Hp * a.Wp * a.Cp + (i - total) * 4) = make_float4(0, 0, 0, 0);
        return;
    }
    const int c = (int)(i % cg) * 4;
    const size_t p = i / cg;
    int iy = (int)(p / a.Wp) - a.pad_lo, ix = (int)(p % a.Wp) - a.pad_lo;
    bool inside = true;
    if (a.reflect) {
        iy = iy < 0 ? -iy : (iy >= a.H ? 2 * (a.H - 1) - iy : iy);
        ix = ix < 0 ? -ix : (ix >= a.W ? 2 * (a.W - 1) - ix : ix);
    } else {
        inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    }
    float4 v = make_float4(0, 0, 0, 0);
    if (inside) {       // padding is applied AFTER the activation: border pixels are exact zeros
        v = *reinterpret_cast<const float4 *>(a.X + ((size_t)iy * a.W + ix) * a.ldx + c);
        const float4 sc = *reinterpret_cast<const float4 *>(a.scale + c), sh = *reinterpret_cast<const float4 *>(a.shift + c);
        float t;
        t = fmaf(v.x, sc.x, sh.x); v.x = fmaxf(t, a.slope * t);
        t = fmaf(v.y, sc.y, sh.y); v.y = fmaxf(t, a.slope * t);
        t = fmaf(v.z, sc.z, sh.z); v.z = fmaxf(t, a.slope * t);
        t = fmaf(v.w, sc.w, sh.w); v.w = fmaxf(t, a.slope * t);
    }
    *reinterpret_cast<float4 *>(a.P + p * a.Cp + c) = v;
}

constexpr int CONV_MAX_SPANS = 8;

struct ConvDmaArgs {
    const float *P;             // padded activated input
    unsigned p_bytes;           // bytes of P incl. slack (buffer bound of the descriptor)
    int Wp, C;                  // padded width, channels per pixel
    int RW, nrows, is;          // logical output grid width, number of output rows, input step
    int nspan;                  // contiguous K runs ("kernel rows")
    int span_off[CONV_MAX_SPANS];     // byte offset of a run's first float relative to the row's base pixel
    int span_steps[CONV_MAX_SPANS];   // 32-float K steps of the run
    const float *W;             // [cout_pad][ldw], K order as above
    int ldw;                    // 32 * total K steps
    const float *bias;          // [cout_pad]
    float *Y; int ldy;
    const float *R; int ldr;    // optional residual
    int cout;
    int os, oy0, ox0, Wout;     // output pixel = (ry * os + oy0, rx * os + ox0)
    int ntm, ntn;
    // split-K (launches with too few tiles to fill the chip): split s accumulates K steps [s * steps_per_split, ...) into
    // part[s][row][col] (raw, no bias); k_conv_dma_reduce sums the splits in a fixed order and applies the epilogue
    int nsplit, steps_per_split, tiles_per_split;
    float *part;                // [nsplit][ntm * BM][ntn * TN]
    int ldp;                    // ntn * TN
    // fused statistics of the values this launch stores (unsplit launches): per (128-row tile, output channel) sum and
    // sum of squares, as the [tile][cout][2] double partials k_stats_finalize reduces - the norm that follows the conv
    // then needs no pass of its own over the tensor
    double *stat;               // or null
    int stat_tile0;             // first partial row of this launch (the four parity classes of a transposed conv share one set)
    // batched GEMM (the 16 positions of a Winograd convolution as ONE launch): row tiles [g * wg_tiles, (g + 1) * wg_tiles)
    // multiply the weight matrix W + g * wg_stride (floats); wg_tiles == 0: one weight matrix for all rows
    int wg_tiles;
    size_t wg_stride;
};

template <int TN>
__device__ __forceinline__ void conv_dma_tile(const ConvDmaArgs &a, const int block_id)
{
    constexpr int WN = TN == 32 ? 1 : 2, WM = 4 / WN;          // wave grid
    constexpr int MT = (BM / WM) / 32, NT = (TN / WN) / 32;    // 32 x 32 accumulators per wave
    constexpr int ATILE = BM * BK, WTILE = TN * BK, NS = BK / 8;
    constexpr int WPIECES = TN / 32;                           // LDS-DMA pieces of the W tile per wave
    __shared__ __attribute__((aligned(16))) float lds[2 * ATILE + 2 * WTILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int split = block_id / a.tiles_per_split, bt = block_id - split * a.tiles_per_split;
    const int xcd = bt & 7, slot = bt >> 3;
    const int tile_n = slot % a.ntn;
    const int tile_m = (slot / a.ntn) * 8 + xcd;
    if (tile_m >= a.ntm) return;
    const int row0 = tile_m * BM, col0 = tile_n * TN;
    const int step_lo = split * a.steps_per_split;
    const int step_hi = min(a.ldw / BK, step_lo + a.steps_per_split);

    const unsigned long long pa = (unsigned long long)a.P,
                             wa = (unsigned long long)(a.W + (size_t)col0 * a.ldw + (a.wg_tiles ? (size_t)(tile_m / a.wg_tiles) * a.wg_stride : 0));
    const u32x4 rP = {(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, a.p_bytes, 0x00020000u};
    const u32x4 rWt = {(unsigned)wa, (unsigned)(wa >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
    // DMA lane mapping: lane (r8 = lane >> 3, p = lane & 7) of wave w fills, in pass q, chunk position p of tile row
    // 32 q + 8 w + r8 with the logical chunk p ^ f(row), f(row) = (row >> 1) & 7 (independent of q)
    const int trow = wave * 8 + (lane >> 3);
    const unsigned chunk = (unsigned)(((lane & 7) ^ ((trow >> 1) & 7)) << 4);
    unsigned voA[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int r = row0 + trow + 32 * q;
        r = r < a.nrows ? r : a.nrows - 1;                     // tile tail: any valid row (its outputs are never stored)
        const int ry = r / a.RW, rx = r - ry * a.RW;
        voA[q] = (unsigned)((ry * a.is) * a.Wp + rx * a.is) * (unsigned)a.C * 4u + chunk;
    }
    const unsigned ldw4 = (unsigned)a.ldw * 4u;
    const unsigned voW = (unsigned)trow * ldw4 + chunk;
    const unsigned a_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)(lds + wave * 8 * BK));
    const unsigned w_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)(lds + 2 * ATILE + wave * 8 * BK));

    // K cursor of the next step to fetch, positioned on this split's first step
    int c_span = 0, c_left = step_lo;
    while (c_left >= a.span_steps[c_span]) { c_left -= a.span_steps[c_span]; ++c_span; }
    unsigned c_soff = (unsigned)a.span_off[c_span] + (unsigned)c_left * (BK * 4u), c_kw4 = (unsigned)step_lo * (BK * 4u);
    c_left = a.span_steps[c_span] - c_left;
#define CONV_FETCH(buf_)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
            lds_dma1(a_lds + (unsigned)(buf_) * (ATILE * 4u) + q * 4096u, voA[q], rP, c_soff);         \
        _Pragma("unroll") for (int q = 0; q < WPIECES; ++q)                                            \
            lds_dma1(w_lds + (unsigned)(buf_) * (WTILE * 4u) + q * 4096u, voW, rWt, c_kw4 + q * 32u * ldw4); \
        c_soff += BK * 4u;                                                                             \
        c_kw4 += BK * 4u;                                                                              \
        if (--c_left == 0 && ++c_span < a.nspan) {                                                     \
            c_left = a.span_steps[c_span];                                                             \
            c_soff = (unsigned)a.span_off[c_span];                                                     \
        }                                                                                              \
    }

    const int h = lane >> 5, j = lane & 31;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    CONV_FETCH(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the DMAs of the first step have landed
    __syncthreads();

    const char *Al = reinterpret_cast<const char *>(lds) + (wm * (BM / WM) + j) * (BK * 4);
    const char *Bl = reinterpret_cast<const char *>(lds + 2 * ATILE) + (wn * (TN / WN) + j) * (BK * 4);
    const unsigned hf = (unsigned)((h ^ ((j >> 1) & 7)) << 4);
    for (int step = step_lo; step < step_hi; ++step) {
        const int buf = (step - step_lo) & 1;
        if (step + 1 < step_hi) CONV_FETCH(buf ^ 1)
        const char *A = Al + buf * (ATILE * 4), *B = Bl + buf * (WTILE * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned off = hf ^ (unsigned)(s * 32);
            float4 af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const float4 *>(A + mt * 32 * BK * 4 + off);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const float4 *>(B + nt * 32 * BK * 4 + off);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float av = t == 0 ? af[mt].x : (t == 1 ? af[mt].y : (t == 2 ? af[mt].z : af[mt].w));
                        const float bv = t == 0 ? bf[nt].x : (t == 1 ? bf[nt].y : (t == 2 ? bf[nt].z : bf[nt].w));
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt][nt], 0, 0, 0);
                    }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the tiles of step + 1 have landed
        __syncthreads();                         // ... and every wave is done with the tiles of this step
    }
#undef CONV_FETCH

    if (a.nsplit > 1) {     // raw partial tile; rows / columns beyond the valid range are written too (never read)
        float *Pt = a.part + ((size_t)split * a.ntm * BM + row0) * (size_t)a.ldp + col0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Pt[(size_t)(wm * (BM / WM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * (size_t)a.ldp + wn * (TN / WN) + nt * 32 + j] =
                        acc[mt][nt][r];
        return;
    }
    const bool plain = a.os == 1 && a.RW == a.Wout && a.oy0 == 0 && a.ox0 == 0;
    // The residual values first, all of them, then the stores (issuing them before the K loop as well was measured: no further
    // gain, 17 more registers): a residual may be the output itself (ConvBlock's 1 x 1
    // downsample adds in place), so with the load inside the store loop the compiler must keep every load behind the store
    // before it - 32 dependent round trips per thread, 1.5 TB/s on the K = 128 downsample convolutions of the fine level.
    float resid[NT][MT][16];
    if (a.R) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + wn * (TN / WN) + nt * 32 + j;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wm * (BM / WM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float rv = 0.f;
                    if (col < a.cout && row < a.nrows) {
                        size_t pix = (size_t)row;
                        if (!plain) {
                            const int ry = row / a.RW, rx = row - ry * a.RW;
                            pix = (size_t)(ry * a.os + a.oy0) * a.Wout + (rx * a.os + a.ox0);
                        }
                        rv = a.R[pix * a.ldr + col];
                    }
                    resid[nt][mt][r] = rv;
                }
        }
    }
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ssum[nt] = 0.f; ssq[nt] = 0.f;
        const int col = col0 + wn * (TN / WN) + nt * 32 + j;
        const bool cok = col < a.cout;
        const float bias = cok ? a.bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * (BM / WM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (cok && row < a.nrows) {
                    size_t pix = (size_t)row;            // plain convolution: output pixel = GEMM row
                    if (!plain) {
                        const int ry = row / a.RW, rx = row - ry * a.RW;
                        pix = (size_t)(ry * a.os + a.oy0) * a.Wout + (rx * a.os + a.ox0);
                    }
                    float v = acc[mt][nt][r] + bias;
                    if (a.R) v += resid[nt][mt][r];
                    a.Y[pix * a.ldy + col] = v;
                    ssum[nt] += v;
                    ssq[nt] = fmaf(v, v, ssq[nt]);
                }
            }
        }
    }
    if (a.stat) {       // (the operand tiles are dead: every wave passed the loop's final barrier)
        float *red = lds;                                    // [2 (sum | squares)][WM][TN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float sv = ssum[nt] + __shfl_xor(ssum[nt], 32), qv = ssq[nt] + __shfl_xor(ssq[nt], 32);
            if (lane < 32) {
                red[(0 * WM + wm) * TN + wn * (TN / WN) + nt * 32 + j] = sv;
                red[(1 * WM + wm) * TN + wn * (TN / WN) + nt * 32 + j] = qv;
            }
        }
        __syncthreads();
        if (tid < TN && col0 + tid < a.cout) {
            double sd = 0.0, qd = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { sd += (double)red[(0 * WM + w) * TN + tid]; qd += (double)red[(1 * WM + w) * TN + tid]; }
            double *dst = a.stat + ((size_t)(a.stat_tile0 + tile_m) * a.cout + col0 + tid) * 2;
            dst[0] = sd;
            dst[1] = qd;
        }
    }
}

template <int TN>
__global__ __launch_bounds__(256, 3) void k_conv_dma(const ConvDmaArgs a)
{
    conv_dma_tile<TN>(a, blockIdx.x);
}

// The four output-parity classes of a transposed convolution (same tile grid, different taps / weights / output offsets) as
// ONE launch, class = blockIdx.y: at 128^2 -> 256^2 and 256^2 -> 512^2 a class alone is 512 blocks of 4 - 16 K steps - two
// thirds of the chip's block slots, done before it is warm (254 / 228 us for the four launches of 9.7 GFLOP).
struct ConvDma4 {
    ConvDmaArgs a[4];
};

template <int TN>
__global__ __launch_bounds__(256, 3) void k_conv_dma4(const ConvDma4 m)
{
    conv_dma_tile<TN>(m.a[blockIdx.y], blockIdx.x);
}

// split-K epilogue of k_conv_dma: Y = sum_s part[s] (fixed order) + bias (+ residual), with the conv's output pixel mapping
__global__ void k_conv_dma_reduce(const ConvDmaArgs a)
{
    const int ldp4 = a.ldp / 4;
    const int c4 = (blockIdx.x * blockDim.x + threadIdx.x) % ldp4;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) / ldp4;
    if (row >= a.nrows || c4 * 4 >= a.cout) return;
    const size_t plane = (size_t)a.ntm * BM * a.ldp;
    float4 v = *reinterpret_cast<const float4 *>(a.part + (size_t)row * a.ldp + c4 * 4);
    for (int s = 1; s < a.nsplit; ++s) {
        const float4 p = *reinterpret_cast<const float4 *>(a.part + s * plane + (size_t)row * a.ldp + c4 * 4);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const int ry = row / a.RW, rx = row - ry * a.RW;
    const size_t pix = (size_t)(ry * a.os + a.oy0) * a.Wout + (rx * a.os + a.ox0);
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = c4 * 4 + k;
        if (col < a.cout) {
            float o = vv[k] + a.bias[col];
            if (a.R) o += a.R[pix * a.ldr + col];
            a.Y[pix * a.ldy + col] = o;
        }
    }
}

// ---- per-channel statistics of an NHWC tensor: partial (sum, sum^2) per block of rows -----------------
__global__ void k_channel_stats(const float *X, int ld, int C, int P, int rows_per_block, double *partial /* [nb][C][2] */)
{
    __shared__ double sh[256 * 2];
    const int cg = C / 4;                     // float4 channel groups
    const int tpr = blockDim.x / cg > 0 ? blockDim.x / cg : 1;   // threads along rows
    const int c4 = threadIdx.x % cg, tr = threadIdx.x / cg;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(P, r0 + rows_per_block);
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (tr < tpr && c4 < cg) {
        for (int r = r0 + tr; r < r1; r += tpr) {
            const float4 v = *reinterpret_cast<const float4 *>(X + (size_t)r * ld + c4 * 4);
            s[0] += v.x; q[0] = fmaf(v.x, v.x, q[0]);
            s[1] += v.y; q[1] = fmaf(v.y, v.y, q[1]);
            s[2] += v.z; q[2] = fmaf(v.z, v.z, q[2]);
            s[3] += v.w; q[3] = fmaf(v.w, v.w, q[3]);
        }
    }
    // reduce over tr in a fixed order through LDS, one channel component at a time
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        sh[threadIdx.x * 2] = s[k];
        sh[threadIdx.x * 2 + 1] = q[k];
        __syncthreads();
        if (tr == 0 && c4 < cg) {
            double ds = 0, dq = 0;
            for (int t = 0; t < tpr; ++t) { ds += sh[(t * cg + c4) * 2]; dq += sh[(t * cg + c4) * 2 + 1]; }
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2] = ds;
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2 + 1] = dq;
        }
    }
}

// out = a + b (the identity shortcut of a ConvBlock) with the statistics of `out` in the same pass: same block geometry and
// partial layout as k_channel_stats
__global__ void k_add_stats(const float *A, int lda, const float *B, int ldb, float *O, int ldo, int C, int P, int rows_per_block,
                            double *partial /* [nb][C][2] */)
{
    __shared__ double sh[256 * 2];
    const int cg = C / 4;
    const int tpr = blockDim.x / cg > 0 ? blockDim.x / cg : 1;
    const int c4 = threadIdx.x % cg, tr = threadIdx.x / cg;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(P, r0 + rows_per_block);
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (tr < tpr && c4 < cg) {
        for (int r = r0 + tr; r < r1; r += tpr) {
            const float4 x = *reinterpret_cast<const float4 *>(A + (size_t)r * lda + c4 * 4);
            const float4 y = *reinterpret_cast<const float4 *>(B + (size_t)r * ldb + c4 * 4);
            const float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
            *reinterpret_cast<float4 *>(O + (size_t)r * ldo + c4 * 4) = v;
            s[0] += v.x; q[0] = fmaf(v.x, v.x, q[0]);
            s[1] += v.y; q[1] = fmaf(v.y, v.y, q[1]);
            s[2] += v.z; q[2] = fmaf(v.z, v.z, q[2]);
            s[3] += v.w; q[3] = fmaf(v.w, v.w, q[3]);
        }
    }
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        sh[threadIdx.x * 2] = s[k];
        sh[threadIdx.x * 2 + 1] = q[k];
        __syncthreads();
        if (tr == 0 && c4 < cg) {
            double ds = 0, dq = 0;
            for (int t = 0; t < tpr; ++t) { ds += sh[(t * cg + c4) * 2]; dq += sh[(t * cg + c4) * 2 + 1]; }
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2] = ds;
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2 + 1] = dq;
        }
    }
}

// one BLOCK per group: (sum, sum^2) over the partial rows and the group's channels -> per-channel scale / shift.  (Round 5 ran
// one wave per group: 2 048 partial rows of a 512^2 map were 32 dependent double loads per lane - 14 - 20 us of pure latency
// on the critical queue, 55 times per pass.  256 threads take 8 each and meet in a fixed-order tree: ~5 us.)
__global__ __launch_bounds__(256) void k_stats_finalize(const double *partial, int nb, int C, int cpg, int groups, double count_per_channel,
                                                        const float *gamma, const float *beta, float eps, float *scale, float *shift)
{
    __shared__ double red[2][256];
    const int grp = blockIdx.x, tid = threadIdx.x;
    double s = 0, q = 0;
    const int total = nb * cpg;
    for (int i = tid; i < total; i += 256) {
        const int b = i / cpg, ch = grp * cpg + i % cpg;
        s += partial[((size_t)b * C + ch) * 2];
        q += partial[((size_t)b * C + ch) * 2 + 1];
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {      // fixed order: the result does not depend on scheduling
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    const double cnt = count_per_channel * cpg;
    const double mean = red[0][0] / cnt;
    double var = red[1][0] / cnt - mean * mean;
    var = var > 0 ? var : 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int i = tid; i < cpg; i += 256) {
        const int ch = grp * cpg + i;
        const double g = gamma ? (double)gamma[ch] : 1.0, b = beta ? (double)beta[ch] : 0.0;
        scale[ch] = (float)(g * rstd);
        shift[ch] = (float)(b - mean * rstd * g);
    }
}

// ---- element-wise helpers (NHWC, float4 over channels) ------------------------------------------------
// out = a + (b * scale + shift)   (scale == null: plain add)
__global__ void k_add(const float *A, int lda, const float *B, int ldb, const float *scale, const float *shift,
                      float *O, int ldo, int C, size_t P)
{
    const int cg = C / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * cg) return;
    const size_t p = i / cg;
    const int c = (int)(i % cg) * 4;
    const float4 a = *reinterpret_cast<const float4 *>(A + p * lda + c);
    float4 b = *reinterpret_cast<const float4 *>(B + p * ldb + c);
    if (scale) {
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
        b.x = fmaf(b.x, sc.x, sh.x); b.y = fmaf(b.y, sc.y, sh.y); b.z = fmaf(b.z, sc.z, sh.z); b.w = fmaf(b.w, sc.w, sh.w);
    }
    *reinterpret_cast<float4 *>(O + p * ldo + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// out = act(x * scale + shift), act(t) = max(t, slope * t)
__global__ void k_affine_act(const float *X, int ldx, const float *scale, const float *shift, float slope, float *O, int ldo,
                             int C, size_t P)
{
    const int cg = C / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * cg) return;
    const size_t p = i / cg;
    const int c = (int)(i % cg) * 4;
    float4 v = *reinterpret_cast<const float4 *>(X + p * ldx + c);
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
    float t;
    t = fmaf(v.x, sc.x, sh.x); v.x = fmaxf(t, slope * t);
    t = fmaf(v.y, sc.y, sh.y); v.y = fmaxf(t, slope * t);
    t = fmaf(v.z, sc.z, sh.z); v.z = fmaxf(t, slope * t);
    t = fmaf(v.w, sc.w, sh.w); v.w = fmaxf(t, slope * t);
    *reinterpret_cast<float4 *>(O + p * ldo + c) = v;
}

// avg_pool2d(x, 2, stride 2)  (Filter.py:113,191)
__global__ void k_avgpool2(const float *X, int ldx, int Win, float *O, int ldo, int C, int Ho, int Wo)
{
    const int cg = C / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Ho * Wo * cg) return;
    const int c = (int)(i % cg) * 4;
    const size_t p = i / cg;
    const int ox = (int)(p % Wo), oy = (int)(p / Wo);
    const float *s = X + ((size_t)(2 * oy) * Win + 2 * ox) * ldx + c;
    const float4 a = *reinterpret_cast<const float4 *>(s), b = *reinterpret_cast<const float4 *>(s + ldx);
    const float4 d = *reinterpret_cast<const float4 *>(s + (size_t)Win * ldx), e = *reinterpret_cast<const float4 *>(s + (size_t)Win * ldx + ldx);
    *reinterpret_cast<float4 *>(O + p * ldo + c) =
        make_float4((a.x + b.x + d.x + e.x) * 0.25f, (a.y + b.y + d.y + e.y) * 0.25f, (a.z + b.z + d.z + e.z) * 0.25f,
                    (a.w + b.w + d.w + e.w) * 0.25f);
}

__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4])
{
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = (1.f - t) + 1.f;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// out = up1 + bicubic_x2(low), align_corners=True (Filter.py:125-127)
__global__ void k_bicubic_up2_add(const float *L, int ldl, int h, int w, const float *U, int ldu, float *O, int ldo, int C)
{
    const int cg = C / 4, H = 2 * h, Wd = 2 * w;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * Wd * cg) return;
    const int c = (int)(i % cg) * 4;
    const size_t p = i / cg;
    const int ox = (int)(p % Wd), oy = (int)(p / Wd);
    const float sy = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = w > 1 ? (float)(w - 1) / (float)(Wd - 1) : 0.f;
    const float fy = sy * oy, fx = sx * ox;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_coeffs(fy - iy, wy);
    cubic_coeffs(fx - ix, wx);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), h - 1);
        float4 row = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), w - 1);
            const float4 v = *reinterpret_cast<const float4 *>(L + ((size_t)yy * w + xx) * ldl + c);
            row.x = fmaf(v.x, wx[b], row.x); row.y = fmaf(v.y, wx[b], row.y);
            row.z = fmaf(v.z, wx[b], row.z); row.w = fmaf(v.w, wx[b], row.w);
        }
        acc.x = fmaf(row.x, wy[a], acc.x); acc.y = fmaf(row.y, wy[a], acc.y);
        acc.z = fmaf(row.z, wy[a], acc.z); acc.w = fmaf(row.w, wy[a], acc.w);
    }
    const float4 u = *reinterpret_cast<const float4 *>(U + p * ldu + c);
    *reinterpret_cast<float4 *>(O + p * ldo + c) = make_float4(u.x + acc.x, u.y + acc.y, u.z + acc.z, u.w + acc.w);
}

// The same two element-wise producers with the statistics of what they write in the same pass (block geometry and partial
// layout of k_channel_stats / k_add_stats: a block owns `rows_per_block` consecutive output pixels, thread (c4, tr) walks its
// rows in a fixed order).  Their outputs are the inputs of ConvBlocks whose first norm works on batch / group statistics
// (train-mode BatchNorm of the fine level, GroupNorm): k_channel_stats used to re-read the whole tensor for them
// (14 launches, 0.5 ms of a pass).
__global__ void k_avgpool2_stats(const float *X, int ldx, int Win, float *O, int ldo, int C, int Ho, int Wo, int rows_per_block,
                                 double *partial /* [nb][C][2] */)
{
    __shared__ double sh[256 * 2];
    const int cg = C / 4, P = Ho * Wo;
    const int tpr = blockDim.x / cg > 0 ? blockDim.x / cg : 1;
    const int c4 = threadIdx.x % cg, tr = threadIdx.x / cg;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (tr < tpr && c4 < cg) {
        for (int r = r0 + tr; r < r1; r += tpr) {
            const int ox = r % Wo, oy = r / Wo;
            const float *src = X + ((size_t)(2 * oy) * Win + 2 * ox) * ldx + c4 * 4;
            const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + ldx);
            const float4 d = *reinterpret_cast<const float4 *>(src + (size_t)Win * ldx), e = *reinterpret_cast<const float4 *>(src + (size_t)Win * ldx + ldx);
            const float4 v = make_float4((a.x + b.x + d.x + e.x) * 0.25f, (a.y + b.y + d.y + e.y) * 0.25f, (a.z + b.z + d.z + e.z) * 0.25f,
                                         (a.w + b.w + d.w + e.w) * 0.25f);
            *reinterpret_cast<float4 *>(O + (size_t)r * ldo + c4 * 4) = v;
            s[0] += v.x; q[0] = fmaf(v.x, v.x, q[0]); s[1] += v.y; q[1] = fmaf(v.y, v.y, q[1]);
            s[2] += v.z; q[2] = fmaf(v.z, v.z, q[2]); s[3] += v.w; q[3] = fmaf(v.w, v.w, q[3]);
        }
    }
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        sh[threadIdx.x * 2] = s[k];
        sh[threadIdx.x * 2 + 1] = q[k];
        __syncthreads();
        if (tr == 0 && c4 < cg) {
            double ds = 0, dq = 0;
            for (int t = 0; t < tpr; ++t) { ds += sh[(t * cg + c4) * 2]; dq += sh[(t * cg + c4) * 2 + 1]; }
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2] = ds;
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2 + 1] = dq;
        }
    }
}

__global__ void k_bicubic_up2_add_stats(const float *L, int ldl, int h, int w, const float *U, int ldu, float *O, int ldo, int C,
                                        int rows_per_block, double *partial /* [nb][C][2] */)
{
    __shared__ double sh[256 * 2];
    const int cg = C / 4, H = 2 * h, Wd = 2 * w, P = H * Wd;
    const int tpr = blockDim.x / cg > 0 ? blockDim.x / cg : 1;
    const int c4 = threadIdx.x % cg, tr = threadIdx.x / cg;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
    const float sy = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = w > 1 ? (float)(w - 1) / (float)(Wd - 1) : 0.f;
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (tr < tpr && c4 < cg) {
        const int c = c4 * 4;
        for (int r = r0 + tr; r < r1; r += tpr) {
            const int ox = r % Wd, oy = r / Wd;
            const float fy = sy * oy, fx = sx * ox;
            const int iy = (int)floorf(fy), ix = (int)floorf(fx);
            float wy[4], wx[4];
            cubic_coeffs(fy - iy, wy);
            cubic_coeffs(fx - ix, wx);
            float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(iy - 1 + a, 0), h - 1);
                float4 row = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int xx = min(max(ix - 1 + b, 0), w - 1);
                    const float4 v = *reinterpret_cast<const float4 *>(L + ((size_t)yy * w + xx) * ldl + c);
                    row.x = fmaf(v.x, wx[b], row.x); row.y = fmaf(v.y, wx[b], row.y);
                    row.z = fmaf(v.z, wx[b], row.z); row.w = fmaf(v.w, wx[b], row.w);
                }
                acc.x = fmaf(row.x, wy[a], acc.x); acc.y = fmaf(row.y, wy[a], acc.y);
                acc.z = fmaf(row.z, wy[a], acc.z); acc.w = fmaf(row.w, wy[a], acc.w);
            }
            const float4 u = *reinterpret_cast<const float4 *>(U + (size_t)r * ldu + c);
            const float4 v = make_float4(u.x + acc.x, u.y + acc.y, u.z + acc.z, u.w + acc.w);
            *reinterpret_cast<float4 *>(O + (size_t)r * ldo + c) = v;
            s[0] += v.x; q[0] = fmaf(v.x, v.x, q[0]); s[1] += v.y; q[1] = fmaf(v.y, v.y, q[1]);
            s[2] += v.z; q[2] = fmaf(v.z, v.z, q[2]); s[3] += v.w; q[3] = fmaf(v.w, v.w, q[3]);
        }
    }
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        sh[threadIdx.x * 2] = s[k];
        sh[threadIdx.x * 2 + 1] = q[k];
        __syncthreads();
        if (tr == 0 && c4 < cg) {
            double ds = 0, dq = 0;
            for (int t = 0; t < tpr; ++t) { ds += sh[(t * cg + c4) * 2]; dq += sh[(t * cg + c4) * 2 + 1]; }
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2] = ds;
            partial[((size_t)blockIdx.x * C + c4 * 4 + k) * 2 + 1] = dq;
        }
    }
}

// The network input [img (Cimg) | nmlF (3) | nmlB (3) | 0...] at H x W from an NCHW image and NCHW normal maps at (Hn x Wn)
// - bilinear align_corners=True if the sizes differ (PIFuNetwNML.py:84-92, PIFuMRNet.py:99-103) - written DIRECTLY in the form
// the first convolution's LDS-DMA reads: border-padded (zeros, or mirrored pixels for the normal nets' ReflectionPad) NHWC
// with Cp = roundup(channels, 4) floats per pixel + 64 floats of slack.  (Round 5 wrote a 32-channel NHWC tensor - 134 MB at
// 1024^2, one thread per pixel storing 128 strided bytes: 203 us - and k_prep_act then copied 12 of the 32 channels into the
// padded form; this is one pass over 50 MB.)
__global__ void k_pack_prepared(const float *img, int Cimg, int H, int W, const float *nF, const float *nB, int Hn, int Wn,
                                float *P, int Hp, int Wp, int Cp, int pad, int reflect)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)Hp * Wp;
    if (p >= total) {
        if (p < total + 16) *reinterpret_cast<float4 *>(P + total * Cp + (p - total) * 4) = make_float4(0, 0, 0, 0);
        return;
    }
    int y = (int)(p / Wp) - pad, x = (int)(p % Wp) - pad;
    bool inside = true;
    if (reflect) {
        y = y < 0 ? -y : (y >= H ? 2 * (H - 1) - y : y);
        x = x < 0 ? -x : (x >= W ? 2 * (W - 1) - x : x);
        inside = y >= 0 && y < H && x >= 0 && x < W;      // (beyond one mirror length: cannot happen for pad < H)
    } else {
        inside = y >= 0 && y < H && x >= 0 && x < W;
    }
    float *o = P + p * Cp;
    int c = 0;
    if (inside) {
        const size_t q = (size_t)y * W + x;
        for (int k = 0; k < Cimg; ++k) o[c++] = img[(size_t)k * H * W + q];
        const float *srcs[2] = {nF, nB};
        for (int s = 0; s < 2; ++s) {
            if (!srcs[s]) continue;
            if (Hn == H && Wn == W) {
                for (int k = 0; k < 3; ++k) o[c++] = srcs[s][(size_t)k * H * W + q];
            } else {
                const float fy = Hn > 1 ? (float)y * ((float)(Hn - 1) / (float)(H - 1)) : 0.f;
                const float fx = Wn > 1 ? (float)x * ((float)(Wn - 1) / (float)(W - 1)) : 0.f;
                const int y0 = min((int)fy, Hn - 1), x0 = min((int)fx, Wn - 1);
                const int y1 = min(y0 + 1, Hn - 1), x1 = min(x0 + 1, Wn - 1);
                const float ly = fy - y0, lx = fx - x0;
                for (int k = 0; k < 3; ++k) {
                    const float *m = srcs[s] + (size_t)k * Hn * Wn;
                    const float v = (1.f - ly) * ((1.f - lx) * m[(size_t)y0 * Wn + x0] + lx * m[(size_t)y0 * Wn + x1]) +
                                    ly * ((1.f - lx) * m[(size_t)y1 * Wn + x0] + lx * m[(size_t)y1 * Wn + x1]);
                    o[c++] = v;
                }
            }
        }
    }
    for (; c < Cp; ++c) o[c] = 0.f;
}

// Last layer of the normal nets: ReflectionPad(3) + Conv7x7 (64 -> 3) + Tanh, NCHW output (networks.py:158-160).
// A block computes 16 x 16 output pixels.  Per group of 16 input channels it stages the 22 x 22 input patch - with the
// InstanceNorm + ReLU prologue applied once per input value instead of once per tap - and the 3 x 49 x 16 weights in LDS
// (pixel stride 20 floats: conflict-free ds_read_b128), then every thread runs its 49 taps from LDS.  (The one-thread-per-
// pixel version re-read and re-normalised every input value 49 times from global memory: 1.3 ms per 512 x 512 image.)
// (Tried in round 3: the tap weights by block-uniform scalar loads instead of LDS - 4 instead of 16 ds_read_b128 per 48 FMAs;
// slower, 323 -> 385 us: the scalar loads' latency sits in front of every tap.)
__global__ __launch_bounds__(256) void k_conv7_to3_tanh(const float *X, int ldx, int H, int W, int Cin, const float *scale,
                                                        const float *shift, const float *Wt /* [3][49][Cin] */, const float *bias,
                                                        float *out_nchw)
{
    constexpr int T = 16, PW = T + 6, CG = 16, PS = 20;
    __shared__ __attribute__((aligned(16))) float patch[PW * PW * PS];
    __shared__ __attribute__((aligned(16))) float wl[3 * 49 * CG];
    const int tiles_x = (W + T - 1) / T;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[3] = {bias[0], bias[1], bias[2]};
    for (int cg = 0; cg < Cin; cg += CG) {
        for (int i = threadIdx.x; i < PW * PW * (CG / 4); i += 256) {
            const int pix = i / (CG / 4), q = i % (CG / 4);
            int iy = by * T + pix / PW - 3, ix = bx * T + pix % PW - 3;
            iy = iy < 0 ? -iy : (iy >= H ? 2 * (H - 1) - iy : iy);
            ix = ix < 0 ? -ix : (ix >= W ? 2 * (W - 1) - ix : ix);
            iy = min(max(iy, 0), H - 1);      // only pixels of a partial tile that lie outside the image get here
            ix = min(max(ix, 0), W - 1);
            const int c = cg + 4 * q;
            float4 v = *reinterpret_cast<const float4 *>(X + ((size_t)iy * W + ix) * ldx + c);
            const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
            v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
            v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
            *reinterpret_cast<float4 *>(patch + pix * PS + 4 * q) = v;
        }
        for (int i = threadIdx.x; i < 3 * 49 * CG; i += 256) {
            const int o = i / (49 * CG), r = i % (49 * CG);
            wl[i] = Wt[((size_t)o * 49 + r / CG) * Cin + cg + r % CG];
        }
        __syncthreads();
        for (int tap = 0; tap < 49; ++tap) {
            const float *pp = patch + ((ty + tap / 7) * PW + tx + tap % 7) * PS;
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(pp + 4 * q);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float *wp = wl + (o * 49 + tap) * CG;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *reinterpret_cast<const float4 *>(wp + 4 * q);
                    acc[o] = fmaf(v[q].x, w.x, acc[o]); acc[o] = fmaf(v[q].y, w.y, acc[o]);
                    acc[o] = fmaf(v[q].z, w.z, acc[o]); acc[o] = fmaf(v[q].w, w.w, acc[o]);
                }
            }
        }
        __syncthreads();
    }
    const int x = bx * T + tx, y = by * T + ty;
    if (x < W && y < H) {
#pragma unroll
        for (int o = 0; o < 3; ++o) out_nchw[(size_t)o * H * W + (size_t)y * W + x] = tanhf(acc[o]);
    }
}

__global__ void k_nhwc_to_nchw(const float *src, int ld, int C, int HW, float *dst)
{
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? src[(size_t)p * ld + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) dst[(size_t)c * HW + p] = tile[tx][r];
    }
}

// ---- Winograd F(2x2, 3x3) for the ResnetBlock trunk of the normal nets (networks.py:162-207) ------------------------------
// 36 of a pass's convolutions are 1024 -> 1024, 3 x 3, stride 1, reflection pad 1 on a 32 x 32 map: 696 of the encoders' 2 002
// GFLOP.  Y = A^T [(G g G^T) . (B^T d B)] A turns each into 16 GEMMs [256 tiles x 1024] x [1024 x 1024] - 2.25 x fewer
// multiplies, exact in real arithmetic, fp32 rounding within 2 x of the direct sum's (6e-7 relative on this shape) - and what
// sits BETWEEN two of them is separable per channel: output transform + bias, InstanceNorm statistics over the 1 024 pixels,
// normalise, ReLU or residual add, reflection pad and the next convolution's input transform.  k_wino_bridge does all of it in
// one launch, a block per 16 channels with the whole 32 x 32 plane of those channels in LDS: the direct path's split-K reduce,
// statistics, finalize and prepared-input kernels (and their launches) between two trunk convolutions are gone.
constexpr int WCG = 16;            // channels per block of the bridge (64 blocks moving 64-byte rows: 18 - 26 us; 8 -> 128 blocks of 32-byte rows: 21 - 32 us)
constexpr int WPX = WCG + 4;       // floats per pixel in LDS (padded: neighbouring tiles land in different banks)
constexpr int WLPT = WCG / 4;      // lanes per tile / pixel: each moves one float4 of the block's channels
constexpr int WTS = 256 / WLPT;    // tiles / pixels a block works on at a time

// U[p][o][c] = (G g G^T)[p / 4][p % 4] of the 3 x 3 kernel g = w[o][c]
__global__ void k_wino_weight(const float *w, int co, int ci, float *U)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)co * ci) return;
    const float *g = w + idx * 9;
    float Gg[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Gg[0][k] = g[k];
        Gg[1][k] = 0.5f * (g[k] + g[3 + k] + g[6 + k]);
        Gg[2][k] = 0.5f * (g[k] - g[3 + k] + g[6 + k]);
        Gg[3][k] = g[6 + k];
    }
    const size_t plane = (size_t)co * ci;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        U[(size_t)(i * 4 + 0) * plane + idx] = Gg[i][0];
        U[(size_t)(i * 4 + 1) * plane + idx] = 0.5f * (Gg[i][0] + Gg[i][1] + Gg[i][2]);
        U[(size_t)(i * 4 + 2) * plane + idx] = 0.5f * (Gg[i][0] - Gg[i][1] + Gg[i][2]);
        U[(size_t)(i * 4 + 3) * plane + idx] = Gg[i][2];
    }
}

struct WinoArgs {
    const float *M;        // [16][T][C] products of the convolution just finished (modes 1, 2)
    float *V;              // [16][T][C] transformed input of the next convolution (null: none follows)
    float *R;              // [H * W][C] the residual stream (mode 0: read; mode 2: read, updated in place)
    const float *bias;     // [C] of the convolution just finished (modes 1, 2)
    int H, W, C, T;        // T = (H / 2) * (W / 2) tiles
    int mode;              // 0: V <- R;  1: V <- relu(IN(conv + bias));  2: R += IN(conv + bias), V <- R
    float eps;
};

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

__global__ __launch_bounds__(256) void k_wino_bridge(const WinoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wino_lds[];
    float *h = wino_lds;                                           // [H * W][WPX]
    double *red = reinterpret_cast<double *>(wino_lds + (size_t)a.H * a.W * WPX);     // [4 waves][WCG][2]
    float *aff = reinterpret_cast<float *>(red + 4 * WCG * 2);     // [2][WCG] scale | shift
    const int tid = threadIdx.x, c4 = tid % WLPT, ts = tid / WLPT;  // WLPT lanes cover a tile's WCG channels (contiguous bytes)
    const int c0 = blockIdx.x * WCG + 4 * c4;
    const int TW = a.W / 2;
    const size_t plane = (size_t)a.T * a.C;
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = make_float4(0.f, 0.f, 0.f, 0.f);

    if (a.mode == 0) {
        for (int px = ts; px < a.H * a.W; px += WTS)
            *reinterpret_cast<float4 *>(h + px * WPX + 4 * c4) = *reinterpret_cast<const float4 *>(a.R + (size_t)px * a.C + c0);
    } else {
        // output transform A^T m A of every tile (+ bias) into the plane, statistics on the way
        const float4 b = *reinterpret_cast<const float4 *>(a.bias + c0);
        for (int t = ts; t < a.T; t += WTS) {
            float4 m[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = *reinterpret_cast<const float4 *>(a.M + p * plane + (size_t)t * a.C + c0);
            float4 t0[4], t1[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                t0[l] = f4_add(f4_add(m[l], m[4 + l]), m[8 + l]);
                t1[l] = f4_sub(f4_sub(m[4 + l], m[8 + l]), m[12 + l]);
            }
            float4 y[4];
            y[0] = f4_add(f4_add(f4_add(t0[0], t0[1]), t0[2]), b);
            y[1] = f4_add(f4_sub(f4_sub(t0[1], t0[2]), t0[3]), b);
            y[2] = f4_add(f4_add(f4_add(t1[0], t1[1]), t1[2]), b);
            y[3] = f4_add(f4_sub(f4_sub(t1[1], t1[2]), t1[3]), b);
            const int ty = t / TW, tx = t - ty * TW;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = (2 * ty + (k >> 1)) * a.W + 2 * tx + (k & 1);
                *reinterpret_cast<float4 *>(h + px * WPX + 4 * c4) = y[k];
                ssum = f4_add(ssum, y[k]);
                ssq.x = fmaf(y[k].x, y[k].x, ssq.x); ssq.y = fmaf(y[k].y, y[k].y, ssq.y);
                ssq.z = fmaf(y[k].z, y[k].z, ssq.z); ssq.w = fmaf(y[k].w, y[k].w, ssq.w);
            }
        }
        // InstanceNorm2d(affine=False) statistics of this block's 16 channels over the whole plane: fixed-order double sums
        double sd[4] = {(double)ssum.x, (double)ssum.y, (double)ssum.z, (double)ssum.w};
        double qd[4] = {(double)ssq.x, (double)ssq.y, (double)ssq.z, (double)ssq.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int o = WLPT; o < 64; o <<= 1) { sd[k] += __shfl_xor(sd[k], o); qd[k] += __shfl_xor(qd[k], o); }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane < WLPT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                red[((wave * WCG) + 4 * lane + k) * 2] = sd[k];
                red[((wave * WCG) + 4 * lane + k) * 2 + 1] = qd[k];
            }
        }
        __syncthreads();
        if (tid < WCG) {
            double s1 = 0.0, q1 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[((w * WCG) + tid) * 2]; q1 += red[((w * WCG) + tid) * 2 + 1]; }
            const double cnt = (double)a.H * a.W, mean = s1 / cnt;
            double var = q1 / cnt - mean * mean;
            var = var > 0 ? var : 0;
            const double rstd = 1.0 / sqrt(var + (double)a.eps);
            aff[tid] = (float)rstd;
            aff[WCG + tid] = (float)(-mean * rstd);
        }
        __syncthreads();
        // normalise in place: ReLU (the second convolution's input) or the residual add (the block's output)
        const float4 sc = *reinterpret_cast<const float4 *>(aff + 4 * c4), sh = *reinterpret_cast<const float4 *>(aff + WCG + 4 * c4);
        for (int px = ts; px < a.H * a.W; px += WTS) {
            float4 v = *reinterpret_cast<const float4 *>(h + px * WPX + 4 * c4);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            if (a.mode == 1) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else {
                float *rp = a.R + (size_t)px * a.C + c0;
                const float4 r = *reinterpret_cast<const float4 *>(rp);
                v = make_float4(r.x + v.x, r.y + v.y, r.z + v.z, r.w + v.w);      // x + conv_block(x): ResnetBlock.forward
                *reinterpret_cast<float4 *>(rp) = v;
            }
            *reinterpret_cast<float4 *>(h + px * WPX + 4 * c4) = v;
        }
    }
    __syncthreads();
    if (!a.V) return;
    // input transform B^T d B of every 4 x 4 patch (stride 2) of the reflection-padded plane
    for (int t = ts; t < a.T; t += WTS) {
        const int ty = t / TW, tx = t - ty * TW;
        float4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int yy = 2 * ty - 1 + i;
            yy = yy < 0 ? -yy : (yy >= a.H ? 2 * a.H - 2 - yy : yy);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int xx = 2 * tx - 1 + j;
                xx = xx < 0 ? -xx : (xx >= a.W ? 2 * a.W - 2 - xx : xx);
                d[i][j] = *reinterpret_cast<const float4 *>(h + (yy * a.W + xx) * WPX + 4 * c4);
            }
        }
        float4 u[4][4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            u[0][l] = f4_sub(d[0][l], d[2][l]);
            u[1][l] = f4_add(d[1][l], d[2][l]);
            u[2][l] = f4_sub(d[2][l], d[1][l]);
            u[3][l] = f4_sub(d[1][l], d[3][l]);
        }
        float *vp = a.V + (size_t)t * a.C + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(vp + (size_t)(i * 4 + 0) * plane) = f4_sub(u[i][0], u[i][2]);
            *reinterpret_cast<float4 *>(vp + (size_t)(i * 4 + 1) * plane) = f4_add(u[i][1], u[i][2]);
            *reinterpret_cast<float4 *>(vp + (size_t)(i * 4 + 2) * plane) = f4_sub(u[i][2], u[i][1]);
            *reinterpret_cast<float4 *>(vp + (size_t)(i * 4 + 3) * plane) = f4_sub(u[i][1], u[i][3]);
        }
    }
}

// ---- weight packing on the device ---------------------------------------------------------------------------------
// dst [cout_pad][ldw] <- a Conv2d weight [cout][d1][kh][kw] (or ConvTranspose2d [ci][cout][kh][kw]): per K run (span) the
// taps (ky, kx[i]) of `nx` consecutive pixels, C channels each (zero beyond the weight's `ci` and where a run is padded to
// the K step).  One thread per destination element: the 1.5 GB of encoder weights are repacked at HBM speed on first use
// instead of in a scalar host loop followed by a pageable upload (1.7 s of the first pass).
struct PackArgs {
    const float *src; float *dst;
    int cout, cout_pad, ci, C, d1, kh, kw, transposed, nspan, ldw;
    int base[CONV_MAX_SPANS + 1], nx[CONV_MAX_SPANS], ky[CONV_MAX_SPANS], kx[CONV_MAX_SPANS][7];
};

__global__ void k_pack_weight(const PackArgs a)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)a.cout_pad * a.ldw) return;
    const int o = (int)(idx / a.ldw), col = (int)(idx % a.ldw);
    int s = 0;
    while (s + 1 < a.nspan && col >= a.base[s + 1]) ++s;
    const int r = col - a.base[s], i = r / a.C, c = r % a.C;
    float v = 0.f;
    if (o < a.cout && i < a.nx[s] && c < a.ci) {
        const int ky = a.ky[s], kx = a.kx[s][i];
        const size_t src = a.transposed ? ((((size_t)c * a.d1 + o) * a.kh + ky) * a.kw + kx)
                                        : ((((size_t)o * a.d1 + c) * a.kh + ky) * a.kw + kx);
        v = a.src[src];
    }
    a.dst[idx] = v;
}

// ======================================================================================================
// host side: tensors, parameter registry, graph execution
// ======================================================================================================
struct Tensor {
    float *p = nullptr;
    int H = 0, W = 0, C = 0, ld = 0;
    // (sum, sum^2) partials [stat_nb][C][2] of exactly these C channels in their CURRENT values, when the producing kernel
    // left them behind (conv epilogue, k_add_stats) or a norm already reduced them; null: the norm makes its own pass
    double *stat = nullptr;
    int stat_nb = 0;
    // a tensor that exists ONLY in the prepared form of its (first) consumer (pack_input): border-padded NHWC with pCp floats
    // per pixel, activation already "applied" (the network input is consumed raw); p is null then
    const float *prepP = nullptr;
    int pHp = 0, pWp = 0, pCp = 0, pPad = 0;
    bool pReflect = false;
    unsigned pBytes = 0;
    size_t pixels() const { return (size_t)H * W; }
};

struct Affine {           // per-channel transform applied by the consumer
    const float *scale = nullptr, *shift = nullptr;
    float slope = 1.f;    // 1 identity, 0 ReLU
};

struct HostParam {
    std::vector<float> data;                     // host values (pifu_enc_set_param); empty for pifu_enc_set_param_device
    std::vector<int64_t> shape;
    float *raw = nullptr;                        // the tensor as given, on the device: what the packing kernels read
    bool owned = false;                          // raw was uploaded by the library from `data` (else: the caller's tensor, borrowed)
    size_t count() const { size_t n = 1; for (int64_t d : shape) n *= (size_t)d; return n; }
};

// A lane = one in-order chain of launches with the buffers only it touches.  Nets that do not depend on each other run in
// different lanes, i.e. on different streams at the same time: the two normal nets (netF || netB), and - when the caller
// lets the coarse call return before its hourglass has finished (PIFU_ENC_DEFER_JOIN) - the coarse hourglass beside the
// fine level's, whose big grids leave room for the coarse level's small ones.
struct Lane {
    DevBuf arena;                                // activations (bump-allocated per call)
    DevBuf splitk;                               // partial tiles of split-K convolutions (reused, stream-ordered)
    DevBuf prep;                                 // activated + padded input of the convolution in flight (k_prep_act)
    size_t used = 0;
    hipStream_t st = nullptr;                    // lane 0: the caller's stream; lanes 1, 2: streams of the library
    hipStream_t own = nullptr;
};

struct EncState {
    std::map<std::string, HostParam> params;
    std::map<std::string, float *> dev;          // packed device arrays (weights, biases, taps, fixed affines)
    Lane lanes[5];                               // 0 caller / netF / fine level, 1 netB, 2 coarse hourglass, 3 - 4 hourglass branches
    Lane *cur = &lanes[0];                       // the lane the host routine is currently issuing into
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_coarse = nullptr;
    // An hourglass level is two independent chains until its final add (Filter.py:104-127): up1 = b1(x) at the level's own
    // resolution, and pool -> b2 -> (next level) -> b3 at half of it.  While `branch_free` lanes are left, a level issues its
    // low chain into a branch lane and its up1 block into the lane it was called on: the low chain's many small launches
    // (statistics, prepared inputs, 32-wide tiles on quarter-size maps) run beside the big convolutions of up1 instead of
    // between them.  Same kernels, same arithmetic: bit-identical results.
    int branch_free = 0, branch_next = 3;
    hipEvent_t ev_bfork[2] = {nullptr, nullptr}, ev_bjoin[2] = {nullptr, nullptr};
    bool coarse_pending = false;                 // the deferred coarse hourglass has been issued and not yet joined
    int norm = 0;                                // 0 batch-eval, 1 batch-train, 2 group, 3 instance (normal nets)
    bool fuse_stats = false;                     // the convs of the net being run leave statistics partials for the next norm
    bool direct_trunk = false;                   // PIFU_ENC_NO_WINOGRAD: the ResnetBlock trunks as direct convolutions
    int wino_state = 0;                          // k_wino_bridge's LDS request on this context's device: 0 not asked, 1 granted, -1 refused
    pifu_ctx *ctx = nullptr;                     // owner (pifu_profile_*: algorithmic FLOPs of the convolutions, slot PIFU_PROF_ENC)
    bool busy = false;                           // a filter pass was issued since the last device synchronisation of set_param
    std::string err;
};

// books one convolution's ALGORITHMIC work (2 * output pixels * Cout * Cin * taps; padding of K / Cout not counted) while
// profiling is on; the launches run on up to three concurrent lanes, so they are counted, not timed (the pass is timed whole)
inline void book_conv(EncState &S, double flops, double canonical = -1.0)
{
    if (S.ctx && S.ctx->prof_on) {
        S.ctx->prof[PIFU_PROF_ENC].work += flops;                 // what the launch executes
        S.ctx->prof[PIFU_PROF_ENC].launches += 1;
        // what the direct convolution would execute (differs for the Winograd trunk of the normal nets: x 2.25)
        S.ctx->prof[PIFU_PROF_ENC_CANONICAL].work += canonical >= 0.0 ? canonical : flops;
        S.ctx->prof[PIFU_PROF_ENC_CANONICAL].launches += 1;
    }
}

std::map<pifu_ctx *, EncState *> g_states;

EncState &state_of(pifu_ctx *ctx)
{
    auto it = g_states.find(ctx);
    if (it == g_states.end()) { it = g_states.emplace(ctx, new EncState()).first; it->second->ctx = ctx; }
    return *it->second;
}

struct Fail {
    std::string msg;
};

float *arena_take(EncState &S, size_t floats)
{
    const size_t bytes = (floats * sizeof(float) + 255) & ~(size_t)255;
    if (S.cur->used + bytes > S.cur->arena.bytes) throw Fail{"encoder arena exhausted"};
    float *p = reinterpret_cast<float *>(static_cast<char *>(S.cur->arena.p) + S.cur->used);
    S.cur->used += bytes;
    return p;
}

Tensor new_tensor(EncState &S, int H, int W, int C)
{
    Tensor t;
    t.H = H; t.W = W; t.C = C; t.ld = C;
    t.p = arena_take(S, (size_t)H * W * C);
    return t;
}

Tensor slice(const Tensor &t, int c0, int c)
{
    Tensor s = t;
    s.p = t.p + c0;
    s.C = c;
    s.stat = nullptr;
    s.stat_nb = 0;
    return s;
}

HostParam &param(EncState &S, const std::string &name)
{
    auto it = S.params.find(name);
    if (it == S.params.end()) throw Fail{"missing encoder parameter " + name};
    return it->second;
}

float *dev_upload(EncState &S, const std::string &key, const std::vector<float> &host)
{
    auto it = S.dev.find(key);
    if (it != S.dev.end()) return it->second;
    float *d = nullptr;
    if (hipMalloc(&d, host.size() * sizeof(float)) != hipSuccess) throw Fail{"hipMalloc failed for " + key};
    if (hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) throw Fail{"hipMemcpy failed for " + key};
    S.dev[key] = d;
    return d;
}

const float *ones(EncState &S) { return dev_upload(S, "#ones", std::vector<float>(PIFU_MAX_WIDTH, 1.f)); }
const float *zeros(EncState &S) { return dev_upload(S, "#zeros", std::vector<float>(PIFU_MAX_WIDTH, 0.f)); }
Affine raw(EncState &S) { return Affine{ones(S), zeros(S), 1.f}; }

inline unsigned nblk(size_t n, int per) { return (unsigned)((n + per - 1) / per); }

int stat_rows_per_block(int P)
{
    int rows = (P + 255) / 256;
    return rows < 8 ? 8 : (rows > 1024 ? 1024 : rows);
}

int stat_threads(int C)
{
    const int cg = C / 4;
    if (cg > 1024) throw Fail{"channel count too large for k_channel_stats"};
    return cg <= 256 ? 256 : (cg <= 512 ? 512 : 1024);
}

// ---- normalisation: returns the per-channel affine that realises `norm(x)`; slope is set by the caller ----
Affine norm_affine(EncState &S, const std::string &prefix, Tensor &x, int norm)
{
    Affine a;
    if (norm == 0) {   // BatchNorm2d eval: running statistics, folded once
        const std::string key = prefix + "#bn_eval";
        if (!S.dev.count(key + ".s")) {
            const auto &g = param(S, prefix + ".weight").data, &b = param(S, prefix + ".bias").data;
            const auto &rm = param(S, prefix + ".running_mean").data, &rv = param(S, prefix + ".running_var").data;
            std::vector<float> sc(g.size()), sh(g.size());
            for (size_t i = 0; i < g.size(); ++i) {
                const double rstd = 1.0 / sqrt((double)rv[i] + 1e-5);
                sc[i] = (float)(g[i] * rstd);
                sh[i] = (float)((double)b[i] - (double)rm[i] * rstd * g[i]);
            }
            dev_upload(S, key + ".s", sc);
            dev_upload(S, key + ".b", sh);
        }
        a.scale = S.dev[key + ".s"];
        a.shift = S.dev[key + ".b"];
        return a;
    }
    // batch statistics over H*W (batch 1): BatchNorm train (per channel), GroupNorm (32 groups), InstanceNorm
    const int C = x.C;
    const int groups = norm == 2 ? 32 : C;
    const int cpg = C / groups;
    const int P = (int)x.pixels();
    int nb = x.stat_nb;
    double *partial = x.stat;
    if (!partial) {
        // ~256 blocks in flight: small maps (32 x 32 x 1024 channels) would otherwise be reduced by a single block
        int rows_per_block = stat_rows_per_block(P);
        nb = (P + rows_per_block - 1) / rows_per_block;
        partial = reinterpret_cast<double *>(arena_take(S, (size_t)nb * C * 2 * 2));
        hipLaunchKernelGGL(k_channel_stats, dim3(nb), dim3(stat_threads(C)), 0, S.cur->st, (const float *)x.p, x.ld, C, P, rows_per_block, partial);
        x.stat = partial;           // a second norm of the same tensor (ConvBlock bn1 / bn4) reuses the reduction
        x.stat_nb = nb;
    }
    float *sc = arena_take(S, C), *sh = arena_take(S, C);
    const float *gamma = nullptr, *beta = nullptr;
    if (norm != 3) {
        gamma = dev_upload(S, prefix + ".weight", param(S, prefix + ".weight").data);
        beta = dev_upload(S, prefix + ".bias", param(S, prefix + ".bias").data);
    }
    hipLaunchKernelGGL(k_stats_finalize, dim3(groups), dim3(256), 0, S.cur->st, (const double *)partial, nb, C,
                       cpg, groups, (double)P, gamma, beta, 1e-5f, sc, sh);
    a.scale = sc;
    a.shift = sh;
    return a;
}

// ---- convolution launcher -----------------------------------------------------------------------------
struct ConvGeom {
    int k = 3, stride = 1, pad = 1;
    bool reflect = false;
};

const float *packed_bias(EncState &S, const std::string &name, int cout_pad, bool has_bias)
{
    const std::string key = name + "#b" + std::to_string(cout_pad);
    if (S.dev.count(key)) return S.dev[key];
    std::vector<float> b(cout_pad, 0.f);
    if (has_bias) {
        const auto &src = param(S, name).data;
        for (size_t i = 0; i < src.size(); ++i) b[i] = src[i];
    }
    return dev_upload(S, key, b);
}

// ---- LDS-DMA convolution path (k_prep_act + k_conv_dma) ---------------------------------------------------
// one contiguous K run of the implicit GEMM: `nx` consecutive pixels of P starting at (dy, dx) relative to the row's base
// pixel, pixel i of the run multiplying kernel tap (ky, kx[i])
struct KSpan {
    int dy = 0, dx = 0, nx = 0, ky = 0;
    int kx[7] = {0};
};

// the parameter as given, on the device: uploaded once from the host copy (pifu_enc_set_param) or already there
// (pifu_enc_set_param_device)
const float *raw_on_device(EncState &S, const std::string &name, HostParam &w)
{
    if (w.raw) return w.raw;
    if (w.data.empty()) throw Fail{name + ": parameter has neither host nor device values"};
    if (hipMalloc(&w.raw, w.data.size() * sizeof(float)) != hipSuccess) throw Fail{"hipMalloc failed for " + name};
    w.owned = true;
    if (hipMemcpyAsync(w.raw, w.data.data(), w.data.size() * sizeof(float), hipMemcpyHostToDevice, S.cur->st) != hipSuccess)
        throw Fail{"upload failed for " + name};
    return w.raw;
}

// weights -> [cout_pad][sum over spans of roundup(nx * C, 32)], zero where a run is padded to the K step (k_pack_weight)
const float *packed_dma_weight(EncState &S, const std::string &key, const std::string &wname, HostParam &w, bool transposed, int C,
                               int cout, int cout_pad, const std::vector<KSpan> &spans, int *ldw_out)
{
    int ldw = 0;
    for (const KSpan &s : spans) ldw += (int)round_up((int64_t)s.nx * C, BK);
    *ldw_out = ldw;
    if (S.dev.count(key)) return S.dev[key];
    if ((int)spans.size() > CONV_MAX_SPANS) throw Fail{key + ": too many K runs"};
    PackArgs a{};
    a.src = raw_on_device(S, wname, w);
    a.cout = cout; a.cout_pad = cout_pad; a.C = C; a.ldw = ldw; a.transposed = transposed ? 1 : 0;
    a.d1 = (int)w.shape[1]; a.kh = (int)w.shape[2]; a.kw = (int)w.shape[3];
    a.ci = transposed ? (int)w.shape[0] : (int)w.shape[1];
    if (a.ci > C) throw Fail{key + ": input channels exceed the prepared width"};
    a.nspan = (int)spans.size();
    int base = 0;
    for (int k = 0; k < a.nspan; ++k) {
        a.base[k] = base; a.nx[k] = spans[k].nx; a.ky[k] = spans[k].ky;
        for (int i = 0; i < 7; ++i) a.kx[k][i] = spans[k].kx[i];
        base += (int)round_up((int64_t)spans[k].nx * C, BK);
    }
    a.base[a.nspan] = base;
    float *d = nullptr;
    const size_t n = (size_t)cout_pad * ldw;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) throw Fail{"hipMalloc failed for " + key};
    a.dst = d;
    // on the lane's stream: ordered before the convolution that reads it; a packed copy is written once and only read after
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S.cur->st, a);
    S.dev[key] = d;
    return d;
}

// N tile of a DMA convolution: no wider than the output needs and never 128 - the 128 x 64 tile keeps three blocks per CU
// (48 KB of LDS) and measured 129 TFLOP/s on a 128 -> 64 layer where the 128 x 128 tile (64 KB, two blocks) reached 70 on
// 128 -> 128 - and narrower while that still helps to put ~2 tiles on every CU
int pick_tile_n(int ntm, int cout, int nsteps)
{
    const int widest = cout > 32 ? 64 : 32;
    if (widest == 64 && nsteps >= 96) return 64;      // deep K: split-K supplies the parallelism, the wider tile the MFMA rate
    for (int tn = widest; tn >= 32; tn /= 2)
        if ((int64_t)ntm * ((cout + tn - 1) / tn) >= 512 || tn == 32) return tn;
    return 32;
}

// act(x) into the padded scratch image (one reusable buffer: launches are stream-ordered)
float *prepare_input(EncState &S, const Tensor &x, const Affine &act, int Cp, int pad_lo, int Hp, int Wp, bool reflect,
                     unsigned *bytes_out)
{
    const size_t floats = (size_t)Hp * Wp * Cp + 64;
    if (floats * sizeof(float) >= ((size_t)1 << 32)) throw Fail{"prepared conv input exceeds 4 GB"};
    if (pifu_ensure(S.cur->prep, floats * sizeof(float)) != PIFU_OK) throw Fail{"prepared conv input allocation failed"};
    PrepArgs p{};
    p.X = x.p; p.H = x.H; p.W = x.W; p.ldx = x.ld;
    p.scale = act.scale; p.shift = act.shift; p.slope = act.slope;
    p.pad_lo = pad_lo; p.reflect = reflect ? 1 : 0;
    p.P = static_cast<float *>(S.cur->prep.p); p.Hp = Hp; p.Wp = Wp; p.Cp = Cp;
    const size_t threads = (size_t)Hp * Wp * (Cp / 4) + 16;
    hipLaunchKernelGGL(k_prep_act, dim3(nblk(threads, 256)), dim3(256), 0, S.cur->st, p);
    *bytes_out = (unsigned)(floats * sizeof(float));
    return p.P;
}

// `stat` (may be null): receives the statistics partials of the stored values, if this launch can produce them (no split-K)
// fills the derived fields of `a` (tile grid, split-K plan and workspace, statistics target); returns the grid size
unsigned plan_conv_dma(EncState &S, ConvDmaArgs &a, int tn, Tensor *stat, int stat_class, int stat_classes)
{
    a.ntm = (a.nrows + BM - 1) / BM;
    a.ntn = (a.cout + tn - 1) / tn;
    a.ldp = a.ntn * tn;
    const int tiles = (int)round_up(a.ntm, 8) * a.ntn;
    const int nsteps = a.ldw / BK;
    // fewer than ~1.5 tiles per CU: split K so that ~768 blocks are in flight, at least 8 K steps per split
    int nsplit = 1;
    if (a.ntm * a.ntn < 384) {
        nsplit = (768 + a.ntm * a.ntn - 1) / (a.ntm * a.ntn);
        if (nsplit > nsteps / 8) nsplit = nsteps / 8;
        if (nsplit < 1) nsplit = 1;
    }
    a.steps_per_split = (nsteps + nsplit - 1) / nsplit;
    nsplit = (nsteps + a.steps_per_split - 1) / a.steps_per_split;
    a.nsplit = nsplit;
    a.tiles_per_split = tiles;
    a.part = nullptr;
    if (nsplit > 1) {
        const size_t bytes = (size_t)nsplit * a.ntm * BM * a.ldp * sizeof(float);
        if (pifu_ensure(S.cur->splitk, bytes) != PIFU_OK) throw Fail{"split-K workspace allocation failed"};
        a.part = static_cast<float *>(S.cur->splitk.p);
    }
    a.stat = nullptr;
    a.stat_tile0 = 0;
    if (stat && nsplit == 1 && S.fuse_stats) {
        if (stat_class == 0) {
            stat->stat_nb = a.ntm * stat_classes;
            stat->stat = reinterpret_cast<double *>(arena_take(S, (size_t)stat->stat_nb * a.cout * 2 * 2));
        }
        a.stat = stat->stat;
        a.stat_tile0 = stat_class * a.ntm;
    } else if (stat) {
        stat->stat = nullptr;
        stat->stat_nb = 0;
    }
    return (unsigned)tiles * nsplit;
}

void launch_conv_dma(EncState &S, ConvDmaArgs a, int tn, Tensor *stat = nullptr, int stat_class = 0, int stat_classes = 1)
{
    const unsigned grid = plan_conv_dma(S, a, tn, stat, stat_class, stat_classes);
    if (tn == 64) hipLaunchKernelGGL(k_conv_dma<64>, dim3(grid), dim3(256), 0, S.cur->st, a);
    else hipLaunchKernelGGL(k_conv_dma<32>, dim3(grid), dim3(256), 0, S.cur->st, a);
    if (a.nsplit > 1) {
        const size_t threads = (size_t)a.nrows * (a.ldp / 4);
        hipLaunchKernelGGL(k_conv_dma_reduce, dim3(nblk(threads, 256)), dim3(256), 0, S.cur->st, a);
    }
}

// y (a channel slice of `out`) = conv(act(x)); out must have the conv's output size
void conv2d(EncState &S, const std::string &wname, const std::string &bname, const Tensor &x, const Affine &act,
            const ConvGeom &g, Tensor &out, const Tensor *residual = nullptr)
{
    if (x.ld % 4 || x.C % BK) throw Fail{wname + ": input channels must be padded to 32"};
    HostParam &wp = param(S, wname);
    if (wp.shape.size() != 4) throw Fail{wname + ": expected a 4-d conv weight"};
    const int co = (int)wp.shape[0], ci = (int)wp.shape[1], kk = (int)wp.shape[2];
    if (co != out.C || kk != g.k) throw Fail{wname + ": shape mismatch"};
    const int nrows = out.H * out.W;
    if (g.k > CONV_MAX_SPANS) throw Fail{wname + ": kernel too large"};
    {
        // the packed network input and the padded `out` of a stack carry more channels than the weight reads: prepare
        // only the channels that exist (rounded to the 16-byte chunk)
        const int Cp = ci < x.C ? (int)round_up(ci, 4) : x.C;
        const int tn = pick_tile_n((nrows + BM - 1) / BM, co, g.k * (int)round_up((int64_t)g.k * Cp, BK) / BK);
        const int cout_pad = (int)round_up(co, tn);
        const int Hp = std::max(x.H + 2 * g.pad, (out.H - 1) * g.stride + g.k), Wp = std::max(x.W + 2 * g.pad, (out.W - 1) * g.stride + g.k);
        std::vector<KSpan> spans(g.k);
        for (int ky = 0; ky < g.k; ++ky) {
            spans[ky].dy = ky; spans[ky].dx = 0; spans[ky].nx = g.k; spans[ky].ky = ky;
            for (int i = 0; i < g.k; ++i) spans[ky].kx[i] = i;
        }
        ConvDmaArgs a{};
        a.W = packed_dma_weight(S, wname + "#dma" + std::to_string(Cp) + "_" + std::to_string(tn), wname, wp, false, Cp, co, cout_pad, spans, &a.ldw);
        a.bias = packed_bias(S, bname.empty() ? wname + "#nobias" : bname, cout_pad, !bname.empty());
        if (x.prepP) {      // the packed network input: already in this convolution's prepared form
            if (x.pCp != Cp || x.pHp != Hp || x.pWp != Wp || x.pPad != g.pad || x.pReflect != g.reflect || act.slope != 1.f)
                throw Fail{wname + ": the packed input was prepared for another geometry"};
            a.P = x.prepP; a.p_bytes = x.pBytes;
        } else
            a.P = prepare_input(S, x, act, Cp, g.pad, Hp, Wp, g.reflect, &a.p_bytes);
        a.Wp = Wp; a.C = Cp; a.RW = out.W; a.nrows = nrows; a.is = g.stride;
        a.nspan = g.k;
        for (int ky = 0; ky < g.k; ++ky) {
            a.span_off[ky] = ky * Wp * Cp * 4;
            a.span_steps[ky] = (int)round_up((int64_t)g.k * Cp, BK) / BK;
        }
        a.Y = out.p; a.ldy = out.ld; a.cout = co;
        if (residual) { a.R = residual->p; a.ldr = residual->ld; }
        a.os = 1; a.oy0 = 0; a.ox0 = 0; a.Wout = out.W;
        launch_conv_dma(S, a, tn, &out);
        book_conv(S, 2.0 * nrows * co * (double)ci * g.k * g.k);
    }
}

// ConvTranspose2d(k 3, stride 2, padding 1, output_padding 1) as 4 output-parity classes (networks.py:155-157)
void conv_transpose2d(EncState &S, const std::string &wname, const std::string &bname, const Tensor &x, const Affine &act,
                      Tensor &out)
{
    HostParam &w = param(S, wname);                  // [Cin][Cout][3][3]
    const int ci = (int)w.shape[0], co = (int)w.shape[1];
    if (ci != x.C || co != out.C || out.H != 2 * x.H || out.W != 2 * x.W) throw Fail{wname + ": transposed conv shape mismatch"};
    {
        // zero row / column behind the last pixel (the taps reach one pixel down / right), activation applied once
        const int Cp = x.C, Hp = x.H + 1, Wp = x.W + 1;
        const int tn = pick_tile_n((x.H * x.W + BM - 1) / BM, co, Cp / BK);
        const int cout_pad = (int)round_up(co, tn);
        const float *bias = packed_bias(S, bname, cout_pad, true);
        unsigned p_bytes = 0;
        const float *P = prepare_input(S, x, act, Cp, 0, Hp, Wp, false, &p_bytes);
        ConvDma4 four{};
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                // oy = 2 ry + py: input row ry uses ky = 1 (py == 0) or ky = 2 (py == 1); input row ry + 1 uses ky = 0 (py == 1)
                std::vector<KSpan> spans;
                for (int iy = 0; iy <= py; ++iy) {
                    KSpan sp;
                    sp.dy = iy; sp.dx = 0; sp.ky = py == 0 ? 1 : (iy == 0 ? 2 : 0);
                    sp.nx = px + 1;
                    for (int ix = 0; ix <= px; ++ix) sp.kx[ix] = px == 0 ? 1 : (ix == 0 ? 2 : 0);
                    spans.push_back(sp);
                }
                ConvDmaArgs a{};
                a.W = packed_dma_weight(S, wname + "#dmaT" + std::to_string(py) + std::to_string(px) + "_" + std::to_string(tn), wname, w, true,
                                        Cp, co, cout_pad, spans, &a.ldw);
                a.bias = bias;
                a.P = P; a.p_bytes = p_bytes; a.Wp = Wp; a.C = Cp; a.RW = x.W; a.nrows = x.H * x.W; a.is = 1;
                a.nspan = (int)spans.size();
                for (int i = 0; i < a.nspan; ++i) {
                    a.span_off[i] = (spans[i].dy * Wp + spans[i].dx) * Cp * 4;
                    a.span_steps[i] = (int)round_up((int64_t)spans[i].nx * Cp, BK) / BK;
                }
                a.Y = out.p; a.ldy = out.ld; a.cout = co;
                a.os = 2; a.oy0 = py; a.ox0 = px; a.Wout = out.W;
                four.a[py * 2 + px] = a;
            }
        // the classes share the tile grid; whether they are split along K depends on that grid only (the deepest class decides
        // the split count of each, so it is looked at per class)
        const int ntiles = ((x.H * x.W + BM - 1) / BM) * ((co + tn - 1) / tn);
        if (ntiles >= 384) {            // no split-K in any class: one launch for all four
            unsigned grid = 0;
            for (int c = 0; c < 4; ++c) grid = plan_conv_dma(S, four.a[c], tn, &out, c, 4);
            if (tn == 64) hipLaunchKernelGGL(k_conv_dma4<64>, dim3(grid, 4), dim3(256), 0, S.cur->st, four);
            else hipLaunchKernelGGL(k_conv_dma4<32>, dim3(grid, 4), dim3(256), 0, S.cur->st, four);
        } else {
            for (int c = 0; c < 4; ++c) launch_conv_dma(S, four.a[c], tn, &out, c, 4);
        }
        book_conv(S, 2.0 * x.H * x.W * (double)co * ci * 9.0);       // every input pixel meets each of the 3 x 3 taps once
    }
}

// ---- building blocks -------------------------------------------------------------------------------------
// ConvBlock (Filter.py:23-69): cat[conv1(a(bn1 x)), conv2(a(bn2 o1)), conv3(a(bn3 o2))] + (x | conv1x1(a(bn4 x)))
Tensor conv_block(EncState &S, const std::string &p, Tensor x, int cout)
{
    Tensor out = new_tensor(S, x.H, x.W, cout);
    const ConvGeom g3{3, 1, 1, false}, g1{1, 1, 0, false};
    Affine a1 = norm_affine(S, p + ".bn1", x, S.norm); a1.slope = 0.f;
    Tensor o1 = slice(out, 0, cout / 2), o2 = slice(out, cout / 2, cout / 4), o3 = slice(out, cout / 2 + cout / 4, cout / 4);
    // o2 / o3 are 32..64-channel slices: their inputs must be 32-aligned slices, which holds for all widths used
    conv2d(S, p + ".conv1.weight", "", x, a1, g3, o1);
    Affine a2 = norm_affine(S, p + ".bn2", o1, S.norm); a2.slope = 0.f;
    conv2d(S, p + ".conv2.weight", "", o1, a2, g3, o2);
    Affine a3 = norm_affine(S, p + ".bn3", o2, S.norm); a3.slope = 0.f;
    conv2d(S, p + ".conv3.weight", "", o2, a3, g3, o3);
    if (x.C != cout) {
        Affine a4 = norm_affine(S, p + ".bn4", x, S.norm); a4.slope = 0.f;      // same statistics as bn1: reduced once
        conv2d(S, p + ".downsample.2.weight", "", x, a4, g1, out, &out);      // out = conv1x1(...) + cat (in place)
    } else if (S.norm == 0) {       // eval-mode BatchNorm: nobody needs statistics of the sum
        hipLaunchKernelGGL(k_add, dim3(nblk(out.pixels() * (cout / 4), 256)), dim3(256), 0, S.cur->st, (const float *)out.p, out.ld,
                           (const float *)x.p, x.ld, (const float *)nullptr, (const float *)nullptr, out.p, out.ld, cout, out.pixels());
    } else {                        // the next block normalises this sum: leave its statistics behind
        const int P = (int)out.pixels(), rows_per_block = stat_rows_per_block(P);
        out.stat_nb = (P + rows_per_block - 1) / rows_per_block;
        out.stat = reinterpret_cast<double *>(arena_take(S, (size_t)out.stat_nb * cout * 2 * 2));
        hipLaunchKernelGGL(k_add_stats, dim3(out.stat_nb), dim3(stat_threads(cout)), 0, S.cur->st, (const float *)out.p, out.ld,
                           (const float *)x.p, x.ld, out.p, out.ld, cout, P, rows_per_block, out.stat);
    }
    return out;
}

// partial rows of a statistics-producing element-wise pass over P pixels (the geometry k_channel_stats would use)
void attach_stats(EncState &S, Tensor &t, int *rows_per_block)
{
    const int P = (int)t.pixels();
    // ~2 048 blocks: these passes also COMPUTE (16 taps per bicubic output), so they want more waves in flight than the pure
    // reduction's 256 blocks; the finalize kernel takes any number of partial rows
    int rows = (P + 2047) / 2048;
    *rows_per_block = rows < 8 ? 8 : rows;
    t.stat_nb = (P + *rows_per_block - 1) / *rows_per_block;
    t.stat = reinterpret_cast<double *>(arena_take(S, (size_t)t.stat_nb * t.C * 2 * 2));
}

Tensor avg_pool(EncState &S, const Tensor &x)
{
    Tensor o = new_tensor(S, x.H / 2, x.W / 2, x.C);
    if (S.norm != 0 && S.fuse_stats) {      // the ConvBlock that follows normalises this tensor: leave its statistics behind
        int rpb = 0;
        attach_stats(S, o, &rpb);
        hipLaunchKernelGGL(k_avgpool2_stats, dim3(o.stat_nb), dim3(stat_threads(x.C)), 0, S.cur->st, (const float *)x.p, x.ld, x.W, o.p, o.ld,
                           x.C, o.H, o.W, rpb, o.stat);
        return o;
    }
    hipLaunchKernelGGL(k_avgpool2, dim3(nblk(o.pixels() * (x.C / 4), 256)), dim3(256), 0, S.cur->st, (const float *)x.p, x.ld, x.W, o.p, o.ld,
                       x.C, o.H, o.W);
    return o;
}

Tensor hourglass(EncState &S, const std::string &p, int level, const Tensor &x)
{
    Tensor up1, low;
    const auto low_chain = [&]() {
        Tensor l = conv_block(S, p + ".b2_" + std::to_string(level), avg_pool(S, x), x.C);
        if (level > 1) l = hourglass(S, p, level - 1, l);
        else l = conv_block(S, p + ".b2_plus_" + std::to_string(level), l, x.C);
        return conv_block(S, p + ".b3_" + std::to_string(level), l, x.C);
    };
    if (S.branch_free > 0) {
        // the low chain on a branch lane (it continues from what this lane has issued: x is complete there), up1 here
        Lane *here = S.cur;
        const int b = S.branch_next++;
        --S.branch_free;
        Lane *br = &S.lanes[b];
        if (hipEventRecord(S.ev_bfork[b - 3], here->st) != hipSuccess || hipStreamWaitEvent(br->st, S.ev_bfork[b - 3], 0) != hipSuccess)
            throw Fail{"hourglass: fork of the branch lane failed"};
        S.cur = br;
        low = low_chain();
        S.cur = here;
        up1 = conv_block(S, p + ".b1_" + std::to_string(level), x, x.C);
        if (hipEventRecord(S.ev_bjoin[b - 3], br->st) != hipSuccess || hipStreamWaitEvent(here->st, S.ev_bjoin[b - 3], 0) != hipSuccess)
            throw Fail{"hourglass: join of the branch lane failed"};
        --S.branch_next;                        // the lane is free again for the next stack's hourglass
        ++S.branch_free;
    } else {
        up1 = conv_block(S, p + ".b1_" + std::to_string(level), x, x.C);
        low = low_chain();
    }
    Tensor out = new_tensor(S, x.H, x.W, x.C);
    if (S.norm != 0 && S.fuse_stats) {      // (see avg_pool)
        int rpb = 0;
        attach_stats(S, out, &rpb);
        hipLaunchKernelGGL(k_bicubic_up2_add_stats, dim3(out.stat_nb), dim3(stat_threads(x.C)), 0, S.cur->st, (const float *)low.p, low.ld,
                           low.H, low.W, (const float *)up1.p, up1.ld, out.p, out.ld, x.C, rpb, out.stat);
        return out;
    }
    hipLaunchKernelGGL(k_bicubic_up2_add, dim3(nblk(out.pixels() * (x.C / 4), 256)), dim3(256), 0, S.cur->st, (const float *)low.p, low.ld,
                       low.H, low.W, (const float *)up1.p, up1.ld, out.p, out.ld, x.C);
    return out;
}

// Filter.forward (Filter.py:186-228); returns the LAST stack's output (what inference keeps)
Tensor hg_filter(EncState &S, const std::string &p, const Tensor &in32, int n_stack, int depth, int last_ch, bool ave_pool,
                 std::vector<Tensor> *stacks = nullptr)
{
    const ConvGeom g1{1, 1, 0, false};
    const int H1 = (in32.H + 2 * 3 - 7) / 2 + 1, W1 = (in32.W + 2 * 3 - 7) / 2 + 1;
    Tensor c1 = new_tensor(S, H1, W1, 64);
    conv2d(S, p + ".conv1.weight", p + ".conv1.bias", in32, raw(S), ConvGeom{7, 2, 3, false}, c1);
    // x = relu(bn1(conv1)) is both normalised again (ConvBlock.bn1/bn4) and used by the shortcut: materialise it
    Affine b1 = norm_affine(S, p + ".bn1", c1, S.norm);
    Tensor x = new_tensor(S, H1, W1, 64);
    hipLaunchKernelGGL(k_affine_act, dim3(nblk(x.pixels() * 16, 256)), dim3(256), 0, S.cur->st, (const float *)c1.p, c1.ld, b1.scale, b1.shift,
                       0.f, x.p, x.ld, 64, x.pixels());
    x = conv_block(S, p + ".conv2", x, 128);
    if (ave_pool) x = avg_pool(S, x);
    x = conv_block(S, p + ".conv3", x, 128);
    x = conv_block(S, p + ".conv4", x, 256);
    Tensor prev = x, out;
    for (int s = 0; s < n_stack; ++s) {
        const std::string ss = std::to_string(s);
        Tensor ll = hourglass(S, p + ".m" + ss, depth, prev);
        ll = conv_block(S, p + ".top_m_" + ss, ll, 256);
        Tensor cl = new_tensor(S, ll.H, ll.W, 256);
        conv2d(S, p + ".conv_last" + ss + ".weight", p + ".conv_last" + ss + ".bias", ll, raw(S), g1, cl);
        Affine be = norm_affine(S, p + ".bn_end" + ss, cl, S.norm); be.slope = 0.f;        // relu(bn_end(.)) as a prologue
        const int lc_pad = (int)round_up(last_ch, BK);
        out = new_tensor(S, ll.H, ll.W, lc_pad);                                             // padded so that `al` can read it
        if (lc_pad != last_ch) (void)hipMemsetAsync(out.p, 0, out.pixels() * lc_pad * sizeof(float), S.cur->st);
        Tensor out_valid = out; out_valid.C = last_ch;
        conv2d(S, p + ".l" + ss + ".weight", p + ".l" + ss + ".bias", cl, be, g1, out_valid);
        if (stacks) stacks->push_back(out_valid);                                            // train mode keeps every stack's output
        if (s < n_stack - 1) {
            Tensor t1 = new_tensor(S, ll.H, ll.W, 256), t2 = new_tensor(S, ll.H, ll.W, 256);
            conv2d(S, p + ".bl" + ss + ".weight", p + ".bl" + ss + ".bias", cl, be, g1, t1, &prev);   // previous + bl(ll)
            conv2d(S, p + ".al" + ss + ".weight", p + ".al" + ss + ".bias", out, raw(S), g1, t2, &t1); // + al(out)
            prev = t2;
        }
    }
    out.C = last_ch;
    return out;
}

// ---- the ResnetBlock trunk through Winograd F(2x2, 3x3) (kernels above) --------------------------------------------------
bool wino_ok(const Tensor &r, int c)
{
    if (r.H % 2 || r.W % 2 || r.ld != c || c % 64 || c % WCG) return false;
    const int T = (r.H / 2) * (r.W / 2);
    // whole 128-row GEMM tiles per Winograd position, and the plane of 16 channels in (half of) a CU's LDS
    return T % BM == 0 && (size_t)r.H * r.W * WPX * 4 + 2048 <= (size_t)96 * 1024;
}

const float *wino_weight(EncState &S, const std::string &wname, int c)
{
    const std::string key = wname + "#wino";
    if (S.dev.count(key)) return S.dev[key];
    HostParam &w = param(S, wname);
    if (w.shape.size() != 4 || w.shape[0] != c || w.shape[1] != c || w.shape[2] != 3 || w.shape[3] != 3)
        throw Fail{wname + ": expected a [C][C][3][3] weight"};
    const float *src = raw_on_device(S, wname, w);
    float *d = nullptr;
    if (hipMalloc(&d, (size_t)16 * c * c * sizeof(float)) != hipSuccess) throw Fail{"hipMalloc failed for " + key};
    hipLaunchKernelGGL(k_wino_weight, dim3(nblk((size_t)c * c, 256)), dim3(256), 0, S.cur->st, src, c, c, d);
    S.dev[key] = d;
    return d;
}

// M[16][T][c] = V[16][T][c] x U[16][c][c]^T: the 16 positions as one launch of the convolution kernel (a 1 x 1 "convolution"
// over 16 T rows whose weight matrix changes every T / 128 row tiles)
void wino_gemm(EncState &S, const float *V, const float *U, float *M, int T, int c, int H, int W)
{
    ConvDmaArgs a{};
    const int rows = 16 * T;
    a.P = V; a.p_bytes = (unsigned)((size_t)rows * c * 4);
    a.Wp = rows; a.C = c; a.RW = rows; a.nrows = rows; a.is = 1;
    a.nspan = 1; a.span_off[0] = 0; a.span_steps[0] = c / BK;
    a.W = U; a.ldw = c; a.bias = zeros(S);
    a.Y = M; a.ldy = c; a.cout = c;
    a.os = 1; a.oy0 = 0; a.ox0 = 0; a.Wout = rows;
    a.wg_tiles = T / BM; a.wg_stride = (size_t)c * c;
    launch_conv_dma(S, a, 64);
    book_conv(S, 2.0 * 16.0 * T * (double)c * c, 2.0 * H * W * (double)c * c * 9.0);
}

// k_wino_bridge keeps the plane of 16 channels in LDS: more than the 64 KB a kernel gets by default.  Asked for once per
// context (i.e. per device); where the device refuses, the trunk stays on the direct convolutions.
bool wino_available(EncState &S)
{
    if (S.wino_state == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_wino_bridge),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) (void)hipGetLastError();
        S.wino_state = e == hipSuccess ? 1 : -1;
    }
    return S.wino_state > 0;
}

void wino_bridge(EncState &S, const WinoArgs &wa)
{
    const size_t lds = (size_t)wa.H * wa.W * WPX * 4 + 4 * WCG * 2 * sizeof(double) + 2 * WCG * sizeof(float);
    hipLaunchKernelGGL(k_wino_bridge, dim3(wa.C / WCG), dim3(256), lds, S.cur->st, wa);
}

// GlobalGenerator (networks.py:131-207) with InstanceNorm(affine=False); returns NCHW [3][H][W] in `dst`
void global_generator(EncState &S, const std::string &p, const Tensor &in32, float *dst_nchw)
{
    const int inst = 3;
    Tensor x = new_tensor(S, in32.H, in32.W, 64);
    conv2d(S, p + ".model.1.weight", p + ".model.1.bias", in32, raw(S), ConvGeom{7, 1, 3, true}, x);
    Affine act = norm_affine(S, "", x, inst); act.slope = 0.f;
    int idx = 4, c = 64;
    for (int i = 0; i < 4; ++i) {
        Tensor y = new_tensor(S, x.H / 2, x.W / 2, 2 * c);
        const std::string m = p + ".model." + std::to_string(idx);
        conv2d(S, m + ".weight", m + ".bias", x, act, ConvGeom{3, 2, 1, false}, y);
        act = norm_affine(S, "", y, inst); act.slope = 0.f;
        x = y; c *= 2; idx += 3;
    }
    // the first ResnetBlock consumes relu(IN(x)) as its RAW input and as the residual: materialise it
    Tensor r = new_tensor(S, x.H, x.W, c);
    hipLaunchKernelGGL(k_affine_act, dim3(nblk(r.pixels() * (c / 4), 256)), dim3(256), 0, S.cur->st, (const float *)x.p, x.ld, act.scale,
                       act.shift, 0.f, r.p, r.ld, c, r.pixels());
    const ConvGeom gr{3, 1, 1, true};
    if (wino_ok(r, c) && !S.direct_trunk && wino_available(S)) {
        // the trunk through Winograd: `r` is the residual stream (updated in place by every block), V / M the transformed
        // input and the products of the convolution in flight (one buffer each: the launches are stream-ordered)
        const int T = (r.H / 2) * (r.W / 2);
        float *V = arena_take(S, (size_t)16 * T * c), *M = arena_take(S, (size_t)16 * T * c);
        WinoArgs wa{};
        wa.M = M; wa.V = V; wa.R = r.p; wa.H = r.H; wa.W = r.W; wa.C = c; wa.T = T; wa.eps = 1e-5f;
        wa.mode = 0; wa.bias = nullptr;
        wino_bridge(S, wa);
        for (int b = 0; b < 9; ++b) {
            const std::string q = p + ".model." + std::to_string(idx) + ".conv_block";
            wino_gemm(S, V, wino_weight(S, q + ".1.weight", c), M, T, c, r.H, r.W);
            wa.mode = 1; wa.bias = dev_upload(S, q + ".1.bias", param(S, q + ".1.bias").data); wa.V = V;
            wino_bridge(S, wa);                                   // relu(IN(conv1)) -> conv2's input
            wino_gemm(S, V, wino_weight(S, q + ".5.weight", c), M, T, c, r.H, r.W);
            wa.mode = 2; wa.bias = dev_upload(S, q + ".5.bias", param(S, q + ".5.bias").data); wa.V = b < 8 ? V : nullptr;
            wino_bridge(S, wa);                                   // r += IN(conv2) -> the next block's input
            idx += 1;
        }
    } else
    for (int b = 0; b < 9; ++b) {
        const std::string q = p + ".model." + std::to_string(idx) + ".conv_block";
        Tensor h1 = new_tensor(S, r.H, r.W, c), h2 = new_tensor(S, r.H, r.W, c), o = new_tensor(S, r.H, r.W, c);
        conv2d(S, q + ".1.weight", q + ".1.bias", r, raw(S), gr, h1);
        Affine a1 = norm_affine(S, "", h1, inst); a1.slope = 0.f;
        conv2d(S, q + ".5.weight", q + ".5.bias", h1, a1, gr, h2);
        Affine a2 = norm_affine(S, "", h2, inst);
        hipLaunchKernelGGL(k_add, dim3(nblk(o.pixels() * (c / 4), 256)), dim3(256), 0, S.cur->st, (const float *)r.p, r.ld, (const float *)h2.p,
                           h2.ld, a2.scale, a2.shift, o.p, o.ld, c, o.pixels());          // x + IN(conv(...))
        r = o; idx += 1;
    }
    x = r;
    act = raw(S);
    for (int i = 0; i < 4; ++i) {
        Tensor y = new_tensor(S, x.H * 2, x.W * 2, c / 2);
        const std::string m = p + ".model." + std::to_string(idx);
        conv_transpose2d(S, m + ".weight", m + ".bias", x, act, y);
        act = norm_affine(S, "", y, inst); act.slope = 0.f;
        x = y; c /= 2; idx += 3;
    }
    // ReflectionPad(3) + Conv7x7 (64 -> 3) + Tanh
    const std::string m = p + ".model." + std::to_string(idx + 1);
    HostParam &w = param(S, m + ".weight");
    const std::string key = m + "#w7";
    if (!S.dev.count(key)) {
        // [3][ci][7][7] -> [3][49][ci]: seven K runs (one per kernel row) of seven taps, ci channels each
        const int ci = (int)w.shape[1];
        if ((7 * ci) % BK) throw Fail{"k_conv7_to3_tanh: 7 * Cin must be a multiple of 32"};
        std::vector<KSpan> spans(7);
        for (int ky = 0; ky < 7; ++ky) {
            spans[ky].nx = 7; spans[ky].ky = ky;
            for (int i = 0; i < 7; ++i) spans[ky].kx[i] = i;
        }
        int ldw = 0;
        packed_dma_weight(S, key, m + ".weight", w, false, ci, 3, 3, spans, &ldw);
    }
    const float *bias = dev_upload(S, m + ".bias", param(S, m + ".bias").data);
    if (x.C % 16 != 0) throw Fail{"k_conv7_to3_tanh needs a multiple of 16 input channels"};
    hipLaunchKernelGGL(k_conv7_to3_tanh, dim3((unsigned)(((x.W + 15) / 16) * ((x.H + 15) / 16))), dim3(256), 0, S.cur->st, (const float *)x.p,
                       x.ld, x.H, x.W, x.C, act.scale, act.shift, (const float *)S.dev[key], bias, dst_nchw);
    book_conv(S, 2.0 * x.H * x.W * 3.0 * (double)w.shape[1] * 49.0);
}

// the network input of a 7 x 7 first convolution (pad 3: zeros for the HGFilters, mirrored for the normal nets), in prepared form
Tensor pack_input(EncState &S, const float *img, int Cimg, int H, int W, const float *nF, const float *nB, int Hn, int Wn, bool reflect)
{
    const int ci = Cimg + (nF ? 3 : 0) + (nB ? 3 : 0), pad = 3;
    if (ci > 32) throw Fail{"more than 32 input channels"};
    Tensor t;
    t.H = H; t.W = W; t.C = 32; t.ld = 32;          // the logical tensor (channels beyond `ci` are zero); it has no unprepared form
    t.pCp = (int)round_up(ci, 4); t.pHp = H + 2 * pad; t.pWp = W + 2 * pad; t.pPad = pad; t.pReflect = reflect;
    const size_t floats = (size_t)t.pHp * t.pWp * t.pCp + 64;
    if (floats * sizeof(float) >= ((size_t)1 << 32)) throw Fail{"prepared network input exceeds 4 GB"};
    float *P = arena_take(S, floats);
    t.prepP = P; t.pBytes = (unsigned)(floats * sizeof(float));
    hipLaunchKernelGGL(k_pack_prepared, dim3(nblk((size_t)t.pHp * t.pWp + 16, 256)), dim3(256), 0, S.cur->st, img, Cimg, H, W, nF, nB, Hn, Wn,
                       P, t.pHp, t.pWp, t.pCp, pad, reflect ? 1 : 0);
    return t;
}

// make lane `k` current for a call on the caller's stream `st`: lane 0 issues into `st` itself, lanes 1 and 2 into streams
// of their own (created on first use); the lane's arena holds at least `arena_bytes` and is empty
int begin(pifu_ctx *ctx, EncState &S, int k, size_t arena_bytes, hipStream_t st)
{
    PIFU_HIP(hipSetDevice(ctx->device));
    Lane &L = S.lanes[k];
    if (k == 0) L.st = st;
    else if (k == 3) {
        // the first hourglass branch lane issues into netB's stream, which is idle by the time a hourglass runs on lane 0: the
        // runtime deals streams out over a few hardware queues (four by default) in order of creation, and a stream that
        // lands on the queue of the lane it is meant to run beside would be serialised with it
        Lane &B = S.lanes[1];
        if (!B.own) PIFU_HIP(hipStreamCreateWithFlags(&B.own, hipStreamNonBlocking));
        L.st = B.own;
    } else {
        if (!L.own) PIFU_HIP(hipStreamCreateWithFlags(&L.own, hipStreamNonBlocking));
        L.st = L.own;
    }
    if (!S.ev_fork) {
        PIFU_HIP(hipEventCreateWithFlags(&S.ev_fork, hipEventDisableTiming));
        PIFU_HIP(hipEventCreateWithFlags(&S.ev_join, hipEventDisableTiming));
        PIFU_HIP(hipEventCreateWithFlags(&S.ev_coarse, hipEventDisableTiming));
    }
    int rc = pifu_ensure(L.arena, arena_bytes);
    if (rc) return rc;
    L.used = 0;
    S.cur = &L;
    return PIFU_OK;
}

// lane `k` continues from what `from` has issued so far
int fork_lane(EncState &S, int k, hipStream_t from)
{
    PIFU_HIP(hipEventRecord(S.ev_fork, from));
    PIFU_HIP(hipStreamWaitEvent(S.lanes[k].st, S.ev_fork, 0));
    return PIFU_OK;
}

// `to` continues only after everything lane `k` has issued so far
int join_lane(EncState &S, int k, hipStream_t to)
{
    PIFU_HIP(hipEventRecord(S.ev_join, S.lanes[k].st));
    PIFU_HIP(hipStreamWaitEvent(to, S.ev_join, 0));
    return PIFU_OK;
}

}  // namespace

// A coarse call issued with PIFU_ENC_DEFER_JOIN leaves its hourglass running on a stream of the library; whoever reads the
// coarse feature map next - every query / grid entry point, pifu_set_features, the next filter call - first makes its own
// stream wait for it here.
int pifu_enc_join(pifu_ctx *ctx, hipStream_t st)
{
    auto it = g_states.find(ctx);
    if (it == g_states.end() || !it->second->coarse_pending) return PIFU_OK;
    // Every stream that asks while the hourglass is still running waits for it - not only the first one: a later query,
    // set_features or filter call on ANOTHER stream is ordered after the deferred work too.  Once the event has completed
    // nobody needs to wait any more and the flag is dropped.
    const hipError_t q = hipEventQuery(it->second->ev_coarse);
    if (q == hipSuccess) { it->second->coarse_pending = false; return PIFU_OK; }
    if (q != hipErrorNotReady) PIFU_HIP(q);
    (void)hipGetLastError();                   // hipErrorNotReady is an answer, not an error to leave behind
    PIFU_HIP(hipStreamWaitEvent(st, it->second->ev_coarse, 0));
    return PIFU_OK;
}

static int set_param(pifu_ctx *ctx, const char *name, const float *host, const float *dev, const int64_t *shape, int rank,
                     hipStream_t st);

extern "C" int pifu_enc_set_param(pifu_ctx *ctx, const char *name, const float *host, const int64_t *shape, int rank)
{
    PIFU_REQUIRE(ctx && name && host && shape && rank >= 0 && rank <= 4, PIFU_ERR_INVALID, "pifu_enc_set_param: bad argument");
    return set_param(ctx, name, host, nullptr, shape, rank, nullptr);
}

extern "C" int pifu_enc_set_param_device(pifu_ctx *ctx, const char *name, const float *dev, const int64_t *shape, int rank,
                                         pifu_stream stream)
{
    PIFU_REQUIRE(ctx && name && dev && shape && rank == 4, PIFU_ERR_INVALID,
                 "pifu_enc_set_param_device: bad argument (rank-4 convolution weights only)");
    return set_param(ctx, name, nullptr, dev, shape, rank, (hipStream_t)stream);
}

static int set_param(pifu_ctx *ctx, const char *name, const float *host, const float *dev, const int64_t *shape, int rank,
                     hipStream_t st)
{
    EncState &S = state_of(ctx);
    HostParam hp;
    size_t n = 1;
    for (int i = 0; i < rank; ++i) { hp.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    PIFU_HIP(hipSetDevice(ctx->device));
    // Whatever this registration replaces (the library's own upload of the previous values, packed copies derived from them)
    // may still be read by a filter pass in flight: ONE device synchronisation per batch of registrations - the first call
    // after a pass was issued pays it, the hundreds that follow in the same weight refresh do not.
    const auto settle = [&]() -> int {
        if (S.busy) { PIFU_HIP(hipDeviceSynchronize()); S.busy = false; }
        return PIFU_OK;
    };
    auto old = S.params.find(name);
    if (old != S.params.end() && old->second.raw && old->second.owned) {
        const int rc = settle();
        if (rc) return rc;
        (void)hipFree(old->second.raw);
        old->second.raw = nullptr;
    }
    if (host) hp.data.assign(host, host + n);
    else {
        // borrowed: the packing kernels read the caller's tensor when a layer first needs a packed form of it (the first pass
        // at a given image size), so it must stay valid until the parameter is registered again or the context is destroyed
        hp.raw = const_cast<float *>(dev);
        hp.owned = false;
        (void)st;
    }
    S.params[name] = std::move(hp);
    // drop every packed copy derived from this parameter: keys `<name>...` (packed weights / biases) and keys
    // `<module>#...` derived from several tensors of the owning module (folded eval-mode BatchNorm `<module>#bn_eval.s/.b`
    // from weight / bias / running_mean / running_var; the repacked 7x7 output conv `<module>#w7`)
    const std::string nm(name);
    const size_t dot = nm.rfind('.');
    const std::string mod = dot == std::string::npos ? std::string() : nm.substr(0, dot) + "#";
    bool dropped = false;
    for (auto it = S.dev.begin(); it != S.dev.end();) {
        const bool hit = it->first.compare(0, nm.size(), nm) == 0 || (!mod.empty() && it->first.compare(0, mod.size(), mod) == 0);
        if (hit) {
            if (!dropped) { const int rc = settle(); if (rc) return rc; dropped = true; }   // a filter may still be reading it
            (void)hipFree(it->second);
            it = S.dev.erase(it);
        } else ++it;
    }
    return PIFU_OK;
}

extern "C" int pifu_enc_clear(pifu_ctx *ctx)
{
    PIFU_REQUIRE(ctx, PIFU_ERR_INVALID, "pifu_enc_clear: null context");
    auto it = g_states.find(ctx);
    if (it == g_states.end()) return PIFU_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &kv : it->second->dev) (void)hipFree(kv.second);
    for (auto &kv : it->second->params)
        if (kv.second.raw && kv.second.owned) (void)hipFree(kv.second.raw);
    for (Lane &L : it->second->lanes) {
        if (L.arena.p) (void)hipFree(L.arena.p);
        if (L.splitk.p) (void)hipFree(L.splitk.p);
        if (L.prep.p) (void)hipFree(L.prep.p);
        if (L.own) (void)hipStreamDestroy(L.own);
    }
    for (hipEvent_t e : {it->second->ev_fork, it->second->ev_join, it->second->ev_coarse, it->second->ev_bfork[0], it->second->ev_bfork[1],
                         it->second->ev_bjoin[0], it->second->ev_bjoin[1]})
        if (e) (void)hipEventDestroy(e);
    delete it->second;
    g_states.erase(it);
    return PIFU_OK;
}

// PIFuNetwNML.filter (PIFuNetwNML.py:73-97)
extern "C" int pifu_filter_coarse(pifu_ctx *ctx, const float *img_nchw, int Cimg, int H, int W, int num_stack, int hg_depth,
                                  int hg_dim, int norm, int ave_pool, int use_front, int use_back, float *nmlF, float *nmlB,
                                  float *feat_nchw, int flags, pifu_stream stream)
{
    PIFU_REQUIRE(ctx && img_nchw, PIFU_ERR_INVALID, "pifu_filter_coarse: null argument");
    PIFU_REQUIRE(H % 16 == 0 && W % 16 == 0 && H >= 32 && W >= 32, PIFU_ERR_INVALID, "pifu_filter_coarse: H, W must be multiples of 16");
    PIFU_REQUIRE((!use_front || nmlF) && (!use_back || nmlB), PIFU_ERR_INVALID, "pifu_filter_coarse: normal-map output buffers missing");
    PIFU_REQUIRE(norm >= 0 && norm <= 2 && hg_dim % 4 == 0, PIFU_ERR_INVALID, "pifu_filter_coarse: bad norm / hg_dim");
    PIFU_USE(ctx, stream);
    const bool all_stacks = (flags & PIFU_ENC_ALL_STACKS) != 0, defer = (flags & PIFU_ENC_DEFER_JOIN) != 0;
    // the two halves of the call on their own (a rank of a sharded gen_mesh runs one normal net, or the hourglass on normal
    // maps another rank computed): NORMALS_ONLY stops after the enabled normal nets, NORMALS_GIVEN reads nmlF / nmlB as inputs
    const bool normals_only = (flags & PIFU_ENC_NORMALS_ONLY) != 0, normals_given = (flags & PIFU_ENC_NORMALS_GIVEN) != 0;
    PIFU_REQUIRE(!(normals_only && normals_given), PIFU_ERR_INVALID, "pifu_filter_coarse: NORMALS_ONLY and NORMALS_GIVEN exclude each other");
    state_of(ctx).direct_trunk = (flags & PIFU_ENC_NO_WINOGRAD) != 0;
    EncState &S = state_of(ctx);
    S.busy = true;
    hipStream_t st = (hipStream_t)stream;
    const size_t px = (size_t)H * W;
    int rc = pifu_enc_join(ctx, st);      // an earlier deferred hourglass still owns the map this call replaces
    if (rc) return rc;
    try {
        // the two normal nets read the same packed image and nothing of each other: netF on the caller's stream (lane 0),
        // netB beside it on lane 1
        if ((use_front || use_back) && !normals_given) {
            rc = begin(ctx, S, 0, px * 4 * (size_t)1000 + ((size_t)320 << 20), st);
            if (rc) return rc;
            Tensor in = pack_input(S, img_nchw, Cimg, H, W, nullptr, nullptr, H, W, true);
            S.fuse_stats = true;          // every conv of the normal nets is followed by an InstanceNorm
            const bool side = use_front && use_back;
            if (side) {         // lane 1 starts from the packed image, BEFORE anything of netF is in the caller's stream
                rc = begin(ctx, S, 1, px * 4 * (size_t)1000 + ((size_t)320 << 20), st);
                if (rc) return rc;
                rc = fork_lane(S, 1, st);
                if (rc) return rc;
                global_generator(S, "netB", in, nmlB);
                S.cur = &S.lanes[0];
                global_generator(S, "netF", in, nmlF);
                rc = join_lane(S, 1, st);
                if (rc) return rc;
            } else if (use_front) {
                global_generator(S, "netF", in, nmlF);
            } else {
                global_generator(S, "netB", in, nmlB);
            }
        }
        if (normals_only) {
            S.cur = &S.lanes[0];
            PIFU_HIP(hipGetLastError());
            return PIFU_OK;
        }
        // the hourglass filter runs on lane 2: with PIFU_ENC_DEFER_JOIN the call returns while it is still running - the fine
        // level (pifu_filter_fine, lane 0) does not read its result and fills the chip beside it - otherwise the caller's
        // stream waits for it before the call returns
        rc = begin(ctx, S, 2, px * 4 * (size_t)(1600 + 700 * num_stack) + ((size_t)320 << 20), st);
        if (rc) return rc;
        rc = fork_lane(S, 2, st);
        if (rc) return rc;
        Tensor in = pack_input(S, img_nchw, Cimg, H, W, use_front ? nmlF : nullptr, use_back ? nmlB : nullptr, H, W, false);
        S.norm = norm;
        S.fuse_stats = norm != 0;         // eval-mode BatchNorm needs no statistics
        std::vector<Tensor> stacks;
        Tensor f = hg_filter(S, "image_filter", in, num_stack, hg_depth, hg_dim, ave_pool != 0, &stacks);
        // install as the coarse query feature map (NHWC) and optionally export NCHW
        FeatMap &fm = ctx->feat[0];
        const size_t bytes = f.pixels() * f.C * sizeof(float);
        if (bytes > fm.bytes) {
            if (fm.nhwc) PIFU_HIP(hipFree(fm.nhwc));
            fm.nhwc = nullptr;
            PIFU_HIP(hipMalloc(&fm.nhwc, bytes));
            fm.bytes = bytes;
        }
        fm.C = f.C; fm.Cpad = f.C; fm.H = f.H; fm.W = f.W;
        ctx->pix.valid = false;           // the pixel table of the point-list queries belongs to the previous map
        ctx->col_cache.valid = false;     // ... and so do the column tables of the lattice evaluations
        PIFU_HIP(hipMemcpy2DAsync(fm.nhwc, f.C * sizeof(float), f.p, f.ld * sizeof(float), f.C * sizeof(float), f.pixels(),
                                  hipMemcpyDeviceToDevice, S.cur->st));
        if (feat_nchw && all_stacks) {      // train mode (PIFuNetwNML.py:94-97 keeps the whole list): [num_stack][C][H/4][W/4]
            for (size_t k = 0; k < stacks.size(); ++k)
                hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(nblk(f.pixels(), 32), nblk(f.C, 32)), dim3(256), 0, S.cur->st, (const float *)stacks[k].p,
                                   stacks[k].ld, f.C, (int)f.pixels(), feat_nchw + k * f.pixels() * f.C);
        } else if (feat_nchw)
            hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(nblk(f.pixels(), 32), nblk(f.C, 32)), dim3(256), 0, S.cur->st, (const float *)f.p, f.ld, f.C,
                               (int)f.pixels(), feat_nchw);
        PIFU_HIP(hipEventRecord(S.ev_coarse, S.cur->st));
        S.coarse_pending = true;
        S.cur = &S.lanes[0];
        if (!defer) {
            rc = pifu_enc_join(ctx, st);
            if (rc) return rc;
        }
    } catch (const Fail &e) {
        S.cur = &S.lanes[0];
        pifu_set_error("pifu_filter_coarse: %s", e.msg.c_str());
        return PIFU_ERR_STATE;
    }
    PIFU_HIP(hipGetLastError());
    return PIFU_OK;
}

// PIFuMRNet.filter_local (PIFuMRNet.py:83-118), rect = None
extern "C" int pifu_filter_fine(pifu_ctx *ctx, const float *img_nchw, int Cimg, int H, int W, const float *nmlF, const float *nmlB,
                                int Hn, int Wn, int num_stack, int hg_depth, int hg_dim, int norm, float *feat_nchw,
                                pifu_stream stream)
{
    PIFU_REQUIRE(ctx && img_nchw, PIFU_ERR_INVALID, "pifu_filter_fine: null argument");
    PIFU_REQUIRE(H % 8 == 0 && W % 8 == 0 && H >= 32 && W >= 32, PIFU_ERR_INVALID, "pifu_filter_fine: H, W must be multiples of 8");
    PIFU_REQUIRE(norm >= 0 && norm <= 2 && hg_dim % 4 == 0, PIFU_ERR_INVALID, "pifu_filter_fine: bad norm / hg_dim");
    PIFU_USE(ctx, stream);
    EncState &S = state_of(ctx);
    S.busy = true;
    const size_t px = (size_t)H * W / 4;      // the hourglass runs at H/2 x W/2
    int rc = begin(ctx, S, 0, px * 4 * (size_t)(3400 + 2600 * num_stack) + ((size_t)320 << 20), (hipStream_t)stream);
    if (rc) return rc;
    // the two upper hourglass levels issue their low chains into branch lanes (EncState::branch_free)
    for (int b = 0; b < 2; ++b) {
        if (!S.ev_bfork[b]) {
            PIFU_HIP(hipEventCreateWithFlags(&S.ev_bfork[b], hipEventDisableTiming));
            PIFU_HIP(hipEventCreateWithFlags(&S.ev_bjoin[b], hipEventDisableTiming));
        }
        // a low chain holds ~6 tensors of 256 channels at a quarter of its level's pixels (1.5 KB per pixel of the level) plus
        // what the levels below it add (a third of that again) and the statistics partials: 3 KB per pixel, per stack
        rc = begin(ctx, S, 3 + b, (px >> (2 * b)) * 4 * (size_t)750 * num_stack + ((size_t)64 << 20), (hipStream_t)stream);
        if (rc) return rc;
    }
    S.cur = &S.lanes[0];
    S.branch_free = hg_depth >= 2 ? 2 : hg_depth;
    S.branch_next = 3;
    try {
        Tensor in = pack_input(S, img_nchw, Cimg, H, W, nmlF, nmlB, Hn, Wn, false);
        S.norm = norm;
        S.fuse_stats = norm != 0;
        Tensor f = hg_filter(S, "fine.image_filter", in, num_stack, hg_depth, hg_dim, false);
        S.branch_free = 0;
        FeatMap &fm = ctx->feat[1];
        const size_t bytes = f.pixels() * f.C * sizeof(float);
        if (bytes > fm.bytes) {
            if (fm.nhwc) PIFU_HIP(hipFree(fm.nhwc));
            fm.nhwc = nullptr;
            PIFU_HIP(hipMalloc(&fm.nhwc, bytes));
            fm.bytes = bytes;
        }
        fm.C = f.C; fm.Cpad = f.C; fm.H = f.H; fm.W = f.W;
        ctx->col_cache.valid = false;
        PIFU_HIP(hipMemcpy2DAsync(fm.nhwc, f.C * sizeof(float), f.p, f.ld * sizeof(float), f.C * sizeof(float), f.pixels(),
                                  hipMemcpyDeviceToDevice, S.cur->st));
        if (feat_nchw)
            hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(nblk(f.pixels(), 32), nblk(f.C, 32)), dim3(256), 0, S.cur->st, (const float *)f.p, f.ld, f.C,
                               (int)f.pixels(), feat_nchw);
    } catch (const Fail &e) {
        S.cur = &S.lanes[0];
        S.branch_free = 0;
        pifu_set_error("pifu_filter_fine: %s", e.msg.c_str());
        return PIFU_ERR_STATE;
    }
    PIFU_HIP(hipGetLastError());
    // a coarse hourglass the previous call left running beside this one: both levels are complete for whatever comes next
    return pifu_enc_join(ctx, (hipStream_t)stream);
}
