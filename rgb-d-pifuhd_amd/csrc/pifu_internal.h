// Internal declarations shared by the translation units of libpifuhd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/pifuhd_hip.h"

void pifu_set_error(const char *fmt, ...);

#define PIFU_HIP(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            pifu_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                           __LINE__);                                                        \
            return PIFU_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

#define PIFU_REQUIRE(cond, code, ...)   \
    do {                                \
        if (!(cond)) {                  \
            pifu_set_error(__VA_ARGS__);\
            return (code);              \
        }                               \
    } while (0)

// GEMM tile geometry of the MLP layers (see DESIGN.md "mlp_gemm")
constexpr int PIFU_BM = 128;   // points per block tile (also the padding quantum of a chunk)
constexpr int PIFU_BN = 128;   // output channels per block tile
constexpr int PIFU_BK = 32;    // K step; every K segment is padded to a multiple of it
constexpr int PIFU_MAX_LAYERS = 8;
constexpr int PIFU_MAX_WIDTH = 8192;  // widest MLP layer / input supported
constexpr int PIFU_PROF_SLOTS = 7;    // PIFU_PROF_GEMM, _SAMPLER, _MC, _ENC, _MC_NODES, _WALK, _ENC_CANONICAL

static inline int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct MlpLayer {
    int cin_prev = 0;      // channels coming from the previous layer (0 for layer 0)
    int cin_feat = 0;      // channels of the skip-concatenated MLP input (0 if not a res layer / layer 0: all)
    int cout = 0, cout_pad = 0;
    int kpad = 0;          // padded K = sum of padded segment widths
    bool has_norm = false; // followed by norm + leaky (all but the last layer)
    float *w = nullptr;    // device [cout_pad][kpad]
    float *bias = nullptr; // device [cout_pad]
    float *gamma = nullptr, *beta = nullptr;       // device [cout_pad] (affine of the norm)
    float *fixed_scale = nullptr, *fixed_shift = nullptr;  // device [cout_pad] for NONE / BATCH_EVAL
    float host_bias0 = 0.f; // bias of a 1-channel (last) layer, kept on the host
    // column sharing (lattice sources): weights without the skip input's per-column part
    std::vector<float> host_w;   // [cout][cin] as given
    float *w_col = nullptr;      // device [cout_pad | 1][kpad_col]
    float *w_colT = nullptr;     // device [kpad_col][cout_pad]: transposed copy for the depth walk (coarse layer 1)
    int kpad_col = 0, cin_col = 0;
    float *wz = nullptr;         // device [cout_pad]: weight of the depth feature (coarse level)
    float host_wz0 = 0.f;
    int col_off = -1;            // first column of this layer's block in the column table (-1: none)
    bool set = false;
};

struct Mlp {
    bool defined = false;
    int ndims = 0;
    int dims[PIFU_MAX_LAYERS + 1] = {0};
    bool res[PIFU_MAX_LAYERS] = {false};
    int merge_layer = 0;
    int norm = PIFU_NORM_GROUP;
    int nlayers = 0;
    MlpLayer layer[PIFU_MAX_LAYERS];
    // column sharing: stacked per-column weights [col_rows][col_k] (image-feature columns of layer 0 / res layers)
    float *colW = nullptr;
    int col_rows = 0, col_k = 0;
    bool col_built = false;
};

struct FeatMap {
    float *nhwc = nullptr;  // device [H][W][Cpad]
    int C = 0, Cpad = 0, H = 0, W = 0;
    size_t bytes = 0;
};

struct McState {
    bool valid = false;
    int n0 = 0, n1 = 0, n2 = 0;
    double level = 0;
    int dtype = 0;
    const void *volume = nullptr;
    int64_t nverts = 0, nfaces = 0;
    int64_t nsurf = 0, list_cap = 0;      // surface-cell records written by the counting pass, and the list's capacity
    // slab mode (pifu_mc_slab_*): first node plane of the slab in the tall volume, numbering offset of its vertices
    bool slab = false, verts_done = false;
    int zoff = 0;
    int64_t vert_base = 0;
};

struct pifu_ctx {
    int device = 0;
    Mlp mlp[2];
    FeatMap feat[2];
    float z_mul = 512.f, z_div = 200.f;
    // query workspace (grown lazily; everything lives in one arena)
    DevBuf arena;
    int64_t ws_rows = 0;
    int64_t ws_limit = 0;                 // pifu_set_workspace_limit: bytes the query workspace may take (0 = no limit)
    // marching-cubes workspace
    DevBuf mc_arena;
    McState mc;
    // octree scratch
    DevBuf oct_arena;
    // pifu_octree_select: block counts / offsets of the running call (own buffer), and the total the last COUNT call brought
    // to the host, so that the fill call of the same level needs no second host synchronisation
    DevBuf sel_counts;
    struct SelState { bool valid = false; int R = 0, reso = 0; const uint8_t *notproc = nullptr; int64_t n = 0; } sel;
    // column tables of the lattice evaluation (column sharing)
    DevBuf col_arena;
    // the tables col_arena holds at the moment (query.hip build_column_tables): the levels of ONE octree evaluation - and every
    // later call on the same maps, weights, lattice and calibration - share them instead of rebuilding them per call.  Dropped
    // by every setter of their inputs (feature maps, MLP weights, precision).
    struct ColCache {
        bool valid = false;
        bool fine = false;
        int res[3] = {0, 0, 0}, ldc = 0, ldf = 0, precision = 0;
        float calib[12] = {0};
        double cinv[12] = {0};
        int64_t col_base = 0, ncols = 0;
    } col_cache;
    int col_enable = 1;
    int walk_enable = 1;                  // depth walk of coarse layer 1 on lattice sources
    // pixel tables of explicit point lists (query.hip "Pixel tables"): G = W_feat * F over the pixels of the coarse map
    int pix_mode = 1;                     // pifu_set_pixel_tables: 0 off, 1 from PIFU_PIXEL_TABLE_MIN_POINTS points on, 2 always
    DevBuf pix_arena;
    struct PixState { bool valid = false; const float *feat = nullptr; int H = 0, W = 0, C = 0, precision = 0, ldg = 0; } pix;
    // Cross-stream ordering of the context's shared state (workspace arenas, feature maps, column / pixel tables, packed
    // weights): every entry point that enqueues work records `use_event` on its stream when it returns, and an entry point
    // called with ANOTHER stream makes that stream wait for it first (PIFU_USE).  A context is still used by one call at a
    // time, but the calls may come on different streams without the caller synchronising them.
    hipEvent_t use_event = nullptr;
    hipStream_t use_stream = nullptr;
    bool use_recorded = false;
    // small persistent device scratch (scan totals etc.)
    DevBuf misc;
    void *pinned = nullptr;   // 256 B of pinned host memory for count read-backs
    float *ident_scale = nullptr, *ident_shift = nullptr;   // device ones / zeros [PIFU_MAX_WIDTH]
    // optional per-launch timing of the dominant kernel (k_mlp_gemm) with HIP events on the launch stream
    int num_cus = 256;
#ifdef PIFU_LAB     // tools/lab builds only
    void *debug_buf = nullptr;            // in-kernel stamp buffer of the diagnostic kernels
    void *tile_ctr = nullptr;             // k_mlp_gemm_dyn work-queue counters: ring of [8] int slots
    int tile_ctr_next = 0;
    int gemm_variant = 5;                 // kernel / schedule selection (env PIFU_GEMM_VARIANT)
    int walk_variant = 0;                 // 1 = round-1 depth-walk kernel (env PIFU_WALK_VARIANT)
#endif
    void *clock_buf = nullptr;            // pifu_clock_probe: [8 XCDs][shader ticks, real-time ticks]
    // arithmetic of the MLP GEMMs: 0 = exact fp32 MFMA (default), 1 = bf16 x 3 split precision on the bf16 matrix cores
    int precision = 0;
    struct Planes { void *hi; void *lo; size_t n; };
    std::map<const void *, Planes> bf16_planes;
    // optional HIP-event timing of the kernels bench.py prices against a roofline (pifu_profile_*), one slot per kernel
    bool prof_on = false;
    struct ProfSlot {
        std::vector<hipEvent_t> ev;       // pairs (start, stop)
        size_t used = 0;                  // events in use
        double work = 0.0;                // algorithmic work of the timed launches (FLOPs or bytes)
        double ms_acc = 0.0;              // time already folded in (events recycled)
        int64_t launches = 0;
    } prof[PIFU_PROF_SLOTS];
};

// records the start event of a timed launch in `slot` (no-op when profiling is off: *stop = nullptr)
int pifu_prof_begin(pifu_ctx *ctx, int slot, hipStream_t st, hipEvent_t *stop);
// records the stop event and books the launch's algorithmic work
int pifu_prof_end(pifu_ctx *ctx, int slot, hipEvent_t stop, double work, hipStream_t st);

// grow-only hipMalloc (synchronises when it grows); `exact`: no slack on top of `bytes` (arenas with a caller-set bound)
int pifu_ensure(DevBuf &b, size_t bytes, bool exact = false);
// PIFU_USE(ctx, stream) at the top of an entry point: `stream` waits for the previous call's work if that was enqueued on
// another stream; when the entry point returns (any path) the event is recorded on `stream`
int pifu_use_begin(pifu_ctx *ctx, hipStream_t st);
void pifu_use_end(pifu_ctx *ctx, hipStream_t st);
struct PifuUse {
    pifu_ctx *c; hipStream_t s;
    PifuUse(pifu_ctx *c_, hipStream_t s_) : c(c_), s(s_) {}
    ~PifuUse() { pifu_use_end(c, s); }
};
#define PIFU_USE(ctx, st)                                         \
    {                                                             \
        const int use_rc_ = pifu_use_begin((ctx), (hipStream_t)(st)); \
        if (use_rc_) return use_rc_;                              \
    }                                                             \
    PifuUse pifu_use_guard_((ctx), (hipStream_t)(st))
// encoder.hip: make `st` wait for a coarse hourglass that pifu_filter_coarse(PIFU_ENC_DEFER_JOIN) left running
int pifu_enc_join(pifu_ctx *ctx, hipStream_t st);

// implemented in query.hip
int pifu_query_impl(pifu_ctx *ctx, int src_mode, const float *points, int64_t n, const int32_t *res,
                    const double *calib_inv, const int64_t *index, int64_t first, const float *calib,
                    int64_t chunk, int flags, float *pred_fine, float *pred_coarse, float *phi,
                    float delta_for_normals, float **normal_pred_out, hipStream_t stream);
