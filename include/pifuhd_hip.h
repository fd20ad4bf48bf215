/* pifuhd_hip.h -- C ABI of the MI355X-native PIFuHD occupancy-query + meshing hot path.
 *
 * The reference (agnJason/RGB-D-PIFuHD) has no FFI/plugin boundary: its hot path is plain Python over
 * PyTorch ops.  This header is therefore the boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md); every entry point names the reference code it replaces (file:line relative to the
 * reference root).  Conventions:
 *   - every function returns 0 (PIFU_OK) or a negative PIFU_ERR_* code; pifu_last_error() gives the
 *     message of the last failure on the calling thread; nothing throws across the boundary;
 *   - the CALLER owns every buffer (device pointers unless a parameter says "host"); the library owns
 *     only the opaque context (packed weights, NHWC feature maps, workspaces);
 *   - all kernels are enqueued on the caller's stream (a hipStream_t passed as void*); only the calls
 *     documented as "synchronises" wait for the device; a context serves one call at a time (it is not
 *     re-entrant: workspaces, feature maps and tables are shared), but the calls may come on different streams:
 *     an entry point called with another stream than the previous call makes it wait for that call's work;
 *   - fp32 network arithmetic (exact-fp32 MFMA), float64 grid bookkeeping, as in the reference.
 */
#ifndef PIFUHD_HIP_H
#define PIFUHD_HIP_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported. */
#if defined(__GNUC__) || defined(__clang__)
#define PIFU_API __attribute__((visibility("default")))
#else
#define PIFU_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pifu_ctx pifu_ctx;
typedef void *pifu_stream;                 /* hipStream_t */

enum {
    PIFU_OK = 0,
    PIFU_ERR_INVALID = -1,                 /* bad argument / shape */
    PIFU_ERR_HIP = -2,                     /* a HIP runtime call failed */
    PIFU_ERR_STATE = -3,                   /* call order violated (e.g. query before features) */
    PIFU_ERR_NOMEM = -4,                   /* workspace too small and cannot grow */
    PIFU_ERR_MC_LEVEL = -5,                /* "Surface level must be within volume data range." */
    PIFU_ERR_MC_EMPTY = -6                 /* "No surface found at the given iso value." */
};

enum { PIFU_LEVEL_COARSE = 0, PIFU_LEVEL_FINE = 1 };

/* MLP normalisation modes (reference: MLP.py:37-41,65-69; train/eval quirk SURVEY C-2) */
enum {
    PIFU_NORM_NONE = 0,                    /* leaky_relu only                                  */
    PIFU_NORM_GROUP = 1,                   /* GroupNorm(32, C): statistics over (C/32 x chunk) */
    PIFU_NORM_BATCH_EVAL = 2,              /* BatchNorm1d with running statistics              */
    PIFU_NORM_BATCH_TRAIN = 3              /* BatchNorm1d with per-chunk batch statistics      */
};

/* pifu_query flags */
enum {
    PIFU_Q_COARSE_ONLY = 1,                /* PIFuNetwNML.query only (PIFuNetwNML.py:99-141)            */
    PIFU_Q_NO_FINE_MASK = 2,               /* calc_normal: fine prediction not masked (PIFuMRNet.py:223-230) */
    PIFU_Q_FULL_COARSE = 4,                /* run the coarse MLP to its last layer even if pred_coarse is NULL (the
                                              reference always does: netG.query inside PIFuMRNet.query) */
    PIFU_Q_NO_COLUMN_SHARING = 8,          /* lattice sources: evaluate every point independently instead of sharing the
                                              image-feature products of an (i, j) column (same result up to fp32
                                              summation order; 26 % more FLOPs) */
    PIFU_Q_NO_DEPTH_WALK = 16,             /* lattice sources: evaluate coarse layer 1 as a per-point GEMM instead of
                                              walking each column along z (layer-0 activations are piecewise linear
                                              in z inside a column; same result up to fp32 rounding) */
    PIFU_Q_NO_COARSE_MASK = 32             /* the coarse prediction not masked by the bounding box (PIFuNetwNML.calc_normal,
                                              PIFuNetwNML.py:207-209, differences the raw MLP output) */
};

/* pifu_filter_coarse flags */
enum {
    PIFU_ENC_ALL_STACKS = 1,               /* feat_nchw receives every stack's output (the module in train mode keeps the
                                              whole list, PIFuNetwNML.py:94-97) instead of the last one */
    PIFU_ENC_NORMALS_ONLY = 4,             /* run the enabled normal nets (use_front / use_back) and stop: no hourglass, the
                                              coarse feature map stays as it is */
    PIFU_ENC_NORMALS_GIVEN = 8,            /* nmlF / nmlB are INPUTS (device NCHW [1][3][H][W], e.g. computed by other ranks):
                                              the normal nets are skipped, the hourglass runs on cat[img, nmlF, nmlB].  The
                                              two halves of PIFuNetwNML.filter for callers that deal them to different GPUs */
    PIFU_ENC_NO_WINOGRAD = 16,             /* the 3 x 3 convolutions of the normal nets' ResnetBlock trunks as direct convolutions
                                              instead of Winograd F(2x2, 3x3) (the default where the trunk's map allows it:
                                              32 x 32, i.e. 512 x 512 images; same result up to fp32 rounding, ~1e-6 relative) */
    PIFU_ENC_DEFER_JOIN = 2                /* return while the coarse hourglass is still running on a stream of the library:
                                              pifu_filter_fine - which reads the normal maps, not the coarse features -
                                              then runs beside it.  The caller's stream is made to wait for the hourglass by
                                              the next call that needs its result (pifu_filter_fine at its end, every query /
                                              grid entry point, pifu_set_features, the next pifu_filter_coarse); until then
                                              feat_nchw must not be read by anyone else */
};

/* Version of the contracts in this header; pifu_abi_version() returns the value the library was built with, and a binding
 * must refuse a library whose value differs (the Python binding does, _lib.load()).
 *   1: rounds 1-2.
 *   2: round 3 - pifu_obj_finish in one-call mode creates / truncates `path` (abandon = nverts < 0); open_edge of
 *      pifu_face_components is a bit mask; the `all_stacks` argument of pifu_filter_coarse became the PIFU_ENC_* flags word.
 *   3: round 4 - explicit point lists (pifu_query, pifu_calc_normal*) of PIFU_PIXEL_TABLE_MIN_POINTS points or more take the
 *      coarse level's image-feature products from per-pixel tables (PIFU_Q_NO_COLUMN_SHARING keeps the per-point GEMMs);
 *      pifu_enc_join semantics: every stream that asks before the deferred hourglass has finished waits for it;
 *      pifu_set_pixel_tables, pifu_set_workspace_limit added.
 *   4: round 5 - the automatic pixel-table threshold follows the coarse map (calls of at least max(16 384, H * W) points: the
 *      same as before for the 128^2 map, later for larger maps); pifu_set_workspace_limit is a hard bound of the query arena (no allocation slack);
 *      pifu_arena_bytes, pifu_set_host_threads, pifu_enc_set_param_device, the PIFU_ENC_NORMALS_* flags and the pifu_comm_* /
 *      pifu_allgather_* calls added;
 *      a context is still used by ONE call at a time, but consecutive calls may be enqueued on different streams without the
 *      caller synchronising them: every entry point first makes its stream wait for the work of the previous call.
 *   5: round 6 - pifu_enc_set_param_device BORROWS the caller's tensor (no copy, no synchronisation; it must outlive the
 *      registration); pifu_calc_normal_coarse, pifu_reserve_auto, pifu_effective_chunk, pifu_drop_tables and PIFU_Q_NO_COARSE_MASK added; the
 *      column tables of a lattice evaluation are kept for later calls on the same inputs; calls of more than
 *      `chunk` points are regrouped when every MLP norm is pointwise (see pifu_effective_chunk); the library exports nothing
 *      but the PIFU_API entry points of this header (-fvisibility=hidden + export map). */
#define PIFU_ABI_VERSION 5
PIFU_API const char *pifu_last_error(void);
PIFU_API int pifu_abi_version(void);

/* ---- context ------------------------------------------------------------------------------- */
PIFU_API int pifu_ctx_create(int device, pifu_ctx **out);
PIFU_API int pifu_ctx_destroy(pifu_ctx *ctx);
/* Pre-size the query workspace for `rows` padded point rows (growth otherwise happens lazily, outside
 * the timed region after the first call).  Synchronises. */
PIFU_API int pifu_reserve(pifu_ctx *ctx, int64_t rows);
/* The same, sized from what the device has free NOW (hipMemGetInfo): rows = the largest multiple of 128 not above `max_rows`
 * whose batch workspace (worst case over the call kinds: lattice with depth walk, point lists with pixel tables) fits into
 * `share` (0 < share <= 1) of the free memory, at least one 128-row tile.  For several contexts on one device (ranks sharing a
 * GPU, or beside other work): with share = 0.5 / contexts none of them runs the card out of memory.  Both MLPs must be defined.
 * *rows_out (may be NULL) receives the choice. */
PIFU_API int pifu_reserve_auto(pifu_ctx *ctx, int64_t max_rows, double share, int64_t *rows_out);
PIFU_API int64_t pifu_workspace_bytes(pifu_ctx *ctx);
/* Bound the query workspace (the arena of one batch of launches, ~12.5 KB per padded point row) to `bytes`; 0 = no bound
 * (default).  A call whose batch would exceed the bound - or whose allocation the device refuses - runs in more, smaller
 * batches of whole chunks instead: the same numbers (a chunk never depends on the batch it runs in), a few per cent slower
 * below ~0.25 M rows.  One chunk is the floor: PIFU_ERR_NOMEM if even that does not fit.  An arena already larger than the
 * bound is released (synchronises).  For devices shared with other work; the marching-cubes, octree and table arenas
 * (R^3-sized) are not covered. */
PIFU_API int pifu_set_workspace_limit(pifu_ctx *ctx, int64_t bytes);
/* Bytes one of the context's device arenas holds at the moment (pifu_workspace_bytes is their sum). */
enum { PIFU_ARENA_QUERY = 0, PIFU_ARENA_MC = 1, PIFU_ARENA_OCTREE = 2, PIFU_ARENA_COLUMN = 3, PIFU_ARENA_PIXEL = 4, PIFU_ARENA_MISC = 5 };
PIFU_API int64_t pifu_arena_bytes(pifu_ctx *ctx, int which);
/* Host worker threads the library may start per call (the OBJ writer's formatting / writing threads): 0 = as many as the
 * machine has, at most 32 (default).  Process-wide.  For several processes sharing one host (one per GPU): cpu_count / ranks. */
PIFU_API int pifu_set_host_threads(int n);

/* ---- SurfaceClassifier definition: replaces MLP.__init__ / load_state_dict (MLP.py:12-41) ---- */
/* dims = filter_channels, res_layers = skip-concat layers, merge_layer as passed to MLP() (<=0 means
 * len(dims)/2, MLP.py:25), norm = PIFU_NORM_*.  Level FINE consumes [fine features | coarse phi]. */
PIFU_API int pifu_mlp_define(pifu_ctx *ctx, int level, const int32_t *dims, int ndims,
                    const int32_t *res_layers, int nres, int merge_layer, int norm);
/* weight [Cout][Cin] (Conv1d k=1 weight squeezed), bias [Cout]; norm_weight/norm_bias [Cout] or NULL
 * (last layer / PIFU_NORM_NONE); running_mean/var only for the batch modes.  Host pointers. */
PIFU_API int pifu_mlp_set_layer(pifu_ctx *ctx, int level, int layer, const float *weight, const float *bias,
                       const float *norm_weight, const float *norm_bias,
                       const float *running_mean, const float *running_var);
/* Arithmetic of the MLP GEMMs: 0 = exact fp32 MFMA (default, ~1e-6 from the reference); 1 = bf16 x 3 split precision
 * on the bf16 matrix cores (hi*hi + lo*hi + hi*lo, fp32 accumulation; ~1e-5 on the occupancy, inside the 1e-4 contract);
 * 2 = fp16 x 3: the same three products on the fp16 matrix cores with the weight planes scaled by 2^11 (operands good to
 * 2^-22: ~1e-6 on the occupancy; needs |w| < 32 and |activations| < 65504, the first is checked). */
PIFU_API int pifu_set_precision(pifu_ctx *ctx, int mode);
/* Pixel tables for explicit point lists (pifu_query, pifu_calc_normal*) and for lattice sources whose calibration rules out
 * column sharing (x or y depends on k).  index() is linear in the feature map, so the
 * products of the coarse MLP's weights with the sampled image features (layer 0 and the skip connections, MLP.py:58-64) are
 * W * sum_t w_t F[tap_t] = sum_t w_t (W * F[tap_t]): the table G = W * F is computed once per feature map (H * W pixel rows)
 * and a point blends four rows of it instead of running a K = 256 GEMM per point - 23 % fewer FLOPs on the calc_normal path,
 * same result up to fp32 summation order (~1e-6 on the occupancy).  mode 0: never; 1 (default): for calls of at least
 * PIFU_PIXEL_TABLE_MIN_POINTS points AND at least H * W points (the map's pixel count: below it building the table costs more
 * than the per-point GEMMs it replaces; once built it serves every later call on the same map) - the two coincide for the
 * reference's 128 x 128 map; 2: always.  PIFU_Q_NO_COLUMN_SHARING on a call keeps the per-point GEMMs whatever the
 * mode.  The table (H * W x 1536 floats: 100 MB for the 128^2 map) lives in the context and is rebuilt when the coarse
 * feature map, the coarse MLP or the precision mode changes. */
#define PIFU_PIXEL_TABLE_MIN_POINTS 16384
PIFU_API int pifu_set_pixel_tables(pifu_ctx *ctx, int mode);
/* Tables derived from the feature maps live in the context until an input changes: the pixel table above and - ABI 5 - the
 * per-column tables of the lattice evaluations (sampled features of every (i, j) column and their products with the skip
 * weights), which the levels of one octree evaluation and every later call on the same maps, weights, lattice and calibration
 * share.  Every setter of an input drops them; pifu_drop_tables does it by hand (a benchmark that must pay for them in every
 * step, as a new subject would). */
PIFU_API int pifu_drop_tables(pifu_ctx *ctx);
/* DepthNormalizer (DepthNormalizer.py:17-24): z_feat = z * z_mul / z_div with z_mul = loadSize // 2,
 * z_div = z_size of the COARSE net's options. */
PIFU_API int pifu_set_depth_normalizer(pifu_ctx *ctx, float z_mul, float z_div);

/* ---- feature maps: the output of filter()/filter_global()/filter_local() -------------------- */
/* feat: device fp32 NCHW [1][C][H][W] (PIFuNetwNML.py:94-97, PIFuMRNet.py:114-117).  Repacked to NHWC
 * inside the context so that one bilinear tap is one contiguous channel row. */
PIFU_API int pifu_set_features(pifu_ctx *ctx, int level, const float *feat_nchw, int C, int H, int W,
                      pifu_stream stream);

/* ---- image encoders: filter() / filter_global() / filter_local() -------------------------------------------
 * Parameters are registered by their reference state-dict names (HOST fp32 arrays): coarse net keys without the
 * `netG.` prefix (`image_filter.*`, `netF.*`, `netB.*`), fine net keys as `fine.image_filter.*`.
 * norm: 0 = BatchNorm2d with running statistics (eval), 1 = BatchNorm2d with batch statistics (the fine level
 * under recon(): SURVEY C-2), 2 = GroupNorm(32, C) (Filter.py:37-46).
 * pifu_filter_coarse = PIFuNetwNML.filter (PIFuNetwNML.py:73-97): img device NCHW [1][Cimg][H][W]; runs netF / netB
 * (networks.py:131-207) when enabled and writes their NCHW [1][3][H][W] outputs, then the HGFilter
 * (Filter.py:186-228) on cat[img, nmlF, nmlB]; the LAST stack's output becomes the coarse query feature map and is
 * optionally exported as NCHW [1][hg_dim][H/4][W/4] (ave_pool) to feat_nchw; with PIFU_ENC_ALL_STACKS in `flags`
 * feat_nchw receives every stack's output, [num_stack][hg_dim][H/4][W/4].  The two normal nets run side by side on two
 * streams, the hourglass on a third (joined before the call returns unless PIFU_ENC_DEFER_JOIN is given).
 * pifu_filter_fine = PIFuMRNet.filter_local (PIFuMRNet.py:83-118, rect = None): normals (NCHW, Hn x Wn) are
 * resized bilinearly (align_corners) to H x W, concatenated, HGFilter('no_down'); output [1][hg_dim][H/2][W/2]. */
PIFU_API int pifu_enc_set_param(pifu_ctx *ctx, const char *name, const float *host_data, const int64_t *shape, int rank);
/* The same registration for a rank-4 convolution weight that already lives on the device (a checkpoint loaded to the GPU).
 * ABI 5: the tensor is BORROWED, not copied - it must stay valid and unchanged until the parameter is registered again or the
 * context is destroyed (the 1.5 GB of encoder weights are then held once by the caller, not a second time by the library), and
 * the call neither copies nor synchronises; `stream` is the stream the tensor was produced on (the library's lanes fork from the
 * caller's stream, so a pass issued later on that stream is ordered behind it).  Either way
 * the [Cout][Cin][k][k] tensors are repacked into the kernels' K-run layout BY A KERNEL on first use (1.5 GB of weights:
 * milliseconds instead of the 1.7 s a host loop + pageable upload took); vectors (biases, norm parameters, running
 * statistics) go through pifu_enc_set_param, their folding happens on the host.  Re-registering parameters after a filter pass
 * synchronises the device ONCE per batch of registrations (the first call pays it). */
PIFU_API int pifu_enc_set_param_device(pifu_ctx *ctx, const char *name, const float *dev_data, const int64_t *shape, int rank,
                              pifu_stream stream);
PIFU_API int pifu_enc_clear(pifu_ctx *ctx);
PIFU_API int pifu_filter_coarse(pifu_ctx *ctx, const float *img_nchw, int Cimg, int H, int W, int num_stack, int hg_depth,
                       int hg_dim, int norm, int ave_pool, int use_front, int use_back, float *nmlF, float *nmlB,
                       float *feat_nchw, int flags, pifu_stream stream);
PIFU_API int pifu_filter_fine(pifu_ctx *ctx, const float *img_nchw, int Cimg, int H, int W, const float *nmlF,
                     const float *nmlB, int Hn, int Wn, int num_stack, int hg_depth, int hg_dim, int norm,
                     float *feat_nchw, pifu_stream stream);

/* ---- query: PIFuMRNet.query / PIFuNetwNML.query (PIFuMRNet.py:119-186, PIFuNetwNML.py:99-141) - */
/* points: device fp32 [3][n] (the reference's [1,3,N] tensor); calib: HOST fp32 4x4 row-major;
 * `chunk` = number of points the reference would pass per query() call (GroupNorm couples the points
 * of one call: SURVEY C-1); the n points are treated as consecutive calls of `chunk` points, remainder
 * last (mesh_util.py:98-114).  Outputs (device, may be NULL): pred_fine [n] = net.get_preds()[0][0],
 * pred_coarse [n] = netG.preds[0][0], phi [Cphi][n] = netG.phi[0] (only when n <= chunk). */
PIFU_API int pifu_query(pifu_ctx *ctx, const float *points, int64_t n, const float *calib, int64_t chunk,
               int flags, float *pred_fine, float *pred_coarse, float *phi, pifu_stream stream);

/* Pointwise MLP norms (PIFU_NORM_NONE - the published pifuhd.pt - or PIFU_NORM_BATCH_EVAL, at every level the call runs): a
 * point's result does not depend on the query() call it is evaluated in, so `chunk` only says how the reference would CUT the
 * work, and the library regroups it: lattice calls (pifu_eval_lattice, pifu_eval_grid) always, point lists (pifu_query,
 * pifu_calc_normal_chunked) when they span more than one chunk.  Groups of PIFU_POINTWISE_CHUNK rows - for a dense lattice
 * range the nearest multiple of lcm(128, res[2]), i.e. whole GEMM tiles and whole lattice columns (tiles only where that lcm
 * exceeds 4 groups, e.g. 1 021 nodes per column) - with no padding rows, no
 * statistics passes, and depth-walk segments that are complete columns.  Same numbers as the reference's cut up to fp32
 * summation order (the depth walk restarts at group boundaries).  The grouping of a lattice call is a function of
 * (chunk, res[2], the norms) alone and is cut relative to `first`; pifu_effective_chunk returns it (`chunk` itself when a norm
 * couples the chunk), so that a sharded caller deals out whole groups and N ranks reproduce one rank's field bit for bit
 * (rgb-d-pifuhd_amd/dist.py).  res2 = res[2] for a dense lattice range (index == NULL), 0 for an index list; of `flags` only
 * PIFU_Q_COARSE_ONLY matters. */
#define PIFU_POINTWISE_CHUNK 16384
PIFU_API int64_t pifu_effective_chunk(pifu_ctx *ctx, int64_t chunk, int res2, int flags);

/* ---- lattice evaluation: create_grid + calib^-1 + eval_func (mesh_util.py:12-38,59-74) -------- */
/* Evaluates `count` points of the C-ordered lattice res[0] x res[1] x res[2] over [-1,1)^3
 * (point (i,j,k) = -1 + (i,j,k) * 2/res, float64, then calib_inv (HOST float64 4x4), then cast to
 * float32 exactly like mesh_util.py:61-70) and writes fp32 predictions.  The points are the flat
 * indices first, first+1, ... when `index` is NULL, else index[first .. first+count) (device int64).
 * Chunks are cut relative to `first`, so `first` must be a multiple of `chunk` for parity. */
PIFU_API int pifu_eval_lattice(pifu_ctx *ctx, const int32_t res[3], const double *calib_inv, const float *calib,
                      const int64_t *index, int64_t first, int64_t count, int64_t chunk, int flags,
                      float *pred, pifu_stream stream);

/* ---- octree bookkeeping: eval_grid_octree (mesh_util.py:124-187), order-free form */
/* state: sdf float64 [R^3] and notprocessed uint8 [R^3], both device, caller-owned. */
PIFU_API int pifu_octree_begin(pifu_ctx *ctx, int R, double *sdf, uint8_t *notprocessed, pifu_stream stream);
/* Ordered (C-order) compaction of lattice(reso) & notprocessed into index_out (device int64, capacity
 * cap); *count_host receives the number of test points.  Call pattern per level: once with index_out = NULL (counting
 * pass; synchronises the stream ONCE to return the count), then with a buffer of that capacity (fill pass: reuses the
 * block counts of the counting pass, launches the compaction only and does not synchronise).  A fill call that does not
 * follow the counting call of the same level recounts and synchronises. */
PIFU_API int pifu_octree_select(pifu_ctx *ctx, int R, int reso, const uint8_t *notprocessed,
                       int64_t *index_out, int64_t cap, int64_t *count_host, pifu_stream stream);
/* Scatter the evaluated values, clear notprocessed there, then (if reso > 1) the 8-corner min/max test
 * with `threshold` and the inclusive block fill, bit-exact with the reference's sequential loop. */
PIFU_API int pifu_octree_commit(pifu_ctx *ctx, int R, int reso, double threshold, const int64_t *index,
                       const float *values, int64_t count, double *sdf, uint8_t *notprocessed,
                       pifu_stream stream);
/* Convenience single-GPU driver of the three calls above + pifu_eval_lattice; dense when
 * init_resolution <= 0 (eval_grid, mesh_util.py:116-120).  *n_evaluated (host) = points evaluated. */
PIFU_API int pifu_eval_grid(pifu_ctx *ctx, int R, const double *calib_inv, const float *calib, int64_t chunk,
                   int init_resolution, double threshold, int flags, double *sdf, uint8_t *notprocessed,
                   int64_t *n_evaluated, pifu_stream stream);

/* ---- marching cubes: skimage.measure.marching_cubes_lewiner as called at mesh_util.py:84 ------- */
/* volume: device [n0][n1][n2], dtype 0 = float32, 1 = float64 (cast to float32 first, as skimage does).
 * pifu_mc_count classifies every cell (33-case Lewiner tables + ambiguity tests) and prefix-sums vertex
 * and triangle counts in the serial scan order; it synchronises and returns the totals, or
 * PIFU_ERR_MC_LEVEL / PIFU_ERR_MC_EMPTY for the two exceptions skimage raises.  pifu_mc_emit then writes
 * verts fp32 [nv][3] (array-axis order) and faces int32 [nf][3] ('descent' order), bit-exact. */
PIFU_API int pifu_mc_count(pifu_ctx *ctx, const void *volume, int dtype, int n0, int n1, int n2, double level,
                  int64_t *nverts, int64_t *nfaces, pifu_stream stream);
PIFU_API int pifu_mc_emit(pifu_ctx *ctx, float *verts, int32_t *faces, pifu_stream stream);

/* ---- the same marching cubes over axis-0 slabs of one volume (one slab per GPU; SURVEY 8e, BASELINE config 5) ----
 * skimage scans cells with axis 0 outermost (mesh_util.py:84), so the cells [z_first, z_first + n0 - 1) of a slab own one
 * contiguous range of the mesh's vertex and face numbers.  A slab is the n0 >= 2 node planes [z_first, z_first + n0) of the
 * volume (neighbouring slabs share one node plane).  Per slab, in this order:
 *   pifu_mc_slab_count  - vertices / faces this slab's cells create (a cell creates the vertex of an edge iff it is the first
 *                         cell of the WHOLE volume touching it) and the slab's value range; synchronises.  The caller sums
 *                         the counts (vert_base = vertices of all lower slabs), reduces the range over the slabs and raises
 *                         skimage's two errors itself (level outside the range; no vertex at all);
 *   pifu_mc_slab_verts  - verts fp32 [nverts][3] of this slab (coordinates of the whole volume) and, in halo_out (device
 *                         int32 [2][n1][n2], may be NULL for the top slab), the mesh-wide numbers of the vertices on the x- /
 *                         y-edges of the slab's top node plane;
 *   pifu_mc_slab_faces  - faces int32 [nfaces][3] in mesh-wide vertex numbers; halo_in = the halo_out of the slab below
 *                         (NULL iff z_first == 0).
 * Concatenating the slabs' arrays in slab order gives pifu_mc_emit's output bit for bit. */
PIFU_API int pifu_mc_slab_count(pifu_ctx *ctx, const void *slab, int dtype, int n0, int n1, int n2, int z_first, double level,
                       int64_t *nverts, int64_t *nfaces, float *vmin, float *vmax, pifu_stream stream);
PIFU_API int pifu_mc_slab_verts(pifu_ctx *ctx, float *verts, int64_t vert_base, int32_t *halo_out, pifu_stream stream);
PIFU_API int pifu_mc_slab_faces(pifu_ctx *ctx, int32_t *faces, const int32_t *halo_in, pifu_stream stream);

/* ---- the exchange step of the sharded evaluation (the reference is single-device) -------------------------------
 * One process per GPU evaluates a contiguous range of whole chunks of a grid / octree level's point list and ONE all-gather
 * of the fp32 predictions rebuilds the list on every rank (rgb-d-pifuhd_amd/dist.py: torch.distributed does it for the Python
 * binding; these calls are the same collective for bindings without it).  A thin wrapper over RCCL's ncclAllGather; RCCL is
 * looked up at run time (the copy already loaded into the process, else /opt/rocm's), PIFU_ERR_STATE if there is none.
 * pifu_comm_unique_id: rank 0 draws the id and hands the 128 bytes to the other ranks over any side channel;
 * pifu_comm_create: collective over all `world` ranks (one per device); pifu_allgather_f32: recv [world][count_per_rank]
 * receives rank r's send [count_per_rank] at row r, enqueued on `stream` (recv + rank * count_per_rank == send is allowed). */
typedef struct pifu_comm pifu_comm;
#define PIFU_COMM_ID_BYTES 128
PIFU_API int pifu_comm_unique_id(uint8_t id[PIFU_COMM_ID_BYTES]);
PIFU_API int pifu_comm_create(int device, int rank, int world, const uint8_t id[PIFU_COMM_ID_BYTES], pifu_comm **out);
PIFU_API int pifu_comm_destroy(pifu_comm *comm);
PIFU_API int pifu_allgather_f32(pifu_comm *comm, const float *send, float *recv, int64_t count_per_rank, pifu_stream stream);
PIFU_API int pifu_allgather_bytes(pifu_comm *comm, const void *send, void *recv, int64_t bytes_per_rank, pifu_stream stream);

/* ---- per-vertex normals: PIFuMRNet.calc_normal (PIFuMRNet.py:188-243) as used by gen_mesh -------- */
/* verts: device fp32 [3][n]; every call is ONE GroupNorm batch of 4n points (p, p+dx, p+dy, p+dz
 * interleaved per point); normals: device fp32 [3][n] unit normals; pred4 (device fp32 [4n], may be NULL) receives the
 * unmasked fine occupancies of those 4n points in that order - the tensor `pred` the reference differences
 * (PIFuMRNet.py:223-230) - so that parity can be checked on the values, not only on the direction of their differences. */
PIFU_API int pifu_calc_normal(pifu_ctx *ctx, const float *verts, int64_t n, const float *calib, float delta,
                     float *normals, float *pred4, pifu_stream stream);

/* PIFuNetwNML.calc_normal (PIFuNetwNML.py:181-218): the same construction on the COARSE net alone - the 4n interleaved points
 * go through index / DepthNormalizer / the coarse MLP (sigmoid output, NOT masked by the bounding box), normals = -normalize of
 * the three forward differences (eps 1e-8).  The reference's default delta is 0.1.  pred4 as above (may be NULL). */
PIFU_API int pifu_calc_normal_coarse(pifu_ctx *ctx, const float *verts, int64_t n, const float *calib, float delta,
                                     float *normals, float *pred4, pifu_stream stream);

/* The colour loop of gen_mesh (reconstruction.py:58-70) in one call: identical to pifu_calc_normal on consecutive
 * slices of `verts_per_call` vertices (the last one shorter) - every slice is its own GroupNorm batch - but several
 * slices share one kernel launch. */
PIFU_API int pifu_calc_normal_chunked(pifu_ctx *ctx, const float *verts, int64_t n, int64_t verts_per_call, const float *calib,
                             float delta, float *normals, float *pred4, pifu_stream stream);

/* ---- vertex back-transform of reconstruction() (mesh_util.py:87-89): verts = (trans_mat[:3,:3] @ verts.T + trans_mat[:3,3:4]).T
 * in float64 on the float32 marching-cubes vertices.  verts: device fp32 [n][3]; mat: HOST double [12] = rows of the 3x4
 * trans_mat; out: device double [n][3].  Per component: the three products accumulated by FMA in k order starting from the
 * first product, the translation added last - the order an FMA-capable BLAS dgemm (what numpy's matmul calls) accumulates
 * K = 3 in; a fixed function of its inputs, compared with numpy on whole arrays by the tests. */
PIFU_API int pifu_transform_verts(pifu_ctx *ctx, const float *verts, int64_t n, const double *mat, double *out, pifu_stream stream);

/* ---- index(feat, uv) as a standalone op (BasePIFuNet.py:11-23; the vertex-colour lookup of gen_mesh_imgColor,
 * reconstruction.py:112-115): bilinear grid_sample, align_corners=True, zero padding.
 * feat: device fp32 NCHW [C][H][W] (batch 1); uv: device fp32 [2][n]; out: device fp32 [C][n]. */
PIFU_API int pifu_index(pifu_ctx *ctx, const float *feat_nchw, int C, int H, int W, const float *uv, int64_t n,
               float *out, pifu_stream stream);

/* ---- image half of readDataset.get_item (readData.py:15-29 addrect, :83-86 cv2.resize, :96-104 ToTensor+Normalize):
 * crop rect = (x, y, w, h) of a BGR uint8 image with a zero border where it leaves the image, resize to size x size
 * with cv2.resize's INTER_LINEAR 8-bit fixed-point arithmetic (INTER_AREA 2x2 mean when the crop is exactly
 * 2*size), BGR -> RGB, v/255, (t-0.5)/0.5.  bgr: device u8 [H][W][3]; out: device fp32 [3][size][size]. */
PIFU_API int pifu_preprocess_image(pifu_ctx *ctx, const uint8_t *bgr, int H, int W, const int32_t rect[4], int size,
                          float *out, pifu_stream stream);

/* ---- mesh clean-up of gen_mesh_imgColor_plus (reconstruction.py:127-229,325-344) ---------------------------------------
 * pifu_face_components: connected components of a triangle list over shared edges - what `meshcleaning` obtains from
 * trimesh's `mesh.split()` (reconstruction.py:331-332; trimesh is a third-party dependency absent from this image, its
 * split semantics are restated: faces are adjacent when they share an edge that exactly two faces use).  faces: device
 * int32 [nfaces][3]; label (device int32 [nfaces]) receives the SMALLEST face index of each face's component; open_edge
 * (device uint8 [nfaces]): bit 0 = the face has an edge no other face uses (its component cannot be watertight), bit 1 =
 * it has an edge that three or more faces use (no adjacency through it; trimesh judges watertightness per component, so
 * such an edge still counts as closed for a component holding exactly two of its faces - the caller, who has the labels,
 * settles these few faces).  A component none of whose faces is flagged is watertight.
 * pifu_esti_color: `esti_color` (reconstruction.py:180-229), exact: xyz = projected vertices, device fp32 [3][n]; color:
 * device uint16 [n][4] (RGBA), updated in place - every vertex with z < 0 receives the mean (uint16 sums, float64 division,
 * truncation) of the colours of its <= 10 nearest silhouette vertices (0 <= z < 0.001) on the left (x_s < x: ordered by
 * |y_s - y| ascending, x_s descending) and <= 10 on the right (x_s >= x: |y_s - y| ascending, x_s ascending), ties in
 * vertex order.  Fails with PIFU_ERR_STATE where the reference raises (no silhouette vertex at all / none on either side).
 * Synchronises the stream. */
PIFU_API int pifu_face_components(pifu_ctx *ctx, const int32_t *faces, int64_t nfaces, int32_t *label, uint8_t *open_edge,
                         pifu_stream stream);
PIFU_API int pifu_esti_color(pifu_ctx *ctx, const float *xyz, int64_t n, uint16_t *color, pifu_stream stream);

/* ---- supervised forward: CustomBCELoss.forward (net_util.py:45-66) as used by PIFuMRNet.get_error (PIFuMRNet.py:253-274).
 * Forward value only (the HIP path has no autograd): mean over [batch][n] of
 *   -w * (g * t * log(p) + (1 - g) * (1 - t) * log(1 - p)),  p = clamp(pred, 1e-5, 1 - 1e-5),  t = gt (brock: 3 gt - 1)
 * pred, gt: device fp32 [batch][n]; gamma: device fp32 [batch], or NULL to use fixed_gamma (the loss object's own gamma);
 * w: NULL (no weighting), device fp32 [batch] (w_per_element = 0) or [batch][n] (w_per_element = 1); out: device fp32 [1].
 * Deterministic (fixed-order double accumulation). */
PIFU_API int pifu_bce_loss(pifu_ctx *ctx, const float *pred, const float *gt, int batch, int64_t n, const float *gamma,
                  float fixed_gamma, const float *w, int w_per_element, int brock, float *out, pifu_stream stream);

/* ---- measurement hook (bench.py / profiles): HIP-event timing, on the launch stream, of the kernels bench.py prices
 * against a roofline.  `which`: PIFU_PROF_GEMM = every MLP GEMM launch (k_mlp_gemm_tri; work = ALGORITHMIC FLOPs,
 * 2 * valid points * Cout * Cin), PIFU_PROF_SAMPLER = every per-point k_sample launch (work = algorithmic bytes: 4 taps x
 * channels gathered + one MLP input row written per valid point), PIFU_PROF_MC = the marching-cubes kernels of
 * pifu_mc_count + pifu_mc_emit (two timed spans per mesh; work = one pass over the volume, n0 * n1 * n2 * sizeof(value), plus the
 * 12-byte vertices and faces written), PIFU_PROF_ENC = the convolutions of pifu_filter_coarse / pifu_filter_fine (work =
 * ALGORITHMIC FLOPs, 2 * output pixels * Cout * Cin * taps; they run on up to three concurrent streams of the library, so they are
 * counted, not timed: ms = 0 - time the pass as a whole), PIFU_PROF_MC_NODES = the node pass of marching cubes alone
 * (k_mc_nodebricks: the one kernel that streams the whole volume; work = n0 * n1 * n2 * sizeof(value)), PIFU_PROF_WALK = the
 * depth walk of coarse layer 1 on lattice sources (k_walk; work = the bytes it must move: rows x (N1 + 1) floats),
 * PIFU_PROF_ENC_CANONICAL = the same launches as PIFU_PROF_ENC priced as DIRECT convolutions (PIFU_PROF_ENC counts what is
 * executed: the 3 x 3 convolutions of the normal nets' ResnetBlock trunk run as Winograd F(2x2, 3x3), 2.25 x fewer multiplies).
 * pifu_profile_read synchronises the device and returns, since the last reset: summed kernel time (ms), launches, work. */
enum { PIFU_PROF_GEMM = 0, PIFU_PROF_SAMPLER = 1, PIFU_PROF_MC = 2, PIFU_PROF_ENC = 3, PIFU_PROF_MC_NODES = 4, PIFU_PROF_WALK = 5,
       PIFU_PROF_ENC_CANONICAL = 6 };
PIFU_API int pifu_profile_enable(pifu_ctx *ctx, int on);
PIFU_API int pifu_profile_read(pifu_ctx *ctx, int which, double *ms, int64_t *launches, double *work, int reset);

/* pifu_clock_probe synchronises the stream and returns, per XCD (8 on MI355X), the shader-clock counter (s_memtime)
 * and the constant 100 MHz real-time counter (s_memrealtime) read by one wave on that XCD: out[2 * xcd] = shader ticks,
 * out[2 * xcd + 1] = real-time ticks (0, 0 for an XCD no probe wave landed on).  Two probes around a timed region give
 * the AVERAGE shader clock the region actually ran at (bench.py reports it beside the nominal-clock roofline). */
PIFU_API int pifu_clock_probe(pifu_ctx *ctx, uint64_t out[16], pifu_stream stream);

/* ---- OBJ writer: save_obj_mesh_with_color (mesh_util.py:189-198); host buffers ------------------ */
PIFU_API int pifu_write_obj(const char *path, const double *verts, const double *colors, int64_t nverts,
                   const int32_t *faces, int64_t nfaces);
/* The same file in two calls, for callers that know the triangles before the vertex colours (gen_mesh: marching cubes, then
 * the colour loop, reconstruction.py:56-72): pifu_obj_begin starts formatting the face lines on worker threads, in memory
 * (faces: HOST int32, triangle i at faces[i * row_stride + k * col_stride], k = 0..2 - a reversed view such as
 * faces[:, ::-1] needs no copy; it must stay alive until pifu_obj_finish); pifu_obj_finish creates the file, formats the
 * vertex lines, writes vertices then faces, closes the file and frees the writer (also when it returns an error).
 * nverts < 0 abandons the writer: the workers are joined, nothing is created or truncated - a run that fails between the
 * two calls leaves `path` as it was, like the reference, which only writes once the colours exist. */
PIFU_API int pifu_obj_begin(const char *path, const int32_t *faces, int64_t nfaces, int64_t row_stride, int64_t col_stride, void **writer);
/* Optional middle steps, for callers that know every vertex POSITION long before the last COLOUR (gen_mesh: marching cubes,
 * then the colour loop, reconstruction.py:56-72).  pifu_obj_layout declares the complete arrays - verts HOST double [nverts][3],
 * colors HOST float32 [nverts][3] (the float32 values the reference copies into its float64 colour array: the same text), both
 * alive until pifu_obj_finish, the colours still to be filled in.  A colour in [0, 10) prints as 6 characters, so the positions
 * alone fix the file offset of every vertex block and of the face section: the face blocks are written in place at once, and
 * pifu_obj_ready(writer, upto) - "the colours of vertices [0, upto) are valid" - releases the vertex blocks below `upto` to be
 * formatted and written while the caller computes the rest.  The text goes to `<path>.part`; pifu_obj_finish(writer, NULL,
 * NULL, 0) declares every colour valid, waits, and renames it to `path` (a line whose length was not the predicted one - a NaN
 * or negative colour - makes it rewrite the file in one pass first); nverts < 0 removes the partial file. */
PIFU_API int pifu_obj_layout(void *writer, const double *verts, const float *colors, int64_t nverts);
PIFU_API int pifu_obj_ready(void *writer, int64_t upto);
PIFU_API int pifu_obj_finish(void *writer, const double *verts, const double *colors, int64_t nverts);

#ifdef __cplusplus
}
#endif
#endif /* PIFUHD_HIP_H */
