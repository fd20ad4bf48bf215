// Context management, weight packing and the host-only entry points of libpifuhd_hip.so.
#include "pifu_internal.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>

#include <algorithm>
#include <math.h>
#include <memory>
#include <string>
#include <thread>
#include <vector>

static thread_local char g_err[1024] = "";

void pifu_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *pifu_last_error(void) { return g_err; }
extern "C" int pifu_abi_version(void) { return PIFU_ABI_VERSION; }

int pifu_ensure(DevBuf &b, size_t bytes, bool exact)
{
    if (bytes <= b.bytes) return PIFU_OK;
    if (b.p) {
        PIFU_HIP(hipDeviceSynchronize());   // growing: nothing may still be using the old arena
        PIFU_HIP(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = exact ? bytes : bytes + bytes / 8;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&b.p, bytes);
        if (e != hipSuccess) {
            pifu_set_error("workspace allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
            (void)hipGetLastError();          // reported through the return code: nothing is left behind for a later check
            b.p = nullptr;
            return PIFU_ERR_NOMEM;
        }
        b.bytes = bytes;
        return PIFU_OK;
    }
    b.bytes = want;
    return PIFU_OK;
}

extern "C" int pifu_ctx_create(int device, pifu_ctx **out)
{
    PIFU_REQUIRE(out, PIFU_ERR_INVALID, "pifu_ctx_create: null out pointer");
    int ndev = 0;
    PIFU_HIP(hipGetDeviceCount(&ndev));
    PIFU_REQUIRE(device >= 0 && device < ndev, PIFU_ERR_INVALID, "pifu_ctx_create: device %d of %d", device, ndev);
    PIFU_HIP(hipSetDevice(device));
    pifu_ctx *c = new pifu_ctx();
    c->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    }
#ifdef PIFU_LAB     // tools/lab builds only: kernel selection and diagnostics through the environment
    if (const char *v = getenv("PIFU_GEMM_VARIANT")) c->gemm_variant = atoi(v);
    if (const char *v = getenv("PIFU_WALK_VARIANT")) c->walk_variant = atoi(v);
    if (const char *v = getenv("PIFU_COLUMN_MODE")) c->col_enable = atoi(v);
    if (const char *v = getenv("PIFU_DEPTH_WALK")) c->walk_enable = atoi(v);
    if (const char *v = getenv("PIFU_PRECISION")) c->precision = atoi(v);
#endif
    if (hipHostMalloc(&c->pinned, 256, hipHostMallocDefault) != hipSuccess) {
        delete c;
        pifu_set_error("pinned host allocation failed");
        return PIFU_ERR_HIP;
    }
    {
        std::vector<float> ones(PIFU_MAX_WIDTH, 1.f);
        if (hipMalloc(&c->ident_scale, PIFU_MAX_WIDTH * sizeof(float)) != hipSuccess ||
            hipMalloc(&c->ident_shift, PIFU_MAX_WIDTH * sizeof(float)) != hipSuccess ||
            hipMemcpy(c->ident_scale, ones.data(), PIFU_MAX_WIDTH * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(c->ident_shift, 0, PIFU_MAX_WIDTH * sizeof(float)) != hipSuccess) {
            pifu_set_error("identity scale/shift allocation failed");
            pifu_ctx_destroy(c);
            return PIFU_ERR_HIP;
        }
    }
#ifdef PIFU_LAB
    if (c->gemm_variant == 9 || getenv("PIFU_DEBUG_STAMPS")) { (void)hipMalloc(&c->debug_buf, (64 * 40 * 4 + 16384 * 16) * 8); (void)hipMemset(c->debug_buf, 0, (64 * 40 * 4 + 16384 * 16) * 8); }
#endif
    *out = c;
    return PIFU_OK;
}

#ifdef PIFU_LAB
// tools/lab builds only (not part of the ABI): copy the in-kernel stamp buffer of the diagnostic kernels to the host
extern "C" int pifu_debug_read(pifu_ctx *ctx, void *host, size_t bytes)
{
    if (!ctx || !ctx->debug_buf) return PIFU_ERR_STATE;
    PIFU_HIP(hipDeviceSynchronize());
    PIFU_HIP(hipMemcpy(host, ctx->debug_buf, bytes, hipMemcpyDeviceToHost));
    return PIFU_OK;
}
#endif

static void free_layer(MlpLayer &l)
{
    float **ptrs[] = {&l.w, &l.bias, &l.gamma, &l.beta, &l.fixed_scale, &l.fixed_shift, &l.w_col, &l.w_colT, &l.wz};
    for (float **p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    l.set = false;
}

extern "C" int pifu_enc_clear(pifu_ctx *ctx);

extern "C" int pifu_ctx_destroy(pifu_ctx *ctx)
{
    if (!ctx) return PIFU_OK;
    (void)pifu_enc_clear(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int L = 0; L < 2; ++L) {
        for (int i = 0; i < PIFU_MAX_LAYERS; ++i) free_layer(ctx->mlp[L].layer[i]);
        if (ctx->feat[L].nhwc) (void)hipFree(ctx->feat[L].nhwc);
    }
    DevBuf *bufs[] = {&ctx->arena, &ctx->mc_arena, &ctx->oct_arena, &ctx->misc, &ctx->col_arena, &ctx->pix_arena, &ctx->sel_counts};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int L = 0; L < 2; ++L)
        if (ctx->mlp[L].colW) (void)hipFree(ctx->mlp[L].colW);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (auto &kv : ctx->bf16_planes) (void)hipFree(kv.second.hi);
    for (auto &slot : ctx->prof)
        for (hipEvent_t e : slot.ev) (void)hipEventDestroy(e);
#ifdef PIFU_LAB
    if (ctx->tile_ctr) (void)hipFree(ctx->tile_ctr);
    if (ctx->debug_buf) (void)hipFree(ctx->debug_buf);
#endif
    if (ctx->use_event) (void)hipEventDestroy(ctx->use_event);
    if (ctx->clock_buf) (void)hipFree(ctx->clock_buf);
    if (ctx->ident_scale) (void)hipFree(ctx->ident_scale);
    if (ctx->ident_shift) (void)hipFree(ctx->ident_shift);
    delete ctx;
    return PIFU_OK;
}

int pifu_use_begin(pifu_ctx *ctx, hipStream_t st)
{
    if (!ctx) return PIFU_OK;                 // the entry point's own argument check reports it
    PIFU_HIP(hipSetDevice(ctx->device));
    if (!ctx->use_event) PIFU_HIP(hipEventCreateWithFlags(&ctx->use_event, hipEventDisableTiming));
    if (ctx->use_recorded && ctx->use_stream != st) PIFU_HIP(hipStreamWaitEvent(st, ctx->use_event, 0));
    return PIFU_OK;
}

void pifu_use_end(pifu_ctx *ctx, hipStream_t st)
{
    if (!ctx || !ctx->use_event) return;
    if (hipEventRecord(ctx->use_event, st) == hipSuccess) { ctx->use_stream = st; ctx->use_recorded = true; }
    else (void)hipGetLastError();
}

int pifu_prof_begin(pifu_ctx *ctx, int slot, hipStream_t st, hipEvent_t *stop)
{
    *stop = nullptr;
    if (!ctx->prof_on) return PIFU_OK;
    pifu_ctx::ProfSlot &p = ctx->prof[slot];
    if (p.used + 2 > p.ev.size()) {
        hipEvent_t a0, a1;
        // device-scope release only: a system-scope fence per event costs ~5 us of idle GPU around every timed launch
        PIFU_HIP(hipEventCreateWithFlags(&a0, hipEventDisableSystemFence));
        PIFU_HIP(hipEventCreateWithFlags(&a1, hipEventDisableSystemFence));
        p.ev.push_back(a0);
        p.ev.push_back(a1);
    }
    hipEvent_t e0 = p.ev[p.used];
    *stop = p.ev[p.used + 1];
    p.used += 2;
    PIFU_HIP(hipEventRecord(e0, st));
    return PIFU_OK;
}

int pifu_prof_end(pifu_ctx *ctx, int slot, hipEvent_t stop, double work, hipStream_t st)
{
    if (!stop) return PIFU_OK;
    PIFU_HIP(hipEventRecord(stop, st));
    ctx->prof[slot].work += work;
    ctx->prof[slot].launches += 1;
    return PIFU_OK;
}

extern "C" int pifu_profile_enable(pifu_ctx *ctx, int on)
{
    PIFU_REQUIRE(ctx, PIFU_ERR_INVALID, "pifu_profile_enable: null context");
    ctx->prof_on = on != 0;
    return PIFU_OK;
}

extern "C" int pifu_profile_read(pifu_ctx *ctx, int which, double *ms_out, int64_t *launches, double *work, int reset)
{
    PIFU_REQUIRE(ctx && which >= 0 && which < PIFU_PROF_SLOTS, PIFU_ERR_INVALID, "pifu_profile_read: bad argument");
    PIFU_HIP(hipSetDevice(ctx->device));
    PIFU_HIP(hipDeviceSynchronize());
    pifu_ctx::ProfSlot &p = ctx->prof[which];
    double ms = p.ms_acc;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        float t = 0.f;
        PIFU_HIP(hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]));
        ms += t;
    }
    p.ms_acc = ms;
    p.used = 0;                              // events can be recycled: their time is folded into the accumulator
    if (ms_out) *ms_out = ms;
    if (launches) *launches = p.launches;
    if (work) *work = p.work;
    if (reset) { p.ms_acc = 0.0; p.work = 0.0; p.launches = 0; }
    return PIFU_OK;
}

extern "C" int pifu_reserve(pifu_ctx *ctx, int64_t rows)
{
    PIFU_REQUIRE(ctx && rows > 0, PIFU_ERR_INVALID, "pifu_reserve: bad argument");
    ctx->ws_rows = round_up(rows, PIFU_BM);
    return PIFU_OK;
}

// upper bound of the query workspace (the arena of a batch of launches); 0 = none.  A smaller bound means more, smaller
// batches of launches with the same results; an existing larger arena is released
extern "C" int pifu_set_workspace_limit(pifu_ctx *ctx, int64_t bytes)
{
    PIFU_REQUIRE(ctx && bytes >= 0, PIFU_ERR_INVALID, "pifu_set_workspace_limit: bad argument");
    ctx->ws_limit = bytes;
    if (bytes > 0 && ctx->arena.bytes > (size_t)bytes && ctx->arena.p) {
        PIFU_HIP(hipSetDevice(ctx->device));
        PIFU_HIP(hipDeviceSynchronize());
        PIFU_HIP(hipFree(ctx->arena.p));
        ctx->arena.p = nullptr;
        ctx->arena.bytes = 0;
    }
    return PIFU_OK;
}

extern "C" int64_t pifu_workspace_bytes(pifu_ctx *ctx)
{
    return ctx ? (int64_t)(ctx->arena.bytes + ctx->mc_arena.bytes + ctx->oct_arena.bytes + ctx->misc.bytes + ctx->col_arena.bytes +
                           ctx->pix_arena.bytes) : 0;
}

extern "C" int64_t pifu_arena_bytes(pifu_ctx *ctx, int which)
{
    if (!ctx) return 0;
    const DevBuf *bufs[] = {&ctx->arena, &ctx->mc_arena, &ctx->oct_arena, &ctx->col_arena, &ctx->pix_arena, &ctx->misc};
    return which >= 0 && which < 6 ? (int64_t)bufs[which]->bytes : 0;
}

// 0 = exact fp32 MFMA (default); 1 = bf16 x 3 split precision (~1e-5 on the occupancy); 2 = fp16 x 3 (~1e-6)
extern "C" int pifu_set_precision(pifu_ctx *ctx, int mode)
{
    PIFU_REQUIRE(ctx && mode >= 0 && mode <= 2, PIFU_ERR_INVALID, "pifu_set_precision: mode must be 0 (fp32), 1 (bf16x3) or 2 (fp16x3)");
    if (mode != ctx->precision) {      // the cached weight planes belong to one mode
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        for (auto &kv : ctx->bf16_planes) (void)hipFree(kv.second.hi);
        ctx->bf16_planes.clear();
    }
    ctx->precision = mode;
    ctx->pix.valid = false;            // the pixel table is a product of the GEMM in the current arithmetic
    ctx->col_cache.valid = false;
    return PIFU_OK;
}

// pixel tables of explicit point lists: 0 never, 1 from PIFU_PIXEL_TABLE_MIN_POINTS points on (default), 2 always
extern "C" int pifu_set_pixel_tables(pifu_ctx *ctx, int mode)
{
    PIFU_REQUIRE(ctx && mode >= 0 && mode <= 2, PIFU_ERR_INVALID, "pifu_set_pixel_tables: mode must be 0 (off), 1 (auto) or 2 (always)");
    ctx->pix_mode = mode;
    return PIFU_OK;
}

// forget the column tables of the lattice evaluations and the pixel table of the point-list queries: the next call rebuilds
// them (what a new subject's feature maps do implicitly)
extern "C" int pifu_drop_tables(pifu_ctx *ctx)
{
    PIFU_REQUIRE(ctx, PIFU_ERR_INVALID, "pifu_drop_tables: null context");
    ctx->col_cache.valid = false;
    ctx->pix.valid = false;
    return PIFU_OK;
}

extern "C" int pifu_set_depth_normalizer(pifu_ctx *ctx, float z_mul, float z_div)
{
    PIFU_REQUIRE(ctx && z_div != 0.f, PIFU_ERR_INVALID, "pifu_set_depth_normalizer: bad argument");
    ctx->z_mul = z_mul;
    ctx->z_div = z_div;
    return PIFU_OK;
}

extern "C" int pifu_mlp_define(pifu_ctx *ctx, int level, const int32_t *dims, int ndims, const int32_t *res_layers,
                               int nres, int merge_layer, int norm)
{
    PIFU_REQUIRE(ctx && dims, PIFU_ERR_INVALID, "pifu_mlp_define: null argument");
    PIFU_REQUIRE(level == 0 || level == 1, PIFU_ERR_INVALID, "pifu_mlp_define: level must be 0 or 1");
    PIFU_REQUIRE(ndims >= 3 && ndims <= PIFU_MAX_LAYERS + 1, PIFU_ERR_INVALID, "pifu_mlp_define: 3 <= len(dims) <= %d", PIFU_MAX_LAYERS + 1);
    PIFU_REQUIRE(norm >= PIFU_NORM_NONE && norm <= PIFU_NORM_BATCH_TRAIN, PIFU_ERR_INVALID, "pifu_mlp_define: unknown norm %d", norm);
    PIFU_REQUIRE(dims[ndims - 1] == 1, PIFU_ERR_INVALID, "pifu_mlp_define: the last layer must have one output channel (occupancy)");
    PIFU_REQUIRE(level == 0 || ctx->mlp[0].defined, PIFU_ERR_STATE, "pifu_mlp_define: define the coarse MLP before the fine one");
    PIFU_HIP(hipSetDevice(ctx->device));
    Mlp &m = ctx->mlp[level];
    for (int i = 0; i < PIFU_MAX_LAYERS; ++i) free_layer(m.layer[i]);
    if (m.colW) { (void)hipFree(m.colW); m.colW = nullptr; }
    if (level == 0) ctx->pix.valid = false;
    ctx->col_cache.valid = false;
    m = Mlp();
    m.ndims = ndims;
    m.nlayers = ndims - 1;
    for (int i = 0; i < ndims; ++i) {
        PIFU_REQUIRE(dims[i] > 0 && dims[i] <= PIFU_MAX_WIDTH - PIFU_BN, PIFU_ERR_INVALID, "pifu_mlp_define: dims[%d] = %d", i, dims[i]);
        m.dims[i] = dims[i];
    }
    for (int i = 0; i < nres; ++i) {
        // a skip-concat at layer 0 would concatenate the input with itself; the reference never does it
        PIFU_REQUIRE(res_layers[i] >= 1 && res_layers[i] < m.nlayers, PIFU_ERR_INVALID, "pifu_mlp_define: res layer %d out of range", res_layers[i]);
        m.res[res_layers[i]] = true;
    }
    m.merge_layer = merge_layer > 0 ? merge_layer : ndims / 2;          // MLP.py:25
    PIFU_REQUIRE(m.merge_layer < m.nlayers - 1, PIFU_ERR_INVALID, "pifu_mlp_define: merge_layer %d must be a hidden layer", m.merge_layer);
    m.norm = norm;
    if (norm == PIFU_NORM_GROUP)
        for (int i = 1; i < ndims - 1; ++i)
            PIFU_REQUIRE(dims[i] % 32 == 0, PIFU_ERR_INVALID, "pifu_mlp_define: GroupNorm(32, %d) is not defined", dims[i]);
    // input feature split: coarse = [image features | z]; fine = [fine features | phi]
    int feat_pad;
    if (level == 0) {
        feat_pad = (int)round_up(dims[0], PIFU_BK);
    } else {
        const Mlp &mc = ctx->mlp[0];
        const int cphi = mc.dims[mc.merge_layer + 1];
        PIFU_REQUIRE(dims[0] > cphi, PIFU_ERR_INVALID, "pifu_mlp_define: fine input %d must exceed the coarse phi width %d", dims[0], cphi);
        feat_pad = (int)round_up(dims[0] - cphi, PIFU_BK) + mc.layer[mc.merge_layer].cout_pad;
    }
    for (int i = 0; i < m.nlayers; ++i) {
        MlpLayer &l = m.layer[i];
        l.cin_prev = i == 0 ? 0 : dims[i];
        l.cin_feat = (i == 0 || m.res[i]) ? dims[0] : 0;
        l.cout = dims[i + 1];
        l.has_norm = i != m.nlayers - 1;
        l.cout_pad = l.has_norm ? (int)round_up(l.cout, PIFU_BN) : l.cout;
        const int prev_pad = i == 0 ? 0 : m.layer[i - 1].cout_pad;
        l.kpad = prev_pad + (l.cin_feat ? feat_pad : 0);
    }
    m.defined = true;
    return PIFU_OK;
}

static int upload(float **dst, const std::vector<float> &host)
{
    if (*dst) {
        PIFU_HIP(hipFree(*dst));
        *dst = nullptr;
    }
    PIFU_HIP(hipMalloc(dst, host.size() * sizeof(float)));
    PIFU_HIP(hipMemcpy(*dst, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return PIFU_OK;
}

extern "C" int pifu_mlp_set_layer(pifu_ctx *ctx, int level, int layer, const float *weight, const float *bias,
                                  const float *norm_weight, const float *norm_bias, const float *running_mean,
                                  const float *running_var)
{
    PIFU_REQUIRE(ctx && weight && bias, PIFU_ERR_INVALID, "pifu_mlp_set_layer: null argument");
    PIFU_REQUIRE(level == 0 || level == 1, PIFU_ERR_INVALID, "pifu_mlp_set_layer: level must be 0 or 1");
    Mlp &m = ctx->mlp[level];
    PIFU_REQUIRE(m.defined, PIFU_ERR_STATE, "pifu_mlp_set_layer: MLP level %d is not defined", level);
    PIFU_REQUIRE(layer >= 0 && layer < m.nlayers, PIFU_ERR_INVALID, "pifu_mlp_set_layer: layer %d out of range", layer);
    PIFU_HIP(hipSetDevice(ctx->device));
    MlpLayer &l = m.layer[layer];
    const int cin = l.cin_prev + l.cin_feat;
    const int prev_pad = layer == 0 ? 0 : m.layer[layer - 1].cout_pad;
    // column map: reference input channel -> padded K column
    std::vector<int> colmap(cin);
    for (int k = 0; k < l.cin_prev; ++k) colmap[k] = k;
    if (l.cin_feat) {
        if (level == 0) {
            for (int k = 0; k < l.cin_feat; ++k) colmap[l.cin_prev + k] = prev_pad + k;
        } else {
            const Mlp &mc = ctx->mlp[0];
            const int cphi = mc.dims[mc.merge_layer + 1];
            const int cfine = m.dims[0] - cphi;
            const int fine_pad = (int)round_up(cfine, PIFU_BK);
            for (int k = 0; k < cfine; ++k) colmap[l.cin_prev + k] = prev_pad + k;
            for (int k = 0; k < cphi; ++k) colmap[l.cin_prev + cfine + k] = prev_pad + fine_pad + k;
        }
    }
    std::vector<float> w((size_t)(l.cout_pad > 0 ? l.cout_pad : 1) * l.kpad, 0.f);
    for (int o = 0; o < l.cout; ++o)
        for (int k = 0; k < cin; ++k) w[(size_t)o * l.kpad + colmap[k]] = weight[(size_t)o * cin + k];
    int rc = upload(&l.w, w);
    if (rc) return rc;
    std::vector<float> b(l.cout_pad > 0 ? l.cout_pad : 1, 0.f);
    for (int o = 0; o < l.cout; ++o) b[o] = bias[o];
    if ((rc = upload(&l.bias, b))) return rc;
    l.host_bias0 = bias[0];
    l.host_w.assign(weight, weight + (size_t)l.cout * cin);
    for (auto &kv : ctx->bf16_planes) (void)hipFree(kv.second.hi);      // device addresses may be reused by new weights
    ctx->bf16_planes.clear();
    m.col_built = false;
    ctx->col_cache.valid = false;
    if (level == 0) { ctx->mlp[1].col_built = false; ctx->pix.valid = false; }
    if (l.has_norm) {
        std::vector<float> g(l.cout_pad, 0.f), be(l.cout_pad, 0.f), fs(l.cout_pad, 0.f), fb(l.cout_pad, 0.f);
        if (m.norm != PIFU_NORM_NONE) {
            PIFU_REQUIRE(norm_weight && norm_bias, PIFU_ERR_INVALID, "pifu_mlp_set_layer: layer %d needs norm weight/bias", layer);
            for (int o = 0; o < l.cout; ++o) { g[o] = norm_weight[o]; be[o] = norm_bias[o]; }
        }
        if (m.norm == PIFU_NORM_BATCH_EVAL) {
            PIFU_REQUIRE(running_mean && running_var, PIFU_ERR_INVALID, "pifu_mlp_set_layer: batch-eval needs running statistics");
            for (int o = 0; o < l.cout; ++o) {
                const double rstd = 1.0 / sqrt((double)running_var[o] + 1e-5);
                fs[o] = (float)(norm_weight[o] * rstd);
                fb[o] = (float)((double)norm_bias[o] - (double)running_mean[o] * rstd * norm_weight[o]);
            }
        }
        if ((rc = upload(&l.gamma, g))) return rc;
        if ((rc = upload(&l.beta, be))) return rc;
        if ((rc = upload(&l.fixed_scale, fs))) return rc;
        if ((rc = upload(&l.fixed_shift, fb))) return rc;
    }
    l.set = true;
    return PIFU_OK;
}

// save_obj_mesh_with_color (mesh_util.py:189-198): 'v %.4f %.4f %.4f %.4f %.4f %.4f', faces 1-based (f0, f2, f1).
// Multi-threaded formatter, byte-identical to printf("%.4f"): the value is scaled by 1e4 with an error-free
// product (p + e = x * 1e4 exactly, via fma) and rounded half-to-even on the exact value.
namespace {

inline char *put_uint(char *o, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *o++ = tmp[--n];
    return o;
}

constexpr int SLOW_CHARS = 320;     // "%.4f" of the largest finite double: 1 sign + 309 digits + 5 = 315 characters

inline char *put_fixed4(char *o, double x)
{
    if (!(fabs(x) < 1e11)) {                                       // huge / inf / nan (the caller budgets SLOW_CHARS for each)
        if (x != x) { memcpy(o, "nan", 3); return o + 3; }         // Python prints nan without a sign
        if (fabs(x) == INFINITY) { const char *t = x < 0 ? "-inf" : "inf"; const size_t n = strlen(t); memcpy(o, t, n); return o + n; }
        const int n = snprintf(o, SLOW_CHARS, "%.4f", x);
        return o + (n < SLOW_CHARS ? n : SLOW_CHARS - 1);
    }
    if (signbit(x)) *o++ = '-';
    const double ax = fabs(x);
    const double p = ax * 1e4;
    const double e = fma(ax, 1e4, -p);
    double fl = floor(p);
    uint64_t n = (uint64_t)fl;
    const double d = p - fl;
    if (d > 0.5 || (d == 0.5 && (e > 0.0 || (e == 0.0 && (n & 1))))) ++n;
    else if (d == 0.0 && e < 0.0) { /* exact value just below an integer: floor(p) is already one too high */
        // p was rounded UP to the integer fl while the exact product is fl - |e|: the digits are those of fl - 1 + (1 - |e|)
        // whose fraction is > 0.5, so the rounded result is fl again: nothing to do
    }
    o = put_uint(o, n / 10000);
    *o++ = '.';
    const unsigned f = (unsigned)(n % 10000);
    *o++ = (char)('0' + f / 1000);
    *o++ = (char)('0' + f / 100 % 10);
    *o++ = (char)('0' + f / 10 % 10);
    *o++ = (char)('0' + f % 10);
    return o;
}

// number of characters put_fixed4 writes for x (same arithmetic, no stores); -1 for values outside its fast range
inline int fixed4_len(double x)
{
    if (!(fabs(x) < 1e11)) return -1;
    const double ax = fabs(x);
    const double p = ax * 1e4;
    const double e = fma(ax, 1e4, -p);
    const double fl = floor(p);
    uint64_t n = (uint64_t)fl;
    const double d = p - fl;
    if (d > 0.5 || (d == 0.5 && (e > 0.0 || (e == 0.0 && (n & 1))))) ++n;
    uint64_t ip = n / 10000;
    int digits = 1;
    while (ip >= 10) { ip /= 10; ++digits; }
    return (signbit(x) ? 1 : 0) + digits + 5;
}

}  // namespace

namespace {

constexpr int64_t OBJ_BLOCK = 16384;        // lines per block of the writer

std::atomic<int> g_host_threads{0};          // pifu_set_host_threads: 0 = the machine's

inline int obj_threads(int64_t nblocks)
{
    unsigned hw = g_host_threads.load() > 0 ? (unsigned)g_host_threads.load() : std::thread::hardware_concurrency();
#ifdef PIFU_LAB     // tools/obj_bench.py: writer threads as a parameter
    if (const char *v = getenv("PIFU_OBJ_THREADS")) hw = (unsigned)atoi(v);
#endif
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(hw ? hw : 1u, 32u), nblocks));
}

inline bool write_all(int fd, const char *src, size_t left, int64_t off)
{
    // in pieces of at most 256 KB: on this image's ext4 a buffered pwrite of 1 MB or more runs at ~0.3 GB/s where 64 - 512 KB
    // calls reach 4 GB/s (18 MB: 62 ms in one call, 36 ms in 4 MB calls, 4 ms in 256 KB calls)
    constexpr size_t PIECE = 256 << 10;
    while (left) {
        const ssize_t w = pwrite(fd, src, left < PIECE ? left : PIECE, (off_t)off);
        if (w <= 0) return false;
        src += w; off += w; left -= (size_t)w;
    }
    return true;
}

// 'v x y z r g b' lines of vertices [lo, hi) into o; returns the end
// (colours: float64 as the reference's array holds them, or the float32 values it was filled from - the same numbers)
template <typename C>
inline char *format_vertex_lines(char *o, const double *verts, const C *colors, int64_t lo, int64_t hi)
{
    for (int64_t i = lo; i < hi; ++i) {
        const double *v = verts + 3 * i;
        const C *c = colors + 3 * i;
        *o++ = 'v';
        for (int k = 0; k < 3; ++k) { *o++ = ' '; o = put_fixed4(o, v[k]); }
        for (int k = 0; k < 3; ++k) { *o++ = ' '; o = put_fixed4(o, (double)c[k]); }
        *o++ = '\n';
    }
    return o;
}

// values of vertices [lo, hi) outside the fast formatter's range (each needs up to SLOW_CHARS characters)
template <typename C>
inline size_t count_slow_values(const double *verts, const C *colors, int64_t lo, int64_t hi)
{
    size_t slow = 0;
    for (int64_t i = lo; i < hi; ++i)
        for (int k = 0; k < 3; ++k) slow += !(fabs(verts[3 * i + k]) < 1e11) + !(fabs((double)colors[3 * i + k]) < 1e11);
    return slow;
}

// 'f a b c' lines of triangles [lo, hi): 1-based (f0, f2, f1); triangle i at faces[i * row_stride + k * col_stride]
inline char *format_face_lines(char *o, const int32_t *faces, int64_t row_stride, int64_t col_stride, int64_t lo, int64_t hi)
{
    for (int64_t i = lo; i < hi; ++i) {
        const int32_t *tr = faces + i * row_stride;
        const int idx[3] = {tr[0] + 1, tr[2 * col_stride] + 1, tr[col_stride] + 1};
        *o++ = 'f';
        for (int k = 0; k < 3; ++k) {
            *o++ = ' ';
            if (idx[k] < 0) { *o++ = '-'; o = put_uint(o, (uint64_t)(-(int64_t)idx[k])); }
            else o = put_uint(o, (uint64_t)idx[k]);
        }
        *o++ = '\n';
    }
    return o;
}

constexpr size_t OBJ_VERTEX_LINE = 168, OBJ_FACE_LINE = 40;      // upper bounds of a line without out-of-range values

// Two-phase writer.  The face lines depend on the triangles only, so pifu_obj_begin starts formatting them on worker threads
// (into memory, block by block) while the caller is still computing the vertex colours; pifu_obj_finish formats the vertex
// lines - each thread takes blocks in file order, learns a block's file offset from its predecessor's (offset + length) as
// soon as that one is FORMATTED, and writes the block at its own offset - and then writes the face blocks behind them.
struct ObjWriter {
    int fd = -1;
    std::string path;
    // Streaming mode (pifu_obj_layout / pifu_obj_ready), for callers that know every vertex POSITION long before the last
    // COLOUR (gen_mesh: marching cubes, then the colour loop).  A colour in [0, 10) always prints as 6 characters, so the length
    // of every vertex line - hence the file offset of every vertex block and of the face section - follows from the positions
    // alone: the pool measures the blocks as soon as the layout is declared, the face blocks go to their final place while
    // the caller is still computing colours, and a vertex block is formatted and written the moment its colours are declared
    // ready.  A block whose text does not have the predicted length (a colour outside [0, 10): NaN, negative) flags a mismatch
    // and the finish call rewrites the file in one pass from the complete arrays.  The text goes to `path + ".part"`.
    bool streaming = false;
    const double *sv = nullptr;
    const float *sc = nullptr;
    int64_t sn = 0, svblocks = 0;
    std::vector<int64_t> vlen, voff;            // predicted bytes per vertex block, their prefix (valid once layout_done)
    std::vector<int64_t> foff;                  // prefix of the face blocks' bytes (valid once faces_ready)
    std::atomic<int64_t> measure_next{0}, measured{0}, vnext{0}, ready_upto{0}, faces_formatted{0}, fwrite_next{0};
    std::atomic<bool> mismatch{false}, vbad{false};
    std::mutex mu;
    std::condition_variable cv;
    bool layout_done = false, faces_ready = false, stop = false;      // guarded by mu
    std::vector<std::thread> vpool;

    int64_t block_lo(int64_t b) const { return b * OBJ_BLOCK; }
    int64_t block_hi(int64_t b) const { return std::min(sn, (b + 1) * OBJ_BLOCK); }

    void vertex_worker()
    {
        // (1) measure: predicted text length of every vertex block from the positions alone
        for (;;) {
            const int64_t b = measure_next.fetch_add(1, std::memory_order_relaxed);
            if (b >= svblocks) break;
            int64_t bytes = 0;
            bool odd = false;
            for (int64_t i = block_lo(b); i < block_hi(b); ++i)
                for (int k = 0; k < 3; ++k) {
                    const int n = fixed4_len(sv[3 * i + k]);
                    if (n < 0) odd = true; else bytes += n;
                }
            if (odd) mismatch.store(true);                       // a position outside the fast formatter's range: one-pass rewrite
            vlen[b] = bytes + 26 * (block_hi(b) - block_lo(b));   // 'v' + 6 blanks + 3 x 6 colour characters + newline
            if (measured.fetch_add(1, std::memory_order_acq_rel) + 1 == svblocks) {
                for (int64_t q = 0; q < svblocks; ++q) voff[q + 1] = voff[q] + vlen[q];
                { std::lock_guard<std::mutex> lk(mu); layout_done = true; }
                cv.notify_all();
            }
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return layout_done || stop; });
            if (!layout_done) return;
        }
        // (2) format + write the vertex blocks, in order of arrival of their colours
        constexpr size_t FAST = (size_t)OBJ_BLOCK * OBJ_VERTEX_LINE + 64;
        std::unique_ptr<char[]> own(new char[FAST]);               // uninitialised on purpose, re-used per block
        for (;;) {
            const int64_t b = vnext.fetch_add(1, std::memory_order_relaxed);
            if (b >= svblocks) break;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready_upto.load() >= block_hi(b) || stop; });
                if (ready_upto.load() < block_hi(b)) return;     // abandoned
            }
            if (mismatch.load()) continue;                       // the file is rewritten as a whole anyway
            if (count_slow_values(sv, sc, block_lo(b), block_hi(b))) { mismatch.store(true); continue; }
            char *o = format_vertex_lines(own.get(), sv, sc, block_lo(b), block_hi(b));
            if ((int64_t)(o - own.get()) != vlen[b]) { mismatch.store(true); continue; }
            if (!write_all(fd, own.get(), (size_t)vlen[b], voff[b])) vbad.store(true);
        }
    }

    // the face threads, once every face block is formatted: wait for the vertex layout, then write the blocks in place
    void write_faces_in_place()
    {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return (layout_done && faces_ready && streaming) || stop; });
            if (stop) return;
        }
        for (;;) {
            const int64_t b = fwrite_next.fetch_add(1, std::memory_order_relaxed);
            if (b >= nblocks) break;
            if (!write_all(fd, text[b].get(), len[b], voff[svblocks] + foff[b])) vbad.store(true);
        }
    }

    const int32_t *faces = nullptr;
    int64_t nfaces = 0, row_stride = 3, col_stride = 1;
    int64_t nblocks = 0;
    std::vector<std::unique_ptr<char[]>> text;
    std::vector<size_t> len;
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;

    void format_faces()
    {
        for (;;) {
            const int64_t b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= nblocks) break;
            const int64_t lo = b * OBJ_BLOCK, hi = std::min(nfaces, lo + OBJ_BLOCK);
            text[b].reset(new char[(size_t)(hi - lo) * OBJ_FACE_LINE + 16]);
            char *o = format_face_lines(text[b].get(), faces, row_stride, col_stride, lo, hi);
            len[b] = (size_t)(o - text[b].get());
            if (faces_formatted.fetch_add(1, std::memory_order_acq_rel) + 1 == nblocks) {
                for (int64_t q = 0; q < nblocks; ++q) foff[q + 1] = foff[q] + (int64_t)len[q];
                { std::lock_guard<std::mutex> lk(mu); faces_ready = true; }
                cv.notify_all();
            }
        }
        write_faces_in_place();      // streaming mode only; returns at once when the writer is finished the one-call way
    }
};

}  // namespace

extern "C" int pifu_set_host_threads(int n)
{
    PIFU_REQUIRE(n >= 0, PIFU_ERR_INVALID, "pifu_set_host_threads: negative thread count");
    g_host_threads.store(n);
    return PIFU_OK;
}

extern "C" int pifu_obj_begin(const char *path, const int32_t *faces, int64_t nfaces, int64_t row_stride, int64_t col_stride,
                              void **writer)
{
    PIFU_REQUIRE(path && writer && (faces || nfaces == 0) && nfaces >= 0, PIFU_ERR_INVALID, "pifu_obj_begin: bad argument");
    // the file itself is created by pifu_obj_finish: like the reference's writer (mesh_util.py:189-198, called once the colours
    // exist) a run that fails between the two steps leaves whatever was at `path` untouched
    ObjWriter *w = new ObjWriter();
    w->path = path; w->faces = faces; w->nfaces = nfaces; w->row_stride = row_stride; w->col_stride = col_stride;
    w->nblocks = (nfaces + OBJ_BLOCK - 1) / OBJ_BLOCK;
    w->text.resize((size_t)w->nblocks);
    w->len.assign((size_t)w->nblocks, 0);
    w->foff.assign((size_t)w->nblocks + 1, 0);
    if (w->nblocks == 0) w->faces_ready = true;
    // half of the cores: the caller's own work (the colour loop's launches, copies) runs beside these threads
    const int nthreads = w->nblocks ? std::max(1, obj_threads(w->nblocks) / 2) : 0;
    for (int t = 0; t < nthreads; ++t) w->pool.emplace_back([w]() { w->format_faces(); });
    *writer = w;
    return PIFU_OK;
}

extern "C" int pifu_obj_layout(void *writer, const double *verts, const float *colors, int64_t nverts)
{
    PIFU_REQUIRE(writer && nverts >= 0 && ((verts && colors) || nverts == 0), PIFU_ERR_INVALID, "pifu_obj_layout: bad argument");
    ObjWriter *w = (ObjWriter *)writer;
    PIFU_REQUIRE(!w->streaming, PIFU_ERR_STATE, "pifu_obj_layout: the layout of this writer was already declared");
    const std::string part = w->path + ".part";
    const int fd = open(part.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    PIFU_REQUIRE(fd >= 0, PIFU_ERR_INVALID, "pifu_obj_layout: cannot open %s", part.c_str());
    w->fd = fd;
    w->sv = verts; w->sc = colors; w->sn = nverts;
    w->svblocks = (nverts + OBJ_BLOCK - 1) / OBJ_BLOCK;
    w->vlen.assign((size_t)w->svblocks, 0);
    w->voff.assign((size_t)w->svblocks + 1, 0);
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->streaming = true;
        if (w->svblocks == 0) w->layout_done = true;
    }
    const int nthreads = std::max(2, obj_threads(1 << 20) / 2);     // beside the face formatters and the caller's launches
    for (int t = 0; t < nthreads; ++t) w->vpool.emplace_back([w]() { w->vertex_worker(); });
    w->cv.notify_all();
    return PIFU_OK;
}

extern "C" int pifu_obj_ready(void *writer, int64_t upto)
{
    PIFU_REQUIRE(writer, PIFU_ERR_INVALID, "pifu_obj_ready: null writer");
    ObjWriter *w = (ObjWriter *)writer;
    PIFU_REQUIRE(w->streaming && upto >= 0 && upto <= w->sn, PIFU_ERR_INVALID, "pifu_obj_ready: %lld vertices declared ready of %lld "
                 "(declare the layout first)", (long long)upto, (long long)w->sn);
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (upto > w->ready_upto.load()) w->ready_upto.store(upto);
    }
    w->cv.notify_all();
    return PIFU_OK;
}

extern "C" int pifu_obj_finish(void *writer, const double *verts, const double *colors, int64_t nverts)
{
    PIFU_REQUIRE(writer, PIFU_ERR_INVALID, "pifu_obj_finish: null writer");
    std::unique_ptr<ObjWriter> w((ObjWriter *)writer);
    if (w->streaming) {                                 // layout mode: (NULL, NULL, 0) = every colour is valid now
        const bool args_ok = nverts < 0 || (nverts == 0 && !verts && !colors);
        {
            std::lock_guard<std::mutex> lk(w->mu);
            if (nverts >= 0 && args_ok) w->ready_upto.store(w->sn);
            else w->stop = true;
        }
        w->cv.notify_all();
        for (auto &th : w->vpool) th.join();
        for (auto &th : w->pool) th.join();
        const std::string part = w->path + ".part";
        bool bad = w->vbad.load();
        if (nverts >= 0 && args_ok && !bad && w->mismatch.load()) {
            // some line did not have the predicted length: one pass over the complete arrays, as the one-call writer does
            bad = ftruncate(w->fd, 0) != 0;
            int64_t off = 0;
            constexpr size_t FAST = (size_t)OBJ_BLOCK * OBJ_VERTEX_LINE + 64;
            std::unique_ptr<char[]> own(new char[FAST]);
            for (int64_t b = 0; b < w->svblocks && !bad; ++b) {
                const size_t slow = count_slow_values(w->sv, w->sc, w->block_lo(b), w->block_hi(b));
                std::unique_ptr<char[]> big;
                char *base = own.get();
                if (slow) { big.reset(new char[FAST + slow * SLOW_CHARS]); base = big.get(); }
                char *o = format_vertex_lines(base, w->sv, w->sc, w->block_lo(b), w->block_hi(b));
                if (!write_all(w->fd, base, (size_t)(o - base), off)) bad = true;
                off += (int64_t)(o - base);
            }
            for (int64_t b = 0; b < w->nblocks && !bad; ++b) {
                if (!write_all(w->fd, w->text[b].get(), w->len[b], off)) bad = true;
                off += (int64_t)w->len[b];
            }
        }
        if (close(w->fd) != 0) bad = true;
        if (nverts < 0 || bad || !args_ok) {            // abandoned or failed: the destination is left as it was
            (void)unlink(part.c_str());
            PIFU_REQUIRE(args_ok, PIFU_ERR_INVALID, "pifu_obj_finish: the layout of this writer was declared; finish it "
                         "with (NULL, NULL, 0)");
            PIFU_REQUIRE(nverts < 0, PIFU_ERR_INVALID, "pifu_obj_finish: write error on %s", part.c_str());
            return PIFU_OK;
        }
        PIFU_REQUIRE(rename(part.c_str(), w->path.c_str()) == 0, PIFU_ERR_INVALID, "pifu_obj_finish: cannot rename %s", part.c_str());
        return PIFU_OK;
    }
    {   // one-call form (or an abandoned writer without a layout): the face threads only format, nothing is written in place
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv.notify_all();
    if (nverts < 0) {                                   // abandon: wait for the face formatters, write nothing
        for (auto &th : w->pool) th.join();
        return PIFU_OK;
    }
    const bool args_ok = (verts && colors) || nverts == 0;
    const int fd = args_ok ? open(w->path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666) : -1;
    if (fd < 0) {
        for (auto &th : w->pool) th.join();
        PIFU_REQUIRE(args_ok, PIFU_ERR_INVALID, "pifu_obj_finish: bad argument");
        PIFU_REQUIRE(false, PIFU_ERR_INVALID, "pifu_obj_finish: cannot open %s", w->path.c_str());
    }
    std::atomic<bool> bad{false};
    int64_t vbytes = 0;
    if (args_ok && nverts > 0) {
        const int64_t nblocks = (nverts + OBJ_BLOCK - 1) / OBJ_BLOCK;
        std::unique_ptr<std::atomic<int64_t>[]> offs(new std::atomic<int64_t>[(size_t)nblocks + 1]);
        for (int64_t i = 0; i <= nblocks; ++i) offs[i].store(i == 0 ? 0 : -1, std::memory_order_relaxed);
        std::atomic<int64_t> next{0};
        std::vector<std::thread> pool;
        const int nthreads = obj_threads(nblocks);
        for (int t = 0; t < nthreads; ++t) {
            pool.emplace_back([&]() {
                constexpr size_t FAST = (size_t)OBJ_BLOCK * OBJ_VERTEX_LINE + 64;
                std::unique_ptr<char[]> own(new char[FAST]);               // uninitialised on purpose, re-used per block
                std::unique_ptr<char[]> big;
                for (;;) {
                    const int64_t b = next.fetch_add(1, std::memory_order_relaxed);
                    if (b >= nblocks) break;
                    const int64_t lo = b * OBJ_BLOCK, hi = std::min(nverts, lo + OBJ_BLOCK);
                    const size_t slow = count_slow_values(verts, colors, lo, hi);
                    char *base = own.get();
                    if (slow) { big.reset(new char[FAST + slow * SLOW_CHARS]); base = big.get(); }
                    char *o = format_vertex_lines(base, verts, colors, lo, hi);
                    int64_t off;
                    while ((off = offs[b].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
                    const size_t n = (size_t)(o - base);
                    offs[b + 1].store(off + (int64_t)n, std::memory_order_release);
                    if (!write_all(fd, base, n, off)) bad.store(true);
                    big.reset();
                }
            });
        }
        for (auto &th : pool) th.join();
        vbytes = offs[nblocks].load();
    }
    for (auto &th : w->pool) th.join();
    if (args_ok && w->nblocks) {
        std::vector<int64_t> off((size_t)w->nblocks + 1, vbytes);
        for (int64_t b = 0; b < w->nblocks; ++b) off[b + 1] = off[b] + (int64_t)w->len[b];
        std::atomic<int64_t> next{0};
        std::vector<std::thread> pool;
        const int nthreads = obj_threads(w->nblocks);
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&]() {
                for (;;) {
                    const int64_t b = next.fetch_add(1, std::memory_order_relaxed);
                    if (b >= w->nblocks) break;
                    if (!write_all(fd, w->text[b].get(), w->len[b], off[b])) bad.store(true);
                    w->text[b].reset();
                }
            });
        for (auto &th : pool) th.join();
    }
    if (close(fd) != 0) bad.store(true);
    PIFU_REQUIRE(!bad.load(), PIFU_ERR_INVALID, "pifu_obj_finish: write error on %s", w->path.c_str());
    return PIFU_OK;
}

extern "C" int pifu_write_obj(const char *path, const double *verts, const double *colors, int64_t nverts,
                              const int32_t *faces, int64_t nfaces)
{
    PIFU_REQUIRE(path && (verts || nverts == 0) && (colors || nverts == 0) && (faces || nfaces == 0), PIFU_ERR_INVALID,
                 "pifu_write_obj: null argument");
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    PIFU_REQUIRE(fd >= 0, PIFU_ERR_INVALID, "pifu_write_obj: cannot open %s", path);
    // one pass: the vertex blocks, then the face blocks, handed out in file order; a thread formats a block into its own
    // (re-used, cache-resident) buffer, takes the block's file offset from its predecessor's (offset + length) - known as
    // soon as the predecessor is formatted - and writes the block at that offset
    const int64_t vblocks = (nverts + OBJ_BLOCK - 1) / OBJ_BLOCK, fblocks = (nfaces + OBJ_BLOCK - 1) / OBJ_BLOCK;
    const int64_t nblocks = vblocks + fblocks;
    std::unique_ptr<std::atomic<int64_t>[]> offs(new std::atomic<int64_t>[(size_t)nblocks + 1]);
    for (int64_t i = 0; i <= nblocks; ++i) offs[i].store(i == 0 ? 0 : -1, std::memory_order_relaxed);
    std::atomic<int64_t> next{0};
    std::atomic<bool> bad{false};
    std::vector<std::thread> pool;
    const int nthreads = obj_threads(nblocks);
    for (int t = 0; t < nthreads; ++t) {
        pool.emplace_back([&]() {
            constexpr size_t FAST = (size_t)OBJ_BLOCK * OBJ_VERTEX_LINE + 64;
            std::unique_ptr<char[]> own(new char[FAST]);               // uninitialised on purpose
            std::unique_ptr<char[]> big;
            for (;;) {
                const int64_t b = next.fetch_add(1, std::memory_order_relaxed);
                if (b >= nblocks) break;
                char *base = own.get(), *o;
                if (b < vblocks) {
                    const int64_t lo = b * OBJ_BLOCK, hi = std::min(nverts, lo + OBJ_BLOCK);
                    const size_t slow = count_slow_values(verts, colors, lo, hi);
                    if (slow) { big.reset(new char[FAST + slow * SLOW_CHARS]); base = big.get(); }
                    o = format_vertex_lines(base, verts, colors, lo, hi);
                } else {
                    const int64_t lo = (b - vblocks) * OBJ_BLOCK, hi = std::min(nfaces, lo + OBJ_BLOCK);
                    o = format_face_lines(base, faces, 3, 1, lo, hi);
                }
                int64_t off;
                while ((off = offs[b].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
                const size_t n = (size_t)(o - base);
                offs[b + 1].store(off + (int64_t)n, std::memory_order_release);
                if (!write_all(fd, base, n, off)) bad.store(true);
                big.reset();
            }
        });
    }
    for (auto &th : pool) th.join();
    if (close(fd) != 0) bad.store(true);
    PIFU_REQUIRE(!bad.load(), PIFU_ERR_INVALID, "pifu_write_obj: write error on %s", path);
    return PIFU_OK;
}
