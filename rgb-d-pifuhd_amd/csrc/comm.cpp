// The one exchange step of the sharded path (an all-gather of the fp32 predictions per grid / octree level, of
// the colours per round of the colour loop) as a C-ABI call for bindings that have no torch.distributed: a thin wrapper
// over RCCL's ncclAllGather.  RCCL is looked up at run time (dlopen) - the copy the process already holds when a framework
// loaded one (torch ships its own librccl.so), /opt/rocm's otherwise - so the library itself loads without it.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

#include "pifu_internal.h"

namespace {

struct NcclId { char internal[PIFU_COMM_ID_BYTES]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *NcclComm;
enum { NCCL_FLOAT32 = 7, NCCL_UINT8 = 1 };                  // ncclDataType_t values of rccl.h

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_why[512] = "";        // dlerror() text of the decisive dlopen attempt, copied when it failed (dlerror() clears on read)

void keep_dlerror()
{
    const char *e = dlerror();
    snprintf(g_rccl_why, sizeof(g_rccl_why), "%s", e && *e ? e : "dlopen gave no reason");
}

void load_rccl()
{
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names)                                            // the copy already in the process, if any
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char *n : names) {
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
            keep_dlerror();
        }
    if (!h && !(h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL))) keep_dlerror();
    if (!h) return;
    g_rccl.handle = h;
    g_rccl.GetUniqueId = (int (*)(NcclId *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(NcclComm *, int, NcclId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(NcclComm))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, NcclComm, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather;
}

int need_rccl(const char *who)
{
    std::call_once(g_rccl_once, load_rccl);
    PIFU_REQUIRE(g_rccl.ok, PIFU_ERR_STATE, "%s: RCCL (librccl.so) could not be loaded: %s", who,
                 g_rccl.handle ? "a symbol is missing" : g_rccl_why);
    return PIFU_OK;
}

#define PIFU_NCCL(call, who)                                                                                     \
    do {                                                                                                         \
        const int r_ = (call);                                                                                   \
        if (r_ != 0) {                                                                                           \
            pifu_set_error("%s: %s failed: %s", who, #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return PIFU_ERR_HIP;                                                                                 \
        }                                                                                                        \
    } while (0)

}  // namespace

struct pifu_comm {
    NcclComm comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

extern "C" int pifu_comm_unique_id(uint8_t id[PIFU_COMM_ID_BYTES])
{
    PIFU_REQUIRE(id, PIFU_ERR_INVALID, "pifu_comm_unique_id: null argument");
    int rc = need_rccl("pifu_comm_unique_id");
    if (rc) return rc;
    NcclId u;
    PIFU_NCCL(g_rccl.GetUniqueId(&u), "pifu_comm_unique_id");
    memcpy((void *)id, (const void *)u.internal, PIFU_COMM_ID_BYTES);
    return PIFU_OK;
}

extern "C" int pifu_comm_create(int device, int rank, int world, const uint8_t id[PIFU_COMM_ID_BYTES], pifu_comm **out)
{
    PIFU_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, PIFU_ERR_INVALID, "pifu_comm_create: bad argument");
    int rc = need_rccl("pifu_comm_create");
    if (rc) return rc;
    PIFU_HIP(hipSetDevice(device));
    NcclId u;
    memcpy((void *)u.internal, (const void *)id, PIFU_COMM_ID_BYTES);
    pifu_comm *c = new pifu_comm();
    c->device = device; c->rank = rank; c->world = world;
    const int r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != 0) {
        pifu_set_error("pifu_comm_create: ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        delete c;
        return PIFU_ERR_HIP;
    }
    *out = c;
    return PIFU_OK;
}

extern "C" int pifu_comm_destroy(pifu_comm *comm)
{
    if (!comm) return PIFU_OK;
    if (comm->comm && g_rccl.ok) {
        (void)hipSetDevice(comm->device);
        (void)g_rccl.CommDestroy(comm->comm);
    }
    delete comm;
    return PIFU_OK;
}

extern "C" int pifu_allgather_f32(pifu_comm *comm, const float *send, float *recv, int64_t count_per_rank, pifu_stream stream)
{
    PIFU_REQUIRE(comm && comm->comm && send && recv && count_per_rank >= 0, PIFU_ERR_INVALID, "pifu_allgather_f32: bad argument");
    if (count_per_rank == 0) return PIFU_OK;
    PIFU_HIP(hipSetDevice(comm->device));
    PIFU_NCCL(g_rccl.AllGather(send, recv, (size_t)count_per_rank, NCCL_FLOAT32, comm->comm, (hipStream_t)stream), "pifu_allgather_f32");
    return PIFU_OK;
}

extern "C" int pifu_allgather_bytes(pifu_comm *comm, const void *send, void *recv, int64_t bytes_per_rank, pifu_stream stream)
{
    PIFU_REQUIRE(comm && comm->comm && send && recv && bytes_per_rank >= 0, PIFU_ERR_INVALID, "pifu_allgather_bytes: bad argument");
    if (bytes_per_rank == 0) return PIFU_OK;
    PIFU_HIP(hipSetDevice(comm->device));
    PIFU_NCCL(g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, NCCL_UINT8, comm->comm, (hipStream_t)stream), "pifu_allgather_bytes");
    return PIFU_OK;
}
