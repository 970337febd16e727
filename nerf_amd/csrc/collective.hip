// The one exchange of data-parallel training at the C boundary: the sum of a gradient bucket over the ranks of an
// RCCL communicator (RCCL = the NCCL API over xGMI / PCIe on ROCm).  The reference is single-process; rays are
// independent, so a step needs nothing else (nerf_amd/dist.py does the same through torch.distributed).
//
// librccl is not a link-time dependency: a process that also hosts PyTorch already holds a copy (torch ships its
// own librccl.so), and a communicator must be used with the copy that created it.  The entry points are therefore
// looked up at first use -- first among the libraries the process has already loaded, then by loading
// librccl.so.1 -- and every call goes through that one set of pointers.
#include <dlfcn.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include <rccl/rccl.h>

#include "common.h"

namespace nerf {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*GetVersion)(int*);
};

static RcclApi g_api;
static std::once_flag g_once;
static bool g_ok = false;
static char g_why[256] = "";

static void* lookup(void* handle, const char* name) {
    void* p = dlsym(handle ? handle : RTLD_DEFAULT, name);
    return p;
}

static void load_rccl() {
    void* handle = nullptr;   // RTLD_DEFAULT first: the copy the process already uses
    if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
        handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!handle) {
            snprintf(g_why, sizeof g_why, "librccl is not loaded and cannot be loaded: %s", dlerror());
            return;
        }
    }
#define NERF_RCCL_SYM(field, name)                                                    \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(lookup(handle, name));      \
    if (!g_api.field) {                                                               \
        snprintf(g_why, sizeof g_why, "librccl has no symbol %s", name);              \
        return;                                                                       \
    }
    NERF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    NERF_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    NERF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    NERF_RCCL_SYM(CommCount, "ncclCommCount")
    NERF_RCCL_SYM(AllReduce, "ncclAllReduce")
    NERF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
    NERF_RCCL_SYM(GetVersion, "ncclGetVersion")
#undef NERF_RCCL_SYM
    g_ok = true;
}

static int rccl(const char* who) {
    std::call_once(g_once, load_rccl);
    if (!g_ok) return fail(NERF_ELAUNCH, "%s: %s", who, g_why);
    return NERF_OK;
}

static int rccl_check(const char* who, ncclResult_t r) {
    if (r == ncclSuccess) return NERF_OK;
    return fail(NERF_ELAUNCH, "%s: RCCL error %d: %s", who, (int)r, g_api.GetErrorString(r));
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_rccl_version(void) {
    if (rccl("nerf_rccl_version")) return -1;
    int v = 0;
    return g_api.GetVersion(&v) == ncclSuccess ? v : -1;
}

extern "C" int nerf_comm_unique_id(void* id128) {
    NERF_REQUIRE(id128, "nerf_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == NERF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (int rc = rccl("nerf_comm_unique_id")) return rc;
    return rccl_check("nerf_comm_unique_id", g_api.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
}

extern "C" int nerf_comm_init_rank(const void* id128, int world, int rank, void** comm) {
    NERF_REQUIRE(id128 && comm, "nerf_comm_init_rank: null pointer");
    NERF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "nerf_comm_init_rank: rank %d of %d", rank, world);
    if (int rc = rccl("nerf_comm_init_rank")) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    if (int rc = rccl_check("nerf_comm_init_rank", g_api.CommInitRank(&c, world, id, rank))) return rc;
    *comm = c;
    return NERF_OK;
}

extern "C" int nerf_comm_destroy(void* comm) {
    if (!comm) return NERF_OK;
    if (int rc = rccl("nerf_comm_destroy")) return rc;
    return rccl_check("nerf_comm_destroy", g_api.CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
}

extern "C" int nerf_allreduce_sum(float* buf, size_t n, void* rccl_comm, void* stream) {
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(buf && rccl_comm, "nerf_allreduce_sum: null pointer");
    if (int rc = rccl("nerf_allreduce_sum")) return rc;
    return rccl_check("nerf_allreduce_sum", g_api.AllReduce(buf, buf, n, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(rccl_comm),
                                                            (hipStream_t)stream));
}
