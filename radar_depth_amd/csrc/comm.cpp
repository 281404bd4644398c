// Data-parallel exchange step of the hot path: RCCL sum-all-reduce of gradient buckets over xGMI on a
// caller-supplied HIP stream.  The reference has no collectives at all (single-process training, main.py:285-447); these
// entry points are the C ABI a one-process-per-GPU launcher binds instead of torch.distributed.
//
// RCCL is resolved at run time (dlopen + dlsym), not at link time: the library builds and loads on hosts without RCCL, a
// process that already holds an RCCL (PyTorch bundles one) reuses THAT copy instead of loading a second one with clashing
// ncclXxx symbols, and single-GPU users never pay for it.
#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace {

// ABI subset of rccl.h (RCCL 2.x; stable since NCCL 2.0)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_comm_destroy)(Comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9;

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_broadcast broadcast = nullptr;
    fn_error_string error_string = nullptr;
    Comm comm = nullptr;
    int rank = -1, world = 0;
};
Rccl g;
std::mutex g_mu;

int load_rccl() {
    if (g.handle) return RD_OK;
    // 1. a copy this process already mapped (torch's), 2. RD_RCCL_LIB, 3. the system ROCm one
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
        if (!g.handle) g.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    if (!g.handle && getenv("RD_RCCL_LIB")) g.handle = dlopen(getenv("RD_RCCL_LIB"), RTLD_NOW | RTLD_LOCAL);
    for (const char* n : names)
        if (!g.handle) g.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!g.handle) g.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!g.handle) {
        rd::set_error("rd_comm: cannot load RCCL (librccl.so): %s", dlerror());
        return RD_ELAUNCH;
    }
    g.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(g.handle, "ncclGetUniqueId"));
    g.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(g.handle, "ncclCommInitRank"));
    g.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(g.handle, "ncclCommDestroy"));
    g.all_reduce = reinterpret_cast<fn_all_reduce>(dlsym(g.handle, "ncclAllReduce"));
    g.broadcast = reinterpret_cast<fn_broadcast>(dlsym(g.handle, "ncclBroadcast"));
    g.error_string = reinterpret_cast<fn_error_string>(dlsym(g.handle, "ncclGetErrorString"));
    if (!g.get_unique_id || !g.comm_init_rank || !g.comm_destroy || !g.all_reduce || !g.broadcast) {
        rd::set_error("rd_comm: librccl.so lacks the NCCL 2.x entry points");
        g.handle = nullptr;
        return RD_ELAUNCH;
    }
    return RD_OK;
}

#define RD_CHECK_NCCL(expr)                                                                          \
    do {                                                                                             \
        const int r__ = (expr);                                                                      \
        if (r__ != 0) {                                                                              \
            rd::set_error("%s: %s", #expr, g.error_string ? g.error_string(r__) : "RCCL error");     \
            return RD_ELAUNCH;                                                                       \
        }                                                                                            \
    } while (0)

int nccl_dtype(int dtype) { return dtype == RD_DTYPE_F32 ? kNcclFloat32 : (dtype == RD_DTYPE_BF16 ? kNcclBfloat16 : -1); }

}  // namespace

// rank 0: fill a 128-byte rendezvous token; the launcher ships it to every rank (torch.distributed store, a file, MPI ...)
extern "C" int rd_comm_unique_id(void* out128) {
    RD_CHECK_ARG(out128 != nullptr, "rd_comm_unique_id: null buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = load_rccl();
    if (rc != RD_OK) return rc;
    UniqueId id;
    RD_CHECK_NCCL(g.get_unique_id(&id));
    memcpy(out128, id.internal, sizeof(id.internal));
    return RD_OK;
}

// every rank, after hipSetDevice: joins the communicator (collective call).  One communicator per process.
extern "C" int rd_comm_init(const void* unique_id128, int32_t rank, int32_t world) {
    RD_CHECK_ARG(unique_id128 && world >= 1 && rank >= 0 && rank < world, "rd_comm_init: bad arguments (rank %d of %d)", rank, world);
    std::lock_guard<std::mutex> lk(g_mu);
    RD_CHECK_ARG(g.comm == nullptr, "rd_comm_init: communicator already initialised (rd_comm_destroy first)");
    int rc = load_rccl();
    if (rc != RD_OK) return rc;
    UniqueId id;
    memcpy(id.internal, unique_id128, sizeof(id.internal));
    RD_CHECK_NCCL(g.comm_init_rank(&g.comm, world, id, rank));
    g.rank = rank;
    g.world = world;
    return RD_OK;
}

extern "C" int rd_comm_world(void) { return g.comm ? g.world : 0; }
extern "C" int rd_comm_rank(void) { return g.comm ? g.rank : -1; }

// In-place sum over all ranks of `count` elements at `ptr`, asynchronous on `stream` (the caller orders it behind the backward
// kernels that produce the bucket with an event, and orders the optimizer step behind it).  No host synchronisation.
extern "C" int rd_allreduce_bucket(void* ptr, int64_t count, int32_t dtype, void* stream) {
    RD_CHECK_ARG(g.comm != nullptr, "rd_allreduce_bucket: rd_comm_init has not been called");
    RD_CHECK_ARG(ptr && count > 0 && nccl_dtype(dtype) >= 0, "rd_allreduce_bucket: bad arguments");
    RD_CHECK_NCCL(g.all_reduce(ptr, ptr, (size_t)count, nccl_dtype(dtype), kNcclSum, g.comm, static_cast<hipStream_t>(stream)));
    return RD_OK;
}

// In-place broadcast from `root` (initial parameters / momentum / BatchNorm buffers, so that replicas start identical).
extern "C" int rd_broadcast(void* ptr, int64_t count, int32_t dtype, int32_t root, void* stream) {
    RD_CHECK_ARG(g.comm != nullptr, "rd_broadcast: rd_comm_init has not been called");
    RD_CHECK_ARG(ptr && count > 0 && nccl_dtype(dtype) >= 0 && root >= 0 && root < g.world, "rd_broadcast: bad arguments");
    RD_CHECK_NCCL(g.broadcast(ptr, ptr, (size_t)count, nccl_dtype(dtype), root, g.comm, static_cast<hipStream_t>(stream)));
    return RD_OK;
}

extern "C" int rd_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.comm) {
        RD_CHECK_NCCL(g.comm_destroy(g.comm));
        g.comm = nullptr;
        g.rank = -1;
        g.world = 0;
    }
    return RD_OK;
}
