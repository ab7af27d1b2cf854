// comm.hip -- the exchange step of the sharded search and index build on RCCL, inside the library.
// one all-gather of per-shard top-k records over xGMI (plus, for the global threshold, one of the shards'
// k largest approximate scores; for the index build one of the cluster sums).  The Python driver may run these through
// torch.distributed; a host without torch -- the Julia shim, a C++ driver -- uses the entry points below: one
// communicator per process and GPU, the unique id handed from rank 0 to the others by whatever the host has (a file,
// MPI, a socket).  librccl is opened at run time (dlopen): the search library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.hpp"

using namespace clb;

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        // a librccl the process has already loaded (e.g. torch's) is reused; otherwise the ROCm one
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names)
            if ((a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
        if (!a.handle)
            for (const char* n : names)
                if ((a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!a.handle) { a.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
#define CLB_SYM(F)                                                                       \
    a.F = reinterpret_cast<decltype(a.F)>(dlsym(a.handle, "nccl" #F));                   \
    if (!a.F) { a.error = "librccl lacks nccl" #F; return; }
        CLB_SYM(GetUniqueId) CLB_SYM(CommInitRank) CLB_SYM(CommDestroy) CLB_SYM(AllGather) CLB_SYM(AllReduce)
        CLB_SYM(GetErrorString)
#undef CLB_SYM
    });
    return a;
}

int rccl_ready() {
    RcclApi& a = api();
    if (!a.error.empty()) return fail(CLB_EHIP, "RCCL unavailable: %s", a.error.c_str());
    return CLB_OK;
}

#define CLB_NCCL(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) return fail(CLB_EHIP, "%s failed: %s", #expr, api().GetErrorString(_r));    \
    } while (0)

}  // namespace

struct clb_comm {
    int device = 0, rank = 0, n_ranks = 1;
    ncclComm_t comm = nullptr;
};

extern "C" {

int64_t clb_comm_unique_id_bytes(void) { return (int64_t)sizeof(ncclUniqueId); }

int clb_comm_unique_id(void* id, int64_t bytes) {
    if (!id || bytes < (int64_t)sizeof(ncclUniqueId)) return fail(CLB_EARGUMENT, "id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    CLB_TRY(rccl_ready());
    ncclUniqueId u;
    CLB_NCCL(api().GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return CLB_OK;
}

int clb_comm_create(int device, int rank, int n_ranks, const void* id, int64_t bytes, clb_comm** out) {
    if (!out) return fail(CLB_EARGUMENT, "out is null");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(CLB_EARGUMENT, "rank %d of %d", rank, n_ranks);
    if (!id || bytes < (int64_t)sizeof(ncclUniqueId)) return fail(CLB_EARGUMENT, "id must be the %zu bytes of clb_comm_unique_id", sizeof(ncclUniqueId));
    CLB_TRY(use_device(device));
    CLB_TRY(rccl_ready());
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    clb_comm* c = new clb_comm();
    c->device = device; c->rank = rank; c->n_ranks = n_ranks;
    ncclResult_t r = api().CommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(CLB_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, n_ranks, api().GetErrorString(r));
    }
    *out = c;
    return CLB_OK;
}

int clb_comm_destroy(clb_comm* c) {
    if (!c) return CLB_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)api().CommDestroy(c->comm);
    delete c;
    return CLB_OK;
}

int clb_comm_rank(const clb_comm* c) { return c ? c->rank : -1; }
int clb_comm_size(const clb_comm* c) { return c ? c->n_ranks : 0; }

int clb_comm_all_gather(clb_comm* c, const void* d_send, void* d_recv, int64_t bytes_per_rank, void* hip_stream) {
    if (!c || !d_send || !d_recv) return fail(CLB_EARGUMENT, "null argument");
    if (bytes_per_rank < 0) return fail(CLB_EARGUMENT, "bytes_per_rank < 0");
    CLB_TRY(use_device(c->device));
    CLB_NCCL(api().AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclUint8, c->comm, (hipStream_t)hip_stream));
    return CLB_OK;
}

int clb_comm_all_reduce_max_f32(clb_comm* c, float* d_buf, int64_t n, void* hip_stream) {
    if (!c || !d_buf) return fail(CLB_EARGUMENT, "null argument");
    if (n < 0) return fail(CLB_EARGUMENT, "n < 0");
    CLB_TRY(use_device(c->device));
    CLB_NCCL(api().AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclMax, c->comm, (hipStream_t)hip_stream));
    return CLB_OK;
}

}  // extern "C"
