// r2l_allreduce.hip — the one collective of data-parallel R2L training for hosts WITHOUT torch.distributed: an in-place SUM
// all-reduce of the flat fp32 gradient buffer over RCCL (xGMI inside a node), enqueued on the caller's HIP stream.
// Replaces nn.DataParallel's gradient reduction (/root/reference/main.py:37-42, 472-479: ReduceAddCoalesced onto GPU 0 +
// next step's parameter broadcast); SURVEY.md §8(b) export row `r2l_allreduce_init / grad_allreduce / destroy`.
//
// RCCL is bound at run time (dlopen): libr2l_hip.so keeps no link-time dependency on it, a process that already loaded an
// RCCL (PyTorch bundles one under the same soname) shares that copy, and single-GPU users never load it at all.
// One communicator = one process = one GPU (the device current at r2l_allreduce_init).  The 128-byte unique id is made
// by rank 0 (r2l_allreduce_unique_id) and handed to the other ranks by the host (file, socket, MPI, env: its business).
#include "r2l_common.h"
#include "r2l_hip.h"
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

namespace {
// the slice of the NCCL API used here (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat = 7, ncclSum = 0)
struct UniqueId { char bytes[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat = 7, kSum = 0;

struct Api {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn get_error_string = nullptr;
};
Api g_api;

int load_api() {
    if (g_api.handle != nullptr) return 0;
    const char* names[] = {getenv("R2L_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        if (n == nullptr || *n == 0) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h != nullptr) break;
    }
    if (h == nullptr) {
        r2l_set_error_msg("r2l_allreduce: cannot dlopen librccl.so (set R2L_RCCL_PATH)");
        return R2L_ERR_RCCL_BASE;
    }
    Api a;
    a.handle = h;
    a.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    a.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    a.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    a.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    a.get_error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!a.get_unique_id || !a.comm_init_rank || !a.all_reduce || !a.comm_destroy) {
        r2l_set_error_msg("r2l_allreduce: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
        return R2L_ERR_RCCL_BASE;
    }
    g_api = a;
    return 0;
}

int fail(const char* what, int rc) {
    char msg[400];
    snprintf(msg, sizeof(msg), "%s -> RCCL error %d: %s", what, rc,
             g_api.get_error_string ? g_api.get_error_string(rc) : "?");
    r2l_set_error_msg(msg);
    return R2L_ERR_RCCL_BASE + rc;
}
}  // namespace

struct r2l_comm {
    Comm comm;
    int world, rank;
};

extern "C" int r2l_allreduce_unique_id(void* id_out128) {
    if (id_out128 == nullptr) { r2l_set_error_msg("r2l_allreduce_unique_id: null output"); return R2L_ERR_RCCL_BASE; }
    if (int rc = load_api()) return rc;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    if (int rc = g_api.get_unique_id(&id)) return fail("ncclGetUniqueId", rc);
    memcpy(id_out128, &id, sizeof(id));
    return 0;
}

extern "C" int r2l_allreduce_init(const void* id128, int world, int rank, r2l_comm** out) {
    if (id128 == nullptr || out == nullptr || world < 1 || rank < 0 || rank >= world) {
        r2l_set_error_msg("r2l_allreduce_init: bad arguments");
        return R2L_ERR_RCCL_BASE;
    }
    if (int rc = load_api()) return rc;
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    if (int rc = g_api.comm_init_rank(&c, world, id, rank)) return fail("ncclCommInitRank", rc);
    r2l_comm* h = new r2l_comm{c, world, rank};
    *out = h;
    return 0;
}

extern "C" int r2l_grad_allreduce(r2l_comm* c, float* buf, int64_t n, void* stream) {
    if (c == nullptr || (buf == nullptr && n > 0) || n < 0) {
        r2l_set_error_msg("r2l_grad_allreduce: bad arguments");
        return R2L_ERR_RCCL_BASE;
    }
    if (n == 0) return 0;
    if (int rc = g_api.all_reduce(buf, buf, (size_t)n, kFloat, kSum, c->comm, (hipStream_t)stream))
        return fail("ncclAllReduce", rc);
    return 0;
}

extern "C" int r2l_allreduce_destroy(r2l_comm* c) {
    if (c == nullptr) return 0;
    const int rc = g_api.comm_destroy ? g_api.comm_destroy(c->comm) : 0;
    delete c;
    return rc ? fail("ncclCommDestroy", rc) : 0;
}
