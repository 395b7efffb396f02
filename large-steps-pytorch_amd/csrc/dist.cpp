// dist.cpp -- the one collective of the sharded direct solver behind the C ABI (SURVEY.md section 8b "ls_dist_*"): a sum over the
// ranks of the updates that the subtrees hand to the replicated top of the elimination tree, issued by RCCL (xGMI) on the solve's
// own stream between the two halves of ls_direct_solve_part -- no host round trip, no Python between the launches.
// The reference has no multi-GPU path (largesteps/solvers.py:26-39 is one process, one device); this serves the north star's
// "meshes shard by vertex blocks across the GPUs of one node ... RCCL over xGMI".
// RCCL is looked up at run time (dlopen; the copy PyTorch already loaded if there is one): the library has no link-time dependency
// on it, and a process that never calls ls_dist_* never touches it.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // the copy already in the process
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
    });
    return r;
}

int rccl_fail(ncclResult_t e, const char* what) {
    Rccl& r = rccl();
    ls::set_error("%s: RCCL error %d (%s)", what, (int)e, r.GetErrorString ? r.GetErrorString(e) : "?");
    return 2000 + (int)e;            // > 0: a captured runtime error code (HIP errors are < 2000)
}

}  // namespace

struct ls_dist {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int ls_dist_unique_id(void* h_id128) {
    LS_REQUIRE(h_id128, LS_E_INVALID, "ls_dist_unique_id: null argument");
    Rccl& r = rccl();
    LS_REQUIRE(r.ok, LS_E_STATE, "ls_dist_unique_id: librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail(e, "ls_dist_unique_id");
    memcpy(h_id128, &id, sizeof(id));
    return LS_OK;
}

extern "C" int ls_dist_create(const void* h_id128, int rank, int world, int device, ls_dist** out) {
    LS_REQUIRE(h_id128 && out && world >= 1 && rank >= 0 && rank < world, LS_E_INVALID, "ls_dist_create: bad argument");
    *out = nullptr;
    Rccl& r = rccl();
    LS_REQUIRE(r.ok, LS_E_STATE, "ls_dist_create: librccl.so could not be loaded");
    ls::DeviceGuard g(device);
    LS_HIP(g.err);
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    ls_dist* c = new ls_dist();
    c->rank = rank; c->world = world; c->device = device;
    const ncclResult_t e = r.CommInitRank(&c->comm, world, id, rank);       // collective: every rank of the communicator calls it
    if (e != ncclSuccess) { delete c; return rccl_fail(e, "ls_dist_create (ncclCommInitRank)"); }
    *out = c;
    return LS_OK;
}

extern "C" int ls_dist_destroy(ls_dist* c) {
    if (!c) return LS_OK;
    ls::DeviceGuard g(c->device);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return LS_OK;
}

extern "C" int ls_dist_info(const ls_dist* c, int* h_rank, int* h_world) {
    LS_REQUIRE(c && c->comm, LS_E_INVALID, "ls_dist_info: bad argument");
    Rccl& r = rccl();
    LS_REQUIRE(r.CommCount && r.CommUserRank, LS_E_STATE, "ls_dist_info: this librccl.so has no ncclCommCount / ncclCommUserRank");
    int rank = -1, world = -1;
    ncclResult_t e = r.CommUserRank(c->comm, &rank);
    if (e == ncclSuccess) e = r.CommCount(c->comm, &world);
    if (e != ncclSuccess) return rccl_fail(e, "ls_dist_info");
    if (h_rank) *h_rank = rank;
    if (h_world) *h_world = world;
    return LS_OK;
}

extern "C" int ls_dist_allreduce_sum(ls_dist* c, float* d_buf, int64_t n, void* stream) {
    LS_REQUIRE(c && (d_buf || n == 0) && n >= 0, LS_E_INVALID, "ls_dist_allreduce_sum: bad argument");
    if (n == 0) return LS_OK;
    ls::DeviceGuard g(c->device);
    LS_HIP(g.err);
    const ncclResult_t e = rccl().AllReduce(d_buf, d_buf, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
    if (e != ncclSuccess) return rccl_fail(e, "ls_dist_allreduce_sum");
    return LS_OK;
}

// One sharded solve as ONE native call: this rank's subtrees upwards, the all-reduce of the exchange region IN PLACE in the
// handle's slot array (a few hundred KB; every entry has exactly one non-zero contributor, so the sum is exact and independent of
// the reduction order), then the replicated levels and this rank's subtrees downwards -- all on `stream`.
extern "C" int ls_dist_direct_solve(ls_dist* c, ls_direct* d, const float* b, float* x, int k, void* stream) {
    LS_REQUIRE(c && d && b && x, LS_E_INVALID, "ls_dist_direct_solve: bad argument");
    float* region = nullptr;
    int64_t floats = 0;
    int rc = ls_direct_exchange_region(d, k, &region, &floats);
    if (rc != LS_OK) return rc;
    rc = ls_direct_solve_part(d, b, x, k, 0, region, stream);              // exchange == the handle's own region: no staging copy
    if (rc != LS_OK) return rc;
    rc = ls_dist_allreduce_sum(c, region, floats, stream);
    if (rc != LS_OK) return rc;
    return ls_direct_solve_part(d, b, x, k, 1, region, stream);
}
