// Collective face of the C ABI: the ONE exchange step of the path -- an in-place SUM all-reduce of the flat
// fp32 gradient buffer -- for hosts that do not bring a communicator of their own (a C++ trainer).  RCCL over xGMI.
// The library is bound at run time (dlopen), preferring a copy that is already loaded in the process (a PyTorch host has
// its own librccl: two copies would fight over the nccl* symbols), so libmmvae_hip.so itself has no link dependency on it
// and single-GPU users never load it.  The Python engine keeps using torch.distributed.
#include "common.h"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);        // a copy the host process already uses
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.h, "ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.h, "ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.h, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.h, "ncclGetErrorString"));
    });
    return (r.h && r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy) ? &r : nullptr;
}

int fail(const char* what, ncclResult_t rc) {
    Rccl* r = rccl();
    mmvae_set_error("%s: %s", what, r && r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
    return MMVAE_EHIP;
}

}  // namespace

struct mmvae_comm { ncclComm_t comm; int rank, world; };

extern "C" {

int mmvae_comm_unique_id(void* out128) {
    MMVAE_REQUIRE(out128 != nullptr, "mmvae_comm_unique_id: null argument");
    Rccl* r = rccl();
    MMVAE_REQUIRE(r != nullptr, "mmvae_comm: librccl.so not found (dlopen)");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
    memcpy(out128, &id, sizeof(id));
    return MMVAE_OK;
}

int mmvae_comm_init(mmvae_comm** out, int rank, int world, const void* unique_id128) {
    MMVAE_REQUIRE(out && unique_id128 && world >= 1 && rank >= 0 && rank < world, "mmvae_comm_init: bad argument");
    Rccl* r = rccl();
    MMVAE_REQUIRE(r != nullptr, "mmvae_comm: librccl.so not found (dlopen)");
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t rc = r->CommInitRank(&c, world, id, rank);     // uses the calling thread's current HIP device
    if (rc != ncclSuccess) return fail("ncclCommInitRank", rc);
    *out = new mmvae_comm{c, rank, world};
    return MMVAE_OK;
}

int mmvae_allreduce_grads(mmvae_comm* c, float* flat, size_t n, void* stream) {
    MMVAE_REQUIRE(c && flat, "mmvae_allreduce_grads: null argument");
    if (n == 0) return MMVAE_OK;
    const ncclResult_t rc = rccl()->AllReduce(flat, flat, n, ncclFloat, ncclSum, c->comm, static_cast<hipStream_t>(stream));
    if (rc != ncclSuccess) return fail("ncclAllReduce", rc);
    return MMVAE_OK;
}

int mmvae_comm_world(const mmvae_comm* c) { return c ? c->world : 0; }

int mmvae_comm_destroy(mmvae_comm* c) {
    if (!c) return MMVAE_OK;
    const ncclResult_t rc = rccl()->CommDestroy(c->comm);
    delete c;
    if (rc != ncclSuccess) return fail("ncclCommDestroy", rc);
    return MMVAE_OK;
}

}  // extern "C"
