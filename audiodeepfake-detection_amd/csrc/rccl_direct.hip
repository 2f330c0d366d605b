// The collectives of the data-parallel step issued straight to RCCL on the caller's stream.
//
// Reference: the NCCL calls behind DistributedDataParallel and nn.SyncBatchNorm (src/audiofakedetect/
// train_classifier.py:319-323, models.py:260-289; SURVEY section 2.2: one gradient all-reduce and sixteen <= 2 KB
// BatchNorm-sum exchanges per step).  Through torch.distributed every one of them is handed to the process group's own
// stream and back (two event waits, ~55 us of GPU idle each: profiles/r04_ddp1_collectives.json); here
// ncclAllReduce runs on the stream the kernels run on.  librccl is taken from the process (PyTorch has loaded it) with
// dlopen -- the library has no link-time dependency on it -- and the communicator is created from a unique id that
// the host layer broadcasts through the existing process group.
// OPT-IN (AFD_RCCL_DIRECT=1 in the Python layer): no box with more than one GPU has run it yet.
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <dlfcn.h>

#include <cstring>

namespace {

struct UniqueId {
    char internal[128];
};
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    Comm comm = nullptr;
    int world = 0;
};
Rccl& rccl() {
    static Rccl r;
    return r;
}

bool load_rccl() {
    Rccl& r = rccl();
    if (r.all_reduce) return true;
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy the process already runs (PyTorch's)
        if (!r.handle) r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return false;
    r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.handle, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.handle, "ncclCommInitRank"));
    r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(r.handle, "ncclAllReduce"));
    r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.handle, "ncclCommDestroy"));
    r.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
        r.all_reduce = nullptr;
        return false;
    }
    return true;
}

int rccl_fail(const char* what, int rc) {
    const Rccl& r = rccl();
    return afd::fail(AFD_ERR_HIP, "rccl %s: %s", what, r.error_string ? r.error_string(rc) : "error");
}

}  // namespace

extern "C" int afd_rccl_unique_id(void* id128) {
    if (!id128) return afd::fail(AFD_ERR_ARG, "rccl: null pointer");
    if (!load_rccl()) return afd::fail(AFD_ERR_UNSUPPORTED, "rccl: librccl.so is not loadable in this process");
    UniqueId id{};
    const int rc = rccl().get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return AFD_OK;
}

extern "C" int afd_rccl_init(const void* id128, int rank, int world) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return afd::fail(AFD_ERR_ARG, "rccl: rank %d of %d", rank, world);
    if (!load_rccl()) return afd::fail(AFD_ERR_UNSUPPORTED, "rccl: librccl.so is not loadable in this process");
    Rccl& r = rccl();
    if (r.comm) return afd::fail(AFD_ERR_ARG, "rccl: communicator already initialised");
    UniqueId id{};
    memcpy(id.internal, id128, sizeof(id.internal));
    const int rc = r.comm_init_rank(&r.comm, world, id, rank);  // collective: every rank calls it, on its own device
    if (rc != 0) {
        r.comm = nullptr;
        return rccl_fail("ncclCommInitRank", rc);
    }
    r.world = world;
    return AFD_OK;
}

extern "C" int afd_rccl_all_reduce_sum(void* buf, long count, int is_double, afd_stream_t stream) {
    Rccl& r = rccl();
    if (!r.comm) return afd::fail(AFD_ERR_ARG, "rccl: no communicator (afd_rccl_init first)");
    if (!buf || count < 1) return afd::fail(AFD_ERR_ARG, "rccl all-reduce: bad buffer");
    // ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0 (rccl.h); in place, on the caller's stream
    const int rc = r.all_reduce(buf, buf, (size_t)count, is_double ? 8 : 7, 0, r.comm, static_cast<hipStream_t>(stream));
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    return AFD_OK;
}

extern "C" int afd_rccl_world(void) { return rccl().comm ? rccl().world : 0; }

extern "C" int afd_rccl_destroy(void) {
    Rccl& r = rccl();
    if (r.comm) {
        const int rc = r.comm_destroy(r.comm);
        r.comm = nullptr;
        r.world = 0;
        if (rc != 0) return rccl_fail("ncclCommDestroy", rc);
    }
    return AFD_OK;
}
