// collective.hip — the data-parallel exchange behind the C ABI: asd_comm_* / asd_allreduce_mean_f32 (SURVEY.md section 8b's asd_allreduce_*).
//
// Replaces what Lightning's DDP does for the reference (launch.py:233-240: mean of every trainable gradient once per optimizer step) for
// hosts that do not go through torch.distributed: one RCCL communicator per process (one process per GPU), an in-place mean all-reduce of
// an fp32 buffer on the caller's stream.  RCCL itself is resolved at RUN TIME from the process image (PyTorch ships and loads its own
// librccl.so; linking a second copy into libasd_hip.so would put two RCCL runtimes into one process) — dlopen("librccl.so") otherwise.
// The gradient buckets of this path are few and large (50 MB hash table, one flat bucket for the MLPs: ring traffic over the seven xGMI links
// of a GPU is per-link bound), so a unit is one ncclAllReduce(ncclAvg).
#include <dlfcn.h>
#include <mutex>
#include <string.h>

#include "asd_common.h"

namespace {
// the handful of RCCL entry points used here, with their published signatures (rccl.h) reduced to plain types
typedef struct { char internal[128]; } rccl_unique_id;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, rccl_unique_id id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream);
typedef const char* (*fn_error_string)(int);
enum { RCCL_FLOAT32 = 7, RCCL_SUM = 0, RCCL_AVG = 4 };      // ncclFloat32 / ncclSum / ncclAvg of rccl.h

struct Rccl {
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    bool ok = false;
};
const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclAllReduce")) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
        r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
        r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
        r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
        r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
    });
    return r;
}
int rccl_fail(const char* what, int code) {
    const Rccl& r = rccl();
    asd_set_error("%s: RCCL error %d (%s)", what, code, r.error_string ? r.error_string(code) : "?");
    return ASD_ERR_LAUNCH;
}
}  // namespace

struct asd_comm {
    void* comm;
    int32_t rank, world;
};

extern "C" {

int asd_comm_unique_id(void* id128) {
    ASD_CHECK_ARG(id128, "null argument");
    const Rccl& r = rccl();
    if (!r.ok) { asd_set_error("asd_comm_unique_id: RCCL is not available in this process"); return ASD_ERR_UNSUPPORTED; }
    rccl_unique_id id;
    const int rc = r.get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, 128);
    return ASD_OK;
}

int asd_comm_create(const void* id128, int32_t rank, int32_t world, asd_comm** comm) {
    ASD_CHECK_ARG(id128 && comm && world >= 1 && rank >= 0 && rank < world, "bad argument");
    const Rccl& r = rccl();
    if (!r.ok) { asd_set_error("asd_comm_create: RCCL is not available in this process"); return ASD_ERR_UNSUPPORTED; }
    rccl_unique_id id;
    memcpy(id.internal, id128, 128);
    void* c = nullptr;
    const int rc = r.comm_init_rank(&c, world, id, rank);      // binds to the current HIP device
    if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
    *comm = new asd_comm{c, rank, world};
    return ASD_OK;
}

int asd_comm_destroy(asd_comm* comm) {
    if (!comm) return ASD_OK;
    const int rc = rccl().comm_destroy(comm->comm);
    delete comm;
    return rc == 0 ? ASD_OK : rccl_fail("ncclCommDestroy", rc);
}

// buf[0..n) <- mean over the ranks, in place, enqueued on `stream` (no host synchronisation)
int asd_allreduce_mean_f32(asd_comm* comm, float* buf, int64_t n, void* stream) {
    ASD_CHECK_ARG(comm && buf && n >= 0, "bad argument");
    if (n == 0) return ASD_OK;
    const int rc = rccl().all_reduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_AVG, comm->comm, (hipStream_t)stream);
    return rc == 0 ? ASD_OK : rccl_fail("ncclAllReduce", rc);
}

}  // extern "C"
