// api_dist.hip — multi-GPU at the C-ABI level (SURVEY.md 8e / 8b `pz_bcast_key`): one process per GPU, the batch block-sharded
// over ranks by the caller, the prepared evaluation key broadcast ONCE over RCCL (xGMI) on the module stream; no other
// collective exists on this path (independent ciphertexts, no reduction).
// RCCL is resolved at run time (dlopen) so that single-GPU users of libpoulpy_hip.so do not load it, and so that a process
// which already carries an RCCL (PyTorch ships one) shares that copy.
#include <dlfcn.h>

#include "api_common.hpp"

namespace {
typedef int rccl_result_t;                       // ncclResult_t, ncclSuccess = 0
typedef void* rccl_comm_t;                       // ncclComm_t
struct rccl_unique_id { char internal[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* so = nullptr;
    rccl_result_t (*get_unique_id)(rccl_unique_id*) = nullptr;
    rccl_result_t (*comm_init_rank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    rccl_result_t (*comm_destroy)(rccl_comm_t) = nullptr;
    rccl_result_t (*broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*get_error_string)(rccl_result_t) = nullptr;
};
Rccl& rccl() {
    static Rccl r;
    return r;
}
int rccl_load() {
    Rccl& r = rccl();
    if (r.broadcast) return PZ_OK;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (r.broadcast) return PZ_OK;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.so) break;
    }
    if (!r.so) return fail(PZ_ERR_RCCL, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.so, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.so, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.so, "ncclCommDestroy");
    r.get_error_string = (decltype(r.get_error_string))dlsym(r.so, "ncclGetErrorString");
    auto bc = (decltype(r.broadcast))dlsym(r.so, "ncclBroadcast");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !bc) return fail(PZ_ERR_RCCL, "RCCL symbols missing in librccl");
    r.broadcast = bc;
    return PZ_OK;
}
int rccl_fail(rccl_result_t e, const char* what) {
    Rccl& r = rccl();
    return fail(PZ_ERR_RCCL, "%s failed: %s", what, r.get_error_string ? r.get_error_string(e) : "RCCL error");
}
}  // namespace

extern "C" {

size_t pz_comm_unique_id_bytes(void) { return sizeof(rccl_unique_id); }

// the cheap probe of the non-root ranks (poulpy_amd/dist.py::broadcast_key_agreed): ncclGetUniqueId opens a bootstrap listener per call
int pz_comm_available(void) { return rccl_load(); }

int pz_comm_unique_id(void* out_id) {
    if (!out_id) return fail(PZ_ERR_INVALID, "null id");
    PZ_TRY(rccl_load());
    rccl_unique_id id;
    const rccl_result_t e = rccl().get_unique_id(&id);
    if (e != 0) return rccl_fail(e, "ncclGetUniqueId");
    memcpy(out_id, &id, sizeof(id));
    return PZ_OK;
}

int pz_comm_init_rank(pz_module* M, int world_size, int rank, const void* unique_id) {
    PZ_ENTER(M);
    PZ_REQUIRE(unique_id != nullptr && world_size >= 1 && rank >= 0 && rank < world_size, "pz_comm_init_rank: bad arguments");
    PZ_REQUIRE(M->comm == nullptr, "pz_comm_init_rank: the module already has a communicator");
    PZ_TRY(rccl_load());
    rccl_unique_id id;
    memcpy(&id, unique_id, sizeof(id));
    rccl_comm_t c = nullptr;
    const rccl_result_t e = rccl().comm_init_rank(&c, world_size, id, rank);
    if (e != 0) return rccl_fail(e, "ncclCommInitRank");
    M->comm = c; M->comm_world = world_size; M->comm_rank = rank;
    return PZ_OK;
}

int pz_comm_destroy(pz_module* M) {
    PZ_ENTER(M);
    if (!M->comm) return PZ_OK;
    PZ_HIP(hipStreamSynchronize(M->stream));
    const rccl_result_t e = rccl().comm_destroy((rccl_comm_t)M->comm);
    M->comm = nullptr;
    if (e != 0) return rccl_fail(e, "ncclCommDestroy");
    return PZ_OK;
}

int pz_comm_rank(const pz_module* M) { return M && M->comm ? M->comm_rank : -1; }
int pz_comm_world_size(const pz_module* M) { return M && M->comm ? M->comm_world : 0; }

// ncclBroadcast of a device buffer (a prepared key: VmpPMat / SvpPPol / a whole blind-rotation key) from `root` to every rank, in
// place, in buckets of at most 64 MiB on the module stream (xGMI is point to point: a bucket keeps every link of the ring busy
// while the next one is queued).  Ordered with the module's other work; pz_module_sync() to wait.
int pz_bcast_key(pz_module* M, void* dev_buf, size_t bytes, int root) {
    PZ_ENTER(M);
    PZ_REQUIRE(M->comm != nullptr, "pz_bcast_key: no communicator (pz_comm_init_rank)");
    PZ_REQUIRE(root >= 0 && root < M->comm_world, "pz_bcast_key: root out of range");
    PZ_REQUIRE(bytes == 0 || is_device_ptr(dev_buf), "pz_bcast_key takes a device pointer");
    const size_t bucket = (size_t)64 << 20;
    for (size_t off = 0; off < bytes; off += bucket) {
        const size_t len = std::min(bucket, bytes - off);
        char* p = (char*)dev_buf + off;
        const rccl_result_t e = rccl().broadcast(p, p, len, /*ncclInt8*/ 0, root, (rccl_comm_t)M->comm, M->stream);
        if (e != 0) return rccl_fail(e, "ncclBroadcast");
    }
    return PZ_OK;
}

}  // extern "C"
