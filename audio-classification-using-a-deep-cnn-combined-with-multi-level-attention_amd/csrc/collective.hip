// collective.hip -- the data-parallel exchange of the training step behind the C ABI (SURVEY.md section 8b/8e:
// the reference itself is single-process, train.py:119-142; north star: "RCCL gradient all-reduce over xGMI").
//
// The library does NOT link RCCL: it binds the RCCL that is already in the process (PyTorch ships one and loads it with
// libtorch_hip.so; a C host links ROCm's) with dlopen(RTLD_NOLOAD) / dlsym at first use, so there is never a second set
// of nccl* symbols or a second RCCL runtime beside the host's. The communicator is created by the host through
// mla_comm_init_rank (ncclCommInitRank with a 128-byte id the host distributes by its own means: torch.distributed's
// store, MPI, a file) or handed in directly: an ncclComm_t obtained elsewhere is accepted as `comm` unchanged.
#include <dlfcn.h>

#include <cstring>

#include <mutex>

#include "common.h"

namespace {

// the slice of rccl.h this file needs (values as in /opt/rocm/include/rccl/rccl.h of ROCm 7.2)
struct UniqueId { char internal[128]; };
using Comm = void*;
enum { kSum = 0, kInt32 = 2, kFloat32 = 7, kFloat64 = 8 };

struct Api {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* origin = "";
};

Api g_api;
std::once_flag g_once;

void bind() {
    // 1. an RCCL the process has already loaded (PyTorch's librccl.so, SONAME librccl.so.1; or the host's own);
    // 2. otherwise ROCm's, loaded privately (a C host that did not link RCCL itself).
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        if ((g_api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) { g_api.origin = "already loaded by the host"; break; }
    }
    if (!g_api.handle) {
        for (const char* n : names) {
            if ((g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) { g_api.origin = "loaded by libmla_hip"; break; }
        }
    }
    if (!g_api.handle) return;
    g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(dlsym(g_api.handle, "ncclGetUniqueId"));
    g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(dlsym(g_api.handle, "ncclCommInitRank"));
    g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(dlsym(g_api.handle, "ncclCommDestroy"));
    g_api.CommCount = reinterpret_cast<decltype(g_api.CommCount)>(dlsym(g_api.handle, "ncclCommCount"));
    g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(dlsym(g_api.handle, "ncclAllReduce"));
    g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(dlsym(g_api.handle, "ncclGetErrorString"));
}

int api_ready() {
    std::call_once(g_once, bind);
    if (!g_api.handle) return mla::fail(MLA_E_LAUNCH, "no RCCL in the process and librccl.so.1 cannot be loaded: %s", dlerror());
    if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.CommDestroy || !g_api.AllReduce)
        return mla::fail(MLA_E_LAUNCH, "the RCCL in the process lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
    return MLA_OK;
}

int nccl_fail(const char* what, int rc) {
    return mla::fail(MLA_E_LAUNCH, "%s: RCCL error %d (%s)", what, rc, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
}

}  // namespace

extern "C" int mla_comm_unique_id(void* id_host_128) {
    MLA_REQUIRE(id_host_128, MLA_E_ARG, "null id buffer");
    if (int rc = api_ready()) return rc;
    UniqueId id;
    if (int rc = g_api.GetUniqueId(&id)) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id_host_128, id.internal, sizeof(id.internal));
    return MLA_OK;
}

extern "C" int mla_comm_init_rank(void** comm_out, int nranks, const void* id_host_128, int rank) {
    MLA_REQUIRE(comm_out && id_host_128 && nranks >= 1 && rank >= 0 && rank < nranks, MLA_E_ARG,
                "bad communicator arguments (nranks %d, rank %d)", nranks, rank);
    if (int rc = api_ready()) return rc;
    UniqueId id;
    memcpy(id.internal, id_host_128, sizeof(id.internal));
    Comm comm = nullptr;
    if (int rc = g_api.CommInitRank(&comm, nranks, id, rank)) return nccl_fail("ncclCommInitRank", rc);
    *comm_out = comm;
    return MLA_OK;
}

extern "C" int mla_comm_destroy(void* comm) {
    if (!comm) return MLA_OK;
    if (int rc = api_ready()) return rc;
    if (int rc = g_api.CommDestroy(comm)) return nccl_fail("ncclCommDestroy", rc);
    return MLA_OK;
}

extern "C" int mla_comm_count(void* comm, int* count) {
    MLA_REQUIRE(comm && count, MLA_E_ARG, "bad comm_count arguments");
    if (int rc = api_ready()) return rc;
    MLA_REQUIRE(g_api.CommCount, MLA_E_LAUNCH, "the RCCL in the process lacks ncclCommCount");
    if (int rc = g_api.CommCount(comm, count)) return nccl_fail("ncclCommCount", rc);
    return MLA_OK;
}

extern "C" const char* mla_comm_library_origin(void) {
    std::call_once(g_once, bind);
    return g_api.handle ? g_api.origin : "none";
}

extern "C" int mla_allreduce_flat(void* buf, int64_t count, int dtype, void* comm, mla_stream_t stream) {
    MLA_REQUIRE(buf && comm && count >= 0, MLA_E_ARG, "bad all-reduce arguments");
    const int dt = dtype == MLA_F32 ? kFloat32 : dtype == MLA_F64 ? kFloat64 : dtype == MLA_I32 ? kInt32 : -1;
    MLA_REQUIRE(dt >= 0, MLA_E_DTYPE, "all-reduce dtype %d (MLA_F32, MLA_F64 or MLA_I32)", dtype);
    if (count == 0) return MLA_OK;
    if (int rc = api_ready()) return rc;
    if (int rc = g_api.AllReduce(buf, buf, size_t(count), dt, kSum, comm, static_cast<hipStream_t>(stream)))
        return nccl_fail("ncclAllReduce", rc);
    return MLA_OK;
}
