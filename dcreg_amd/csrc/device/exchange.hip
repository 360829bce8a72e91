// The path's one real exchange step (point sharding of ONE scan pair): every rank linearises its slice of the
// source, then the 32-double rows (21 H + 6 g + 4 sums + flag) of all ranks are all-gathered and added in rank order, so
// every rank holds bitwise the same totals and takes the same host step - no broadcast.  Here the exchange is native:
// ncclAllGather (RCCL over xGMI) on the ctx's stream, inside the C++ engine loop; a 256-byte message is latency-bound, not
// link-bound.  RCCL is loaded lazily with dlopen so that libdcreg_hip.so does not drag the collective library into
// single-GPU processes.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>

#include <cstring>
#include <mutex>
#include <string>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../../include/dcreg.h"
#include "context.hpp"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;        // what went wrong while loading (the loader's message of the first failed dlopen, captured at once: dlerror()
                            // clears itself when read and is overwritten by the next attempt)
    bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllGather && GetErrorString; }
};

// A process may hold two HIP runtimes (PyTorch ships its own libamdhip64 next to its own librccl; /opt/rocm has another
// pair).  Device pointers and streams of one runtime mean nothing to the other, so the RCCL to use is the one that sits next
// to the HIP runtime THIS library is bound to - found by asking the loader where hipMalloc lives.
RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;         // (dcreg_icp_run_many runs one host thread per ctx: the first users may arrive together)
    std::call_once(once, [] {
        std::string dir;
        Dl_info info;
        if (dladdr((void *)static_cast<hipError_t (*)(void **, size_t)>(&hipMalloc), &info) && info.dli_fname) {
            dir = info.dli_fname;
            const size_t slash = dir.rfind('/');
            dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
        }
        const char *names[] = {"librccl.so.1", "librccl.so"};
        auto attempt = [&](const std::string &path) {
            if (api.handle) return;
            api.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!api.handle) { const char *e = dlerror(); if (api.why.empty()) api.why = e ? e : "dlopen failed"; }
        };
        if (!dir.empty()) for (const char *n : names) attempt(dir + n);
        for (const char *n : names) attempt(n);
        if (!api.handle) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        if (!api.ok()) api.why = "librccl was loaded but lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllGather / GetErrorString";
    });
    return api;
}

// ncclCommInitRank prints a version banner on stdout; stdout belongs to the caller (bench.py prints exactly one JSON line there), so
// file descriptor 1 points at stderr while it runs.  Descriptors are process-wide: one initialisation at a time.
std::mutex g_init_mutex;

}  // namespace

extern "C" {

int dcreg_comm_unique_id(void *id128) {
    if (!id128) return DCREG_E_INVALID;
    RcclApi &A = rccl();
    if (!A.ok()) return DCREG_E_DEVICE;
    ncclUniqueId id;
    if (A.GetUniqueId(&id) != ncclSuccess) return DCREG_E_DEVICE;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return DCREG_OK;
}

int dcreg_comm_destroy(dcreg_ctx *c) {
    if (!c) return DCREG_E_INVALID;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        RcclApi &A = rccl();
        if (A.ok()) (void)A.CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    if (c->d_xrow) { (void)hipFree(c->d_xrow); c->d_xrow = nullptr; }
    if (c->d_xall) { (void)hipFree(c->d_xall); c->d_xall = nullptr; }
    if (c->h_xrow) { (void)hipHostFree(c->h_xrow); c->h_xrow = nullptr; }
    if (c->h_xall) { (void)hipHostFree(c->h_xall); c->h_xall = nullptr; }
    c->comm_world = 0; c->comm_rank = 0;
    return DCREG_OK;
}

int dcreg_comm_init(dcreg_ctx *c, const void *id128, int rank, int world) {
    if (!c) return DCREG_E_INVALID;
    if (!id128 || world < 1 || rank < 0 || rank >= world) { c->fail("invalid communicator arguments"); return DCREG_E_INVALID; }
    RcclApi &A = rccl();
    if (!A.ok()) { c->fail("RCCL is not available (%s)", A.why.empty() ? "librccl.so.1 not found" : A.why.c_str()); return DCREG_E_DEVICE; }
    (void)dcreg_comm_destroy(c);
    if (hipSetDevice(c->device) != hipSuccess) { c->fail("hipSetDevice failed"); return DCREG_E_DEVICE; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r;
    {
        std::lock_guard<std::mutex> lock(g_init_mutex);
        std::fflush(stdout);
        const int saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
        r = A.CommInitRank(&comm, world, id, rank);
        std::fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); (void)close(saved); }
    }
    if (r != ncclSuccess) { c->fail("ncclCommInitRank failed: %s", A.GetErrorString(r)); return DCREG_E_DEVICE; }
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    const size_t row = 32 * sizeof(double);
    if (hipMalloc((void **)&c->d_xrow, row) != hipSuccess || hipMalloc((void **)&c->d_xall, row * (size_t)world) != hipSuccess ||
        hipHostMalloc((void **)&c->h_xrow, row, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&c->h_xall, row * (size_t)world, hipHostMallocDefault) != hipSuccess) {
        (void)dcreg_comm_destroy(c);
        c->fail("allocating the exchange buffers failed");
        return DCREG_E_NOMEM;
    }
    return DCREG_OK;
}

// row[32] of this rank -> sum over all ranks, added in rank order (identical on every rank)
int dcreg_comm_allgather_sum(dcreg_ctx *c, double row[32]) {
    if (!c || !row) return DCREG_E_INVALID;
    if (!c->comm) { c->fail("no communicator: call dcreg_comm_init first"); return DCREG_E_STATE; }
    RcclApi &A = rccl();
    const size_t bytes = 32 * sizeof(double);
    std::memcpy(c->h_xrow, row, bytes);
    hipError_t e = hipMemcpyAsync(c->d_xrow, c->h_xrow, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const ncclResult_t r = A.AllGather(c->d_xrow, c->d_xall, 32, ncclDouble, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess) { c->fail("ncclAllGather failed: %s", A.GetErrorString(r)); return DCREG_E_DEVICE; }
        e = hipMemcpyAsync(c->h_xall, c->d_xall, bytes * (size_t)c->comm_world, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->fail("exchange failed: %s", hipGetErrorString(e)); return DCREG_E_DEVICE; }
    for (int k = 0; k < 32; ++k) row[k] = 0.0;
    for (int r = 0; r < c->comm_world; ++r)              // fixed association order
        for (int k = 0; k < 32; ++k) row[k] += c->h_xall[(size_t)r * 32 + k];
    return DCREG_OK;
}

// count doubles of this rank -> recv[world * count], rank-major, on every rank: ONE ncclAllGather on the ctx's stream (the Monte-Carlo
// experiment's statistics gather, engine.cpp dcreg_montecarlo_job).  Without a communicator: a job of one rank (recv = send).
int dcreg_comm_allgather(dcreg_ctx *c, const double *send, double *recv, int64_t count) {
    if (!c || !send || !recv || count < 0) return DCREG_E_INVALID;
    if (!c->comm || c->comm_world <= 1) { std::memcpy(recv, send, sizeof(double) * (size_t)count); return DCREG_OK; }
    if (count == 0) return DCREG_OK;
    RcclApi &A = rccl();
    if (hipSetDevice(c->device) != hipSuccess) { c->fail("hipSetDevice failed"); return DCREG_E_DEVICE; }
    const size_t bytes = sizeof(double) * (size_t)count, all = bytes * (size_t)c->comm_world;
    double *d_send = nullptr, *d_recv = nullptr;
    if (hipMalloc((void **)&d_send, bytes) != hipSuccess || hipMalloc((void **)&d_recv, all) != hipSuccess) {
        if (d_send) (void)hipFree(d_send);
        c->fail("allocating %zu bytes for the gather failed", bytes + all);
        return DCREG_E_NOMEM;
    }
    int rc = DCREG_OK;
    hipError_t e = hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const ncclResult_t r = A.AllGather(d_send, d_recv, (size_t)count, ncclDouble, (ncclComm_t)c->comm, c->stream);
        if (r != ncclSuccess) { c->fail("ncclAllGather failed: %s", A.GetErrorString(r)); rc = DCREG_E_DEVICE; }
        else e = hipMemcpyAsync(recv, d_recv, all, hipMemcpyDeviceToHost, c->stream);
    }
    if (rc == DCREG_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (rc == DCREG_OK && e != hipSuccess) { c->fail("gather failed: %s", hipGetErrorString(e)); rc = DCREG_E_DEVICE; }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_send); (void)hipFree(d_recv);
    return rc;
}
int dcreg_comm_info(const dcreg_ctx *c, int *rank, int *world) {
    if (!c) return DCREG_E_INVALID;
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    if (world) *world = c->comm ? c->comm_world : 1;
    return DCREG_OK;
}

}  // extern "C"
