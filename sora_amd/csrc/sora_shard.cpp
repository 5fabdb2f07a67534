// sora_shard.cpp -- multi-GPU sharding for a C host (include/sora_hip.h "sora_shard_*").
//
// Captures are the natural shard: the reference resets its context per frame (kernel/bb/demod11/fb11ademod_config.hpp:68-95) and its
// harness walks one dump at a time (fb11a_demod.cpp:29-81), so rank r of W runs the whole receive path on a contiguous block of
// captures in its own HBM and there is NO collective on the data path.  The only exchange is the one a host that wants a single result
// table needs: one ncclAllGather of the fixed-size result rows (36 bytes each, padded to max_rows_per_rank per rank), one of the
// per-rank row counts, and an ncclAllReduce(sum) of the counters -- RCCL over xGMI, a few KB per rank, so the per-link bound of the
// ring (7 links x ~153 GB/s) never matters.  One process per GPU; the 128-byte unique id travels by whatever the host has (a file,
// a socket, MPI): sora_shard_unique_id on rank 0, sora_shard_create on every rank.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first sora_shard_* call) so that single-GPU users of libsora_hip.so do
// not load it; a process that already holds an RCCL (PyTorch's) shares that copy.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "../../include/sora_hip.h"

namespace {
// the slice of rccl.h this file uses (ABI-stable since NCCL 2.x: opaque comm, 128-byte id, C enums)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt32 = 2, ncclUint32 = 3, ncclUint64 = 5 };
enum { ncclSum = 0 };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl; std::once_flag g_rccl_once;

const Rccl* rccl()
{
    std::call_once(g_rccl_once, [] {
        void* h = nullptr;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return;
        Rccl r; r.lib = h;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
        r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce) g_rccl = r;
    });
    return g_rccl.lib ? &g_rccl : nullptr;
}
int rccl_fail(const char* what, int rc) { return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, what, rc); }
}  // namespace

struct sora_shard {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    uint32_t* d_count = nullptr;           // this rank's row count, staged for the count gather
};

#define SHARD_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, #call, (int)_e); } while (0)

extern "C" {

int sora_shard_unique_id(uint8_t id[SORA_SHARD_ID_BYTES])
{
    if (!id) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_unique_id: null argument", 0);
    const Rccl* R = rccl();
    if (!R) return sora_internal_fail(SORA_ERR_NO_DEVICE, "sora_shard: librccl.so.1 could not be loaded", 0);
    ncclUniqueId u; const int rc = R->GetUniqueId(&u);
    if (rc != ncclSuccess) return rccl_fail("ncclGetUniqueId", rc);
    static_assert(sizeof(u) == SORA_SHARD_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, sizeof(u));
    return SORA_OK;
}

int sora_shard_create(const uint8_t id[SORA_SHARD_ID_BYTES], int world_size, int rank, int device, sora_shard_t** out)
{
    if (!id || !out || world_size <= 0 || rank < 0 || rank >= world_size) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_create: bad argument", 0);
    const Rccl* R = rccl();
    if (!R) return sora_internal_fail(SORA_ERR_NO_DEVICE, "sora_shard: librccl.so.1 could not be loaded", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (device < 0 || device >= ndev) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_create: device ordinal out of range", 0);
    SHARD_HIP(hipSetDevice(device));
    sora_shard* sh = new sora_shard(); sh->world = world_size; sh->rank = rank; sh->device = device;
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    const int rc = R->CommInitRank(&sh->comm, world_size, u, rank);
    if (rc != ncclSuccess) { delete sh; return rccl_fail("ncclCommInitRank", rc); }
    if (hipMalloc((void**)&sh->d_count, sizeof(uint32_t)) != hipSuccess) { R->CommDestroy(sh->comm); delete sh; return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_shard_create: hipMalloc", 0); }
    *out = sh;
    return SORA_OK;
}

void sora_shard_destroy(sora_shard_t* sh)
{
    if (!sh) return;
    (void)hipSetDevice(sh->device);
    if (sh->comm) { const Rccl* R = rccl(); if (R) R->CommDestroy(sh->comm); }
    (void)hipFree(sh->d_count);
    delete sh;
}

int sora_shard_world(const sora_shard_t* sh, int* world_size, int* rank)
{
    if (!sh) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_world: null handle", 0);
    if (world_size) *world_size = sh->world;
    if (rank) *rank = sh->rank;
    return SORA_OK;
}

void sora_shard_partition(size_t n_items, int world_size, int rank, size_t* first, size_t* count)
{
    if (world_size <= 0 || rank < 0 || rank >= world_size) { if (first) *first = 0; if (count) *count = 0; return; }
    const size_t base = n_items / (size_t)world_size, extra = n_items % (size_t)world_size, r = (size_t)rank;
    if (first) *first = r * base + (r < extra ? r : extra);
    if (count) *count = base + (r < extra ? 1 : 0);
}

int sora_shard_gather_rows(sora_shard_t* sh, const sora_frame_result* d_rows, const uint32_t* d_nrows, size_t max_rows_per_rank,
                           sora_frame_result* d_all_rows, uint32_t* d_all_counts, void* stream)
{
    if (!sh || !d_rows || !d_nrows || !d_all_rows || !d_all_counts || max_rows_per_rank == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_gather_rows: bad argument", 0);
    const Rccl* R = rccl();
    SHARD_HIP(hipSetDevice(sh->device));
    hipStream_t st = (hipStream_t)stream;
    static_assert(sizeof(sora_frame_result) == 36, "row layout");
    int rc = R->AllGather(d_nrows, d_all_counts, 1, ncclUint32, sh->comm, st);
    if (rc != ncclSuccess) return rccl_fail("ncclAllGather(counts)", rc);
    rc = R->AllGather(d_rows, d_all_rows, max_rows_per_rank * (sizeof(sora_frame_result) / 4), ncclInt32, sh->comm, st);
    if (rc != ncclSuccess) return rccl_fail("ncclAllGather(rows)", rc);
    return SORA_OK;
}

int sora_shard_reduce_counters(sora_shard_t* sh, uint64_t* d_counters, size_t n, void* stream)
{
    if (!sh || !d_counters || n == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_reduce_counters: bad argument", 0);
    const Rccl* R = rccl();
    SHARD_HIP(hipSetDevice(sh->device));
    const int rc = R->AllReduce(d_counters, d_counters, n, ncclUint64, ncclSum, sh->comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return rccl_fail("ncclAllReduce(counters)", rc);
    return SORA_OK;
}

int sora_shard_gather_results(sora_shard_t* sh, sora_rx_t* rx, int ticket, size_t max_rows_per_rank,
                              sora_frame_result* h_all_rows, uint32_t* h_counts, size_t* n_total)
{
    if (!sh || !rx || !h_all_rows || !h_counts || !n_total || max_rows_per_rank == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_gather_results: bad argument", 0);
    *n_total = 0;
    const sora_frame_result* d_rows = nullptr; const uint32_t* d_nrows = nullptr;
    int rc = ticket > 0 ? sora_rx_results_dev_of(rx, ticket, &d_rows, &d_nrows, nullptr) : sora_rx_results_dev(rx, &d_rows, &d_nrows, nullptr);
    if (rc != SORA_OK) return rc;
    hipStream_t st = (hipStream_t)(ticket > 0 ? sora_rx_stream_of(rx, ticket) : sora_rx_stream(rx));
    SHARD_HIP(hipSetDevice(sh->device));
    const size_t W = (size_t)sh->world;
    sora_frame_result* d_mine = nullptr; sora_frame_result* d_all = nullptr; uint32_t* d_counts = nullptr;
    hipError_t e = hipMalloc((void**)&d_mine, sizeof(sora_frame_result) * max_rows_per_rank);
    if (e == hipSuccess) e = hipMalloc((void**)&d_all, sizeof(sora_frame_result) * max_rows_per_rank * W);
    if (e == hipSuccess) e = hipMalloc((void**)&d_counts, sizeof(uint32_t) * W);
    // this rank's rows, padded to the common block size (the library's row table may be shorter or longer than max_rows_per_rank)
    uint32_t mine = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&mine, d_nrows, sizeof(mine), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && mine > max_rows_per_rank) { (void)hipFree(d_mine); (void)hipFree(d_all); (void)hipFree(d_counts); return sora_internal_fail(SORA_ERR_CAPACITY, "sora_shard_gather_results: this rank has more rows than max_rows_per_rank", 0); }
    if (e == hipSuccess) e = hipMemsetAsync(d_mine, 0, sizeof(sora_frame_result) * max_rows_per_rank, st);
    if (e == hipSuccess && mine) e = hipMemcpyAsync(d_mine, d_rows, sizeof(sora_frame_result) * mine, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) {
        rc = sora_shard_gather_rows(sh, d_mine, d_nrows, max_rows_per_rank, d_all, d_counts, st);
        if (rc == SORA_OK) {
            e = hipMemcpyAsync(h_counts, d_counts, sizeof(uint32_t) * W, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            size_t n = 0;
            for (size_t r = 0; r < W && e == hipSuccess; r++) {            // compact: rank order, then (capture, time) order inside a rank
                const uint32_t c = h_counts[r] < max_rows_per_rank ? h_counts[r] : (uint32_t)max_rows_per_rank;
                if (c) e = hipMemcpy(h_all_rows + n, d_all + r * max_rows_per_rank, sizeof(sora_frame_result) * c, hipMemcpyDeviceToHost);
                n += c;
            }
            *n_total = n;
        }
    }
    (void)hipFree(d_mine); (void)hipFree(d_all); (void)hipFree(d_counts);
    if (rc != SORA_OK) return rc;
    if (e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_shard_gather_results", (int)e);
    return SORA_OK;
}

}  // extern "C"
