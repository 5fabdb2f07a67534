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
#include <vector>
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "../../include/sora_hip.h"

namespace sora {
// The MPDUs of a call's dense rows, packed densely for the exchange: offsets = running sum of the lengths of the rows that carry an MPDU
// (FRAME_OK / CRC32_FAIL); the rows' mpdu_offset (slot0 * 32 in the call's own MPDU array) is rewritten to the dense offset.
__global__ void __launch_bounds__(1024) k_shard_mpdu_offsets(sora_frame_result* rows, uint32_t n, uint32_t* src_off, uint32_t* total)
{
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_base;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + t;
        const bool has = i < n && (rows[i].error_code == E_FRAME_OK || rows[i].error_code == E_CRC32_FAIL);
        const uint32_t len = has ? rows[i].length : 0u;
        s_scan[t] = len;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) { const uint32_t v = t >= o ? s_scan[t - o] : 0u; __syncthreads(); s_scan[t] += v; __syncthreads(); }
        if (i < n) { src_off[i] = rows[i].mpdu_offset; rows[i].mpdu_offset = s_base + s_scan[t] - len; }
        __syncthreads();
        if (t == 1023) s_base += s_scan[1023];
        __syncthreads();
    }
    if (t == 0) *total = s_base;
}
__global__ void __launch_bounds__(256) k_shard_mpdu_pack(const sora_frame_result* rows, uint32_t n, const uint32_t* src_off, const uint8_t* mpdu, uint8_t* dense, uint32_t cap)
{
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const sora_frame_result r = rows[i];
    if (r.error_code != E_FRAME_OK && r.error_code != E_CRC32_FAIL) return;
    if ((uint64_t)r.mpdu_offset + r.length > cap) return;                       // (the host reports the overflow from the total)
    for (uint32_t k = lane; k < r.length; k += 64) dense[r.mpdu_offset + k] = mpdu[src_off[i] + k];
}
}  // namespace sora

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
    // staging of sora_shard_gather_results (grow-only): this rank's padded row block, every rank's, {rows, MPDU bytes} per rank (+ this rank's),
    // this rank's dense MPDU block, every rank's, the rows' dense MPDU offsets
    sora_frame_result* d_mine = nullptr; size_t mine_bytes = 0;
    sora_frame_result* d_all = nullptr; size_t all_bytes = 0;
    uint32_t* d_pair = nullptr; size_t pair_bytes = 0;
    uint8_t* d_mpdu_mine = nullptr; size_t mpdu_mine_bytes = 0;
    uint8_t* d_mpdu_all = nullptr; size_t mpdu_all_bytes = 0;
    uint32_t* d_off = nullptr; size_t off_bytes = 0;
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
    *out = sh;
    return SORA_OK;
}

void sora_shard_destroy(sora_shard_t* sh)
{
    if (!sh) return;
    (void)hipSetDevice(sh->device);
    if (sh->comm) { const Rccl* R = rccl(); if (R) R->CommDestroy(sh->comm); }
    (void)hipFree(sh->d_mine); (void)hipFree(sh->d_all); (void)hipFree(sh->d_pair); (void)hipFree(sh->d_mpdu_mine); (void)hipFree(sh->d_mpdu_all); (void)hipFree(sh->d_off);
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
    if (!sh || !d_rows || !d_nrows || !d_all_rows || !d_all_counts || max_rows_per_rank == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM,
            "sora_shard_gather_rows: bad argument", 0);
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

// ---- the result exchange of one call, rows AND MPDUs (every MPDU reaches the one host buffer, fb11a_demod.cpp:64-70).
// Three collectives, which EVERY rank enters whatever happened to it locally (a rank that returned early would leave the others hanging in
// ncclAllGather): {row count, MPDU bytes} per rank, the row blocks, the dense MPDU blocks.  A rank with a local error contributes the count
// 0xFFFFFFFF; after the exchange every rank returns an error.  The staging buffers live in the handle (grow-only).
static int shard_reserve(sora_shard* sh, void** p, size_t* have, size_t need)
{
    if (*have >= need) return SORA_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    if (hipMalloc(p, need) != hipSuccess) { (void)hipGetLastError(); return SORA_ERR_HARDWARE_FAILED; }
    *have = need;
    (void)sh;
    return SORA_OK;
}

int sora_shard_gather_results_mpdu(sora_shard_t* sh, sora_rx_t* rx, int ticket, size_t max_rows_per_rank, sora_frame_result* h_all_rows, uint32_t* h_counts,
                                   size_t* n_total, size_t max_mpdu_bytes_per_rank, uint8_t* h_all_mpdu, size_t* mpdu_total)
{
    if (!sh || !rx || !h_all_rows || !h_counts || !n_total || max_rows_per_rank == 0 || (h_all_mpdu && (max_mpdu_bytes_per_rank == 0 || !mpdu_total)))
        return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_shard_gather_results: bad argument", 0);
    *n_total = 0; if (mpdu_total) *mpdu_total = 0;
    const Rccl* R = rccl();
    if (!R) return sora_internal_fail(SORA_ERR_NO_DEVICE, "sora_shard: librccl.so.1 could not be loaded", 0);
    SHARD_HIP(hipSetDevice(sh->device));
    const size_t W = (size_t)sh->world;
    const size_t mpdu_block = h_all_mpdu ? (max_mpdu_bytes_per_rank + 15) / 16 * 16 : 0;
    // staging (a failure here is the one local error that cannot take part in the exchange: the buffers are what the collectives read)
    if (shard_reserve(sh, (void**)&sh->d_mine, &sh->mine_bytes, sizeof(sora_frame_result) * max_rows_per_rank) != SORA_OK ||
        shard_reserve(sh, (void**)&sh->d_all, &sh->all_bytes, sizeof(sora_frame_result) * max_rows_per_rank * W) != SORA_OK ||
        shard_reserve(sh, (void**)&sh->d_pair, &sh->pair_bytes, 8 * (W + 1)) != SORA_OK ||
        (mpdu_block && (shard_reserve(sh, (void**)&sh->d_mpdu_mine, &sh->mpdu_mine_bytes, mpdu_block) != SORA_OK ||
                        shard_reserve(sh, (void**)&sh->d_mpdu_all, &sh->mpdu_all_bytes, mpdu_block * W) != SORA_OK ||
                        shard_reserve(sh, (void**)&sh->d_off, &sh->off_bytes, 4 * max_rows_per_rank) != SORA_OK)))
        return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_shard_gather_results: staging buffers (no exchange was started)", 0);
    // ---- this rank's contribution; any failure from here on is carried into the exchange as the count 0xFFFFFFFF
    int lrc = SORA_OK; const char* lwhat = "";
    const sora_frame_result* d_rows = nullptr; const uint32_t* d_nrows = nullptr; const uint8_t* d_mpdu = nullptr;
    hipStream_t st = nullptr;
    uint32_t mine[2] = { 0, 0 };
    if (sora_internal_rx_device(rx) != sh->device) { lrc = SORA_ERR_INVALID_PARAM;
        lwhat = "sora_shard_gather_results: the receive handle lives on another device than the shard handle"; }
    if (lrc == SORA_OK) {
        lrc = ticket > 0 ? sora_rx_results_dev_of(rx, ticket, &d_rows, &d_nrows, &d_mpdu) : sora_rx_results_dev(rx, &d_rows, &d_nrows, &d_mpdu);
        if (lrc != SORA_OK) lwhat = "sora_shard_gather_results: the call's results are not available (stale ticket?)";
    }
    if (lrc == SORA_OK) {
        st = (hipStream_t)(ticket > 0 ? sora_rx_stream_of(rx, ticket) : sora_rx_stream(rx));
        hipError_t e = hipMemcpyAsync(&mine[0], d_nrows, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { lrc = SORA_ERR_HARDWARE_FAILED; lwhat = "sora_shard_gather_results: reading the row count"; }
        else if (mine[0] > max_rows_per_rank) { lrc = SORA_ERR_CAPACITY; lwhat = "sora_shard_gather_results: this rank has more rows than max_rows_per_rank"; }
    }
    if (lrc == SORA_OK) {
        hipError_t e = hipMemsetAsync(sh->d_mine, 0, sizeof(sora_frame_result) * max_rows_per_rank, st);
        if (e == hipSuccess && mine[0]) e = hipMemcpyAsync(sh->d_mine, d_rows, sizeof(sora_frame_result) * mine[0], hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && mpdu_block && mine[0]) {
            // dense MPDU block of this rank: row i's bytes at the running sum of the lengths of the rows before it; the rows' mpdu_offset is rewritten to it
            hipLaunchKernelGGL(sora::k_shard_mpdu_offsets, dim3(1), dim3(1024), 0, st, sh->d_mine, mine[0], sh->d_off, sh->d_pair + 2 * W);
            // the CALLER's limit: the 16-byte rounding is only the staging / AllGather granule
            hipLaunchKernelGGL(sora::k_shard_mpdu_pack, dim3((mine[0] + 3) / 4), dim3(256), 0, st, sh->d_mine, mine[0], (const uint32_t*)sh->d_off, d_mpdu,
                    sh->d_mpdu_mine, (uint32_t)max_mpdu_bytes_per_rank);
            if (e == hipSuccess) e = hipMemcpyAsync(&mine[1], sh->d_pair + 2 * W, 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess && mine[1] > max_mpdu_bytes_per_rank) { lrc = SORA_ERR_CAPACITY;
                lwhat = "sora_shard_gather_results: this rank's MPDUs exceed max_mpdu_bytes_per_rank"; }
        }
        if (e != hipSuccess) { lrc = SORA_ERR_HARDWARE_FAILED; lwhat = "sora_shard_gather_results: staging this rank's rows"; }
    }
    if (!st) st = nullptr;                                                       // (the null stream carries the exchange of a rank without a valid call)
    if (lrc != SORA_OK) { mine[0] = 0xFFFFFFFFu; mine[1] = 0; }
    // ---- the exchange: every rank, always
    hipError_t e = hipMemcpyAsync(sh->d_pair + 2 * W, mine, 8, hipMemcpyHostToDevice, st);
    int rc = ncclSuccess;
    if (e == hipSuccess) rc = R->AllGather(sh->d_pair + 2 * W, sh->d_pair, 2, ncclUint32, sh->comm, st);
    if (e == hipSuccess && rc == ncclSuccess) rc = R->AllGather(sh->d_mine, sh->d_all, max_rows_per_rank * (sizeof(sora_frame_result) / 4), ncclInt32, sh->comm, st);
    if (e == hipSuccess && rc == ncclSuccess && mpdu_block) rc = R->AllGather(sh->d_mpdu_mine, sh->d_mpdu_all, mpdu_block / 4, ncclUint32, sh->comm, st);
    std::vector<uint32_t> pairs(2 * W);
    if (e == hipSuccess && rc == ncclSuccess) e = hipMemcpyAsync(pairs.data(), sh->d_pair, 8 * W, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (rc != ncclSuccess) return rccl_fail("sora_shard_gather_results: ncclAllGather", rc);
    if (e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_shard_gather_results: exchange", (int)e);
    if (lrc != SORA_OK) return sora_internal_fail(lrc, lwhat, 0);
    for (size_t r = 0; r < W; r++)
        if (pairs[2 * r] == 0xFFFFFFFFu) return sora_internal_fail(SORA_ERR_FAILED, "sora_shard_gather_results: another rank reported an error (nothing was gathered)", (int)r);
    // ---- compact into the caller's tables: rank order, then (capture, time) order inside a rank; mpdu_offset indexes h_all_mpdu
    size_t n = 0, moff = 0;
    for (size_t r = 0; r < W; r++) {
        const uint32_t c = pairs[2 * r], mb = pairs[2 * r + 1];
        h_counts[r] = c;
        if (c) SHARD_HIP(hipMemcpy(h_all_rows + n, sh->d_all + r * max_rows_per_rank, sizeof(sora_frame_result) * c, hipMemcpyDeviceToHost));
        if (mpdu_block) {
            if (mb) SHARD_HIP(hipMemcpy(h_all_mpdu + moff, sh->d_mpdu_all + r * mpdu_block, mb, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < c; i++) h_all_rows[n + i].mpdu_offset += (uint32_t)moff;
            moff += mb;
        }
        n += c;
    }
    *n_total = n; if (mpdu_total) *mpdu_total = moff;
    return SORA_OK;
}

int sora_shard_gather_results(sora_shard_t* sh, sora_rx_t* rx, int ticket, size_t max_rows_per_rank,
                              sora_frame_result* h_all_rows, uint32_t* h_counts, size_t* n_total)
{
    return sora_shard_gather_results_mpdu(sh, rx, ticket, max_rows_per_rank, h_all_rows, h_counts, n_total, 0, nullptr, nullptr);
}

}  // extern "C"
