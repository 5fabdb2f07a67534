// k_deliver.hip -- result delivery without a host wait for the handles whose calls leave Rx11bRow tables (802.11b, 802.11n, 40 MHz HT):
// what sora_rx_deliver_async does for the 802.11a handle.  The reference hands every frame to its MAC as it is decoded
// (kernel/bb/demod11/fb11b_demod.cpp, fb11n_demod.cpp:30-85: the RxThread loop looks at the error code and the frame buffer after every
// source call); with several calls in flight the equivalent is: every call's rows AND MPDUs reach the host, in order, behind the call's
// kernels, without the host blocking on the device.
//
//   k_dense_rows   one block: per capture (or per 40 MHz frame) the rows it filled -> dense sora_frame_result rows in (capture, time)
//                  order, with each row's MPDU placed at the running sum of the MPDU lengths before it (two block scans)
//   k_dense_mpdu   one wave per dense row: the MPDU bytes from the row's 4096-byte slot -> the dense MPDU block
// then three copies into the caller's page-locked buffers: {rows, MPDU bytes}, the row table, the MPDU block.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "kernels.h"
#include "../../include/sora_hip.h"

namespace sora {

__global__ void __launch_bounds__(1024) k_dense_rows(const Rx11bRow* __restrict__ rows, const uint32_t* __restrict__ nframes, const CapDesc* __restrict__ caps,
                                                     const sora_frame_result* __restrict__ tmpl, uint32_t ncaps_bound, uint32_t mf,
                                                     sora_frame_result* __restrict__ out, uint32_t* __restrict__ src_slot, uint32_t* __restrict__ meta, uint32_t mpdu_cap,
                                                     const uint32_t* __restrict__ ncaps_dev, const uint32_t* __restrict__ evbase)
{
    // evbase (the 40 MHz handle's raw-capture calls): "capture" c is an EVENT of the front end; evbase[c] = the row of rows[] its first row comes from, or 0xFFFFFFFF for
    // an event without a decoded frame (a PLCP header that failed): its one row is the template row as it stands (error code, position; no MPDU)
    // (the 40 MHz handle plans its frames on the device: the host knows only a bound)
    const uint32_t ncaps = ncaps_dev ? min(ncaps_bound, *ncaps_dev) : ncaps_bound;
    __shared__ uint32_t s_a[1024], s_b[1024];
    __shared__ uint32_t s_base[2];
    const uint32_t t = threadIdx.x;
    if (t < 2) s_base[t] = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < ncaps; c0 += 1024) {
        const uint32_t c = c0 + t;
        const uint32_t found = c < ncaps ? (nframes ? nframes[c] : mf) : 0u;
        const uint32_t n = min(found, mf);
        const uint32_t base = c < ncaps ? (evbase ? evbase[c] : c * mf) : 0u;
        uint32_t bytes = 0;
        for (uint32_t i = 0; i < n && base != 0xFFFFFFFFu; i++) {
            const Rx11bRow& r = rows[(size_t)base + i];
            if (r.error_code == E_FRAME_OK || r.error_code == E_CRC32_FAIL) bytes += min(r.length, 4096u);
        }
        s_a[t] = n; s_b[t] = bytes;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {                                // Hillis-Steele inclusive scans of both
            const uint32_t va = t >= o ? s_a[t - o] : 0u, vb = t >= o ? s_b[t - o] : 0u;
            __syncthreads();
            s_a[t] += va; s_b[t] += vb;
            __syncthreads();
        }
        uint32_t row = s_base[0] + s_a[t] - n, off = s_base[1] + s_b[t] - bytes;
        for (uint32_t i = 0; i < n; i++, row++) {
            sora_frame_result o;
            if (base == 0xFFFFFFFFu) {                                           // an event without a frame: the template row is the row
                o = tmpl[(size_t)c * mf + i]; o.length = 0; o.crc32 = 0; o.mpdu_offset = off;
                out[row] = o; src_slot[row] = 0xFFFFFFFFu;
                continue;
            }
            const Rx11bRow& r = rows[(size_t)base + i];
            uint16_t tflags = 0;
            // (40 MHz HT: capture_id = frame id, start_sample = spatial stream, rate, symbols; a raw-capture call's truncation flag)
            if (tmpl) { o = tmpl[(size_t)c * mf + i]; tflags = o.flags; }
            else { o.capture_id = caps[c].capture_id; o.start_sample = 0; o.nsym = 0; o.cfo_est = 0; o.rate_kbps = r.rate_kbps; o.end_sample = r.end_sample; }
            o.error_code = r.error_code; o.length = (uint16_t)r.length; o.crc32 = r.crc32;
            o.flags = (uint16_t)(tflags | ((i + 1 == mf && found > mf) ? SORA_ROW_TRUNCATED : 0));
            const bool has = r.error_code == E_FRAME_OK || r.error_code == E_CRC32_FAIL;
            const uint32_t len = has ? min(r.length, 4096u) : 0u;
            o.mpdu_offset = off;
            out[row] = o;
            // (an MPDU that does not fit is not copied; the host sees it from the total)
            src_slot[row] = (has && off + len <= mpdu_cap) ? (uint32_t)((size_t)base + i) : 0xFFFFFFFFu;
            off += len;
        }
        __syncthreads();
        if (t == 1023) { s_base[0] += s_a[1023]; s_base[1] += s_b[1023]; }
        __syncthreads();
    }
    if (t == 0) { meta[0] = s_base[0]; meta[1] = s_base[1]; }
}

__global__ void __launch_bounds__(256) k_dense_mpdu(const sora_frame_result* __restrict__ out, const uint32_t* __restrict__ src_slot, const uint32_t* __restrict__ meta,
                                                    const uint8_t* __restrict__ slots, uint8_t* __restrict__ dense)
{
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= meta[0]) return;
    const uint32_t s = src_slot[i];
    if (s == 0xFFFFFFFFu) return;
    const uint32_t len = min((uint32_t)out[i].length, 4096u), off = out[i].mpdu_offset;
    const uint8_t* p = slots + (size_t)s * 4096;
    for (uint32_t k = lane; k < len; k += 64) dense[off + k] = p[k];
}

}  // namespace sora

using namespace sora;

static int reserve(void** p, size_t* have, size_t need)
{
    if (*have >= need) return SORA_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    if (hipMalloc(p, need) != hipSuccess) { (void)hipGetLastError(); return SORA_ERR_HARDWARE_FAILED; }
    *have = need;
    return SORA_OK;
}

void sora_internal_dense_free(DenseStage* D)
{
    if (!D) return;
    (void)hipFree(D->d_rows); (void)hipFree(D->d_src); (void)hipFree(D->d_mpdu); (void)hipFree(D->d_meta); (void)hipFree(D->d_tmpl);
    *D = DenseStage();
}

int sora_internal_dense_deliver(DenseStage* D, const Rx11bRow* d_rows, const uint32_t* d_nframes, const CapDesc* d_caps, const sora_frame_result* h_tmpl,
                                uint32_t ncaps, uint32_t mf, const uint8_t* d_slots, hipStream_t st,
                                sora_frame_result* h_rows, size_t max_rows, uint32_t* h_meta, uint8_t* h_mpdu, size_t mpdu_cap,
                                const sora_frame_result* d_tmpl, const uint32_t* d_ncaps, const uint32_t* d_evbase)
{
    if (!D || !h_rows || !h_meta || (h_mpdu && mpdu_cap == 0)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "deliver_async: null argument", 0);
    const size_t cap_rows = (size_t)ncaps * mf;
    if (max_rows < cap_rows) return sora_internal_fail(SORA_ERR_CAPACITY, "deliver_async: h_rows must have room for max_captures x max_frames_per_capture rows of this call", 0);
    if (mpdu_cap >= (1ull << 32)) return sora_internal_fail(SORA_ERR_CAPACITY, "deliver_async: MPDU buffer of 4 GB or more", 0);
    if (ncaps == 0) { h_meta[0] = h_meta[1] = 0; return SORA_OK; }
    const size_t mcap = h_mpdu ? (mpdu_cap + 15) / 16 * 16 : 16;
    if (reserve((void**)&D->d_rows, &D->rows_bytes, sizeof(sora_frame_result) * cap_rows) || reserve((void**)&D->d_src, &D->src_bytes, 4 * cap_rows) ||
        reserve((void**)&D->d_mpdu, &D->mpdu_bytes, mcap) || reserve((void**)&D->d_meta, &D->meta_bytes, 16) ||
        (h_tmpl && reserve((void**)&D->d_tmpl, &D->tmpl_bytes, sizeof(sora_frame_result) * cap_rows)))
        return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "deliver_async: staging buffers", 0);
    hipError_t e = hipSuccess;
    if (h_tmpl) e = hipMemcpyAsync(D->d_tmpl, h_tmpl, sizeof(sora_frame_result) * cap_rows, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "deliver_async: row templates", (int)e);
    hipLaunchKernelGGL(k_dense_rows, dim3(1), dim3(1024), 0, st, d_rows, d_nframes, d_caps, (const sora_frame_result*)(d_tmpl ? d_tmpl : h_tmpl ? D->d_tmpl : nullptr), ncaps, mf,
                       D->d_rows, D->d_src, D->d_meta, (uint32_t)(h_mpdu ? mpdu_cap : 0), d_ncaps, d_evbase);
    if (h_mpdu) hipLaunchKernelGGL(k_dense_mpdu, dim3((unsigned)((cap_rows + 3) / 4)), dim3(256), 0, st, (const sora_frame_result*)D->d_rows, (const uint32_t*)D->d_src,
                                   (const uint32_t*)D->d_meta, d_slots, D->d_mpdu);
    e = hipMemcpyAsync(h_meta, D->d_meta, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h_rows, D->d_rows, sizeof(sora_frame_result) * cap_rows, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && h_mpdu) e = hipMemcpyAsync(h_mpdu, D->d_mpdu, mpdu_cap, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "deliver_async", (int)e);
    return SORA_OK;
}
