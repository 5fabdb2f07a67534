// kernels.h -- argument blocks and declarations of the receive-path kernels (shared by host and device code).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "dev_arith.h"
#include "rx_types.h"
#include "../../include/sora_hip.h"

// Experiment and probe switches (SORA_EXP_*, SORA_DBG_*) belong to the TOOLS variant of the library (sora_amd.build.build_variant adds -DSORA_TOOLS for them): the product
// build refuses them, so no measurement scaffolding can reach it by accident.
#if !defined(SORA_TOOLS) && (defined(SORA_EXP_NORING) || defined(SORA_EXP_LB) || defined(SORA_EXP_STAGGER) || defined(SORA_EXP_VIT_PRIO) || defined(SORA_DBG_TRACK_TH) || \
                             defined(SORA_DBG_NO_TRACE) || defined(SORA_DBG_KFRAME_PRIVATE) || defined(SORA_DBG_NO_SYMBOLS) || defined(SORA_DBG_NO_TRELLIS) || defined(SORA_SCAN_PROBE) || \
                             defined(SORA_EXP_FIN))
#error "SORA_EXP_* / SORA_DBG_* switches need -DSORA_TOOLS (sora_amd.build.build_variant)"
#endif

namespace sora {

struct ScanArgs {
    const uint32_t* iq;         // packed COMPLEX16
    const CapDesc*  caps;
    uint32_t        ncaps;
    uint32_t        str;        // 2: 40 MHz input (keep even samples), 1: 20 MHz input
    uint32_t        keep_queue; // 1: the 44 MHz graph (TDownSample44_40 has no Reset/Flush: its queued samples survive a frame reset)
    uint32_t        thr;        // cca_pwr_threshold
    uint32_t        max_frames; // per capture
    Tables          T;
    FrameRow*       frames;     // [ncaps*max_frames]
    FrameCtx*       fctx;
    uint32_t*       nframes;    // [ncaps]
    uint32_t*       njobs;      // [3] frames whose data symbols must be decoded, per code rate (1/2, 2/3, 3/4)
    uint32_t*       joblist;    // [3][nrows] their frame-table rows, compacted: consecutive workgroups of the per-frame
                                //            kernels then carry live work (workgroup b runs on XCD b % 8)
    uint32_t        nrows;      // list stride = ncaps * max_frames
    uint32_t*       slot_row;   // [total slots] frame-table row that owns a symbol slot (the data symbols of every queued frame); the host presets 0xFFFFFFFF
    // stream continuation (sora_rx_set_stream_mode): capture k of this call continues capture k of the call before it
    // [ncaps][kContWords] the carrier-sense state at the capture's last resume point (read at entry when valid, rewritten at every later one); null = off
    uint32_t*       cont;
    uint32_t*       consumed;   // [ncaps] input-rate samples of this capture that are final: the host submits the stream from there on next time
    // what the NEXT call needs cleared, done here instead of by a fill kernel in front of every call (sora_hip.cpp: a pipeline's calls alternate between two sets of
    // job counters; workgroup 0 zeroes the set this call does not use, and k_pipe's hand-off words): null = the host's fill has done it
    uint32_t*       zero_a; uint32_t nzero_a;
    uint32_t*       zero_b; uint32_t nzero_b;
    uint32_t        own_slots;  // 1: every capture's workgroup presets its own symbol slots' owners (0xFFFFFFFF) itself
};
constexpr int kContWords = 64;

struct RxArgs {
    const uint32_t* iq;
    const CapDesc*  caps;
    uint32_t        str;
    uint32_t        total_slots;
    uint32_t        nrows;          // ncaps*max_frames
    Tables          T;
    FrameRow*       frames;
    const FrameCtx* fctx;
    uint8_t*        soft;           // [slots*108] the frames' packed soft streams (three bits per value, rx_types.h); unused by the fused decode kernel
    uint8_t*        vout;           // [slots*32]
    uint8_t*        mpdu;           // [slots*32]
    VitJob*         jobs;           // [3][nrows] (indexed by job); unused by the fused decode kernel
    const uint32_t* njobs;
    const uint32_t* joblist;
    const uint32_t* slot_row;       // [total_slots] owner row of a symbol slot, 0xFFFFFFFF = none (k_scan)
    uint32_t*       eq;             // [total_slots][64] equalised bins, packed COMPLEX16 (k_sym_front -> k_track, k_sym_back)
    TrackRec*       track;          // [total_slots] rotation parameters of a data symbol (k_track -> k_sym_back)
    uint32_t*       pil;            // [total_slots][4] the four pilot bins (43, 57, 7, 21) of eq[] once more, densely: all k_track reads
    const uint32_t* pipe_flags;     // k_finish behind k_pipe: word 0 != 0 = a hand-off inside that launch gave up (else null)
    // sora_rx_bind_mpdu: the caller's page-locked MPDU array (the geometry of mpdu[]): the frame sink writes every MPDU there as well, over PCIe, as it finishes the frame (else null)
    uint8_t*        mpdu_host;
};

// k_pipe (k_rx.hip): the data field of a handful of frames as ONE launch.  Workgroups [0, nfront) are k_sym_front's, [nfront, nfront + ntrack) one frame's tracker and
// everything behind it each, the rest four waves of the window-parallel trellis.  flags (zeroed before every call): [0] a wait gave up; [4 + 4 f + h] quads of symbols
// published by helper wave h of frame row f; [4 + 4 nrows + b] front workgroup b is done.
struct PipeArgs {
    uint32_t  nfront, ntrack;
    uint32_t* flags;
    uint32_t  target, vstride;     // the window-parallel trellis's unit target and its vectors' stride per code-rate list
    uint16_t* vecs;
    uint32_t  stamp_base;          // (tools variant, SORA_DBG_PIPE_TIMELINE: where the launch's time stamps go, in words from flags)
    uint32_t  lanes64;             // 1: the trellis role decodes its units two per wave in the 64-lane layout (a lone capture: a third faster per unit), 0: eight per wave
    uint32_t  wait_ticks;          // bound of every wait inside the launch, in ticks of the 100 MHz counter (sora_rx_set_pipe_wait_us)
};

__global__ void k_scan(ScanArgs A);
__global__ void k_frame(RxArgs A);
__global__ void k_sym_front(RxArgs A);
__global__ void k_track(RxArgs A);
__global__ void k_track_lds(RxArgs A);
__global__ void k_sym_back(RxArgs A);
__global__ void k_pipe(RxArgs A, PipeArgs P);
__global__ void k_decode(RxArgs A);
__global__ void k_viterbi(const VitJob* jobs, const uint32_t* njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* soft, uint8_t* out);
__global__ void k_viterbi11n(const VitJob* jobs, const uint32_t* njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* soft, uint8_t* out);
// k_vit16.hip
__global__ void k_viterbi16(const VitJob* jobs, const uint32_t* njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* soft, uint8_t* out);
__global__ void k_viterbi16_11n(const VitJob* jobs, const uint32_t* njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* soft, uint8_t* out);
// k_vitwin.hip: the window-parallel trellis.  hdr = the call's counter block (njobs per code rate in its first three words); jstride = capacity of a list of jobs;
// target = units the call is cut into at least, frames permitting; vstride = vectors per code-rate list
__global__ void k_viterbi16w(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint8_t* soft, uint8_t* out, uint16_t* vecs);
#ifdef SORA_EXP_FIN
__global__ void k_viterbi16w_fin(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint8_t* soft, uint8_t* out, uint16_t* vecs,
                                 uint32_t* wdone, RxArgs A);
#endif
__global__ void k_viterbi16w_11n(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint8_t* soft, uint8_t* out, uint16_t* vecs);
__global__ void k_win_redo_11n(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint16_t* vecs,
        const uint8_t* soft, uint8_t* out, unsigned long long* stats);
__global__ void k_win_redo(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint16_t* vecs,
        const uint8_t* soft, uint8_t* out, unsigned long long* stats);
__global__ void k_win_redo_finish(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint16_t* vecs,
        const uint8_t* soft, uint8_t* out, unsigned long long* stats,
                                  RxArgs A);   // ... and k_finish behind it, in the same waves   // k_rx.hip
// ... behind k_pipe: the same, and the plain chain's code for the whole call if a hand-off inside k_pipe gave up (host_note: a host-mapped word that is set then)
__global__ void k_win_redo_finish_pipe(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint16_t* vecs,
        const uint8_t* soft, uint8_t* out, unsigned long long* stats, RxArgs A, uint32_t* host_note);
__global__ void k_finish(RxArgs A);
struct PackedRow;
__global__ void k_pack(const FrameRow* frames, const uint32_t* nframes, const CapDesc* caps, uint32_t ncaps, uint32_t max_frames, PackedRow* rows, uint32_t* nrows_out);
__global__ void k_fft64_batch(const uint32_t* in, uint32_t* out, uint32_t n, Tables T);
template <int NB> __global__ void k_demap_batch(const uint32_t* in, uint8_t* soft, uint32_t n, Tables T);
template <int NB> __global__ void k_deint_batch(const uint8_t* in, uint8_t* out, uint32_t n, Tables T);
__global__ void k_lts_batch(const uint32_t* in, uint32_t* ctx, uint32_t n, Tables T);
__global__ void k_symfront_batch(const uint32_t* in, const uint32_t* ctx, const uint32_t* ctx_index, uint32_t* eq, uint32_t n, Tables T);
template <bool PHASE> __global__ void k_ptrack_batch(const uint32_t* eq, const uint32_t* first, const uint32_t* nsym, uint32_t* state, uint32_t* out, uint32_t nframes, Tables T);
template <int KIND> __global__ void k_cmul64_batch(const uint32_t* in, const uint32_t* coef, uint32_t cstride, uint32_t coff, const uint32_t* cindex, uint32_t* out, uint32_t n);
__global__ void k_fft128_batch(const uint32_t* in, uint32_t* out, uint32_t n, Tables T);
struct TxArgs {                // sora_hip_tx11a (k_tx.hip)
    const uint8_t*  mpdu;      // MPDUs without FCS, frame f at mpdu + off[f]
    const uint32_t* off;
    const uint32_t* len;       // bytes without FCS (LENGTH = len + 4)
    const uint32_t* rate;      // kbps
    const uint8_t*  seed;      // scrambler register before the first byte (the harness uses 0xFF)
    int8_t*         out8;      // COMPLEX8 stream
    const uint64_t* out_off;   // first sample of frame f
    const int8_t*   preamble;  // 640 samples
    Tables          T;
};
__global__ void k_tx_preamble(int8_t* out8, Tables T);
__global__ void k_tx11a(TxArgs A);
__global__ void k_ingest(const uint8_t* raw, uint32_t* out, uint64_t m0, uint64_t n_out, unsigned flags);
__global__ void k_ingest_tile(const uint8_t* raw, uint32_t* out, unsigned flags, uint32_t tiles);
__global__ void k_soft_pack3(const uint8_t* soft8, const uint32_t* off8, const uint32_t* nsoft, const uint16_t* flen, const uint32_t* out_off,
                             int code_rate, uint8_t* packed, VitJob* jobs);

// ---- 802.11b receive graph (k_rx11b.hip)
struct Rx11bRow { uint32_t end_sample, error_code, rate_kbps, length, crc32; };
struct Rx11bArgs {
    const uint32_t* iq;         // packed COMPLEX16 @44 MHz
    const CapDesc*  caps;
    uint32_t        ncaps;
    uint32_t        thr;        // cca_pwr_threshold
    uint32_t        max_frames; // rows per capture
    Rx11bRow*       rows;       // [ncaps*max_frames]
    uint32_t*       nframes;    // [ncaps]
    uint8_t*        mpdu;       // [ncaps*max_frames][4096]
    const uint32_t* crc;        // CRC-32 table
    // [ncaps]: set by the first pass for a capture in which a header announces 5.5 / 11 Mbps; such captures are redone by k_rx11b_cck
    uint32_t*       needs_cck;
};
__global__ void k_rx11b(Rx11bArgs A);
__global__ void k_rx11b_cck(Rx11bArgs A);

// one event of the 40 MHz HT front end (k_scan_ht40 in k_rx11n.hip), row cap * max_frames + i
struct Ht40Found {
    uint32_t a20;                  // 20 MHz index (in the capture) of the first HT-STF sample: HT-LTF 1 starts 2 * a20 + 160 samples @40 MHz into the capture
    uint32_t mcs, ht_len, nsym;    // 0 unless the frame was recorded
    int32_t  cfo;                  // phase step per 20 MHz sample, 65536 = 2 pi (TFreqOffsetEst_11n's convention)
    float    noise_var;            // per carrier of the FFT<128> of the 40 MHz stream, LSB^2 (from the L-LTF pair)
    uint32_t end_sample;           // 40 MHz source position at which the event is seen
    uint32_t error_code;           // 0: recorded (the data field decides), else E_PLCP
};


// ---- completion order for the handles whose calls sit in a small array of slots (sora_rx11b_*, sora_ht40_*): the rules of sora_rx_wait_any
// (include/sora_hip.h).  A slot type has { hipStream_t stream; int ticket; hipEvent_t ev_done; bool delivered, released; }.
// the slot of the next call: unused, else the released call with the oldest ticket, else the oldest call
template <typename Slot> inline int slots_next(const Slot* s, int n)
{
    int best = 0, best_rel = -1;
    for (int i = 0; i < n; i++) {
        if (s[i].ticket == 0) return i;
        if (s[i].released && (best_rel < 0 || s[i].ticket < s[best_rel].ticket)) best_rel = i;
        if (s[i].ticket < s[best].ticket) best = i;
    }
    return best_rel >= 0 ? best_rel : best;
}
template <typename Slot> inline hipError_t slots_mark_delivered(Slot& s)       // behind the last copy of a delivery
{
    if (!s.ev_done) { const hipError_t e = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming); if (e != hipSuccess) return e; }
    const hipError_t e = hipEventRecord(s.ev_done, s.stream);
    if (e == hipSuccess) s.delivered = true;
    return e;
}
// -> the finished delivered slot with the oldest ticket (nullptr: none yet); *pending = some delivered call has not been released
template <typename Slot> inline Slot* slots_poll(Slot* s, int n, bool* pending, hipError_t* err)
{
    Slot* done = nullptr; *pending = false; *err = hipSuccess;
    for (int i = 0; i < n; i++) {
        if (s[i].ticket == 0 || !s[i].delivered || s[i].released) continue;
        *pending = true;
        const hipError_t q = hipEventQuery(s[i].ev_done);
        if (q == hipSuccess) { if (!done || s[i].ticket < done->ticket) done = &s[i]; }
        else if (q != hipErrorNotReady) { (void)hipGetLastError(); *err = q; return nullptr; }
    }
    if (!done) (void)hipGetLastError();                                         // (hipErrorNotReady is sticky for hipGetLastError)
    return done;
}

}  // namespace sora

int sora_internal_scan_ht40(const uint32_t* iq0, const uint32_t* iq1, const sora::CapDesc* d_caps, uint32_t ncaps, uint32_t max_frames, sora::Rx11bRow* d_rows, uint32_t* d_nframes,
                            sora::Ht40Found* d_found, const sora::Tables& T, const uint32_t* sincos, const short* atan, hipStream_t st);

// k_deliver.hip: dense rows + MPDUs of a call of the Rx11bRow-table handles into page-locked host memory, behind the call's kernels
struct DenseStage {                  // per slot / pipeline (grow-only device staging)
    sora_frame_result* d_rows = nullptr; size_t rows_bytes = 0;
    uint32_t* d_src = nullptr; size_t src_bytes = 0;
    uint8_t* d_mpdu = nullptr; size_t mpdu_bytes = 0;
    uint32_t* d_meta = nullptr; size_t meta_bytes = 0;
    sora_frame_result* d_tmpl = nullptr; size_t tmpl_bytes = 0;
};
void sora_internal_dense_free(DenseStage* D);
int sora_internal_dense_deliver(DenseStage* D, const sora::Rx11bRow* d_rows, const uint32_t* d_nframes, const sora::CapDesc* d_caps, const sora_frame_result* h_tmpl,
                                uint32_t ncaps, uint32_t mf, const uint8_t* d_slots, hipStream_t st,
                                sora_frame_result* h_rows, size_t max_rows, uint32_t* h_meta, uint8_t* h_mpdu, size_t mpdu_cap,
                                // (a template already on the device; the real number of "captures" where the host
                                // only knows a bound; per "capture" the row its rows start at, k_dense_rows)
                                const sora_frame_result* d_tmpl = nullptr, const uint32_t* d_ncaps = nullptr, const uint32_t* d_evbase = nullptr);

// sora_hip.cpp: the i-th stream of a handle (its i-th pipeline / slot), non-blocking, on priority level i % 3: the runtime keeps GPU_MAX_HW_QUEUES
// hardware queues per level, so a handle's streams get a hardware queue each without the application setting an environment variable
hipError_t sora_internal_stream_create(hipStream_t* out, int index);
// sora_hip.cpp: records the message sora_hip_last_error() returns; hip_error = 0 for none
int sora_internal_fail(int code, const char* what, int hip_error);
const uint32_t* sora_internal_crc_table(int device);
struct sora_rx;
extern "C" int sora_internal_rx_device(sora_rx* rx);                           // device ordinal of a receive handle (sora_shard.cpp)
int sora_internal_tables(int device, sora::Tables* out);                  // the per-device tables of the stage entry points (uploaded on first use)
void sora_internal_dsp_host_tables(std::vector<uint32_t>& sincos, std::vector<short>& atan);   // k_11n.hip: the two dsp_math tables as the host generates them
int sora_internal_pin_table(const char* name, const void* data, size_t bytes);   // sora_hip.cpp: SORA_OK iff the bytes are the pinned sha256 of table `name`
// dsp_math tables of the current device (k_11n.hip)   // device pointer to the 256-entry CRC-32 table of `device` (uploaded on first use), or nullptr
int sora_internal_dsp_tables(const uint32_t** sincos, const short** atan);
