/* sora_hip.h -- C ABI of the MI355X-native 802.11a receive PHY (libsora_hip.so).
 *
 * The reference (microsoft/Sora) has no runtime FFI: its "operator API" is the compile-time BRICK
 * protocol (kernel/brick/inc/brick.h:151-475 -- Process / Reset / Flush on typed burst ports, shared
 * state in context facades, errors as bool + CF_Error::error_code).  This header is the boundary a
 * maintainer binds instead: every entry point below names the reference brick(s) it replaces.  Bricks
 * fire per 4/28/64 samples, a kernel launch per Process() is impossible, so the ports are kept but the
 * BURST is a batch (BRICK allows any BURST/NSTREAM: brick.h:182-238).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns a BK_ERROR_* / E_ERROR_* compatible code
 *     (kernel/brick/inc/dspcomm.h:22-31, kernel/bb/Brick11/src/ieee80211facade.hpp:10-19)
 *   - pointers named d_* are DEVICE (HBM) pointers, h_* are host pointers; the caller owns every buffer
 *   - calls on one handle must be serialised by the caller, different handles are independent and may be used from
 *     different threads (no hidden globals, unlike the reference's BB11aDemodCtx,
 *     kernel/bb/demod11/fb11ademod_config.hpp:123; the per-device look-up tables of the stage entry points are
 *     created once under a lock)
 *   - the library never falls back to a CPU path: without a usable HIP device every compute call
 *     returns SORA_ERR_NO_DEVICE.
 */
#ifndef SORA_HIP_H
#define SORA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)     /* libsora_hip.so is built with -fvisibility=hidden: what this header declares is what it exports */
#endif

/* 3 (round 5).  Against 2: (i) behaviour that round 4 changed under the old number: a call that has been delivered (sora_rx_deliver_async) AND waited for
 * (sora_rx_wait / _wait_any; the same for the rx11b / rx11n / ht40 handles) is RELEASED -- its pipeline is the first to be reused, so its ticket is valid for `depth`
 * further calls only if the host has not both delivered and waited for it; sora_rx_set_fused(1) answers SORA_E_NOT_SUPPORTED in the default build;
 * sora_rx11b_set_single_pass defaults to 2 (automatic); sora_ht40_deliver_async needs max_rows >= 2 x captures x max_frames.  (ii) new this round, all additive:
 * SORA_TRELLIS_WINDOWED and sora_rx_window_stats, sora_rx_set_front / sora_rx_front, sora_hip_table_*, sora_hip_freq_comp11a / _equalize11a / _phase_comp11a,
 * and the automatic choices of sora_rx_set_trellis / sora_rx_set_front (results are identical whichever kernels run).  INTEGRATION.md section 1 lists them. */
/* 4 (round 6).  Against 3: no row is ever delivered with SORA_E_INTERNAL_TIMEOUT (see sora_rx_set_front, form 4); new, additive: sora_rx_set_pipe_wait_us, sora_rx_pipe_stats,
 * sora_hip_pilot11a, sora_rx_bind_mpdu, sora_rx11n_trellis, sora_rx11n_window_stats, sora_rx_call_front, sora_rx_set_ordered, sora_hip_set_share_window_us, SORA_TRELLIS_WINDOWED and
 * the automatic choice for sora_rx11n_set_trellis. */
#define SORA_HIP_ABI_VERSION 4

/* COMPLEX16: kernel/core/inc/complex.h */
typedef struct { int16_t re, im; } sora_complex16;

/* status codes (values of the reference where one exists) */
#define SORA_OK                  0           /* BK_ERROR_SUCCESS / E_ERROR_SUCCESS */
#define SORA_E_FRAME_OK          0x00000001  /* E_ERROR_FRAME_OK */
#define SORA_E_PARAMETER         ((int)0x80000001)
#define SORA_E_PLCP_HEADER_FAIL  ((int)0x80000005)
#define SORA_E_CRC32_FAIL        ((int)0x80000006)
/* retired: ABI 3 reported a frame with it when a bounded wait inside k_pipe expired; since ABI 4 no row carries it (the call is made again on the device) */
#define SORA_E_INTERNAL_TIMEOUT  ((int)0x8000F001)
#define SORA_ERR_FAILED          ((int)0x8000FFFF)  /* BK_ERROR_FAILED */
#define SORA_ERR_HARDWARE_FAILED ((int)0x8000FFFE)  /* BK_ERROR_HARDWARE_FAILED: a HIP call failed */
#define SORA_ERR_INVALID_PARAM   (-1)               /* BK_ERROR_INVALID_PARAM */
#define SORA_ERR_NO_DEVICE       (-5)
#define SORA_ERR_CAPACITY        (-6)               /* a caller-provided buffer or a configured limit is too small */

/* code rates: kernel/bb/Brick11/src/ieee80211const.h:14-20 */
enum { SORA_CR_12 = 0, SORA_CR_23 = 1, SORA_CR_34 = 2 };

/* ------------------------------------------------------------------------------------------------
 * Whole-path receiver = the demod graph CreateDemodGraph11a_40M (kernel/bb/demod11/fb11ademod_config.hpp:
 * 168-233) driven by RxThread (kernel/bb/demod11/fb11a_demod.cpp:29-81), over a BATCH of independent captures.
 * ------------------------------------------------------------------------------------------------ */
typedef struct sora_rx sora_rx_t;

typedef struct {
    uint32_t struct_size;           /* sizeof(sora_rx_cfg) */
    int32_t  device;                /* HIP device ordinal */
    uint32_t sample_rate_mhz;       /* 40: dump rate, TDownSample2 first (samples.hpp:9-47); 20: the even samples;
                                     * 44: CreateDemodGraph11a_44M (fb11ademod_config.hpp:236-300) -- iq is the 40 MHz stream
                                     * sora_hip_ingest(SORA_INGEST_44TO40) makes of a 44 MHz dump; as in that graph, the
                                     * resampler's queued samples survive the reset after a frame (TDownSample44_40 has no
                                     * Reset/Flush).  Positions are 20 MHz-rate samples of the resampled stream. */
    uint32_t max_captures;          /* captures per sora_rx_process call */
    uint64_t max_total_samples;     /* sum of capture lengths per call (input-rate samples) */
    uint32_t max_frames_per_capture;/* frame-table rows per capture */
    uint32_t cca_pwr_threshold;     /* CF_11CCA::cca_pwr_threshold; 0 -> 1000*1000 (fb11ademod_config.hpp:107) */
} sora_rx_cfg;

typedef struct {                    /* one capture = one TMemSamples source (brick/inc/memsource.hpp:16-114) */
    uint64_t offset;                /* first sample, in samples, from the iq base pointer; multiple of 4 (16-byte aligned) */
    uint32_t nsamples;              /* input-rate samples */
    uint32_t capture_id;            /* echoed into the results */
} sora_capture_desc;

typedef struct {                    /* one row per frame the reference's RxThread would have reported */
    uint32_t capture_id;
    uint32_t start_sample;          /* 20 MHz-rate index (in the capture) of the first sample given to T11aLTS */
    uint32_t end_sample;            /* one past the last 20 MHz-rate sample of the frame */
    uint32_t error_code;            /* SORA_E_FRAME_OK / SORA_E_CRC32_FAIL / SORA_E_PLCP_HEADER_FAIL */
    uint32_t rate_kbps;             /* CF_11aRxVector::data_rate_kbps */
    uint16_t length;                /* CF_11aRxVector::frame_length (MPDU incl. FCS) */
    uint16_t nsym;                  /* data symbols */
    uint32_t crc32;                 /* FCS found in the frame (CF_11aRxVector::crc32) */
    int16_t  cfo_est;               /* CF_CFOffset::CFO_est (FP_RAD per sample) */
    uint16_t flags;                 /* SORA_ROW_TRUNCATED on the LAST row of a capture that held more frames than
                                     * max_frames_per_capture: the reference's RxThread reports every frame, the frames after
                                     * this one were found but have no row */
    uint32_t mpdu_offset;           /* byte offset of the MPDU in the mpdu buffer */
} sora_frame_result;

#define SORA_ROW_TRUNCATED 1u

/* create/destroy the graph: CreateDemodGraph11a_40M + BB11aDemodCtx.Init  /  IReferenceCounting::Release */
int  sora_rx_create(const sora_rx_cfg* cfg, sora_rx_t** out);
void sora_rx_destroy(sora_rx_t* rx);
/* ISource::Reset / ISource::Flush (brick.h:343-353): Reset forgets all per-call state; Flush waits for the stream */
int  sora_rx_reset(sora_rx_t* rx);
int  sora_rx_flush(sora_rx_t* rx);
/* The stream every kernel of this handle is launched on (a hipStream_t), for event timing by the caller. */
void* sora_rx_stream(sora_rx_t* rx);

/* ISource::Process over a batch: enqueue the whole receive path for `ncaps` captures whose samples are
 * resident in HBM (d_iq).  Asynchronous; results become available after sora_rx_flush / sora_rx_results. */
int  sora_rx_process_dev(sora_rx_t* rx, const sora_complex16* d_iq, const sora_capture_desc* h_caps, size_t ncaps);
/* Same with a host buffer (copied to HBM first; the PCIe time is then part of the call). */
int  sora_rx_process(sora_rx_t* rx, const sora_complex16* h_iq, size_t total_samples, const sora_capture_desc* h_caps, size_t ncaps);
/* LoadSoraDumpFile -> graph -> MPDU buffer as ONE stream-ordered path (brickutil.h:20-58 in front of fb11a_demod.cpp:88-120): the raw dump
 * bytes in host memory (page-locked for a copy that overlaps the other calls in flight: sora_hip_host_alloc) go up to the call's
 * pipeline, sora_hip_ingest (flags SORA_INGEST_*: RX_BLOCK de-framing, 14 -> 16 bit, 44 -> 40 MHz, TDownSample2) runs on the device and
 * the receive chain decodes the result; no host wait anywhere.  The capture descriptors address the INGESTED sample stream (its length is
 * sora_hip_ingest_count(dump_bytes, flags), at the handle's sample_rate_mhz).  The host buffer must stay untouched until the call has
 * completed (sora_rx_wait).  Ticket, results and delivery as for sora_rx_process_dev. */
int  sora_rx_process_dump(sora_rx_t* rx, const void* h_dump, size_t dump_bytes, unsigned ingest_flags, const sora_capture_desc* h_caps, size_t ncaps);
/* Stream continuation (the live-source case: TRxStream hands the graph an endless stream, brick/inc/rxstream.hpp:34-66, and the graph's
 * state -- the DC estimate, which integrates for ever (dc.hpp:92-166), the carrier-sense windows and counters (cca.hpp:126-158) -- carries
 * over from one read to the next).  With sora_rx_set_stream_mode(rx, 1) capture k of a process call CONTINUES capture k of the call before it:
 *   - after a call, sora_rx_stream_consumed(rx, ticket, h, n) gives, per capture, the RESUME POINT: the number of input-rate samples of that
 *     capture that are final (a position where a burst boundary of the graph falls on a source-call boundary while it is in plain carrier
 *     sense; 0 if the capture holds none).  Every frame the call reported ends in front of it; a frame that was still running when the
 *     capture ended lies behind it and has NOT been reported;
 *   - the host builds the next call's capture k from the stream FROM THAT POINT on: the unconsumed tail of what it submitted plus whatever
 *     has arrived since (each capture still a whole number of 28-sample source bursts; 14 at 20 MHz).  The library starts it with the state
 *     the graph had at the resume point, so the rows of all the calls together (positions are relative to their own capture: add the
 *     stream position of its first sample) are exactly the events the graph reports on the uncut stream -- tests/test_gpu_stream.py holds a
 *     40 MHz stream cut at arbitrary 28-sample boundaries to the compiled reference graph's events on the whole;
 *   - calls of a handle in stream mode run one after the other (a process call first waits for the one before it: it needs its records);
 *     throughput comes from many streams (captures) per call.  sora_rx_reset, and switching the mode, start every stream afresh.  Not
 *     available for sample_rate_mhz = 44 (SORA_E_NOT_SUPPORTED).  Returns the previous mode; a negative argument only queries. */
int  sora_rx_set_stream_mode(sora_rx_t* rx, int enable);
int  sora_rx_stream_consumed(sora_rx_t* rx, int ticket, uint32_t* h_consumed, size_t ncaps);

/* TBB11aFrameSink's frame buffer + CF_Error per frame: copies results of the last process call to the host.
 * h_mpdu may be NULL (descriptors only).  *nout = rows written. Frames appear in (capture, time) order. */
int  sora_rx_results(sora_rx_t* rx, sora_frame_result* h_out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);

/* Tickets.  With several calls in flight (sora_rx_set_depth) every process call has a TICKET, a positive number that
 * identifies the call until its pipeline is reused: by the `depth`-th process call after it, or earlier once the call is
 * RELEASED -- its delivery was enqueued (sora_rx_deliver_async) and it has been waited for (sora_rx_wait / sora_rx_wait_any),
 * so that everything it produced is in the caller's memory.
 * What fb11a_demod.cpp:37-71 does per frame -- look at the result, hand the MPDU to the MAC -- is done per call with these:
 *   sora_rx_ticket          ticket of the most recent process call (0: none yet)
 *   sora_rx_wait            block until that call has finished (its kernels and any sora_rx_deliver_async copies)
 *   sora_rx_results_of      sora_rx_results for that call
 *   sora_rx_results_dev_of  sora_rx_results_dev for that call
 *   sora_rx_stream_of       the HIP stream that call runs on
 *   sora_rx_deliver_async   enqueue, behind the call's kernels and without blocking the host, the delivery of its results
 *                           into host memory: the dense rows (capture, time order; at most max_rows are copied), their
 *                           number, and -- h_mpdu != NULL -- the MPDU array (row.mpdu_offset indexes it; mpdu_bytes must be
 *                           at least sora_rx_mpdu_bytes(rx, ticket)).  The buffers should be page-locked
 *                           (sora_hip_host_alloc) so that the copies overlap later calls; they are valid after sora_rx_wait.
 *   sora_rx_wait_any        block until SOME call whose delivery has been enqueued has finished, and return its ticket (the oldest
 *                           one if several have).  Calls in flight overtake one another (their streams sit on different dispatch
 *                           priorities and share the chip); a host that always waits for its oldest ticket leaves the pipelines of the
 *                           calls that finished first idle until the slowest one is through (measured: a third of every pipeline's
 *                           time, profiles/r04_t_own_timeline.txt).  RxThread's loop (fb11a_demod.cpp:37-71) with this call:
 *                               process_dev -> deliver_async -> [depth calls in flight?] wait_any -> hand that call's MPDUs to the MAC.
 *                           The next process call reuses a released pipeline first.  SORA_ERR_FAILED if no such call is in flight.
 * A stale ticket gives SORA_ERR_INVALID_PARAM. */
int    sora_rx_ticket(sora_rx_t* rx);
int    sora_rx_wait(sora_rx_t* rx, int ticket);
int    sora_rx_wait_any(sora_rx_t* rx, int* ticket);
int    sora_rx_results_of(sora_rx_t* rx, int ticket, sora_frame_result* h_out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);
void*  sora_rx_stream_of(sora_rx_t* rx, int ticket);
size_t sora_rx_mpdu_bytes(sora_rx_t* rx, int ticket);
int    sora_rx_deliver_async(sora_rx_t* rx, int ticket, sora_frame_result* h_rows, size_t max_rows, uint32_t* h_nrows, uint8_t* h_mpdu, size_t mpdu_bytes);
/* A lone call's delivery without the copy behind it (round 6, ABI 4).  sora_rx_bind_mpdu names, for the NEXT process call of the handle only, a page-locked array from
 * sora_hip_host_alloc of at least that call's sora_rx_mpdu_bytes() (SORA_ERR_CAPACITY from the process call otherwise; the array's geometry is the device array's: 32 bytes per
 * symbol slot): the call's frame sink then writes every MPDU there as well, straight over PCIe while the frames are being finished, and sora_rx_deliver_async(ticket, ..) with
 * the SAME h_mpdu copies rows and count only.  The bytes are valid when the ticket has been waited for; words of the array that no frame of the call owns are left as they were.
 * For a host that keeps ONE or TWO calls in flight (the MPDU copy is a lone 4096-frame call's 0.22 ms tail); with many calls in flight their copies overlap anyway and
 * waves that wait on PCIe writes hold compute units the other calls' kernels want (measured in round 3): do not bind there.  h_mpdu = NULL cancels a binding not yet used. */
int    sora_rx_bind_mpdu(sora_rx_t* rx, uint8_t* h_mpdu, size_t mpdu_bytes);
/* Per-kernel timing with HIP events recorded on the streams the kernels run on (SoraStopwatch / MACStopwatch analogue,
 * kernel/bb/demod11/MACStopwatch.h:84-128).  With profiling enabled every process call brackets each kernel launch
 * with events; sora_rx_kernel_times waits for the calls in flight and returns, in launch order, the MEAN duration (ms)
 * of each kernel over all process calls since profiling was (re-)enabled.  Calls in flight on different pipelines
 * share the CUs, so these are the durations the kernels really had, not what each would take alone. */
int  sora_rx_set_profiling(sora_rx_t* rx, int enable);
int  sora_rx_kernel_times(sora_rx_t* rx, float* h_ms, size_t cap, size_t* nout);
const char* sora_rx_kernel_name(size_t index);

/* Consecutive process calls rotate over `depth` internal pipelines (own stream, own intermediate arrays), so the
 * latency-bound front end of one call overlaps the trellis kernel of the call before it -- what the reference gets from
 * running ViterbiThread beside RxThread (fb11a_demod.cpp:117-120).  Default 8, 1 = strictly
 * one call at a time, at most 16.  (The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, one of them
 * the null stream's: a host that wants more than three calls to actually run side by side sets that variable, as bench.py does -- 16; measured:
 * the step time of the benchmark batch stops improving at eight calls in flight on twelve or more queues.)  Returns the previous value; depth <= 0 only queries.  sora_rx_results / _results_dev /
 * _stream refer to the MOST RECENT process call, the *_of forms to the call whose ticket is given; sora_rx_flush waits for
 * every call in flight; an input buffer must stay untouched until the call that reads it has finished (as with any
 * asynchronous call). */
int  sora_rx_set_depth(sora_rx_t* rx, int depth);

/* Two trellis kernels decode the soft stream of the split data-field path, bit for bit the same T11aViterbi (viterbi.hpp:103-237):
 *   64  k_viterbi    the 64 states of a frame pair in the 64 lanes of a wave: two frames per wave, the most waves -- the faster
 *                    one while few frames are in flight (one 4096-frame call: 2048 waves for 1024 SIMDs);
 *   16  k_viterbi16  a frame pair in 16 lanes x 4 registers, eight frames per wave: half the vector instructions per frame and
 *                    no cross-row exchanges, but a quarter of the waves -- the faster one once several calls are in flight.
 *   SORA_TRELLIS_WINDOWED (1)  k_viterbi16w (round 5): the frame's 256-bit trace-back windows (viterbi.hpp:196-214) decoded side by side, each run from
 *                    all-equal metrics a warm-up ahead of its first window and PROVEN afterwards -- its metric vector at the preceding normalisation
 *                    point must equal its predecessor's there (k_win_redo); a frame with a mismatch is decoded again serially by the same kernel.  Bit-exact by
 *                    construction; the kernel for few frames in flight (one capture, one lone call), where a frame per wave-slot leaves the chip idle.
 *   0   (default)    chosen by the library from the handle's capacity in flight: k_viterbi16 when depth x max_captures >= 32768
 *                    (eight 4096-capture calls, two 16384-capture calls), the window-parallel form below that.
 * Returns the previous setting; a negative argument only queries. */
#define SORA_TRELLIS_WINDOWED 1
int  sora_rx_set_trellis(sora_rx_t* rx, int lanes_per_pair);
int  sora_rx_trellis(sora_rx_t* rx);            /* the kernel the next process call will use: 64, 16 or SORA_TRELLIS_WINDOWED (resolves the automatic choice) */
/* The window-parallel trellis's proof record since the handle was created: out[0] unit boundaries compared, out[1] boundaries whose vectors differed,
 * out[2] frames decoded again by the serial kernel because of that, out[3] units.  Waits for the handle's calls in flight. */
int  sora_rx_window_stats(sora_rx_t* rx, unsigned long long out[4]);
/* Three forms of the symbol chain T11aDataSymbol .. T11aDeinterleave (fb11ademod_config.hpp:200-222) write the same soft stream:
 *   1   k_frame      one wave per frame, four symbols per pass, the pilot tracker's loop-carried chain in the same wave: the cheaper one when the chip is full of frames;
 *   3   k_sym_front -> k_track_lds -> k_sym_back   per symbol slot in front of and behind the tracker (TFreqCompensation, TFFT64, TChannelEqualization | TPhaseCompensate's
 *                    rotation, T11aDemap, T11aDeinterleave), and the tracker's chain (freqoffset.hpp:28-30, pilot.hpp:166-233) alone, four lanes per frame, with its
 *                    three look-up tables folded into LDS: a frame's symbols spread over the chip -- the one for few, long frames (fsample-6: 465 symbols);
 *   4   k_pipe       form 3 AND the window-parallel trellis as ONE launch whose workgroups hand symbols on as the tracker passes them (the reference's demod || Viterbi
 *                    overlap, fb11a_demod.cpp:109-112, inside a frame): the one for a handful of frames (a single capture).  (Its trellis units run two per wave in k_viterbi's
 *                    64-lane layout while the handle's calls in flight are few enough for that many workgroups -- a lone capture with up to three calls in flight --, eight per wave
 *                    otherwise.)  Used only with the window-parallel trellis and
 *                    where every workgroup of the handle's calls in flight is resident at once (at most three quarters of the device's compute units: 192 on an MI355X); otherwise a request
 *                    for 4 runs as 3.  Its hand-offs
 *                    are bounded waits (sora_rx_set_pipe_wait_us): should one expire -- another process or other GPU work holds the compute units its workgroups need -- the
 *                    launch's finishing kernel makes the call's data field again with form 1's code and the serial trellis, so the call still delivers the reference's rows
 *                    (never an error code of this library's own), and the handle keeps to form 3 for its next 64 calls (sora_rx_pipe_stats counts both);
 *                    it is the form for an otherwise idle chip -- each of its workgroups takes a whole CU's LDS, so beside a chip kept full by other handles its launch waits for CUs
 *                    to drain and form 3 is the faster one (measured: 3.3 against 1.3 ms median beside eight 4096-capture calls in flight; tools/pipe_under_load.py);
 *   0   (default)    chosen by the library: 4 while depth x max_captures x max_frames_per_capture <= 16 (and it fits, and no batch-sized handle of this process has used
 *                    the device within the last 20 ms), 3 up to 512, else 1.
 * Returns the previous setting; a negative argument only queries. */
int  sora_rx_set_front(sora_rx_t* rx, int kernels);
int  sora_rx_front(sora_rx_t* rx);              /* 1, 3 or 4: what the next process call will use */
/* The automatic choice looks at the clock (has another, batch-sized handle of the process taken a call on this device within the share window?), so sora_rx_front is a
 * forecast.  What a call WAS launched with is latched at its process call: sora_rx_call_front (ticket 0 = the most recent call) answers 1, 3 or 4 for as long as the handle
 * holds the ticket.  sora_hip_set_share_window_us sets the window for the whole process (default 20000; 0 = other handles are never looked at) and returns the old value. */
int  sora_rx_call_front(sora_rx_t* rx, int ticket);
/* Completion in submission order (round 6; 0 / 1, returns the previous setting; default 0).  Calls in flight share the chip and, left alone, finish together: a host that hands a
 * batch over in pieces to get its first frames early gets nothing.  With 1 the trellis kernel of a call starts behind the trellis kernel of the call submitted before it: the
 * calls complete one after the other, a call's front kernels still run beside the trellis of the one before it and its delivery beside the trellis of the one after it.  For the
 * host that wants frames as early as possible (the reference's per-frame hand-over, fb11a_demod.cpp:37-71); the throughput-oriented loop (eight unordered calls) is faster per
 * sample.  Results are identical either way.  (An ordered call is launched kernel by kernel, not as a recorded hipGraph.) */
int  sora_rx_set_ordered(sora_rx_t* rx, int on);
uint32_t sora_hip_set_share_window_us(uint32_t us);
/* k_pipe's safety net.  The bound of every wait inside its launch, in microseconds (default 20000; 0 = a wait that is not satisfied at first look gives up: every call
 * then takes the redo path -- what tests/test_gpu_pipe.py does to prove that path delivers the reference's rows).  Returns the previous bound; a negative argument only queries. */
int  sora_rx_set_pipe_wait_us(sora_rx_t* rx, long long us);
/* out[0] = calls since the handle was created whose data field was made again because a wait inside k_pipe gave up, out[1] = times the handle then switched itself to
 * form 3 for 64 calls.  Waits for the handle's calls in flight. */
int  sora_rx_pipe_stats(sora_rx_t* rx, unsigned long long out[2]);
/* Identical consecutive calls (same buffer, same capture set) may be replayed as ONE hipGraph launch instead of a chain of
 * kernel launches: 1 = on, 0 = off (default).  Returns the previous setting; a negative argument only queries. */
int  sora_rx_set_graph(sora_rx_t* rx, int enable);

/* The data field (T11aDataSymbol .. T11aViterbi) is decoded by k_frame (symbol chain, soft values to HBM packed three bits each)
 * followed by k_viterbi16 / k_viterbi (sora_rx_set_trellis).  Rounds 2-3 also shipped k_decode -- both in one kernel, symbol waves
 * feeding trellis waves through a ring in LDS the way the reference's RxThread feeds its ViterbiThread through TThreadSeparator
 * (stdbrick.hpp:89-248) -- which lost to the split form at every depth (0.80 against 0.46 ms per step, BENCH_r03) and is a BUILD
 * VARIANT since round 4 (sora_amd.build.build_variant("fused", ["SORA_WITH_K_DECODE"])): in the default library
 * sora_rx_set_fused(rx, 1) returns SORA_E_NOT_SUPPORTED and changes nothing; enable = 0 and the query (enable < 0) still work and
 * return the previous value.  sora_rx_kernel_name_fused(i) names the kernels of the fused chain ("" = slot not used). */
int  sora_rx_set_fused(sora_rx_t* rx, int enable);
const char* sora_rx_kernel_name_fused(size_t index);

/* Device-side views of the last call's outputs (valid until the next process/reset/destroy). */
int  sora_rx_results_dev(sora_rx_t* rx, const sora_frame_result** d_rows, const uint32_t** d_nrows, const uint8_t** d_mpdu);
int  sora_rx_results_dev_of(sora_rx_t* rx, int ticket, const sora_frame_result** d_rows, const uint32_t** d_nrows, const uint8_t** d_mpdu);

/* ------------------------------------------------------------------------------------------------
 * Per-stage entry points with the brick port shapes, batched (n = number of bursts).  All pointers are
 * device pointers; `stream` is a hipStream_t (NULL = default stream).  Each can sit behind a BRICK-shaped
 * adapter (include/sora_brick.hpp) where the SSE brick sits today.
 * ------------------------------------------------------------------------------------------------ */
/* TFFT64  (Brick11/src/fft.hpp:108-135 -> core/inc/fft_r4dif.h FFT<64>): IPORT COMPLEX16x64 -> OPORT COMPLEX16x64 */
int sora_hip_fft64(const sora_complex16* d_in, sora_complex16* d_out, size_t n, void* stream);
/* FFT<128> (core/inc/fft_r4dif.h: 4 x 32, 8-point terminal stage; used by the reference's oversampled TX and the
 * shape a 40 MHz HT receiver needs): IPORT COMPLEX16x128 -> OPORT COMPLEX16x128 */
int sora_hip_fft128(const sora_complex16* d_in, sora_complex16* d_out, size_t n, void* stream);
/* T11aLTS (channel_11a.hpp:33-230): IPORT COMPLEX16 x 144 -> the context facades it fills:
 * CF_CFOffset::CFO_est, CF_FreqCompensate::Coeffs[16 vcs], CF_Channel_11a::ChannelCoeffs[16 vcs] */
typedef struct { int16_t cfo_est; int16_t reserved; sora_complex16 freq[64]; sora_complex16 chan[64]; } sora_lts11a_ctx;
int sora_hip_lts11a(const sora_complex16* d_in, sora_lts11a_ctx* d_ctx, size_t n, void* stream);
/* T11aDataSymbol -> TFreqCompensation -> TFFT64 -> TChannelEqualization (PHY_11a.hpp:361-430, channel_11a.hpp:532-653):
 * IPORT COMPLEX16 x 80 -> OPORT COMPLEX16 x 64; symbol i uses d_ctx[d_ctx_index[i]] (d_ctx_index NULL: d_ctx[0]) */
int sora_hip_symfront11a(const sora_complex16* d_in, const sora_lts11a_ctx* d_ctx, const uint32_t* d_ctx_index, sora_complex16* d_eq, size_t n, void* stream);
/* The three one-multiply bricks of the symbol chain on their own (the receive path fuses them; here each can replace its SSE brick alone), IPORT COMPLEX16 x 64 ->
 * OPORT COMPLEX16 x 64, symbol i with the coefficients of d_ctx[d_ctx_index[i]] (index NULL: d_ctx[0]):
 *   TFreqCompensation     (channel_11a.hpp:614-653)  out = ((in >> 1) x CF_FreqCompensate::Coeffs) >> 15.  The brick also leaves in >> 1 in its input queue; the
 *                                                   queue belongs to the caller here and is not written.
 *   TChannelEqualization  (channel_11a.hpp:534-604)  out = (in x CF_Channel_11a::ChannelCoeffs) >> 8, bins 28 .. 35 zero
 *   TPhaseCompensate      (freqoffset.hpp:16-66)     out = (in x CF_PhaseCompensate::CompCoeffs) >> 15 -- sora_track11a_state::comp of d_state[d_state_index[i]] */
int sora_hip_freq_comp11a(const sora_complex16* d_in, const sora_lts11a_ctx* d_ctx, const uint32_t* d_ctx_index, sora_complex16* d_out, size_t n, void* stream);
int sora_hip_equalize11a(const sora_complex16* d_in, const sora_lts11a_ctx* d_ctx, const uint32_t* d_ctx_index, sora_complex16* d_out, size_t n, void* stream);
/* TPhaseCompensate + TPilotTrack (freqoffset.hpp:14-66, pilot.hpp:121-269) over the symbols of n frames, in order:
 * frame f owns symbols d_first[f] .. d_first[f]+d_nsym[f]-1 of d_eq; d_state[f] = CF_PhaseCompensate + CF_PilotTrack,
 * read at entry and written back.  Bins 0 and 27..37 of the output are not defined by the reference and are written 0. */
typedef struct { int16_t cfo_comp, sfo_comp, cfo_tracker, sfo_tracker; uint32_t symbol_count; sora_complex16 comp[64]; } sora_track11a_state;
int sora_hip_pilot_track11a(const sora_complex16* d_eq, const uint32_t* d_first, const uint32_t* d_nsym, sora_track11a_state* d_state,
                            sora_complex16* d_out, size_t nframes, void* stream);
int sora_hip_phase_comp11a(const sora_complex16* d_in, const sora_track11a_state* d_state, const uint32_t* d_state_index, sora_complex16* d_out, size_t n, void* stream);
/* TPilotTrack alone (pilot.hpp:121-269): the same frame tables, d_in = TPhaseCompensate's output (what sora_hip_phase_comp11a wrote for the symbol with the state as it
 * was); rotates the symbol by the pilots' mean phase and slope and advances d_state (the trackers, CFO_comp / SFO_comp, symbol_count, CompCoeffs for the NEXT symbol).
 * The brick that replaces TPilotTrack where TPhaseCompensate stays the reference's (or sora_hip_phase_comp11a's): new in ABI 4. */
int sora_hip_pilot11a(const sora_complex16* d_in, const uint32_t* d_first, const uint32_t* d_nsym, sora_track11a_state* d_state,
                      sora_complex16* d_out, size_t nframes, void* stream);
/* T11aDemap<N_BPSC>::Filter (demapper11a.hpp:10-79): IPORT COMPLEX16x64 -> OPORT uchar x 48*n_bpsc */
int sora_hip_demap11a(const sora_complex16* d_in, uint8_t* d_soft, int n_bpsc, size_t n, void* stream);
/* T11aDeinterleave{BPSK,QPSK,QAM16,QAM64} (deinterleaver.hpp): IPORT uchar x N_CBPS -> OPORT uchar x N_CBPS */
int sora_hip_deinterleave11a(const uint8_t* d_in, uint8_t* d_out, int n_bpsc, size_t n, void* stream);
/* T11aViterbi<5000*8,48,256,24>::Filter (viterbi.hpp:103-237) over n frames: frame i reads nsoft[i] soft values at
 * d_soft + soft_off[i] and writes frame_len[i]+2 decoded bytes at d_out + out_off[i]. */
int sora_hip_viterbi11a(const uint8_t* d_soft, const uint32_t* d_soft_off, const uint32_t* d_nsoft,
                        const uint16_t* d_frame_len, int code_rate, uint8_t* d_out, const uint32_t* d_out_off,
                        size_t n, void* stream);
/* The same brick for a caller that streams bursts through it (a BRICK adapter calls once per burst): no allocation, no host
 * wait, everything in stream order.  soft_span_bytes = the extent of the caller's soft buffer that the n jobs address
 * (max over i of soft_off[i] + nsoft[i]; the jobs' ranges must not overlap -- any offsets, any order, any nsoft >= 24: a
 * job with less than one OFDM symbol of soft values is not a frame); d_workspace = 16-byte aligned device memory of
 * at least sora_hip_viterbi11a_workspace_bytes(soft_span_bytes, n), owned by the caller and free for reuse once the call's
 * work on `stream` has completed.  (sora_hip_viterbi11a is this entry point over a grow-only workspace the library caches
 * per device; it reads the extent back from the device and waits for the stream before it returns.) */
size_t sora_hip_viterbi11a_workspace_bytes(size_t soft_span_bytes, size_t n);
int sora_hip_viterbi11a_ws(const uint8_t* d_soft, size_t soft_span_bytes, const uint32_t* d_soft_off, const uint32_t* d_nsoft,
                           const uint16_t* d_frame_len, int code_rate, uint8_t* d_out, const uint32_t* d_out_off,
                           size_t n, void* d_workspace, size_t workspace_bytes, int lanes_per_pair, void* stream);
/* lanes_per_pair: which of the two trellis kernels decodes (identical results): 64 (or 0) = k_viterbi, 16 = k_viterbi16 (see
 * sora_rx_set_trellis). */

/* Capture ingest in front of the receive graph, one streaming pass on the device:
 *   SORA_INGEST_RXBLOCK    the input is a Sora dump: 128-byte RX_BLOCKs = 16-byte descriptor + 28 COMPLEX16
 *                          (LoadSoraDumpFile, kernel/brick/inc/brickutil.h:20-58; kernel/core/inc/_rx_manager.h:96-137)
 *   SORA_INGEST_RAW14      samples are 14-bit two's complement, zero-extended: x = (int16)(raw << 2)
 *   SORA_INGEST_44TO40     TDownSample44_40 / Down44to40 (Brick11/src/sampling.hpp:35-66, 44MTo40M.hpp:62-123)
 *   SORA_INGEST_DECIMATE2  TDownSample2 (Brick11/src/samples.hpp:9-47): keep the even samples (40 -> 20 MHz)
 * applied in that order.  d_raw: the dump bytes (or, without RXBLOCK, a COMPLEX16 stream) in device memory;
 * sora_hip_ingest_count gives the number of samples the call writes (a function of the size and flags only). */
#define SORA_INGEST_RXBLOCK   1u
#define SORA_INGEST_RAW14     2u
#define SORA_INGEST_44TO40    4u
#define SORA_INGEST_DECIMATE2 8u
size_t sora_hip_ingest_count(size_t raw_bytes, unsigned flags);
int sora_hip_ingest(const void* d_raw, size_t raw_bytes, unsigned flags, sora_complex16* d_out, size_t out_capacity,
                    size_t* n_out, void* stream);

/* 802.11a transmitter: the modulation graph CreateModGraph11a_* of the reference
 * (kernel/bb/demod11/fb11amod_config.hpp:74-110: TBB11aSrc -> T11aSc -> TConvEncode_* -> T11aInterleave* -> TMap11a* ->
 * T11aAddPilot -> TIFFTx -> TPackSample16to8) plus the preamble (Brick11/src/preamble11a.hpp:19-140), a batch of frames per
 * call.  Frame f: MPDU WITHOUT FCS (the FCS is appended, PHY_11a.hpp:160-170) of d_len[f] bytes at d_mpdu + d_off[f], data
 * rate d_rate_kbps[f] (6000..54000), scrambler seed d_seed[f] (the reference harness uses 0xFF, fb11amod_config.hpp:53);
 * writes sora_hip_tx11a_samples(len, rate) COMPLEX8 samples (int8 re, int8 im) at 40 MHz -- what `demod11 -m` writes --
 * at sample d_out_off[f] of d_out.  A frame with an unknown rate must not be submitted (sora_hip_tx11a_samples = 0). */
size_t sora_hip_tx11a_samples(uint32_t mpdu_len_nofcs, uint32_t rate_kbps);
int sora_hip_tx11a(const uint8_t* d_mpdu, const uint32_t* d_off, const uint32_t* d_len, const uint32_t* d_rate_kbps,
                   const uint8_t* d_seed, size_t nframes, int8_t* d_out, const uint64_t* d_out_off, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 802.11n 2x2, stage level (the whole-path graph is sora_rx11n_* below).  Batched bricks, n symbols per call:
 * T11nDemap{BPSK,QPSK,QAM16,QAM64} (kernel/bb/Brick11/src/demapper11n.hpp:89-309): IPORT COMPLEX16 x 64 (one pilot-tracked
 * symbol of one spatial stream) -> OPORT uint8 x 52*N_BPSC soft values 0..7;
 * T11nDeinterleave{...}_S0/_S1 (deinterleaver_11n.hpp:4-1618): 52*N_BPSC soft values of spatial stream 0 or 1 -> de-interleaved.
 * ------------------------------------------------------------------------------------------------ */
int sora_hip_demap11n(const sora_complex16* d_in, uint8_t* d_soft, int n_bpsc, size_t n, void* stream);
int sora_hip_deinterleave11n(const uint8_t* d_in, uint8_t* d_out, int n_bpsc, int spatial_stream, size_t n, void* stream);
/* TMimoChannelEst (channel_11n.hpp:329-443), a batch of frames: d_ltf_r[f][128] = the two HT-LTF symbols of RX chain r after the
 * FFT (first 64, second 64) -> CF_ChannelMimo per frame: d_h[f][2][128] (row = RX chain; columns 0..63 / 64..127 = spatial stream
 * 1 / 2) and d_hinv[f][2][128] = its 2x2 inverse x 2^16 per carrier, computed in single-precision floats operation for operation as
 * the reference's SSE code (brick/inc/sora_matrix.h:134-148,305-313) and packed with saturation.
 * TMimoChannelComp (channel_11n.hpp:445-521), a batch of symbols: x = (Hinv y) >> 9 with saturation; symbol s uses
 * d_hinv[d_frame_index[s]] (d_frame_index NULL: frame 0); y_r = RX chain r after the FFT, x_k = spatial stream k. */
/* The CFO / phase bricks (Brick11/src/dsp_math.h tables, generated with the C library as the reference does at start-up).
 * d_state[f][24] = CF_FreqOffset_11n of frame f: vfo_delta_i[8] | vfo_step_i[8] | vfo_theta_i[8] (int16).
 * sora_hip_cfo_est11n     TFreqEstimator_11n (freqoffset_11n.hpp:42-160): d_lltf_r[f][128] = the L-LTF of RX chain r (two 64-sample
 *                         halves) -> d_state[f] (vfo_delta_i = {0..7} x CFO, vfo_step_i = 8 x CFO, vfo_theta_i = 0; CFO_est = state[1])
 * sora_hip_freq_comp11n   TFreqComp_11n (freqoffset_11n.hpp:162-280): frame f owns d_nbursts[f] bursts of 8 samples starting at sample
 *                         d_first[f] of both chains; out = sat((in * sincos(vfo_delta_i - vfo_theta_i)) >> 15); vfo_delta_i advances by
 *                         vfo_step_i per burst and is written back.  max_bursts >= every d_nbursts[f].
 * sora_hip_pilot_track11n TPilotTrack_11n (pilot_11n.hpp:84-141): frame f owns symbols d_first[f] .. +d_nsym[f]-1 of the two spatial
 *                         streams (after TMimoChannelComp); vfo_theta_i of d_state[f] accumulates the mean pilot phase; d_theta[s][8]
 *                         (optional) receives vfo_theta_i as it stands after symbol s. */
int sora_hip_cfo_est11n(const sora_complex16* d_lltf0, const sora_complex16* d_lltf1, int16_t* d_state, size_t nframes, void* stream);
int sora_hip_freq_comp11n(const sora_complex16* d_in0, const sora_complex16* d_in1, sora_complex16* d_out0, sora_complex16* d_out1,
                          const uint32_t* d_first, const uint32_t* d_nbursts, int16_t* d_state, size_t nframes, size_t max_bursts, void* stream);
int sora_hip_pilot_track11n(const sora_complex16* d_x0, const sora_complex16* d_x1, const uint32_t* d_first, const uint32_t* d_nsym, int16_t* d_state,
                            int16_t* d_theta, size_t nframes, void* stream);
/* The legacy part of the HT-mixed preamble (both RX chains), a batch of frames:
 * sora_hip_siso_est11n   TSisoChannelEst (channel_11n.hpp:33-231): d_lltf_r[f][128] = the two L-LTF symbols of chain r after the FFT ->
 *                        d_ch[f][2][64], 1 / H per chain in Q16 with the reference's rounding lanes; bins 28..35, which the brick
 *                        leaves unwritten, are 0.
 * sora_hip_siso_comp11n  TSisoChannelComp (channel_11n.hpp:233-297) + TMrcCombine (PHY_11n.hpp:362-398): x_r = sat((y_r c_r) >> 9),
 *                        mrc = (x_0 + x_1) >> 1; symbol s uses d_ch[d_frame_index[s]] (NULL: frame 0); any of d_x0 / d_x1 / d_mrc may be
 *                        NULL (not all three).
 * sora_hip_sig_demap11n  T11nSigDemap (demapper11n.hpp:6-87): d_sym[f][3][64] = L-SIG, HT-SIG1, HT-SIG2 after MRC -> d_soft[f][144]
 *                        (L-SIG demapped on I, the HT-SIG symbols on Q; 48 carriers each). */
int sora_hip_siso_est11n(const sora_complex16* d_lltf0, const sora_complex16* d_lltf1, sora_complex16* d_ch, size_t nframes, void* stream);
int sora_hip_siso_comp11n(const sora_complex16* d_ch, const uint32_t* d_frame_index, const sora_complex16* d_y0, const sora_complex16* d_y1,
                          sora_complex16* d_x0, sora_complex16* d_x1, sora_complex16* d_mrc, size_t nsym, void* stream);
int sora_hip_sig_demap11n(const sora_complex16* d_sym, uint8_t* d_soft, size_t nframes, void* stream);
/* sora_hip_sig_decode11n  T11aDeinterleaveBPSK x3 -> T11nViterbiSig (viterbi.hpp:51-99) -> T11nSigParser (PHY_11n.hpp:432-513):
 *                        d_soft[f][144] (sora_hip_sig_demap11n's output) -> d_rec[f][12] = error_code (0 or SORA_E_PLCP_HEADER_FAIL),
 *                        data_rate_kbps, frame_length, ht_frame_mcs, ht_frame_length, code_rate, total_symbols, remain_symbols,
 *                        symbol_type (the context fields, from 0, as the parser leaves them -- partial on a failed parse),
 *                        then the decoded L-SIG (24 bits) and HT-SIG (bits 0..31, bits 32..41). */
int sora_hip_sig_decode11n(const uint8_t* d_soft, uint32_t* d_rec, size_t nframes, void* stream);
int sora_hip_mimo_est11n(const sora_complex16* d_ltf0, const sora_complex16* d_ltf1, sora_complex16* d_h, sora_complex16* d_hinv, size_t nframes, void* stream);
int sora_hip_mimo_comp11n(const sora_complex16* d_hinv, const uint32_t* d_frame_index, const sora_complex16* d_y0, const sora_complex16* d_y1,
                          sora_complex16* d_x0, sora_complex16* d_x1, size_t nsym, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 802.11n 2x2 receive graph = CreateDemodGraph11n (kernel/bb/demod11/fb11ndemod_config.hpp:166-257) driven by
 * RxThread (kernel/bb/demod11/fb11n_demod.cpp:30-85) over a batch of independent captures of TWO RX chains at 40 MHz (the two
 * dump files Test11N_FB_Demod loads).  cfg->sample_rate_mhz must be 40; capture lengths are whole 28-sample source bursts; one
 * descriptor addresses the same range of both chain buffers.  Results as sora_rx_results reports them, with rate_kbps = the MCS
 * index (8..10: what the reference's SIG parser accepts), end_sample = the 40 MHz source position when RxThread sees the event,
 * start_sample / nsym / cfo_est 0.  A frame cut short by the end of its capture is decoded from zero-padded symbols, as the
 * reference's final flush does. */
typedef struct sora_rx11n sora_rx11n_t;
int   sora_rx11n_create(const sora_rx_cfg* cfg, sora_rx11n_t** out);
void  sora_rx11n_destroy(sora_rx11n_t* rx);
void* sora_rx11n_stream(sora_rx11n_t* rx);
int   sora_rx11n_process_dev(sora_rx11n_t* rx, const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_capture_desc* caps, size_t ncaps);
int   sora_rx11n_process(sora_rx11n_t* rx, const sora_complex16* h_iq0, const sora_complex16* h_iq1, size_t nsamples_per_chain, const sora_capture_desc* caps, size_t ncaps);
int   sora_rx11n_results(sora_rx11n_t* rx, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);
/* Calls in flight, as for sora_rx_t: consecutive process calls rotate over `depth` pipelines (own stream and result arrays; default 1, at most 8), so
 * the latency-bound scan / symbol kernels of one call overlap the issue-bound trellis kernel of the call before it.  Every call has a ticket;
 * sora_rx11n_results / _stream refer to the most recent call, sora_rx11n_wait / _results_of to the call whose ticket is given (valid until `depth`
 * further calls have been made, or until the call is released: delivered and waited for, see sora_rx_wait_any); an input buffer must stay untouched until the call that reads it has
 * finished.  sora_rx11n_set_depth returns the
 * previous value (depth <= 0 only queries) and waits for the calls in flight; sora_rx11n_process (host buffers) also does. */
int   sora_rx11n_set_depth(sora_rx11n_t* rx, int depth);
/* As sora_rx_set_trellis, for T11aViterbi<..,192,36>: 64 = k_viterbi11n, 16 = k_viterbi16_11n, SORA_TRELLIS_WINDOWED = the frame's 192-bit trace-back windows decoded side
 * by side and proven afterwards (k_viterbi16w_11n + k_win_redo_11n: bit-exact by construction like the 802.11a form; new in ABI 4), 0 (default since ABI 4; it was 64) =
 * chosen by the library: window-parallel while depth x max_captures x max_frames_per_capture <= 2048 (a frame per wave-slot leaves the chip idle: a lone capture's
 * frame), k_viterbi11n above that.  Returns the previous setting, a negative argument only queries; sora_rx11n_trellis resolves the automatic choice;
 * sora_rx11n_window_stats is sora_rx_window_stats for this handle. */
int   sora_rx11n_set_trellis(sora_rx11n_t* rx, int lanes_per_pair);
int   sora_rx11n_trellis(sora_rx11n_t* rx);
int   sora_rx11n_window_stats(sora_rx11n_t* rx, unsigned long long out[4]);
int   sora_rx11n_ticket(sora_rx11n_t* rx);
int   sora_rx11n_synchronize(sora_rx11n_t* rx);                                                       /* every call issued so far has finished */
int   sora_rx11n_wait(sora_rx11n_t* rx, int ticket);
/* as sora_rx_wait_any: the oldest FINISHED call with an enqueued delivery; it is then released and its pipeline reused first */
int   sora_rx11n_wait_any(sora_rx11n_t* rx, int* ticket);
int   sora_rx11n_results_of(sora_rx11n_t* rx, int ticket, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);
/* as sora_rx11b_deliver_async */
int   sora_rx11n_deliver_async(sora_rx11n_t* rx, int ticket, sora_frame_result* h_rows, size_t max_rows, uint32_t* h_counts, uint8_t* h_mpdu, size_t mpdu_cap);

/* ------------------------------------------------------------------------------------------------
 * The data field of an HT-mixed 40 MHz, two-stream frame (BASELINE.json configs[3]: 128-point FFT, MMSE MIMO detection, one decoder per
 * spatial stream).  PARITY UNPINNED: the reference has no such graph (its 802.11n receiver is 20 MHz, zero-forcing, one decoder, MCS 8-10:
 * kernel/bb/Brick11/src/PHY_11n.hpp:497, channel_11n.hpp:423-433); every brick it does have is used with the reference's arithmetic
 * (FFT<128>, TFreqComp_11n, TMimoChannelEst / TMimoChannelComp -- noise_var = 0 gives exactly their zero-forcing weights --, TPilotTrack_11n,
 * T11nDemap*, T11aViterbi<..,192,36>, T11aDesc, CRC-32), the 40 MHz carrier plan / HT-LTF / interleaver come from IEEE 802.11n-2009 and
 * are modelled independently in oracle/py_ht40.py.  The caller supplies what the (20 MHz, reference-pinned) front end finds: where the first
 * HT-LTF symbol starts, modulation, code rate, PSDU lengths, CFO.  A frame = 2 HT-LTF symbols + nsym data symbols of 160 samples at 40 MHz on
 * two RX chains; each spatial stream carries its own PSDU (SERVICE + PSDU incl. FCS + tail, scrambled, K = 7 coded, punctured, interleaved).
 * Results: two rows per frame (start_sample = spatial stream 0 / 1, error_code FRAME_OK / CRC32_FAIL, length, crc32, nsym;
 * rate_kbps = 10 n_bpsc + code_rate); d_weights (optional, [nframes][4][128] COMPLEX16) receives the detection weights x 2^16.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t offset;        /* 40 MHz samples from both iq bases to the first sample (cyclic prefix) of HT-LTF 1 */
    uint32_t n_bpsc;        /* 1, 2, 4, 6 */
    uint32_t code_rate;     /* SORA_CR_12 / _23 / _34 */
    uint32_t length[2];     /* PSDU bytes (FCS included) of spatial stream 0 / 1, <= 4000 */
    int32_t  cfo;           /* phase step per 40 MHz sample, 65536 = 2 pi (TFreqComp_11n: the running phase is n * cfo - theta) */
    float    noise_var;     /* noise variance per carrier in LSB^2 of the FFT<128> output; 0 = zero forcing; > 0: unbiased MMSE,
                             * W = diag((W'H)_ss)^-1 W', W' = (H^H H + noise_var I)^-1 H^H */
    uint32_t frame_id;      /* echoed into the rows' capture_id */
} sora_ht40_frame;
typedef struct sora_ht40 sora_ht40_t;
uint32_t sora_ht40_symbols(uint32_t length0, uint32_t length1, uint32_t n_bpsc, uint32_t code_rate);   /* data symbols of such a frame (0: bad arguments) */
/* max_soft_values >= sum over frames of 2 x nsym x 108 n_bpsc (+ 64 per frame) */
int   sora_ht40_create(int device, uint32_t max_frames, uint64_t max_soft_values, sora_ht40_t** out);
void  sora_ht40_destroy(sora_ht40_t* rx);
void* sora_ht40_stream(sora_ht40_t* rx);                                                               /* the stream of the most recent process call */
/* 16 (default: the handle keeps eight calls in flight) / 64: as sora_rx_set_trellis (the two streams of a frame are one pair) */
int   sora_ht40_set_trellis(sora_ht40_t* rx, int lanes_per_pair);
int   sora_ht40_synchronize(sora_ht40_t* rx);                                                          /* every call issued so far has finished (a handle keeps eight calls in flight:
                                                                                                        * process_dev waits only for the call eight calls back; results reports the most recent one) */
int   sora_ht40_process_dev(sora_ht40_t* rx, const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_ht40_frame* h_frames, size_t nframes, sora_complex16* d_weights);
int   sora_ht40_results(sora_ht40_t* rx, sora_frame_result* h_out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);
/* The same receiver on RAW CAPTURES, as every other handle takes its input: two-chain 40 MHz captures (whole 28-sample source bursts,
 * offsets a multiple of 4) in, the front end finds the frames.  The legacy preamble and HT-SIG of an HT-mixed 40 MHz frame are the 20 MHz
 * waveforms sent on both halves of the channel (the upper one rotated by 90 degrees), so the even samples of x[n] j^n are (1 + j) times
 * the 20 MHz legacy waveform and the reference's own bricks apply unchanged: TCCA11n (cca_11n.hpp:5-171), the L-LTF frequency offset
 * (freqoffset_11n.hpp:86-280), TSisoChannelEst / Comp, T11nSigDemap, the SIG decoder and T11nSigParser (PHY_11n.hpp:400-514) with one
 * gate relaxed -- MCS 8..14 at CBW 40 and LENGTH <= 4000 are accepted where PHY_11n.hpp:497-505 accepts MCS 8..10 and 1500.  Own
 * definitions (parity unpinned): each spatial stream carries its own PSDU of HT-SIG's LENGTH through its own encoder (configs[3]'s
 * "dual Viterbi"), the CFO handed to the data field is the L-LTF estimate, the noise variance the MMSE detector uses is estimated from
 * the difference of the two L-LTF symbols.  Rows: per event in (capture, time) order; a recorded frame reports two rows (start_sample =
 * spatial stream, rate_kbps = MCS, end_sample = 40 MHz source position of the event, FRAME_OK / CRC32_FAIL), a header that fails one row
 * (SORA_E_PLCP_HEADER_FAIL); a frame the capture cuts off raises no event.  Tickets / results_of / deliver_async as above.  The call is one
 * chain of kernels (the front end's records become the data field's tables on the device, k_ht40_plan): no host wait inside it; captures that
 * hold more frames / soft values than the handle was created for are reported by sora_ht40_wait / _results_of (SORA_ERR_CAPACITY). */
int   sora_ht40_process_captures_dev(sora_ht40_t* rx, const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_capture_desc* h_caps, size_t ncaps,
                                     uint32_t max_frames_per_capture);
/* Tickets, as for sora_rx_t: every process call has one; it stays valid until sora_ht40_calls_in_flight() (8) further calls have reused its
 * slot (or, once the call is released -- delivered and waited for -- until the next call, which takes a released slot first) -- so back-to-back calls are all collectable, each by its own
 * ticket, while later ones run.  The INPUT buffers of a call must stay
 * untouched until sora_ht40_wait(its ticket) (or _results_of, or _synchronize) has returned. */
int   sora_ht40_ticket(sora_ht40_t* rx);                       /* ticket of the most recent process call (0: none) */
int   sora_ht40_calls_in_flight(sora_ht40_t* rx);              /* how many calls the handle keeps addressable */
int   sora_ht40_wait(sora_ht40_t* rx, int ticket);
/* as sora_rx_wait_any: the oldest FINISHED call with an enqueued delivery; it is released and its slot reused first */
int   sora_ht40_wait_any(sora_ht40_t* rx, int* ticket);
void* sora_ht40_stream_of(sora_ht40_t* rx, int ticket);
int   sora_ht40_results_of(sora_ht40_t* rx, int ticket, sora_frame_result* h_out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);
/* As sora_rx11b_deliver_async.  The delivered table is the one sora_ht40_results_of reports, in (capture, time) order: two rows per RECORDED frame
 * (one per spatial stream, start_sample = the stream) and -- raw-capture calls -- one row per event that ended in the front end (an HT-SIG whose CRC-8
 * fails, an unsupported MCS: E_ERROR_PLCP_HEADER_FAIL, no PSDU); the rows of the last event a full capture could hold carry SORA_ROW_TRUNCATED.
 * max_rows: at least 2 x the call's frames (descriptor calls) / 2 x ncaps x max_frames_per_capture (raw-capture calls: the host knows only that bound). */
int   sora_ht40_deliver_async(sora_ht40_t* rx, int ticket, sora_frame_result* h_rows, size_t max_rows, uint32_t* h_counts, uint8_t* h_mpdu, size_t mpdu_cap);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU sharding for a C host.  Captures are independent -- the reference resets its context per
 * frame (kernel/bb/demod11/fb11ademod_config.hpp:68-95) and RxThread walks one dump at a time (fb11a_demod.cpp:29-81) -- so
 * rank r of W runs sora_rx_* on its own block of captures in its own HBM (sora_shard_partition) and nothing is exchanged on
 * the data path.  These calls are the result exchange: ncclAllGather of the dense 36-byte result rows and of the per-rank row
 * counts, ncclAllReduce(sum) of counters, on RCCL over xGMI, one process per GPU.  The 128-byte id is made on rank 0 and
 * carried to the other ranks by the host (file, socket, MPI ...).  RCCL is loaded on the first sora_shard_* call.
 *   sora_shard_gather_rows     device buffers, asynchronous on `stream`: d_rows must hold max_rows_per_rank rows (rows past
 *                              *d_nrows are padding), d_all_rows world x max_rows_per_rank rows, d_all_counts world counts
 *   sora_shard_gather_results  convenience for sora_rx_t: the rows of a finished process call (ticket, or 0 = the most recent)
 *                              of every rank, compacted in rank order into h_all_rows (world x max_rows_per_rank rows of room),
 *                              h_counts[world], *n_total; blocks until done.  capture_id is the caller's own numbering, so a
 *                              host that numbers captures globally gets a table it can use as it is.
 * ------------------------------------------------------------------------------------------------ */
#define SORA_SHARD_ID_BYTES 128
typedef struct sora_shard sora_shard_t;
int  sora_shard_unique_id(uint8_t id[SORA_SHARD_ID_BYTES]);
int  sora_shard_create(const uint8_t id[SORA_SHARD_ID_BYTES], int world_size, int rank, int device, sora_shard_t** out);
void sora_shard_destroy(sora_shard_t* sh);
int  sora_shard_world(const sora_shard_t* sh, int* world_size, int* rank);
void sora_shard_partition(size_t n_items, int world_size, int rank, size_t* first, size_t* count);
int  sora_shard_gather_rows(sora_shard_t* sh, const sora_frame_result* d_rows, const uint32_t* d_nrows, size_t max_rows_per_rank,
                            sora_frame_result* d_all_rows, uint32_t* d_all_counts, void* stream);
int  sora_shard_reduce_counters(sora_shard_t* sh, uint64_t* d_counters, size_t n, void* stream);
int  sora_shard_gather_results(sora_shard_t* sh, sora_rx_t* rx, int ticket, size_t max_rows_per_rank,
                               sora_frame_result* h_all_rows, uint32_t* h_counts, size_t* n_total);
/* The same with the MPDUs (what fb11a_demod.cpp:64-70 hands to the MAC): every rank's MPDU bytes, densely packed in row order, gathered
 * with one more ncclAllGather of max_mpdu_bytes_per_rank bytes per rank; the gathered rows' mpdu_offset indexes h_all_mpdu
 * (world x max_mpdu_bytes_per_rank bytes of room), *mpdu_total = bytes used.  h_all_mpdu = NULL: rows only.
 * Every rank must make the call (three collectives); a rank whose own part fails still takes part and every rank then returns an error. */
int  sora_shard_gather_results_mpdu(sora_shard_t* sh, sora_rx_t* rx, int ticket, size_t max_rows_per_rank, sora_frame_result* h_all_rows,
                                    uint32_t* h_counts, size_t* n_total, size_t max_mpdu_bytes_per_rank, uint8_t* h_all_mpdu, size_t* mpdu_total);

/* ------------------------------------------------------------------------------------------------
 * 802.11b receive graph = CreateDemodGraph (kernel/bb/demod11/fb11bdemod_config.hpp:122-172) driven by
 * MAC11b_Receive (kernel/bb/demod11/fb11b_demod.cpp:27-76) over a batch of independent 44 MHz captures: TDCRemove,
 * TEnergyDetect / TDCEstimator, TSymTiming, TBarkerSync, TBB11bDespread, TSFDSync, TDBPSKDemap / TDQPSKDemap,
 * TCCK5P5Decoder / TCCK11Decoder (kernel/bb/Brick11/src/cck.hpp), TDesc741, TBB11bPlcpParser, TBB11bFrameSink.  cfg->sample_rate_mhz must be 44; capture lengths are whole 28-sample bursts.
 * Result rows: end_sample = CF_MemSamples::mem_sample_index() when the harness sees the event (44 MHz samples);
 * error_code also takes SORA_E_SFD_FAIL / SORA_E_SFD_TIMEOUT / SORA_E_SYNC_TIMEOUT; crc32 = the reference's FCS word (three
 * FCS bytes and one stale buffer byte, PHY_11b.hpp:725-731); start_sample, nsym and cfo_est are 0.  Long preamble; 1 Mbps
 * DBPSK, 2 Mbps DQPSK, 5.5 and 11 Mbps CCK payloads (all four rates of the reference graph).
 * Two kernels per call: the first has no CCK decoders in it (more resident waves) and hands a capture over when a header announces
 * 5.5 / 11 Mbps; the second redoes the handed-over captures.  (A build variant, SORA_VARIANT_11B_ONE_KERNEL, sends every capture through the
 * second one only: 1 / 2 Mbps traffic 3-6 % slower, CCK traffic 28 % faster, same rows either way.)
 * ------------------------------------------------------------------------------------------------ */
#define SORA_E_NOT_SUPPORTED     ((int)0x80000003)
#define SORA_E_SFD_FAIL          ((int)0x80000004)
#define SORA_E_SFD_TIMEOUT       ((int)0x80000008)
#define SORA_E_SYNC_TIMEOUT      ((int)0x80000009)
typedef struct sora_rx11b sora_rx11b_t;
int  sora_rx11b_create(const sora_rx_cfg* cfg, sora_rx11b_t** out);
void sora_rx11b_destroy(sora_rx11b_t* rx);
void* sora_rx11b_stream(sora_rx11b_t* rx);            /* the HIP stream of the most recent process call */
int   sora_rx11b_synchronize(sora_rx11b_t* rx);       /* every call issued so far has finished.  A handle keeps two calls in flight (own stream and result
                                                       * buffers each): process_dev waits only for the call before the previous one -- the caller's d_iq must
                                                       * stay untouched until then --, results reports the most recent call */
int  sora_rx11b_process_dev(sora_rx11b_t* rx, const sora_complex16* d_iq, const sora_capture_desc* caps, size_t ncaps);
int  sora_rx11b_process(sora_rx11b_t* rx, const sora_complex16* h_iq, size_t nsamples, const sora_capture_desc* caps, size_t ncaps);
int  sora_rx11b_results(sora_rx11b_t* rx, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);
/* Tickets, as for sora_rx_t: every process call has one; it stays valid until sora_rx11b_calls_in_flight() (2) further calls have reused its
 * slot (or, once the call is released -- delivered and waited for -- until the next call, which takes a released slot first) -- back-to-back calls are all collectable, each by its own
 * ticket, while the next one runs (what fb11b_demod.cpp does per frame,
 * per call).  The INPUT buffer of a call must stay untouched until sora_rx11b_wait(its ticket) (or _results_of, or _synchronize) has returned. */
int   sora_rx11b_ticket(sora_rx11b_t* rx);                     /* ticket of the most recent process call (0: none) */
int   sora_rx11b_calls_in_flight(sora_rx11b_t* rx);            /* how many calls the handle keeps addressable */
/* The graph runs as two kernels: the Barker-rate instantiation decodes every capture and hands those whose PLCP header announces 5.5 / 11 Mbps to the
 * CCK-capable one, which redoes them from their first sample.  enable = 1: every capture goes straight through the CCK-capable instantiation (all
 * four rates, identical rows) -- the better choice when most frames are CCK; 0: always two passes; 2 (DEFAULT since round 4): automatic -- a
 * two-pass call counts on the device how many captures its first pass handed over, and once such a count has come back (the handle never waits for
 * it) and says "more than half", the following calls take the single pass, except every 16th, which measures again; negative: query.  Returns the
 * previous setting (0, 1 or 2). */
int   sora_rx11b_set_single_pass(sora_rx11b_t* rx, int enable);
int   sora_rx11b_wait(sora_rx11b_t* rx, int ticket);
/* as sora_rx_wait_any: the oldest FINISHED call with an enqueued delivery; it is released and its slot reused first */
int   sora_rx11b_wait_any(sora_rx11b_t* rx, int* ticket);
void* sora_rx11b_stream_of(sora_rx11b_t* rx, int ticket);
int   sora_rx11b_results_of(sora_rx11b_t* rx, int ticket, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap);
/* Result delivery without a host wait (as sora_rx_deliver_async): behind the call's kernels, on its stream, the dense rows in (capture, time)
 * order -- h_rows must have room for captures x max_frames_per_capture of them --, h_counts[0] = rows, h_counts[1] = MPDU bytes, and the MPDUs
 * densely packed in row order (the rows' mpdu_offset indexes h_mpdu; an MPDU that would reach past mpdu_cap is left out and h_counts[1] says
 * so).  Page-locked buffers (sora_hip_host_alloc); valid once sora_rx11b_wait(ticket) has returned.  h_mpdu may be NULL.
 * The copies are enqueued before the call's sizes are known to the host, so they move the WHOLE of what the caller offers -- max
 * captures x frames rows and all mpdu_cap bytes, every call: give mpdu_cap the size the traffic needs (frames x MTU), not a worst case
 * (rows x 4096 costs tens of MB over the host link per call and delivers stale bytes behind h_counts[1]). */
int   sora_rx11b_deliver_async(sora_rx11b_t* rx, int ticket, sora_frame_result* h_rows, size_t max_rows, uint32_t* h_counts, uint8_t* h_mpdu, size_t mpdu_cap);

/* ------------------------------------------------------------------------------------------------
 * Small device-memory helpers so a pure-C host needs no HIP headers.
 * ------------------------------------------------------------------------------------------------ */
int   sora_hip_device_count(void);

/* The look-up tables (usin / ucos / uatan2 of core/inc/intalglut.h:4,3648,7332, the FFT twiddles of fft_lut_twiddle.h:61433-61600, the demapper's step tables of
 * Brick11/src/demapper.h:55-130, dsp_math.h:215-245's sincos / atan, ...) are regenerated from closed forms when a process first needs them, with the host's libm.
 * Each has a PINNED sha256 compiled into the library; a table that does not hash to its pin (another libm, -ffast-math) makes every create / stage call fail with
 * SORA_ERR_FAILED instead of decoding differently.  sora_hip_table_digest needs no device; sora_hip_table_read copies the device-resident table of the current
 * device back (h_out == NULL: only *bytes). */
int   sora_hip_table_count(void);
const char* sora_hip_table_name(int index);
const char* sora_hip_table_pin(const char* name);                 /* the pinned sha256 (64 hex digits), NULL: no such table */
int   sora_hip_table_digest(const char* name, char hex65[65]);    /* what this build on this host generates */
int   sora_hip_table_read(const char* name, void* h_out, size_t cap, size_t* bytes);
void* sora_hip_malloc(size_t bytes);
void  sora_hip_free(void* d_ptr);
int   sora_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int   sora_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int   sora_hip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);   /* asynchronous on `stream` */
/* page-locked host memory (the target of asynchronous result delivery) */
void* sora_hip_host_alloc(size_t bytes);
void  sora_hip_host_free(void* h_ptr);
/* wait for a stream (NULL = the null stream); the streams of a sora_rx_t do not follow the null stream */
int   sora_hip_stream_synchronize(void* stream);
int   sora_hip_abi_version(void);
const char* sora_hip_last_error(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SORA_HIP_H */
