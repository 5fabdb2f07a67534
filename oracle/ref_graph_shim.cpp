// ref_graph_shim.cpp -- TEST INFRASTRUCTURE ONLY (not shipped, not linked into libsora_hip.so).
//
// extern "C" entry points around the REFERENCE's own BRICK graphs, compiled from the sources where they lie under
// /root/reference by oracle/build_ref.sh (clang -fms-compatibility over a scratch copy patched by oracle/ref_flatten.py)
// into oracle/_ref/libsora_refgraph.so.  This is the reference's receive path itself -- every brick, pin queue and
// facade of kernel/bb/demod11/fb11ademod_config.hpp:168-233 -- so tests/ can pin the C restatement (oracle/so_rx11a.c)
// and the GPU path on arbitrary captures, and bench.py can time "the reference SSE path on this box's host cores".
#include "MACStopwatch.h"      // scratch stand-in (ref_flatten.py); the .cpp files of demod11 include it before the configs
#include "stdbrick.hpp"
#include "fb11ademod_config.hpp"

#define EXPORT extern "C" __attribute__((visibility("default")))
#ifndef REF_CREATE_40M                                // (oracle/ref_graph_hip_shim.cpp includes this file with its own graph builder)
#define REF_CREATE_40M CreateDemodGraph11a_40M
#endif

struct ref_frame { uint32_t error_code, sample_index, rate_kbps, length, crc32, mpdu_offset; };

static ISource* g_src; static ISource* g_vit; static IControlPoint* g_cs;
static unsigned char g_out[4096];                    // OUTPUTBUF_SIZE (fb11a_demod.cpp:20)
static COMPLEX16* g_buf; static uint32_t g_cap;

// Test11A_FB_Demod + RxThread (fb11a_demod.cpp:29-81, 88-120) over a capture in memory, the events recorded instead
// of printed.  sample_index = CF_MemSamples::mem_sample_index() when RxThread sees the event (40 MHz samples).
// Returns the number of events (frames and header failures); CS time-outs are handled as RxThread handles them.
static ISource* g_src44; static ISource* g_vit44; static IControlPoint* g_cs44;
static int rx11a_run(int mhz, const int16_t* iq, uint32_t nsamples40, ref_frame* res, int max_res, uint8_t* mpdu, uint32_t mpdu_cap)
{
    ISource*& g_src = mhz == 44 ? g_src44 : ::g_src; ISource*& g_vit = mhz == 44 ? g_vit44 : ::g_vit; IControlPoint*& g_cs = mhz == 44 ? g_cs44 : ::g_cs;
    if (g_cap < nsamples40 + 64) { free(g_buf); g_cap = nsamples40 + 64; g_buf = (COMPLEX16*)aligned_alloc(16, ((size_t)g_cap * 4 + 15) & ~(size_t)15); }
    memcpy(g_buf, iq, (size_t)nsamples40 * 4);
    BB11aDemodCtx.Init(g_buf, nsamples40 * sizeof(COMPLEX16), g_out, sizeof(g_out));
    if (!g_src) { if (mhz == 44) CreateDemodGraph11a_44M(g_src, g_vit, g_cs); else REF_CREATE_40M(g_src, g_vit, g_cs); }
    else g_src->Seek(ISource::START_POS);             // the graph is built once; rewind the memory source
    g_src->Flush(); BB11aDemodCtx.Reset(); g_src->Reset();
    int n = 0; uint32_t used = 0; uint nWaitCounter = 12;
    for (;;) {
        bool rc = g_src->Process();
        ulong err = BB11aDemodCtx.CF_Error::error_code();
        if (err != E_ERROR_SUCCESS) {
            if (err == E_ERROR_CS_TIMEOUT) {
                BB11aDemodCtx.ResetCarrierSense(); g_cs->Reset();
                if (nWaitCounter > 0) { nWaitCounter--; continue; }
                nWaitCounter = 12; continue;         // RxThread returns TRUE here and the thread wrapper calls it again (AllocStartThread)
            }
            if (n < max_res) {
                ref_frame& f = res[n++];
                f.error_code = err; f.sample_index = BB11aDemodCtx.CF_MemSamples::mem_sample_index();
                f.rate_kbps = BB11aDemodCtx.CF_11aRxVector::data_rate_kbps(); f.length = BB11aDemodCtx.CF_11aRxVector::frame_length();
                f.crc32 = BB11aDemodCtx.CF_11aRxVector::crc32(); f.mpdu_offset = used;
                if ((err == E_ERROR_FRAME_OK || err == E_ERROR_CRC32_FAIL) && used + f.length <= mpdu_cap) { memcpy(mpdu + used, g_out, f.length); used += f.length; }
            }
            g_src->Flush(); BB11aDemodCtx.Reset(); g_src->Reset();
        }
        if (!rc) break;
    }
    return n;
}

// A fresh graph per capture, as the harness has (one dump per process): with a graph kept across captures one event in
// 130,000 was seen one source call early -- something an earlier capture left in a brick.
EXPORT int ref_rx11a_capture(const int16_t* iq, uint32_t nsamples40, ref_frame* res, int max_res, uint8_t* mpdu, uint32_t mpdu_cap)
{
    if (g_src) { IReferenceCounting::Release(g_src); g_src = NULL; }
    return rx11a_run(40, iq, nsamples40, res, max_res, mpdu, mpdu_cap);
}
// CreateDemodGraph11a_44M (fb11ademod_config.hpp:236-300): TDownSample44_40 in front of the same graph; positions in 44 MHz samples.
// A fresh resampler is needed per capture (the brick has no Reset), so the graph is rebuilt on every call.
EXPORT int ref_rx11a_capture44(const int16_t* iq, uint32_t nsamples44, ref_frame* res, int max_res, uint8_t* mpdu, uint32_t mpdu_cap)
{
    if (g_src44) { IReferenceCounting::Release(g_src44); g_src44 = NULL; }
    return rx11a_run(44, iq, nsamples44, res, max_res, mpdu, mpdu_cap);
}

// The same loop over `ncap` equal-sized captures laid end to end, `reps` times, with nothing but a counter coming
// back: what bench.py times as the reference path on a host core (no per-capture binding overhead).
EXPORT uint32_t ref_rx11a_bench(const int16_t* iq, uint32_t ncap, uint32_t nsamples40, uint32_t reps)
{
    static uint8_t mpdu[4096]; ref_frame res[8]; uint32_t ok = 0;
    for (uint32_t r = 0; r < reps; r++)
        for (uint32_t c = 0; c < ncap; c++) {
            int n = rx11a_run(40, iq + (size_t)c * nsamples40 * 2, nsamples40, res, 8, mpdu, sizeof(mpdu));   // graph kept: only the rate is of interest here
            for (int i = 0; i < n; i++) ok += res[i].error_code == E_ERROR_FRAME_OK;
        }
    return ok;
}

// ---------------------------------------------------------------- 802.11a transmitter: Test11A_FB_Mod (fb11a_mod.cpp:26-70)
#include "CRC32.h"
#include "scramble.hpp"
#include "samples.hpp"
#include "fb11amod_config.hpp"

// Preamble graph, then the modulation graph, into one COMPLEX8 buffer at 40 MHz -- what "demod11 -m" writes.
// mpdu: the payload file of the harness (FCS is appended by TBB11aSrc); seed: CF_ScramblerSeed (the harness uses 0xFF).
// Returns the number of complex samples, or -1.
EXPORT int ref_tx11a(const uint8_t* mpdu, uint32_t len, uint32_t rate_kbps, uint32_t seed, int8_t* out8, uint32_t max_samples)
{
    static ISource* ssrc; static ISource* tssrc;
    static unsigned char data[4096 + 16];
    if (len > 4096 - 4) return -1;
    if (!ssrc) { ssrc = CreateModGraph11a_40M(); tssrc = CreatePreamble11a_40M(); }
    COMPLEX8* buf = (COMPLEX8*)out8;
    BB11aModCtx.set_mod_buffer(buf, max_samples);
    tssrc->Reset();
    if (BB11aModCtx.CF_Error::error_code() != E_ERROR_SUCCESS) return -1;
    tssrc->Process();
    uint ts_len = BB11aModCtx.CF_TxSampleBuffer::tx_sample_cnt();
    memset(data, 0, sizeof(data)); memcpy(data, mpdu, len);
    BB11aModCtx.init(rate_kbps, data, (ushort)len, buf + ts_len, max_samples - ts_len);
    BB11aModCtx.CF_ScramblerSeed::sc_seed() = (uchar)seed;
    ssrc->Reset();
    if (BB11aModCtx.CF_Error::error_code() != E_ERROR_SUCCESS) return -1;
    ssrc->Process(); ssrc->Flush();
    return (int)(ts_len + BB11aModCtx.CF_TxSampleBuffer::tx_sample_cnt());
}

// ---------------------------------------------------------------- 802.11b (SURVEY row f4 / BASELINE configs[0]): the reference's own graphs
#include "bb/bbb.h"
#include "bb/DataRate.h"
#include "dot11_plcp.h"
#include "pulse.hpp"
#include "phy_11b.hpp"
#include "barkerspread.hpp"
#include "fb11bdemod_config.hpp"
#include "fb11bmod_config.hpp"

// Test11B_FB_Mod (fb11b_mod.cpp:40-70): CreateModGraph (fb11bmod_config.hpp:28-50) over one MPDU; COMPLEX8 at 44 MHz.
// rate_kbps 1000/2000/5500/11000.  Returns the number of complex samples, or -1.
EXPORT int ref_tx11b(const uint8_t* mpdu, uint32_t len, uint32_t rate_kbps, int8_t* out8, uint32_t max_samples)
{
    static ISource* ssrc;
    static unsigned char data[4096 + 16];
    if (len > 4096 - 4) return -1;
    if (!ssrc) ssrc = CreateModGraph();
    memset(data, 0, sizeof(data)); memcpy(data, mpdu, len);
    InitModGraphCtx(rate_kbps, data, (int)len, (COMPLEX8*)out8, max_samples);
    if (BB11bModCtx.CF_Error::error_code() != E_ERROR_SUCCESS) return -1;
    *(PULONG)(data + len) = BB11bModCtx.CF_11bTxVector::crc32();        // "append CRC32 to the data stream"
    ssrc->Reset();
    if (BB11bModCtx.CF_Error::error_code() != E_ERROR_SUCCESS) return -1;
    ssrc->Process(); ssrc->Flush();
    return (int)BB11bModCtx.CF_TxSampleBuffer::tx_sample_cnt();
}

// Test11B_FB_Demod + MAC11b_Receive (fb11b_demod.cpp:27-76, 79-117) over a 44 MHz capture in memory.
EXPORT int ref_rx11b_capture(const int16_t* iq, uint32_t nsamples44, ref_frame* res, int max_res, uint8_t* mpdu, uint32_t mpdu_cap)
{
    static unsigned char out[4096];                                      // OUTPUTBUF_SIZE
    if (g_cap < nsamples44 + 64) { free(g_buf); g_cap = nsamples44 + 64; g_buf = (COMPLEX16*)aligned_alloc(16, ((size_t)g_cap * 4 + 15) & ~(size_t)15); }
    memcpy(g_buf, iq, (size_t)nsamples44 * 4);
    BB11bDemodCtx.init(g_buf, nsamples44 * sizeof(COMPLEX16), out, sizeof(out));
    memset(out, 0, sizeof(out));                                         // as at the start of the harness process: static buffers and
    memset(&BB11bDemodCtx.CF_DifferentialDemap::last_symbol(), 0, sizeof(COMPLEX16));   // the two context fields that no reset touches
    BB11bDemodCtx.CF_Descramber::byte_reg() = 0;
    if (pRxSource) { IReferenceCounting::Release(pRxSource); pRxSource = NULL; }   // a fresh graph per capture, as the harness has
    pRxSource = CreateDemodGraph();
    pRxSource->Flush(); BB11bDemodCtx.reset(); pRxSource->Reset();
    int n = 0; uint32_t used = 0;
    for (;;) {                                                           // MAC11b_Receive is called until it returns false
        bool rc = pRxSource->Process();
        ulong err = BB11bDemodCtx.CF_Error::error_code();
        if (err != BK_ERROR_SUCCESS) {
            if (err != E_ERROR_CS_TIMEOUT && n < max_res) {
                ref_frame& f = res[n++];
                f.error_code = err; f.sample_index = BB11bDemodCtx.CF_MemSamples::mem_sample_index();
                f.rate_kbps = BB11bDemodCtx.CF_11bRxVector::data_rate_kbps(); f.length = BB11bDemodCtx.CF_11bRxVector::frame_length();
                f.crc32 = BB11bDemodCtx.CF_11bRxVector::crc32(); f.mpdu_offset = used;
                if ((err == E_ERROR_FRAME_OK || err == E_ERROR_CRC32_FAIL) && used + f.length <= mpdu_cap) { memcpy(mpdu + used, out, f.length); used += f.length; }
            }
            if (err == E_ERROR_FRAME_OK || err == E_ERROR_CRC32_FAIL) {  // "TRICK: jump advance of the last CRC byte"
                switch (BB11bDemodCtx.CF_11bRxVector::data_rate_kbps()) {
                case 1000:  pRxSource->Seek(8 * 11 * 4); break;
                case 2000:  pRxSource->Seek(4 * 11 * 4); break;
                case 5500:  pRxSource->Seek(8 * 2 * 4); break;
                case 11000: pRxSource->Seek(8 * 1 * 4); break;
                }
            }
            pRxSource->Flush(); BB11bDemodCtx.reset(); pRxSource->Reset();
            continue;
        }
        if (!rc) break;
    }
    return n;
}

// Drop the cached 40 MHz receive graph so that the next call builds a fresh one (tests use it to show that no result
// depends on what an earlier capture left in the bricks).
EXPORT void ref_rx11a_fresh_graph(void) { if (g_src) { IReferenceCounting::Release(g_src); g_src = NULL; } }

// ---------------------------------------------------------------- 802.11n 2x2: the reference's own graphs
#include "phy_11n.hpp"
#include "fb11nmod_config.hpp"
#include "fb11ndemod_config.hpp"

// Test11N_FB_Mod (fb11n_mod.cpp:28-70): L-STF/L-LTF, L-SIG/HT-SIG, HT-STF/HT-LTF and data graphs into two COMPLEX16
// streams (one per TX chain) at 40 MHz.  mcs: what the harness passes as "bit rate" (8..15 in this tree).
// Returns samples per chain, or -1.
EXPORT int ref_tx11n(const uint8_t* mpdu, uint32_t len, uint32_t mcs, int16_t* out0, int16_t* out1, uint32_t max_samples)
{
    static ISource *lsrc, *htsrc, *sigsrc, *ssrc;
    static unsigned char data[4096 + 16];
    if (len > 4096 - 4) return -1;
    if (!ssrc) { CreatePreambleGraph11n(lsrc, htsrc); sigsrc = CreateSigGraph11n(); ssrc = CreateModGraph11n(); }
    COMPLEX16* bufs[2] = { (COMPLEX16*)out0, (COMPLEX16*)out1 };
    memset(data, 0, sizeof(data)); memcpy(data, mpdu, len);
    BB11nModCtx.init((ushort)mcs, data, (ushort)len, bufs, max_samples);
    ISource* seq[4] = { lsrc, sigsrc, htsrc, ssrc };
    for (int i = 0; i < 4; i++) {
        seq[i]->Reset();
        if (BB11nModCtx.CF_Error::error_code() != E_ERROR_SUCCESS) return -1;
        seq[i]->Process();
    }
    ssrc->Flush();
    return (int)BB11nModCtx.GetSinkSampleCount();
}

// Test11N_FB_Demod + RxThread (fb11n_demod.cpp:30-85, 92-140) over two 40 MHz captures (one per RX chain) in memory.
EXPORT int ref_rx11n_capture(const int16_t* iq0, const int16_t* iq1, uint32_t nsamples40, ref_frame* res, int max_res, uint8_t* mpdu, uint32_t mpdu_cap)
{
    static ISource *ssrc, *svit; static IControlPoint* scs;
    static unsigned char out[4096];
    static COMPLEX16* buf[2]; static uint32_t cap;
    if (cap < nsamples40 + 64) {
        for (int k = 0; k < 2; k++) { free(buf[k]); buf[k] = (COMPLEX16*)aligned_alloc(16, ((size_t)(nsamples40 + 64) * 4 + 15) & ~(size_t)15); }
        cap = nsamples40 + 64;
    }
    memcpy(buf[0], iq0, (size_t)nsamples40 * 4); memcpy(buf[1], iq1, (size_t)nsamples40 * 4);
    BB11nDemodCtx.Init(out, sizeof(out));
    if (ssrc) { IReferenceCounting::Release(ssrc); ssrc = NULL; }     // a fresh graph per capture (the harness builds one per run)
    CreateDemodGraph11n(ssrc, svit, scs);
    IQuery* q;
    if (!ssrc->TraverseGraph(&q, "TMemSamples2", 1)) return -1;
    MemSamplesDesc* ms = q->QueryInterface<MemSamplesDesc>();
    ms->Init(2, buf, nsamples40);
    ssrc->Reset();
    int n = 0; uint32_t used = 0; uint nWaitCounter = 12;
    for (;;) {
        bool rc = ssrc->Process();
        ulong err = BB11nDemodCtx.CF_Error::error_code();
        if (err != E_ERROR_SUCCESS) {
            if (err == E_ERROR_CS_TIMEOUT) {
                BB11nDemodCtx.ResetCarrierSense(); scs->Reset();
                if (nWaitCounter > 0) { nWaitCounter--; continue; }
                nWaitCounter = 12; continue;
            }
            if (n < max_res) {
                ref_frame& f = res[n++];
                f.error_code = err; f.sample_index = BB11nDemodCtx.CF_MemSamples::mem_sample_index(); f.rate_kbps = BB11nDemodCtx.CF_HTRxVector::ht_frame_mcs();
                f.length = BB11nDemodCtx.CF_HTRxVector::ht_frame_length(); f.crc32 = BB11nDemodCtx.CF_11aRxVector::crc32(); f.mpdu_offset = used;
                if ((err == E_ERROR_FRAME_OK || err == E_ERROR_CRC32_FAIL) && used + f.length <= mpdu_cap) { memcpy(mpdu + used, out, f.length); used += f.length; }
            }
            ssrc->Flush(); BB11nDemodCtx.Reset(); ssrc->Reset();
        }
        if (!rc) break;
    }
    return n;
}

// The 802.11n loop over `ncap` equal-sized two-chain captures, `reps` times (bench.py: the reference 11n path on a host core).
EXPORT uint32_t ref_rx11n_bench(const int16_t* iq0, const int16_t* iq1, uint32_t ncap, uint32_t nsamples40, uint32_t reps)
{
    static uint8_t mpdu[16384]; ref_frame res[8]; uint32_t ok = 0;
    for (uint32_t r = 0; r < reps; r++)
        for (uint32_t c = 0; c < ncap; c++) {
            int n = ref_rx11n_capture(iq0 + (size_t)c * nsamples40 * 2, iq1 + (size_t)c * nsamples40 * 2, nsamples40, res, 8, mpdu, sizeof(mpdu));
            for (int i = 0; i < n; i++) ok += res[i].error_code == E_ERROR_FRAME_OK;
        }
    return ok;
}

// The 802.11b loop over `ncap` equal-sized captures, `reps` times (bench.py: the reference 11b path on a host core).
EXPORT uint32_t ref_rx11b_bench(const int16_t* iq, uint32_t ncap, uint32_t nsamples44, uint32_t reps)
{
    static uint8_t mpdu[8192]; ref_frame res[8]; uint32_t ok = 0;
    for (uint32_t r = 0; r < reps; r++)
        for (uint32_t c = 0; c < ncap; c++) {
            int n = ref_rx11b_capture(iq + (size_t)c * nsamples44 * 2, nsamples44, res, 8, mpdu, sizeof(mpdu));
            for (int i = 0; i < n; i++) ok += res[i].error_code == E_ERROR_FRAME_OK;
        }
    return ok;
}

// ---------------------------------------------------------------- single 802.11n bricks (row f1, stage level): one burst through one brick
// What a brick's Process() needs of its input pin (pinqueue.h), over a caller's buffer.
template <class T, size_t N> struct OneBurstPin {
    typedef T DataType; static const size_t nstream = 1; static const size_t rsize = N;
    const T* p; bool full;
    bool check_read() const { return full; }
    const T* peek() const { return p; }
    const T* peek(size_t) const { return p; }
    void pop() { full = false; }
    void clear() { full = false; }
};
template <class T_CTX, size_t N> class TCaptureSink : public TSink<T_CTX> {           // keeps the last burst it was handed
public:
    DEFINE_IPORT(uchar, N);
    uchar last[N];
    TCaptureSink(T_CTX& ctx) : TSink<T_CTX>(ctx) { memset(last, 0, N); }
    template <class T_IPIN> bool Process(T_IPIN& ipin) { while (ipin.check_read()) { memcpy(last, ipin.peek(), N); ipin.pop(); } return true; }
};
template <template <class, class> class BRICK, size_t NIN, class TIN, size_t NOUT>
static void run_brick_once(const TIN* in, uint8_t* out)
{
    typedef TCaptureSink<BB11nDemodContext, NOUT> Sink;
    typedef BRICK<BB11nDemodContext, Sink> Brick;
    static Sink* sink = new Sink(BB11nDemodCtx);
    static Brick* brick = new Brick(BB11nDemodCtx, sink);
    OneBurstPin<TIN, NIN> pin = { in, true };
    brick->Process(pin);
    memcpy(out, sink->last, NOUT);
}
// T11nDemap{BPSK,QPSK,QAM16,QAM64} (demapper11n.hpp:89-309): 64 COMPLEX16 (one pilot-tracked symbol of one stream) -> 52*nbpsc soft bits
EXPORT int ref_11n_demap(int nbpsc, const int16_t* sym64, uint8_t* soft)
{
    A16 COMPLEX16 x[64]; memcpy(x, sym64, sizeof(x));
    switch (nbpsc) {
    case 1: run_brick_once<T11nDemapBPSK, 64, COMPLEX16, 52>(x, soft); return 52;
    case 2: run_brick_once<T11nDemapQPSK, 64, COMPLEX16, 104>(x, soft); return 104;
    case 4: run_brick_once<T11nDemapQAM16, 64, COMPLEX16, 208>(x, soft); return 208;
    case 6: run_brick_once<T11nDemapQAM64, 64, COMPLEX16, 312>(x, soft); return 312;
    }
    return -1;
}
// T11nDeinterleave{BPSK,QPSK,QAM16,QAM64}_S{0,1} (deinterleaver_11n.hpp): 52*nbpsc soft bits of spatial stream `stream` -> de-interleaved
EXPORT int ref_11n_deinterleave(int nbpsc, int stream, const uint8_t* in, uint8_t* out)
{
    switch (nbpsc * 2 + stream) {
    case 2:  run_brick_once<T11nDeinterleaveBPSK_S0, 52, uchar, 52>(in, out); return 52;
    case 3:  run_brick_once<T11nDeinterleaveBPSK_S1, 52, uchar, 52>(in, out); return 52;
    case 4:  run_brick_once<T11nDeinterleaveQPSK_S0, 104, uchar, 104>(in, out); return 104;
    case 5:  run_brick_once<T11nDeinterleaveQPSK_S1, 104, uchar, 104>(in, out); return 104;
    case 8:  run_brick_once<T11nDeinterleaveQAM16_S0, 208, uchar, 208>(in, out); return 208;
    case 9:  run_brick_once<T11nDeinterleaveQAM16_S1, 208, uchar, 208>(in, out); return 208;
    case 12: run_brick_once<T11nDeinterleaveQAM64_S0, 312, uchar, 312>(in, out); return 312;
    case 13: run_brick_once<T11nDeinterleaveQAM64_S1, 312, uchar, 312>(in, out); return 312;
    }
    return -1;
}

// two-stream variants of the helpers above (the MIMO bricks take one burst per RX chain)
template <class T, size_t N> struct TwoStreamPin {
    typedef T DataType; static const size_t nstream = 2; static const size_t rsize = N;
    const T* p[2]; bool full;
    bool check_read() const { return full; }
    const T* peek(size_t iss = 0) const { return p[iss]; }
    void pop() { full = false; }
    void clear() { full = false; }
};
template <class T_CTX, size_t N> class TCaptureSink2 : public TSink<T_CTX> {
public:
    DEFINE_IPORT(COMPLEX16, N, 2);
    COMPLEX16 last[2][N];
    TCaptureSink2(T_CTX& ctx) : TSink<T_CTX>(ctx) { memset(last, 0, sizeof(last)); }
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    { while (ipin.check_read()) { memcpy(last[0], ipin.peek(0), N * sizeof(COMPLEX16)); memcpy(last[1], ipin.peek(1), N * sizeof(COMPLEX16)); ipin.pop(); } return true; }
};
// TMimoChannelEst (channel_11n.hpp:329-443): the two HT-LTF symbols of each RX chain after the FFT (chain r: ltf_r[0..63] = first
// HT-LTF, [64..127] = second) -> CF_ChannelMimo: h[2][128] and its scaled inverse hinv[2][128] (float 2x2 inverse x 2^16)
EXPORT void ref_11n_mimo_est(const int16_t* ltf0, const int16_t* ltf1, int16_t* h, int16_t* hinv)
{
    static TMimoChannelEst<BB11nDemodContext>* est = new TMimoChannelEst<BB11nDemodContext>(BB11nDemodCtx);
    A16 COMPLEX16 a[128], b[128]; memcpy(a, ltf0, sizeof(a)); memcpy(b, ltf1, sizeof(b));
    TwoStreamPin<COMPLEX16, 128> pin = { { a, b }, true };
    est->Process(pin);
    memcpy(h, BB11nDemodCtx.CF_ChannelMimo::dot11n_2x2_channel(), sizeof(MIMO_2x2_H));
    memcpy(hinv, BB11nDemodCtx.CF_ChannelMimo::dot11n_2x2_channel_inv(), sizeof(MIMO_2x2_H));
}
// TMimoChannelComp (channel_11n.hpp:445-521): one data symbol of both RX chains (after the FFT) x hinv -> the two spatial streams
EXPORT void ref_11n_mimo_comp(const int16_t* hinv, const int16_t* y0, const int16_t* y1, int16_t* x0, int16_t* x1)
{
    typedef TCaptureSink2<BB11nDemodContext, 64> Sink;
    static Sink* sink = new Sink(BB11nDemodCtx);
    static TMimoChannelComp<BB11nDemodContext, Sink>* comp = new TMimoChannelComp<BB11nDemodContext, Sink>(BB11nDemodCtx, sink);
    memcpy(BB11nDemodCtx.CF_ChannelMimo::dot11n_2x2_channel_inv(), hinv, sizeof(MIMO_2x2_H));
    A16 COMPLEX16 a[64], b[64]; memcpy(a, y0, sizeof(a)); memcpy(b, y1, sizeof(b));
    TwoStreamPin<COMPLEX16, 64> pin = { { a, b }, true };
    comp->Process(pin);
    memcpy(x0, sink->last[0], 256); memcpy(x1, sink->last[1], 256);
}

// ---- dsp_math (Brick11/src/dsp_math.h): the 11n path's trigonometry, tables generated at start-up with libm
EXPORT int  ref_dsp_atan(int x, int y, int wide) { const dsp_math& dm = dsp_math::singleton(); return wide ? dm.atan(x, y) : dm.atan((short)x, (short)y); }
EXPORT void ref_dsp_sincos_table(int16_t* out) { const dsp_math& dm = dsp_math::singleton(); for (int i = 0; i < 65536; i++) { COMPLEX16 c = dm.sincos((short)i); out[2 * i] = c.re; out[2 * i + 1] = c.im; } }
// TFreqEstimator_11n (freqoffset_11n.hpp:86-160): the L-LTF of both RX chains (128 samples each: two 64-sample halves) -> CFO_est;
// state[0..7] = vfo_delta_i, [8..15] = vfo_step_i, [16..23] = vfo_theta_i afterwards
EXPORT int ref_11n_cfo_est(const int16_t* l0, const int16_t* l1, int16_t* state)
{
    typedef TCaptureSink2<BB11nDemodContext, 128> Sink;
    static Sink* sink = new Sink(BB11nDemodCtx);
    static TFreqEstimator_11n<BB11nDemodContext, Sink>* est = new TFreqEstimator_11n<BB11nDemodContext, Sink>(BB11nDemodCtx, sink);
    A16 COMPLEX16 a[128], b[128]; memcpy(a, l0, sizeof(a)); memcpy(b, l1, sizeof(b));
    TwoStreamPin<COMPLEX16, 128> pin = { { a, b }, true };
    est->Process(pin);
    memcpy(state, &BB11nDemodCtx.CF_FreqOffset_11n::vfo_delta_i(), 16); memcpy(state + 8, &BB11nDemodCtx.CF_FreqOffset_11n::vfo_step_i(), 16);
    memcpy(state + 16, &BB11nDemodCtx.CF_FreqOffset_11n::vfo_theta_i(), 16);
    return (short)BB11nDemodCtx.CF_CFOffset::CFO_est();
}
// TFreqComp_11n (freqoffset_11n.hpp:218-280): nbursts x 8 samples of both RX chains with the context state (layout as above) going in
// and coming out
EXPORT void ref_11n_freq_comp(int16_t* state, const int16_t* in0, const int16_t* in1, int16_t* out0, int16_t* out1, int nbursts)
{
    typedef TCaptureSink2<BB11nDemodContext, 8> Sink;
    static Sink* sink = new Sink(BB11nDemodCtx);
    static TFreqComp_11n<BB11nDemodContext, Sink>* comp = new TFreqComp_11n<BB11nDemodContext, Sink>(BB11nDemodCtx, sink);
    memcpy(&BB11nDemodCtx.CF_FreqOffset_11n::vfo_delta_i(), state, 16); memcpy(&BB11nDemodCtx.CF_FreqOffset_11n::vfo_step_i(), state + 8, 16);
    memcpy(&BB11nDemodCtx.CF_FreqOffset_11n::vfo_theta_i(), state + 16, 16);
    for (int k = 0; k < nbursts; k++) {
        A16 COMPLEX16 a[8], b[8]; memcpy(a, in0 + 16 * k, 32); memcpy(b, in1 + 16 * k, 32);
        TwoStreamPin<COMPLEX16, 8> pin = { { a, b }, true };
        comp->Process(pin);
        memcpy(out0 + 16 * k, sink->last[0], 32); memcpy(out1 + 16 * k, sink->last[1], 32);
    }
    memcpy(state, &BB11nDemodCtx.CF_FreqOffset_11n::vfo_delta_i(), 16);
}
// TPilotTrack_11n (pilot_11n.hpp:97-141): one symbol of both spatial streams; theta[8] = vfo_theta_i going in and coming out
EXPORT void ref_11n_pilot_track(int16_t* theta, const int16_t* x0, const int16_t* x1)
{
    typedef TCaptureSink2<BB11nDemodContext, 64> Sink;
    static Sink* sink = new Sink(BB11nDemodCtx);
    static TPilotTrack_11n<BB11nDemodContext, Sink>* trk = new TPilotTrack_11n<BB11nDemodContext, Sink>(BB11nDemodCtx, sink);
    memcpy(&BB11nDemodCtx.CF_FreqOffset_11n::vfo_theta_i(), theta, 16);
    A16 COMPLEX16 a[64], b[64]; memcpy(a, x0, sizeof(a)); memcpy(b, x1, sizeof(b));
    TwoStreamPin<COMPLEX16, 64> pin = { { a, b }, true };
    trk->Process(pin);
    memcpy(theta, &BB11nDemodCtx.CF_FreqOffset_11n::vfo_theta_i(), 16);
}

// TSisoChannelEst (channel_11n.hpp:33-231): the L-LTF of both RX chains after the FFT (two 64-bin halves each) -> CF_Channel_11n;
// bins 28..35 are never written by the brick (left 0 here).  TSisoChannelComp (:233-297), TMrcCombine (PHY_11n.hpp:362-398),
// T11nSigDemap (demapper11n.hpp:6-87) on the three SIG symbols.
EXPORT void ref_11n_siso_est(const int16_t* l0, const int16_t* l1, int16_t* ch)      // ch[2][64] complex
{
    static TSisoChannelEst<BB11nDemodContext>* est = new TSisoChannelEst<BB11nDemodContext>(BB11nDemodCtx);
    A16 COMPLEX16 a[128], b[128]; memcpy(a, l0, sizeof(a)); memcpy(b, l1, sizeof(b));
    TwoStreamPin<COMPLEX16, 128> pin = { { a, b }, true };
    est->Process(pin);
    memcpy(ch, BB11nDemodCtx.CF_Channel_11n::dot11a_siso_channel_1(), 256); memcpy(ch + 128, BB11nDemodCtx.CF_Channel_11n::dot11a_siso_channel_2(), 256);
    memset(ch + 2 * 28, 0, 32); memset(ch + 128 + 2 * 28, 0, 32);
}
struct CSink : public TSink<BB11nDemodContext> {                           // one stream of 64 COMPLEX16
    DEFINE_IPORT(COMPLEX16, 64); COMPLEX16 last[64];
    CSink(BB11nDemodContext& c) : TSink<BB11nDemodContext>(c) { memset(last, 0, sizeof(last)); }
    template <class P> bool Process(P& ipin) { while (ipin.check_read()) { memcpy(last, ipin.peek(), 256); ipin.pop(); } return true; }
};
EXPORT void ref_11n_siso_comp_mrc(const int16_t* ch, const int16_t* y0, const int16_t* y1, int16_t* x0, int16_t* x1, int16_t* mrc)
{
    typedef TCaptureSink2<BB11nDemodContext, 64> Sink;
    static Sink* sink = new Sink(BB11nDemodCtx);
    static TSisoChannelComp<BB11nDemodContext, Sink>* comp = new TSisoChannelComp<BB11nDemodContext, Sink>(BB11nDemodCtx, sink);
    memcpy(BB11nDemodCtx.CF_Channel_11n::dot11a_siso_channel_1(), ch, 256); memcpy(BB11nDemodCtx.CF_Channel_11n::dot11a_siso_channel_2(), ch + 128, 256);
    A16 COMPLEX16 a[64], b[64]; memcpy(a, y0, sizeof(a)); memcpy(b, y1, sizeof(b));
    TwoStreamPin<COMPLEX16, 64> pin = { { a, b }, true };
    comp->Process(pin);
    memcpy(x0, sink->last[0], 256); memcpy(x1, sink->last[1], 256);
    static CSink* s1 = new CSink(BB11nDemodCtx);
    static TMrcCombine<BB11nDemodContext, CSink>* m = new TMrcCombine<BB11nDemodContext, CSink>(BB11nDemodCtx, s1);
    A16 COMPLEX16 c[64], d[64]; memcpy(c, x0, 256); memcpy(d, x1, 256);
    TwoStreamPin<COMPLEX16, 64> pin2 = { { c, d }, true };
    m->Process(pin2);
    memcpy(mrc, s1->last, 256);
}
EXPORT void ref_11n_sig_demap(const int16_t* sym3, uint8_t* soft144)
{
    A16 COMPLEX16 x[192]; memcpy(x, sym3, sizeof(x));
    run_brick_once<T11nSigDemap, 192, COMPLEX16, 144>(x, soft144);
}

// T11aDeinterleaveBPSK x3 -> T11nViterbiSig (viterbi.hpp:51-99) -> T11nSigParser (PHY_11n.hpp:400-514) on the 144 soft values of
// one frame's L-SIG + HT-SIG.  The context fields the parser writes are zeroed first; fields[] = error_code, data_rate_kbps,
// frame_length, ht_frame_mcs, ht_frame_length, code_rate, total_symbols, remain_symbols, symbol_type.  Returns the parser's verdict.
EXPORT int ref_11n_sig_decode(const uint8_t* soft144, uint8_t* out9, uint32_t* fields)
{
    uint8_t di[144];
    for (int s = 0; s < 3; s++) run_brick_once<T11aDeinterleaveBPSK, 48, uchar, 48>(soft144 + 48 * s, di + 48 * s);
    run_brick_once<T11nViterbiSig, 144, uchar, 9>(di, out9);
    static T11nSigParser<BB11nDemodContext>* parser = new T11nSigParser<BB11nDemodContext>(BB11nDemodCtx);
    BB11nDemodContext& c = BB11nDemodCtx;
    c.CF_Error::error_code() = 0; c.CF_11aRxVector::data_rate_kbps() = 0; c.CF_11aRxVector::frame_length() = 0;
    c.CF_HTRxVector::ht_frame_mcs() = 0; c.CF_HTRxVector::ht_frame_length() = 0; c.CF_11aRxVector::code_rate() = 0;
    c.CF_11aRxVector::total_symbols() = 0; c.CF_11aRxVector::remain_symbols() = 0; c.CF_11nSymState::symbol_type() = 0;
    OneBurstPin<uchar, 9> pin = { out9, true };
    const bool ok = parser->Process(pin);
    fields[0] = c.CF_Error::error_code(); fields[1] = c.CF_11aRxVector::data_rate_kbps(); fields[2] = c.CF_11aRxVector::frame_length();
    fields[3] = c.CF_HTRxVector::ht_frame_mcs(); fields[4] = c.CF_HTRxVector::ht_frame_length(); fields[5] = c.CF_11aRxVector::code_rate();
    fields[6] = c.CF_11aRxVector::total_symbols(); fields[7] = c.CF_11aRxVector::remain_symbols(); fields[8] = c.CF_11nSymState::symbol_type();
    return ok ? 1 : 0;
}

// MimoAutoCorr (autocorr.hpp:5-147) and TCCA11n (cca_11n.hpp) over a run of 4-sample bursts of both RX chains at 20 MHz.
// ref_11n_autocorr: the four (auto-correlation energy, mean energy squared) pairs per burst from a fresh MimoAutoCorr.
EXPORT void ref_11n_autocorr(const int16_t* iq0, const int16_t* iq1, uint32_t nbursts, int64_t* acorr, int64_t* energy)
{
    MimoAutoCorr* core = new MimoAutoCorr();
    for (uint32_t b = 0; b < nbursts; b++) {
        A16 COMPLEX16 x0[4], x1[4]; memcpy(x0, iq0 + 8 * b, 16); memcpy(x1, iq1 + 8 * b, 16);
        vq a[2], e[2];
        core->CalcAutoCorrAndEnergy(*(vcs*)x0, *(vcs*)x1, a, e);
        memcpy(acorr + 4 * b, a, 32); memcpy(energy + 4 * b, e, 32);
    }
    delete core;
}
// ref_11n_cca: a fresh TCCA11n brick; after each detection the next `skip` bursts are withheld (the graph routes them to the frame
// bricks), then the context and the brick are Reset as RxThread does after a frame.  CS_TIMEOUT is cleared as RxThread clears it.
// Returns the number of detections; detect[k] = index of the burst in which detection k was raised.
EXPORT int ref_11n_cca(const int16_t* iq0, const int16_t* iq1, uint32_t nbursts, uint32_t skip, uint32_t* detect, int max_detect)
{
    TCCA11n<BB11nDemodContext>* cca = new TCCA11n<BB11nDemodContext>(BB11nDemodCtx);
    BB11nDemodCtx.Reset();
    int n = 0;
    for (uint32_t b = 0; b < nbursts; b++) {
        A16 COMPLEX16 x0[4], x1[4]; memcpy(x0, iq0 + 8 * b, 16); memcpy(x1, iq1 + 8 * b, 16);
        TwoStreamPin<COMPLEX16, 4> pin = { { x0, x1 }, true };
        cca->Process(pin);
        if (BB11nDemodCtx.CF_Error::error_code() == E_ERROR_CS_TIMEOUT) { BB11nDemodCtx.ResetCarrierSense(); cca->Reset(); }
        if (BB11nDemodCtx.CF_11CCA::cca_state() == CF_11CCA::power_detected) {
            if (n < max_detect) detect[n] = b;
            n++; b += skip;
            BB11nDemodCtx.Reset(); cca->Reset();
        }
    }
    return n;                                                        // the brick is left to the process (its allocator is not delete's)
}
