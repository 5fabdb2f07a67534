// ref_graph_mt_shim.cpp -- TEST INFRASTRUCTURE ONLY (not shipped, not linked into libsora_hip.so).
//
// The reference's 802.11a receive graph with its REAL thread boundary: CreateDemodGraph11a_40M as the reference has it
// (kernel/bb/demod11/fb11ademod_config.hpp:168-233), i.e. with TThreadSeparator (kernel/brick/inc/stdbrick.hpp:89-248)
// in front of the Viterbi sub-graph, the RxThread loop (fb11a_demod.cpp:29-81) on the calling thread and the
// ViterbiThread body (fb11a_demod.cpp:83-86: svit->Process(), called again and again by the thread wrapper) on a second,
// joined thread.  oracle/build_ref.sh compiles it into oracle/_ref/libsora_refgraph_mt.so from a scratch copy in which
// oracle/ref_flatten.py applies every patch EXCEPT the TThreadSeparator -> TNoInline substitution.  Its only use:
// tests/test_oracle_vs_refgraph.py shows that the two-thread harness reports the same events as the same-thread
// build (libsora_refgraph.so) that everything else is pinned to.
#include "MACStopwatch.h"
#include "stdbrick.hpp"
#include "fb11ademod_config.hpp"

#include <sched.h>

#define EXPORT extern "C" __attribute__((visibility("default")))

// core/inc/thread_if.h:22 declares it; in user mode the reference links it from its thread library (it yields the CPU)
extern "C" void SoraThreadYield(BOOLEAN) { sched_yield(); }

struct ref_frame { uint32_t error_code, sample_index, rate_kbps, length, crc32, mpdu_offset; };

static unsigned char g_out[4096];
static COMPLEX16* g_buf; static uint32_t g_cap;

struct VitThread { ISource* vit; volatile int stop; };
static void* viterbi_thread(void* p)                                     // ViterbiThread (fb11a_demod.cpp:83-86), re-entered by the thread wrapper until stopped
{
    VitThread* t = (VitThread*)p;
    while (!t->stop) t->vit->Process();
    return NULL;
}

EXPORT int ref_rx11a_capture_mt(const int16_t* iq, uint32_t nsamples40, ref_frame* res, int max_res, uint8_t* mpdu, uint32_t mpdu_cap)
{
    ISource* src = NULL; ISource* vit = NULL; IControlPoint* cs = NULL;
    if (g_cap < nsamples40 + 64) { free(g_buf); g_cap = nsamples40 + 64; g_buf = (COMPLEX16*)aligned_alloc(16, ((size_t)g_cap * 4 + 15) & ~(size_t)15); }
    memcpy(g_buf, iq, (size_t)nsamples40 * 4);
    BB11aDemodCtx.Init(g_buf, nsamples40 * sizeof(COMPLEX16), g_out, sizeof(g_out));
    CreateDemodGraph11a_40M(src, vit, cs);                               // a fresh graph per capture, as the harness has
    if (!vit) return -2;                                                 // built without the separator: not the two-thread graph
    VitThread vt; vt.vit = vit; vt.stop = 0;
    pthread_t viterbi;
    if (pthread_create(&viterbi, NULL, viterbi_thread, &vt) != 0) { IReferenceCounting::Release(src); return -3; }
    src->Reset();                                                        // Test11A_FB_Demod: after the Viterbi thread has started
    BB11aDemodCtx.Reset();
    int n = 0; uint32_t used = 0; uint nWaitCounter = 12;
    for (;;) {                                                           // RxThread
        bool rc = src->Process();
        ulong err = BB11aDemodCtx.CF_Error::error_code();
        if (err != E_ERROR_SUCCESS) {
            if (err == E_ERROR_CS_TIMEOUT) {
                BB11aDemodCtx.ResetCarrierSense(); cs->Reset();
                if (nWaitCounter > 0) { nWaitCounter--; continue; }
                nWaitCounter = 12; continue;
            }
            if (n < max_res) {
                ref_frame& f = res[n++];
                f.error_code = err; f.sample_index = BB11aDemodCtx.CF_MemSamples::mem_sample_index();
                f.rate_kbps = BB11aDemodCtx.CF_11aRxVector::data_rate_kbps(); f.length = BB11aDemodCtx.CF_11aRxVector::frame_length();
                f.crc32 = BB11aDemodCtx.CF_11aRxVector::crc32(); f.mpdu_offset = used;
                if ((err == E_ERROR_FRAME_OK || err == E_ERROR_CRC32_FAIL) && used + f.length <= mpdu_cap) { memcpy(mpdu + used, g_out, f.length); used += f.length; }
            }
            src->Flush(); BB11aDemodCtx.Reset(); src->Reset();
        }
        if (!rc) break;
    }
    vt.stop = 1;
    pthread_join(viterbi, NULL);
    IReferenceCounting::Release(src);
    return n;
}

// The two-thread harness as bench.py times it (cpu_baseline.two_thread_value): the graph and the ViterbiThread are created ONCE -- as Test11A_FB_Demod creates them
// once per dump (fb11a_demod.cpp:88-120) -- and `ncap` equal-sized captures laid end to end go through the RxThread loop `reps` times; only the number of
// frames with a good FCS comes back.  Two host cores are busy: RxThread here, ViterbiThread spinning on the separator's queue as the reference's does.
EXPORT uint32_t ref_rx11a_bench_mt(const int16_t* iq, uint32_t ncap, uint32_t nsamples40, uint32_t reps)
{
    static ISource* src; static ISource* vit; static IControlPoint* cs; static VitThread vt; static pthread_t viterbi; static int started;
    if (g_cap < nsamples40 + 64) { free(g_buf); g_cap = nsamples40 + 64; g_buf = (COMPLEX16*)aligned_alloc(16, ((size_t)g_cap * 4 + 15) & ~(size_t)15); }
    uint32_t ok = 0;
    for (uint32_t r = 0; r < reps; r++)
        for (uint32_t c = 0; c < ncap; c++) {
            memcpy(g_buf, iq + (size_t)c * nsamples40 * 2, (size_t)nsamples40 * 4);
            BB11aDemodCtx.Init(g_buf, nsamples40 * sizeof(COMPLEX16), g_out, sizeof(g_out));
            if (!src) {
                CreateDemodGraph11a_40M(src, vit, cs);
                if (!vit) return 0xFFFFFFFFu;
                vt.vit = vit; vt.stop = 0;
                if (pthread_create(&viterbi, NULL, viterbi_thread, &vt) != 0) return 0xFFFFFFFEu;
                started = 1;
            } else src->Seek(ISource::START_POS);
            src->Flush(); BB11aDemodCtx.Reset(); src->Reset();
            uint nWaitCounter = 12;
            for (;;) {
                bool rc = src->Process();
                ulong err = BB11aDemodCtx.CF_Error::error_code();
                if (err != E_ERROR_SUCCESS) {
                    if (err == E_ERROR_CS_TIMEOUT) {
                        BB11aDemodCtx.ResetCarrierSense(); cs->Reset();
                        if (nWaitCounter > 0) { nWaitCounter--; continue; }
                        nWaitCounter = 12; continue;
                    }
                    ok += err == E_ERROR_FRAME_OK;
                    src->Flush(); BB11aDemodCtx.Reset(); src->Reset();
                }
                if (!rc) break;
            }
        }
    (void)started;
    return ok;
}
