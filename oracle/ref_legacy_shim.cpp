// ref_legacy_shim.cpp -- TEST INFRASTRUCTURE ONLY: the reference's LEGACY 802.11a receiver, compiled from its own sources where they lie under
// /root/reference (oracle/build_ref.sh -> oracle/_ref/libsora_reflegacy.so), as the second cross-check oracle of SURVEY.md section 8 f4:
//   kernel/bb/dot11a/dot11/{a_init,arx_cs,arx_fd,arx_bg1,arx_vdc6..54}.c, kernel/bb/dot11a/mod/{viterbi,ademap,async}.c, the static look-up
//   tables kernel/bb/dot11a/lutst/*.c -- the C path `demod11 -d` runs WITHOUT --802.11a.brick (kernel/bb/demod11/demod11a.cpp).
// It is a different implementation from the BRICK graph the product is held to (own carrier sense and symbol sync, a 4-bit channel
// estimate scaling bb/mod/achannel.h:10-13, Viterbi windows of 36 / 216 columns instead of 24 / 256, a real second thread): its decoded MPDUs
// must agree with the brick graph's wherever both decode, which is what tests/test_oracle_legacy.py checks; event positions are not comparable.
// The sources are compiled as C++ (the reference builds them with /TP), one translation unit, in the order the WDK `sources` file lists them.
#include "dot11/a_init.c"
#include "dot11/arx_fd.c"
#include "dot11/arx_cs.c"
#include "dot11/arx_bg1.c"
#include "dot11/arx_vdc6.c"
#undef TB_DEPTH
#undef TB_OUTPUT
#undef NOR_MASK
#include "dot11/arx_vdc9.c"
#undef TB_DEPTH
#undef TB_OUTPUT
#undef NOR_MASK
#include "dot11/arx_vdc12.c"
#undef TB_DEPTH
#undef TB_OUTPUT
#undef NOR_MASK
#include "dot11/arx_vdc18.c"
#undef TB_DEPTH
#undef TB_OUTPUT
#undef NOR_MASK
#include "dot11/arx_vdc24.c"
#undef TB_DEPTH
#undef TB_OUTPUT
#undef NOR_MASK
#include "dot11/arx_vdc36.c"
#undef TB_DEPTH
#undef TB_OUTPUT
#undef NOR_MASK
#include "dot11/arx_vdc48.c"
#undef TB_DEPTH
#undef TB_OUTPUT
#undef NOR_MASK
#include "dot11/arx_vdc54.c"
#include "dot11/viterbi.c"
#include "dot11/ademap.c"
#include "dot11/async.c"
#include "dot11/44MTo40M.c"
extern "C" {
#include "lutst/arg.c"
#include "lutst/atan64.c"
#include "lutst/cos0xffff.c"
#include "lutst/sin0xffff.c"
#include "lutst/pilotsgn.c"
#include "lutst/scramble_11a.c"
}

#include <pthread.h>
// CsFrameDemod (kernel/bb/demod11/demod11a.cpp:52-185) over a buffer of 40 MHz samples: carrier sense, frame demodulation, the Viterbi worker on
// its own thread (AllocStartThread(BB11ARxViterbiWorker, ..)), until the stream runs dry.
static volatile FLAG g_work = 1;
static void* vit_thread(void* ctx) { while (g_work) { BB11ARxViterbiWorker(ctx); _mm_pause(); } return 0; }
struct LegacyEvent { int hr; unsigned rate_code, length, crc_ok; unsigned long long block_pos; };
extern "C" __attribute__((visibility("default")))
int ref_legacy_rx11a(const COMPLEX16* iq40, unsigned nsamples, LegacyEvent* ev, int max_ev, unsigned char* frames, unsigned frame_cap)
{
    static BB11A_RX_CONTEXT* ctx = 0;
    static unsigned char* fbuf = 0;
    if (!ctx) { ctx = (BB11A_RX_CONTEXT*)_aligned_malloc(sizeof(BB11A_RX_CONTEXT), 64); fbuf = (unsigned char*)malloc(1 << 20); }
    memset(ctx, 0, sizeof(*ctx));
    _SORA_RADIO_RX_STREAM st; st.base = iq40; st.nblocks = nsamples / 28; st.pos = 0;
    g_work = 1;
    BB11ARxContextInit(ctx, 40, 250000, 150, 112, (PFLAG)&g_work);               // PrepareRxContext (demod11a.cpp:47-51)
    pthread_t th; pthread_create(&th, 0, vit_thread, ctx);
    int n = 0; unsigned used = 0; bool finished = false;
    while (!finished) {
        HRESULT hr = BB11ARxCarrierSense(ctx, &st);
        switch (hr) {
        case BB11A_CHANNEL_CLEAN: case BB11A_E_PD_LAG: break;
        case BB11A_OK_POWER_DETECTED:
            BB11APrepareRx(ctx, (char*)fbuf, 1 << 20);
            hr = BB11ARxFrameDemod(ctx, &st);
            if (hr == E_FETCH_SIGNAL_HW_TIMEOUT) { finished = true; break; }
            if (n < max_ev) {
                LegacyEvent& e = ev[n++];
                e.hr = (int)hr; e.rate_code = ctx->bRate; e.length = ctx->__usLength; e.crc_ok = hr == BB11A_OK_FRAME; e.block_pos = st.pos;
                if ((hr == BB11A_OK_FRAME || hr == BB11A_E_CRC32) && ctx->__usLength >= 4 && used + ctx->__usLength <= frame_cap) { memcpy(frames + used, fbuf, ctx->__usLength); used += ctx->__usLength; }
            }
            break;
        default: finished = true; break;                                            // E_FETCH_SIGNAL_HW_TIMEOUT, E_INVALIDARG (demod11a.cpp:176-179)
        }
    }
    g_work = 0; pthread_join(th, 0);
    BB11ARxContextCleanup(ctx);
    return n;
}
