/* TEST INFRASTRUCTURE -- CPU restatement of the reference's 802.11n 2x2 receive graph.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product path never does.
 *
 * CreateDemodGraph11n (kernel/bb/demod11/fb11ndemod_config.hpp:166-257) driven as RxThread drives it (fb11n_demod.cpp:30-85):
 *   TMemSamples2 -> TDownSample2 -> RxSwitch -> TCCA11n                                              (carrier sense)
 *                                            -> TFreqEstimator_11n -> TFreqComp_11n -> 2 x TFFT64 -> TSisoChannelEst   (L-LTF)
 *                                            -> TFreqComp_11n -> T11nDataSymbol -> 2 x TFFT64 -> T11nSymSel
 *        SIG:    TSisoChannelComp -> TMrcCombine -> T11nSigDemap -> T11aDeinterleaveBPSK -> T11nViterbiSig -> T11nSigParser
 *        HT-STF: dropped;   HT-LTF: TMimoChannelEst
 *        DATA:   TMimoChannelComp -> TPilotTrack_11n -> T11nDemap* -> T11nDeinterleave*_S0/_S1 -> TStreamJoin -> TStreamConcat
 *                -> T11aViterbi<5000*8, 312, 192, 36> -> T11aDesc -> TBB11aFrameSink
 * The stage functions are those of so_11n.c / so_rx11a.c; this file adds MimoAutoCorr + TCCA11n (autocorr.hpp:5-147, cca_11n.hpp:25-170)
 * and the glue (queues, symbol state machine, frame reset).  Pinned against the compiled reference graph (oracle/_ref,
 * ref_rx11n_capture) in tests/test_oracle_11n_graph.py and by the recorded events in tests/golden/refgraph_11n.npz. */
#include <stdlib.h>
#include <string.h>
#include "so_oracle.h"
#include "so_internal.h"

/* ------------------------------------------------------------------ MimoAutoCorr: 32-sample moving sums, per RX chain */
typedef struct {
    so_c16  his[2][32];                 /* vHisSample / vHisSample2 */
    int32_t hcr[2][32], hci[2][32];     /* vHisCorr: the products still inside the window */
    int32_t he[2][32];                  /* vHisEnergy */
    int32_t sr[2], si[2], se[2];        /* running sums (vAverageCorrSum / vAverageEnergySum, last lane) */
    int     idx;                        /* vHisIdx, in vectors of four */
} acorr_t;

void so_autocorr11n_reset(void* p) { memset(p, 0, sizeof(acorr_t)); }
size_t so_autocorr11n_size(void) { return sizeof(acorr_t); }

/* CalcAutoCorrAndEnergy (autocorr.hpp:49-74): four samples of both chains -> acorr[4] = |mean of the two chains' moving
 * auto-correlation|^2, energy[4] = (mean of the two chains' moving energy)^2, everything >> 5 on the way in (vShift = log2(32)) */
void so_autocorr11n_burst(void* p, const so_c16 x0[4], const so_c16 x1[4], int64_t acorr[4], int64_t energy[4])
{
    acorr_t* a = (acorr_t*)p;
    const so_c16* x[2] = { x0, x1 };
    int32_t pr[2][4], pi[2][4], pe[2][4];
    for (int r = 0; r < 2; r++)
        for (int j = 0; j < 4; j++) {
            const int s = a->idx * 4 + j;
            int32_t cr, ci;
            so_conj_mul32(x[r][j], a->his[r][s], &cr, &ci);                /* conj_mul: x * conj(delayed x) */
            cr >>= 5; ci >>= 5;
            a->his[r][s] = x[r][j];
            a->sr[r] = so_w32((int64_t)a->sr[r] + so_w32((int64_t)cr - a->hcr[r][s]));
            a->si[r] = so_w32((int64_t)a->si[r] + so_w32((int64_t)ci - a->hci[r][s]));
            a->hcr[r][s] = cr; a->hci[r][s] = ci;
            pr[r][j] = a->sr[r]; pi[r][j] = a->si[r];
            const int32_t e = so_sqnorm(x[r][j]) >> 5;
            a->se[r] = so_w32((int64_t)a->se[r] + so_w32((int64_t)e - a->he[r][s]));
            a->he[r][s] = e;
            pe[r][j] = a->se[r];
        }
    for (int j = 0; j < 4; j++) {
        const int32_t re = so_w32((int64_t)(pr[0][j] >> 1) + (pr[1][j] >> 1)), im = so_w32((int64_t)(pi[0][j] >> 1) + (pi[1][j] >> 1));
        acorr[j] = (int64_t)((uint64_t)((int64_t)re * re) + (uint64_t)((int64_t)im * im));
        const int32_t v = so_w32((int64_t)(pe[0][j] >> 1) + (pe[1][j] >> 1));
        energy[j] = (int64_t)v * v;
    }
    a->idx = (a->idx + 1) % 8;
}

/* ------------------------------------------------------------------ TCCA11n */
typedef struct {
    acorr_t core;
    int64_t his_e[64]; int his_index;   /* energy 64 samples ago; LLONG_MAX until written (cca_11n.hpp:157) */
    int peak_found, peak_count; uint32_t sense_count;
} cca_t;

static void cca_init(cca_t* c)
{
    memset(c, 0, sizeof(*c));
    for (int i = 0; i < 64; i++) c->his_e[i] = INT64_MAX;
}
static void cca_reset(cca_t* c) { c->sense_count = 0; c->peak_found = 0; c->peak_count = 0; }     /* _reset(), cca_11n.hpp:164-169 */

/* one burst (cca_11n.hpp:25-131): returns 1 when the plateau of the L-STF ended (OnPowerDetected); *timeout = carrier-sense timeout */
static int cca_burst(cca_t* c, const so_c16 x0[4], const so_c16 x1[4], int* timeout)
{
    int64_t acorr[4], energy[4];
    int detected = 0;
    so_autocorr11n_burst(&c->core, x0, x1, acorr, energy);
    for (int i = 0; i < 4; i++) {
        const int64_t den = (int64_t)((uint64_t)c->his_e[c->his_index] + 1u);     /* LLONG_MAX + 1 wraps, as the compiled code does */
        const int64_t eb = (den == -1 && energy[i] == INT64_MIN) ? 0 : energy[i] / (den == 0 ? 1 : den);
        if (!c->peak_found) {
            c->sense_count += 1;
            if (eb > 5 && acorr[i] > (energy[i] >> 1)) { c->sense_count = 0; c->peak_count++; c->peak_found = 1; }
            else c->peak_count = 0;
        } else if (acorr[i] < (energy[i] >> 3)) {
            const int good = c->peak_count > 96 && c->peak_count < 160;
            c->peak_found = 0; c->peak_count = 0;
            if (good) { detected = 1; break; }                                 /* the rest of the burst is not looked at, nor recorded */
        } else {
            c->peak_count++;
            if (c->peak_count > 160) { c->peak_found = 0; c->peak_count = 0; }
        }
        c->his_e[c->his_index++] = energy[i];
        c->his_index %= 64;
    }
    *timeout = (c->sense_count >= 84 && !detected);
    return detected;
}

/* test hook: the detections of a run of bursts, `skip` bursts withheld after each (what the graph routes to the frame bricks) */
int so_cca11n(const so_c16* iq0, const so_c16* iq1, uint32_t nbursts, uint32_t skip, uint32_t* detect, int max_detect)
{
    cca_t* c = (cca_t*)malloc(sizeof(cca_t)); int n = 0, to;
    cca_init(c);
    for (uint32_t b = 0; b < nbursts; b++) {
        if (cca_burst(c, iq0 + 4 * b, iq1 + 4 * b, &to)) {
            if (n < max_detect) detect[n] = b;
            n++; b += skip; cca_reset(c);
        } else if (to) cca_reset(c);
    }
    free(c);
    return n;
}

/* ------------------------------------------------------------------ the graph */
enum { SYM_L_LTF = 1, SYM_SIG, SYM_HT_STF, SYM_HT_LTF, SYM_DATA };       /* CF_11nSymState (ieee80211facade.hpp:274-289) */

typedef struct {
    cca_t cca;
    uint32_t error_code; int cca_detected; int symbol_type;
    int16_t vfo[24];                                  /* CF_FreqOffset_11n: vfo_delta_i | vfo_step_i | vfo_theta_i */
    so_c16 ch[2][64];                                 /* CF_Channel_11n::dot11a_siso_channel_{1,2} */
    so_c16 h[2][128], hinv[2][128];                   /* TMimoChannelEst */
    uint32_t fields[9];                               /* T11nSigParser's context fields (so_sig_decode11n order) */
    uint16_t remain_symbols; uint32_t mcs, ht_length, code_rate;
    so_c16 lq[2][128]; int ln;                        /* TFreqEstimator_11n's input queue */
    so_c16 fq[2][8]; int fn;                          /* TFreqComp_11n's */
    so_c16 sq[2][80]; int sn;                         /* T11nDataSymbol's */
    so_c16 sig[192]; int nsig;                        /* T11nSigDemap's (three MRC symbols) */
    so_c16 ltf[2][128]; int nltf;                     /* TMimoChannelEst's (two HT-LTF symbols per chain) */
    uint8_t* soft; uint32_t soft_n, soft_cap;
    uint32_t frame_crc;
    so_frame_result* res; int nres, max_res;
    uint8_t* mpdu_buf; uint32_t mpdu_used, mpdu_cap;
} rx_t;

static void frame_reset(rx_t* rx)                     /* ssrc->Flush(); BB11nDemodCtx.Reset(); ssrc->Reset() (fb11n_demod.cpp:60-66) */
{
    rx->error_code = SO_E_SUCCESS; rx->cca_detected = 0; rx->symbol_type = SYM_L_LTF;
    rx->remain_symbols = 0;
    cca_reset(&rx->cca);
    rx->ln = rx->fn = rx->sn = rx->nsig = rx->nltf = 0; rx->soft_n = 0;
}

static void soft_push(rx_t* rx, const uint8_t* p, uint32_t n)
{
    if (rx->soft_n + n > rx->soft_cap) { rx->soft_cap = (rx->soft_n + n) * 2 + 1024; rx->soft = (uint8_t*)realloc(rx->soft, rx->soft_cap); }
    memcpy(rx->soft + rx->soft_n, p, n); rx->soft_n += n;
}

/* one OFDM symbol of both chains behind T11nDataSymbol (PHY_11n.hpp:312-356): CP dropped, TFFT64 each, T11nSymSel */
static void ofdm_symbol(rx_t* rx)
{
    so_c16 y0[64], y1[64];
    so_fft64(rx->sq[0] + 16, y0); so_fft64(rx->sq[1] + 16, y1);
    switch (rx->symbol_type) {
    case SYM_SIG: {
        so_c16 x0[64], x1[64];
        so_siso_comp11n((const so_c16 (*)[64])rx->ch, y0, y1, x0, x1);
        so_mrc11n(x0, x1, rx->sig + 64 * rx->nsig);
        if (++rx->nsig == 3) {
            uint8_t soft[144], out9[9];
            rx->nsig = 0;
            so_sig_demap11n(rx->sig, soft);
            if (so_sig_decode11n(soft, out9, rx->fields)) {
                rx->mcs = rx->fields[3]; rx->ht_length = rx->fields[4]; rx->code_rate = rx->fields[5];
                rx->remain_symbols = (uint16_t)rx->fields[7];
                rx->symbol_type = SYM_HT_STF;
            } else {
                rx->error_code = SO_E_PLCP_HEADER_FAIL;
            }
        }
        break; }
    case SYM_HT_STF: rx->symbol_type = SYM_HT_LTF; break;                    /* sym_selector_11n, then TDropAny */
    case SYM_HT_LTF:
        memcpy(rx->ltf[0] + 64 * rx->nltf, y0, sizeof(y0)); memcpy(rx->ltf[1] + 64 * rx->nltf, y1, sizeof(y1));
        if (++rx->nltf == 2) { rx->nltf = 0; so_mimo_est11n(rx->ltf[0], rx->ltf[1], rx->h, rx->hinv); rx->symbol_type = SYM_DATA; }
        break;
    default: {
        if (rx->error_code != SO_E_SUCCESS) break;                           /* T11aViterbi drops its input once the frame is over */
        so_c16 x0[64], x1[64];
        const int nb = rx->mcs == 8 ? 1 : 2;                                 /* rate_selector: MCS 8 BPSK, 9 and 10 QPSK */
        uint8_t s0[104], s1[104], d0[104], d1[104], joined[208];
        so_mimo_comp11n((const so_c16 (*)[128])rx->hinv, y0, y1, x0, x1);
        so_pilot_track11n(rx->vfo + 16, x0, x1);
        so_demap11n(nb, x0, s0); so_deinterleave11n(nb, 0, s0, d0);
        so_demap11n(nb, x1, s1); so_deinterleave11n(nb, 1, s1, d1);
        for (int k = 0; k < 52 * nb; k++) { joined[2 * k] = d0[k]; joined[2 * k + 1] = d1[k]; }     /* TStreamJoin<2,52nb> -> TStreamConcat<2,1> */
        soft_push(rx, joined, (uint32_t)(104 * nb));
        break; }
    }
    rx->remain_symbols--;                                                    /* PHY_11n.hpp:331 (ushort, wraps before the parser sets it) */
    if (rx->remain_symbols == 0 && rx->error_code == SO_E_SUCCESS) {
        /* Next()->Flush(): the padded last burst takes the Viterbi past frame_length * 8 + 16 + 6 steps */
        uint8_t* dec = (uint8_t*)malloc((size_t)rx->ht_length + 64);
        uint8_t* tmp = NULL; uint8_t* mpdu = rx->mpdu_buf + rx->mpdu_used;
        if (rx->mpdu_used + rx->ht_length > rx->mpdu_cap) { tmp = (uint8_t*)malloc((size_t)rx->ht_length + 8); mpdu = tmp; }
        so_viterbi_frame_ex(rx->soft, rx->soft_n, (int)rx->code_rate, rx->ht_length, dec, 192, 36);
        rx->error_code = so_desc_sink(dec, rx->ht_length, mpdu, &rx->frame_crc);
        free(dec); free(tmp);
    }
}

/* RxSwitch (fb11ndemod_config.hpp:102-114) for one 4-sample burst of both chains */
static void push_burst(rx_t* rx, const so_c16 x0[4], const so_c16 x1[4])
{
    if (!rx->cca_detected) {
        int to;
        if (cca_burst(&rx->cca, x0, x1, &to)) rx->cca_detected = 1;
        else if (to && rx->error_code == SO_E_SUCCESS) rx->error_code = SO_E_CS_TIMEOUT;
        return;
    }
    if (rx->symbol_type == SYM_L_LTF) {
        memcpy(rx->lq[0] + rx->ln, x0, 16); memcpy(rx->lq[1] + rx->ln, x1, 16); rx->ln += 4;
        if (rx->ln == 128) {
            so_c16 c0[128], c1[128], l0[128], l1[128];
            rx->ln = 0;
            so_cfo_est11n(rx->lq[0], rx->lq[1], rx->vfo);
            so_freq_comp11n(rx->vfo, rx->lq[0], rx->lq[1], c0, c1, 16);
            so_fft64(c0, l0); so_fft64(c0 + 64, l0 + 64); so_fft64(c1, l1); so_fft64(c1 + 64, l1 + 64);
            so_siso_est11n(l0, l1, rx->ch);
            rx->symbol_type = SYM_SIG;
        }
        return;
    }
    memcpy(rx->fq[0] + rx->fn, x0, 16); memcpy(rx->fq[1] + rx->fn, x1, 16); rx->fn += 4;
    if (rx->fn == 8) {
        rx->fn = 0;
        so_freq_comp11n(rx->vfo, rx->fq[0], rx->fq[1], rx->sq[0] + rx->sn, rx->sq[1] + rx->sn, 1);
        rx->sn += 8;
        if (rx->sn == 80) { rx->sn = 0; ofdm_symbol(rx); }
    }
}

/* The end of the capture: TMemSamples2 finds nothing left and flushes the graph (memsource.hpp:212-216).  A flush pads every
 * partly filled pin queue on the way down with zero items and processes it (brick.h:461, pinqueue.h:133-145), along the ports the
 * selectors pick at that moment -- so a frame cut short by the end of the capture can still raise its event from zero-padded
 * symbols.  (The same flush after a frame event changes nothing that is reported: the context is Reset right after it.) */
static void flush_graph(rx_t* rx)
{
    static const so_c16 z[4] = { {0, 0}, {0, 0}, {0, 0}, {0, 0} };
    if (!rx->cca_detected) return;                                           /* RxSwitch flushes the port it would route to: TCCA11n, a sink */
    if (rx->symbol_type == SYM_L_LTF) { while (rx->ln) push_burst(rx, z, z); return; }
    if (rx->fn) push_burst(rx, z, z);                                        /* TFreqComp_11n's queue, then T11nDataSymbol's */
    while (rx->sn) push_burst(rx, z, z);
    switch (rx->symbol_type) {                                               /* T11nSymSel::Flush */
    case SYM_SIG:
        if (rx->nsig) {                                                      /* T11nSigDemap's queue holds MRC output: the missing symbols are zeros */
            uint8_t soft[144], out9[9];
            memset(rx->sig + 64 * rx->nsig, 0, (size_t)(3 - rx->nsig) * 64 * sizeof(so_c16)); rx->nsig = 0;
            so_sig_demap11n(rx->sig, soft);
            if (!so_sig_decode11n(soft, out9, rx->fields)) rx->error_code = SO_E_PLCP_HEADER_FAIL;
        }
        break;
    case SYM_DATA:
        if (rx->error_code == SO_E_SUCCESS && rx->soft_n % 312) {            /* T11aViterbi's input burst, padded with zero soft values */
            const uint32_t n = (rx->soft_n + 311) / 312 * 312;
            uint8_t* padded = (uint8_t*)calloc(n, 1); uint8_t* dec = (uint8_t*)malloc((size_t)rx->ht_length + 64);
            uint8_t* tmp = NULL; uint8_t* mpdu = rx->mpdu_buf + rx->mpdu_used;
            if (rx->mpdu_used + rx->ht_length > rx->mpdu_cap) { tmp = (uint8_t*)malloc((size_t)rx->ht_length + 8); mpdu = tmp; }
            memcpy(padded, rx->soft, rx->soft_n);
            if (so_viterbi_frame_ex(padded, n, (int)rx->code_rate, rx->ht_length, dec, 192, 36) == (int)rx->ht_length + 2)
                rx->error_code = so_desc_sink(dec, rx->ht_length, mpdu, &rx->frame_crc);
            free(padded); free(dec); free(tmp);
        }
        break;
    default: break;                                                          /* HT-STF: dropped; HT-LTF: a channel estimate nobody uses */
    }
}

/* iq0 / iq1: the two RX chains at 40 MHz, nsamples each (a whole number of 28-sample source bursts keeps the last burst free of
 * stale queue memory, memsource.hpp:205-222).  Events as ref_rx11n_capture reports them: rate_kbps carries the MCS index. */
int so_rx11n_capture(const so_c16* iq0, const so_c16* iq1, uint32_t nsamples, so_frame_result* res, int max_res, uint8_t* mpdu_buf, uint32_t mpdu_cap)
{
    rx_t* rx = (rx_t*)calloc(1, sizeof(rx_t));
    rx->res = res; rx->max_res = max_res; rx->mpdu_buf = mpdu_buf; rx->mpdu_cap = mpdu_cap;
    cca_init(&rx->cca);
    frame_reset(rx);
    /* TMemSamples2 appends 28 raw samples per chain and call; TDownSample2 pops them in eights and keeps the even ones (samples.hpp:27-47);
     * the queue between them holds lcm(28, 8) = 56 */
    so_c16 q[2][56]; memset(q, 0, sizeof(q));
    uint32_t w = 0, r = 0, src = 0, remain = nsamples;
    int ret = 1;
    while (ret) {
        if (remain > 28) { memcpy(q[0] + w, iq0 + src, 28 * sizeof(so_c16)); memcpy(q[1] + w, iq1 + src, 28 * sizeof(so_c16)); w += 28; src += 28; remain -= 28; }
        else if (remain == 0) {
            ret = 0;
            if (w - r) {                                                     /* TDownSample2's queue: 4 raw samples left of a 28-sample call */
                so_c16 a[4], b[4];
                for (int e = 0; e < 4; e++) { const uint32_t i = r + 2 * e; a[e] = i < w ? q[0][i] : so_c(0, 0); b[e] = i < w ? q[1][i] : so_c(0, 0); }
                r = w = 0;
                push_burst(rx, a, b);
            }
            flush_graph(rx);
        }
        else { memcpy(q[0] + w, iq0 + src, remain * sizeof(so_c16)); memcpy(q[1] + w, iq1 + src, remain * sizeof(so_c16)); w += 28; src += remain; remain = 0; }
        if (ret)
            while (w - r >= 8) {
                so_c16 a[4], b[4];
                for (int e = 0; e < 4; e++) { a[e] = q[0][r + 2 * e]; b[e] = q[1][r + 2 * e]; }
                r += 8;
                if (r == w) r = w = 0;
                push_burst(rx, a, b);
            }
        const uint32_t err = rx->error_code;
        if (err != SO_E_SUCCESS) {
            if (err == SO_E_CS_TIMEOUT) { rx->error_code = SO_E_SUCCESS; cca_reset(&rx->cca); }      /* ResetCarrierSense(); scs->Reset() */
            else {
                if (rx->nres < rx->max_res) {
                    so_frame_result* f = &rx->res[rx->nres++];
                    memset(f, 0, sizeof(*f));
                    f->error_code = err; f->end_sample = src;                      /* 40 MHz source position as RxThread sees the event */
                    if (err != SO_E_PLCP_HEADER_FAIL) {
                        f->rate_kbps = rx->mcs; f->length = (uint16_t)rx->ht_length; f->crc32 = rx->frame_crc; f->mpdu_offset = rx->mpdu_used;
                        if (rx->mpdu_used + rx->ht_length <= rx->mpdu_cap) rx->mpdu_used += rx->ht_length;
                    }
                }
                w = r = 0;
                frame_reset(rx);
            }
        }
    }
    const int n = rx->nres;
    free(rx->soft); free(rx);
    return n;
}
