/* so_rx11a.c -- 802.11a receive chain of the oracle (TEST INFRASTRUCTURE).
 *
 * Restates, brick by brick and single-threaded, the demod graph
 *   kernel/bb/demod11/fb11ademod_config.hpp:168-233   (CreateDemodGraph11a_40M)
 * and the offline frame loop kernel/bb/demod11/fb11a_demod.cpp:29-81 (RxThread).
 * The reference runs the Viterbi sub-graph on a second thread behind TThreadSeparator
 * (brick/inc/stdbrick.hpp:89-248); data-wise that is a FIFO, so it is restated here in line.
 */
#include <string.h>
#include <stdlib.h>
#include "so_oracle.h"
#include "so_internal.h"

/* ------------------------------------------------------------------ context reset */
void so_rx11a_ctx_reset(so_rx11a_ctx* c)
{
    /* ieee80211facade.hpp:166-236: CF_CFOffset, CF_Channel_11a, CF_FreqCompensate, CF_PhaseCompensate, CF_PilotTrack */
    memset(c, 0, sizeof(*c));
    for (int i = 0; i < 64; i++) c->CompCoeffs[i].re = 0x7fff;
    c->symbol_count = 127;
}

/* ------------------------------------------------------------------ T11aLTS (channel_11a.hpp:33-230) */
static const char LTS_Sequence_11a[64] = {       /* channel_11a.hpp:13-18 (= 802.11a L-LTF signs, FFT order) */
    0,1,0,0,1,1,0,1,0,1,0,0,0,0,0,1, 1,0,0,1,0,1,0,1,1,1,1,0,0,0,0,0,
    0,0,0,0,0,0,1,1,0,0,1,1,0,1,0,1, 1,1,1,1,1,0,0,1,1,0,1,0,1,1,1,1 };

void so_lts(so_rx11a_ctx* c, const so_c16 in144[144])
{
    so_init();
    const int16_t *usin = so_usin_lut(), *ucos = so_ucos_lut();
    so_c16 x[136];
    memcpy(x, in144 + 8, sizeof(x));                                   /* skip_cp = 8 (:211) */
    for (int i = 0; i < 64; i++) x[i] = so_sra(x[i], 1);               /* rep_shift_right<16>(pvi,pvi,1) (:216) */

    /* FreqOffsetEstimate<16>(input, input+16): brick/inc/dspalg.hpp:226-243 */
    int sum_re = 0, sum_im = 0;
    for (int v = 0; v < 16; v++) {
        int32_t sr = 0, si = 0;
        for (int k = 0; k < 4; k++) {
            int32_t re, im;
            so_conj_mul32(x[64 + 4 * v + k], x[4 * v + k], &re, &im);
            sr = so_w32((int64_t)sr + (re >> 5));
            si = so_w32((int64_t)si + (im >> 5));
        }
        sum_re = so_w32((int64_t)sum_re + sr); sum_im = so_w32((int64_t)sum_im + si);
    }
    int16_t arg = so_uatan2(sum_im, sum_re);
    /* "arg / (LEN*vcs::size)": the divisor is a size_t, so arg is converted to UNSIGNED first and the quotient, cut
     * back to 16 bits, is the FLOOR of arg/64 (-438 -> -7, not -6) -- on 32- and 64-bit builds alike (dspalg.hpp:242) */
    c->CFO_est = (int16_t)(arg >> 6);
    c->CFO_comp = c->SFO_comp = 0; c->CFO_tracker = c->SFO_tracker = 0;/* :107-112 */

    /* BuildFrequencyShiftCoeffs<64>(FreqCoeffs, 0, CFO_est): dspalg.hpp:200-208 */
    int16_t ph = 0;
    for (int i = 0; i < 64; i++) {
        c->FreqCoeffs[i].re = ucos[(uint16_t)ph];
        c->FreqCoeffs[i].im = (int16_t)(-usin[(uint16_t)ph]);
        ph = (int16_t)(ph + c->CFO_est);
    }
    for (int i = 0; i < 64; i++) x[i] = so_mul_q15(x[i], c->FreqCoeffs[i]);   /* FrequencyShift (:120) */

    /* _channel_estimation (:125-178) */
    so_c16 Y[64];
    so_fft64(x, Y);
    for (int i = 0; i < 64; i++) {
        if (i >= 28 && i < 36) continue;                               /* vcs 7,8 untouched */
        int32_t e = so_sqnorm(Y[i]) >> 8;                              /* norm_shift = 8 */
        so_c16 L = so_c(LTS_Sequence_11a[i] ? 1600 : -1600, 0);        /* norm_one = 1600 */
        int32_t re, im; so_conj_mul32(L, Y[i], &re, &im);
        int32_t rre = 0, rim = 0;
        if (e != 0) { rre = re / e; rim = im / e; }
        c->ChannelCoeffs[i] = so_c(so_w16(rre), so_w16(rim));
    }
}

/* ------------------------------------------------------------------ data symbol, up to the equaliser: brick by brick */
void so_freq_comp(const so_rx11a_ctx* c, const so_c16 in[64], so_c16 out[64])           /* TFreqCompensation (channel_11a.hpp:636-652) */
{
    for (int i = 0; i < 64; i++) out[i] = so_mul_q15(so_sra(in[i], 1), c->FreqCoeffs[i]);   /* rep_shift_right<16>(pi, pi, 1) :643; FrequencyShift :644 */
}
void so_equalize(const so_rx11a_ctx* c, const so_c16 in[64], so_c16 out[64])            /* TChannelEqualization (channel_11a.hpp:548-574) */
{
    for (int i = 0; i < 64; i++) {
        if (i >= 28 && i < 36) { out[i] = so_c(0, 0); continue; }
        int32_t re, im; so_mul32(in[i], c->ChannelCoeffs[i], &re, &im);
        out[i] = so_c(so_w16(re >> 8), so_w16(im >> 8));
    }
}
void so_sym_front(const so_rx11a_ctx* c, const so_c16 in80[80], so_c16 eq[64])
{
    so_c16 x[64], Y[64];
    so_freq_comp(c, in80 + 8, x);                                      /* T11aDataSymbol skip_cp=8 (PHY_11a.hpp:393-397) */
    so_fft64(x, Y);                                                    /* TFFT64 (fft.hpp:121-134) */
    so_equalize(c, Y, eq);
}

/* ------------------------------------------------------------------ TPhaseCompensate + TPilotTrack */
static const char PilotSgn[128] = {              /* pilot.hpp:10-28 (0 / -1 : polarity of p_n) */
     0, 0, 0,-1,-1,-1, 0,-1, -1,-1,-1, 0, 0,-1, 0,-1, -1, 0, 0,-1, 0, 0,-1, 0,  0, 0, 0, 0, 0,-1, 0, 0,
     0,-1, 0, 0,-1,-1, 0, 0,  0,-1, 0,-1,-1,-1, 0,-1,  0,-1,-1, 0,-1,-1, 0, 0,  0, 0, 0,-1,-1, 0, 0,-1,
    -1, 0,-1, 0,-1, 0, 0,-1, -1,-1, 0, 0,-1,-1,-1,-1,  0,-1,-1, 0,-1, 0, 0, 0,  0,-1, 0,-1, 0,-1, 0,-1,
    -1,-1,-1,-1, 0,-1, 0, 0, -1, 0,-1, 0, 0, 0,-1,-1,  0,-1,-1,-1, 0, 0, 0,-1, -1,-1,-1,-1,-1,-1, 0, 0 };

static void build_coeff(so_c16* p, int16_t ave, int16_t delta)        /* pilot.hpp:138-164 */
{
    const int16_t *usin = so_usin_lut(), *ucos = so_ucos_lut();
    int16_t th = (int16_t)(ave - delta * 26);
    for (int i = 64 - 26; i < 64; i++) { p[i].re = ucos[(uint16_t)th]; p[i].im = (int16_t)(-usin[(uint16_t)th]); th = (int16_t)(th + delta); }
    th = (int16_t)(th + delta);
    for (int i = 1; i <= 26; i++)      { p[i].re = ucos[(uint16_t)th]; p[i].im = (int16_t)(-usin[(uint16_t)th]); th = (int16_t)(th + delta); }
}

void so_phase_comp(const so_rx11a_ctx* c, const so_c16 in[64], so_c16 out[64])          /* TPhaseCompensate: rep_mul<16> (freqoffset.hpp:28-30) */
{
    for (int i = 0; i < 64; i++) out[i] = so_mul_q15(in[i], c->CompCoeffs[i]);
}
void so_sym_track(so_rx11a_ctx* c, const so_c16 eq[64], so_c16 out[64])
{
    so_c16 pc[64];
    so_phase_comp(c, eq, pc);
    so_pilot_track(c, pc, out);
}
void so_pilot_track(so_rx11a_ctx* c, const so_c16 pc[64], so_c16 out[64])               /* TPilotTrack alone: pc = TPhaseCompensate's output */
{
    so_init();
    /* _pilot_track (pilot.hpp:166-233) */
    int16_t th1 = so_uatan2(pc[64 - 21].im, pc[64 - 21].re);
    int16_t th2 = so_uatan2(pc[64 - 7].im,  pc[64 - 7].re);
    int16_t th3 = so_uatan2(pc[7].im,       pc[7].re);
    int16_t th4 = so_uatan2(-(int)pc[21].im, -(int)pc[21].re);
    if (PilotSgn[c->symbol_count]) {
        th1 = (int16_t)(th1 + 0x8000); th2 = (int16_t)(th2 + 0x8000);
        th3 = (int16_t)(th3 + 0x8000); th4 = (int16_t)(th4 + 0x8000);
    }
    c->symbol_count++;
    if (c->symbol_count >= 127) c->symbol_count = 0;

    int16_t avg = (int16_t)(((int)th1 + th2 + th3 + th4) / 4);
    int16_t del = (int16_t)((((int)th3 - th1) / 28 + ((int)th4 - th2) / 28) >> 1);

    so_c16 rot[64];
    memset(rot, 0, sizeof(rot));
    build_coeff(rot, avg, del);
    memset(out, 0, 64 * sizeof(so_c16));                               /* bins 28..35 are never written by _rotate; see note */
    for (int i = 0; i < 28; i++)  out[i] = so_mul_q15(pc[i], rot[i]);  /* rep_mul<7>  vcs 0..6 */
    for (int i = 36; i < 64; i++) out[i] = so_mul_q15(pc[i], rot[i]);  /* rep_mul<7>  vcs 9..15 */
    /* note: rot[] is an uninitialised stack array in the reference for bins 0 and 27 (pilot.hpp:205):
     * those two outputs are not defined by the reference and are never demapped; here rot=0 there. */

    c->CFO_tracker = (int16_t)(c->CFO_tracker + (avg >> 2));
    c->SFO_tracker = (int16_t)(c->SFO_tracker + (del >> 2));
    c->CFO_comp = (int16_t)(c->CFO_comp + avg + c->CFO_tracker);
    c->SFO_comp = (int16_t)(c->SFO_comp + del + c->SFO_tracker);
    build_coeff(c->CompCoeffs, c->CFO_comp, c->SFO_comp);
}

/* ------------------------------------------------------------------ T11aDemap<N_BPSC> (demapper11a.hpp:10-79, demapper.h) */
void so_demap(int nbpsc, const so_c16 in[64], uint8_t* soft)
{
    so_init();
    const uint8_t *l0 = so_demap_lut(0), *l16 = so_demap_lut(1), *l64_2 = so_demap_lut(2), *l64_3 = so_demap_lut(3);
    for (int pass = 0; pass < 2; pass++) {
        int lo = pass ? 1 : 64 - 26, hi = pass ? 27 : 64;
        for (int i = lo; i < hi; i++) {
            if (i == 64 - 21 || i == 64 - 7 || i == 7 || i == 21) continue;
            int re = in[i].re >> 4, im = in[i].im >> 4;                /* demap_limit<64>: >>4, clamp (demapper.h:141-151) */
            re = re < -128 ? -128 : (re > 127 ? 127 : re); im = im < -128 ? -128 : (im > 127 ? 127 : im);
            uint8_t r = (uint8_t)re, m = (uint8_t)im;
            switch (nbpsc) {
            case 1: soft[0] = l0[r]; break;
            case 2: soft[0] = l0[r]; soft[1] = l0[m]; break;
            case 4: soft[0] = l0[r]; soft[1] = l16[r]; soft[2] = l0[m]; soft[3] = l16[m]; break;
            default: soft[0] = l0[r]; soft[1] = l64_2[r]; soft[2] = l64_3[r];
                     soft[3] = l0[m]; soft[4] = l64_2[m]; soft[5] = l64_3[m]; break;
            }
            soft += nbpsc;
        }
    }
}

/* ------------------------------------------------------------------ T11aDeinterleave* (deinterleaver.hpp) */
void so_deinterleave(int nbpsc, const uint8_t* in, uint8_t* out)
{
    const int N = 48 * nbpsc, s = nbpsc / 2 > 1 ? nbpsc / 2 : 1;
    for (int k = 0; k < N; k++) {
        int i = (N / 16) * (k % 16) + k / 16;
        int j = s * (i / s) + (i + N - (16 * i) / N) % s;
        out[k] = in[j];
    }
}

/* ------------------------------------------------------------------ K=7 Viterbi (viterbicore.h, viterbilut.h) */
/* Expected coded bits for the transition into state n through branch br (predecessor (n>>1)+32*br):
 * 7-bit register r = br<<6 | n (LSB = newest bit); A = parity(r & 0155), B = parity(r & 0117)
 * -- VIT_MA / VIT_MB (viterbilut.h:50-185) are exactly  cost = bit ? 2*(7-soft) : 2*soft. */
static int parity7(int v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1; }
static uint8_t g_cA[64][2], g_cB[64][2];
static int g_vit_init;
static void vit_init(void)
{
    if (g_vit_init) return;
    for (int n = 0; n < 64; n++) for (int br = 0; br < 2; br++) {
        int r = (br << 6) | n;
        g_cA[n][br] = (uint8_t)parity7(r & 0155); g_cB[n][br] = (uint8_t)parity7(r & 0117);
    }
    g_vit_init = 1;
}
static inline uint8_t bm(uint8_t soft, int bit) { return (uint8_t)(bit ? 2 * (7 - soft) : 2 * soft); }

/* one ACS step; which: 0 = (A,B), 1 = A only, 2 = B only.  Returns the 64 decision bits. */
static uint64_t acs(uint8_t m[64], int which, uint8_t sa, uint8_t sb)
{
    uint8_t nm[64]; uint64_t dec = 0;
    for (int n = 0; n < 64; n++) {
        uint8_t c0 = m[n >> 1], c1 = m[32 + (n >> 1)];
        if (which != 2) { c0 = (uint8_t)(c0 + bm(sa, g_cA[n][0])); c1 = (uint8_t)(c1 + bm(sa, g_cA[n][1])); }
        if (which != 1) { c0 = (uint8_t)(c0 + bm(sb, g_cB[n][0])); c1 = (uint8_t)(c1 + bm(sb, g_cB[n][1])); }
        c0 &= 0xFE; c1 |= 0x01;                                        /* viterbicore.h:310,315 */
        nm[n] = c0 < c1 ? c0 : c1;                                     /* _mm_min_epu8 */
        dec |= (uint64_t)(nm[n] & 1) << n;
    }
    memcpy(m, nm, 64);
    return dec;
}
static void normalize(uint8_t m[64])                                   /* viterbicore.h:444-465 */
{
    uint8_t mn = 255;
    for (int n = 0; n < 64; n++) if (m[n] < mn) mn = m[n];
    mn &= 0xFE;
    for (int n = 0; n < 64; n++) m[n] = (uint8_t)(m[n] - mn);
}
/* argmin with the reference's tie-break: key = metric<<8 | index<<2 (viterbicore.h:479-524) */
static int argmin_state(const uint8_t m[64])
{
    int best = 0; uint32_t bk = 0xFFFFFFFFu;
    for (int n = 0; n < 64; n++) { uint32_t k = ((uint32_t)m[n] << 8) | ((uint32_t)n << 2); if (k < bk) { bk = k; best = n; } }
    return best;
}
/* Traceback (viterbicore.h:468-555): dec[t] = decisions of trellis column t (1-based), cur = current column */
static void traceback(const uint64_t* dec, uint32_t cur, const uint8_t m[64], uint8_t* out, uint32_t bits, uint32_t lookahead)
{
    int st = argmin_state(m);
    int pos = st | ((m[st] & 1) << 6);
    uint32_t col = cur;
    for (uint32_t i = 0; i < lookahead; i++) {
        col--; pos = (pos >> 1) & 0x3F;
        pos |= (int)((dec[col] >> pos) & 1) << 6;
    }
    uint8_t* po = out + (bits >> 3);
    for (uint32_t i = 0; i < bits >> 3; i++) {
        uint8_t oc = 0;
        for (int j = 0; j < 8; j++) {
            oc = (uint8_t)((oc << 1) | ((pos >> 6) & 1));
            col--; pos = (pos >> 1) & 0x3F;
            pos |= (int)((dec[col] >> pos) & 1) << 6;
        }
        *--po = oc;
    }
}
static void vit_reset(uint8_t m[64]) { for (int n = 0; n < 64; n++) m[n] = 0x30; m[0] = 0; }   /* ALL_INIT0/ALL_INIT viterbilut.h:22-30 */

uint32_t so_viterbi_sig(const uint8_t soft48[48])                      /* Viterbi_sig11 (viterbicore.h:35-261) + viterbi.hpp:38-39 */
{
    vit_init();
    uint8_t m[64]; uint64_t dec[26]; uint8_t out[4] = {0, 0, 0, 0};
    vit_reset(m);
    /* column 0 holds the initial metrics; its LSBs are the "decisions" read when the walk reaches it */
    dec[0] = 0; for (int n = 0; n < 64; n++) dec[0] |= (uint64_t)(m[n] & 1) << n;
    for (uint32_t t = 1; t <= 24; t++) {
        dec[t] = acs(m, 0, soft48[2 * (t - 1)], soft48[2 * (t - 1) + 1]);
        if ((t & 7) == 0) normalize(m);
    }
    normalize(m);
    traceback(dec, 24, m, out, 24, 0);
    uint32_t v = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16);
    return v >> 6;
}

/* Viterbi_sig11 with its output_bit argument (viterbicore.h:36): nbits = 24 or 48 trellis steps from 2*nbits soft values, the 6-bit
 * zero prefix still in place; out receives nbits/8 bytes.  T11nViterbiSig (viterbi.hpp:82-90) uses it for L-SIG (24) and HT-SIG (48). */
void so_viterbi_sig_bits(const uint8_t* soft, int nbits, uint8_t* out)
{
    vit_init();
    uint8_t m[64]; uint64_t dec[50];
    vit_reset(m);
    dec[0] = 0; for (int n = 0; n < 64; n++) dec[0] |= (uint64_t)(m[n] & 1) << n;
    for (int t = 1; t <= nbits; t++) {
        dec[t] = acs(m, 0, soft[2 * (t - 1)], soft[2 * (t - 1) + 1]);
        if ((t & 7) == 0) normalize(m);
    }
    normalize(m);
    traceback(dec, (uint32_t)nbits, m, out, (uint32_t)nbits, 0);
}

int so_viterbi_frame(const uint8_t* soft, uint32_t nsoft, int code_rate, uint32_t frame_length, uint8_t* out)
{
    return so_viterbi_frame_ex(soft, nsoft, code_rate, frame_length, out, 256, 24);       /* T11aViterbi<5000*8,48,256,24> (11a graph) */
}

int so_viterbi_frame_ex(const uint8_t* soft, uint32_t nsoft, int code_rate, uint32_t frame_length, uint8_t* out, uint32_t DEPTH, uint32_t LOOK)
{
    /* T11aViterbi<TRELLIS_MAX, N_INPUT, TRELLIS_DEPTH, TRELLIS_LOOKAHEAD>::Filter::Process (viterbi.hpp:148-235); N_INPUT only sets
     * the burst size and has no effect on the output */
    vit_init();
    const uint32_t PREFIX = 6;
    uint32_t maxcol = nsoft + 8;
    uint64_t* dec = (uint64_t*)malloc((size_t)maxcol * sizeof(uint64_t));
    uint8_t m[64], buf[256 / 8 + 1];
    vit_reset(m);
    dec[0] = 0; for (int n = 0; n < 64; n++) dec[0] |= (uint64_t)(m[n] & 1) << n;
    uint32_t tr = 0, ob = 0; int nout = 0;
    const uint8_t* p = soft; const uint8_t* end = soft + nsoft;
    const uint32_t tr_end = frame_length * 8 + 16 + PREFIX;
    while (p < end) {
        if (code_rate == SO_CR_12)      { dec[tr + 1] = acs(m, 0, p[0], p[1]); tr += 1; p += 2; }
        else if (code_rate == SO_CR_34) { dec[tr + 1] = acs(m, 0, p[0], p[1]); dec[tr + 2] = acs(m, 1, p[2], 0); dec[tr + 3] = acs(m, 2, 0, p[3]); tr += 3; p += 4; }
        else                            { dec[tr + 1] = acs(m, 0, p[0], p[1]); dec[tr + 2] = acs(m, 1, p[2], 0); tr += 2; p += 3; }
        if ((tr & 7) == 0) normalize(m);
        uint32_t cnt = 0, look = 0;
        if (tr >= tr_end) { cnt = tr_end - ob - PREFIX; look = tr - tr_end; }
        else if (tr >= ob + DEPTH + LOOK + PREFIX) { uint32_t remain = (tr - (ob + DEPTH + LOOK + PREFIX)) % 8; cnt = DEPTH; look = LOOK + remain; }
        if (cnt) {
            if (cnt <= DEPTH) { traceback(dec, tr, m, buf, cnt, look); memcpy(out + nout, buf, cnt >> 3); }
            else { /* frames shorter than one window: single final traceback of all bits */
                uint8_t* big = (uint8_t*)malloc((cnt >> 3) + 1);
                traceback(dec, tr, m, big, cnt, look); memcpy(out + nout, big, cnt >> 3); free(big);
            }
            ob += cnt; nout += (int)(cnt >> 3);
            if (tr >= tr_end) break;       /* the sink raises FRAME_OK/CRC32_FAIL; later bursts are dropped (viterbi.hpp:156-161) */
        }
    }
    free(dec);
    return nout;
}

/* ------------------------------------------------------------------ SIGNAL parser (PHY_11a.hpp:548-580) */
static const uint32_t RateLUT[16] = { 0,0,0,0,0,0,0,0, 48000,24000,12000,6000,54000,36000,18000,9000 };  /* ieee80211a_cmn.h:97-107 */
static int ndbps_of(uint32_t kbps)
{
    switch (kbps) { case 6000: return 24; case 9000: return 36; case 12000: return 48; case 18000: return 72;
                    case 24000: return 96; case 36000: return 144; case 48000: return 192; case 54000: return 216; }
    return 0;
}
int so_parse_plcp(uint32_t sig, uint32_t* rate_kbps, uint16_t* length, uint16_t* code_rate, uint16_t* nsym)
{
    sig &= 0xFFFFFF;
    if (sig & 0xFC0010) return 0;
    uint32_t par = (sig >> 16) ^ sig; par = (par >> 8) ^ par; par = (par >> 4) ^ par; par = (par >> 2) ^ par; par = (par >> 1) ^ par;
    if (par & 1) return 0;
    uint32_t kbps = RateLUT[sig & 0xF];
    if (kbps == 0) return 0;
    uint16_t cr = SO_CR_12;
    if (kbps == 48000) cr = SO_CR_23;
    else if (kbps == 9000 || kbps == 18000 || kbps == 36000 || kbps == 54000) cr = SO_CR_34;
    uint32_t len = (sig >> 5) & 0xFFF;
    if (len > 2500) return 0;
    int nd = ndbps_of(kbps);
    int ns = ((int)len * 8 + 16 + 6 + nd - 1) / nd;                    /* B11aGetSymbolCount ieee80211a_cmn.h:151-157 */
    *rate_kbps = kbps; *length = (uint16_t)len; *code_rate = cr; *nsym = (uint16_t)ns;
    return 1;
}

/* ------------------------------------------------------------------ T11aDesc + TBB11aFrameSink */
uint32_t so_desc_sink(const uint8_t* dec, uint32_t frame_length, uint8_t* mpdu, uint32_t* crc_in_frame)
{
    so_init();
    /* scramble.hpp:319-349: byte 0 dropped, byte 1 >> 1 seeds the register */
    uint8_t reg = (uint8_t)(dec[1] >> 1);
    uint32_t crc = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < frame_length; i++) {
        reg = so_g_scr_lut[reg & 0x7F];
        uint8_t o = dec[2 + i] ^ reg;
        reg >>= 1;
        mpdu[i] = o;
        if (i + 4 < frame_length) crc = (crc >> 8) ^ so_g_crc_lut[(o ^ crc) & 0xFF];      /* PHY_11a.hpp:668-673 */
    }
    uint32_t fcs = 0;
    if (frame_length >= 4) memcpy(&fcs, mpdu + frame_length - 4, 4);   /* little-endian host, as the reference (PHY_11a.hpp:683) */
    if (crc_in_frame) *crc_in_frame = fcs;
    return (~crc) == fcs ? SO_E_FRAME_OK : SO_E_CRC32_FAIL;            /* PHY_11a.hpp:688-692 */
}

/* ================================================================== the offline harness */
typedef struct { so_c16 v[4]; } vcs_t;

typedef struct {                                 /* CMovingWindow<T,4> / CAccumulator<int,4> (dspalg.hpp:5-98) */
    int32_t e[4]; int idx; int32_t reg;
} acc4;
static void acc_clear(acc4* a) { memset(a, 0, sizeof(*a)); }
static void acc_push(acc4* a, int32_t d) { a->reg = so_w32((int64_t)a->reg + d - a->e[a->idx]); a->e[a->idx] = d; a->idx = (a->idx + 1) & 3; }

typedef struct {
    /* TCCA11a state (cca.hpp:126-158) */
    vcs_t his[4]; int his_idx;
    acc4 ac_re, ac_im, energy;
    uint32_t auto_count, sense_count, high_count;
    int sync_high;                                /* sync_state: 0 no_energy, 1 high_energy */
    int peak_corr, peak_index;
    /* TDCEstimator (dc.hpp:92-166) */
    uint32_t dc_update_cnt; so_c16 sum_dc[4];
} cs_state;

typedef struct {
    /* context (BB11aDemodContext) */
    uint32_t error_code;
    int cca_detected;                             /* CF_11CCA::cca_state */
    uint32_t cca_pwr_threshold;
    so_c16 dc[4];                                 /* CF_VecDC::direct_current (survives frame resets) */
    int symbol_is_data;                           /* CF_11aSymState */
    int plcp_is_data;                             /* CF_11RxPLCPSwitch */
    uint32_t rate_kbps; uint16_t frame_length, code_rate, total_symbols, remain_symbols;
    so_rx11a_ctx fc;
    cs_state cs;
    /* pin queues */
    so_c16 lts_q[144]; int lts_n;
    so_c16 sym_q[80];  int sym_n;
    /* viterbi FIFO: whole frame of de-interleaved soft values (restates the 48-byte bursts) */
    uint8_t* soft; uint32_t soft_n, soft_cap;
    /* bookkeeping */
    uint32_t pos20;                               /* 20 MHz-rate index of the vcs being pushed */
    uint32_t frame_start, frame_end;
    int sig_symbol_done;
    uint16_t nsym;
    int16_t cfo_est;
    /* outputs */
    so_frame_result* res; int nres, max_res;
    uint8_t* mpdu_buf; uint32_t mpdu_cap, mpdu_used;
    so_trace* trace; int traced;
    uint32_t sym_idx;
    uint32_t frame_crc;
} rx_t;

static void cs_brick_reset(cs_state* s)          /* TCCA11a::__init (cca.hpp:279-295) + TDCEstimator::__init */
{
    memset(s->his, 0, sizeof(s->his)); s->his_idx = 0;
    acc_clear(&s->ac_re); acc_clear(&s->ac_im); acc_clear(&s->energy);
    s->auto_count = s->sense_count = s->high_count = 0; s->sync_high = 0;
    s->peak_corr = 0; s->peak_index = 0;
    s->dc_update_cnt = 8; memset(s->sum_dc, 0, sizeof(s->sum_dc));
}

static int iabs(int v) { return v < 0 ? -v : v; }

static int cross_corr(const cs_state* s, int k, const so_c16* pattern)   /* GetCrossCorrelation (cca.hpp:202-218) */
{
    int32_t sre[4] = {0,0,0,0}, sim[4] = {0,0,0,0};
    for (int v = 0; v < 4; v++) {
        for (int e = 0; e < 4; e++) {
            int32_t re, im; so_conj_mul32(pattern[4 * v + e], s->his[k].v[e], &re, &im);
            sre[e] = so_w32((int64_t)sre[e] + re); sim[e] = so_w32((int64_t)sim[e] + im);
        }
        k = (k + 1) & 3;
    }
    int32_t r = so_w32((int64_t)sre[0] + sre[1] + sre[2] + sre[3]);
    int32_t i = so_w32((int64_t)sim[0] + sim[1] + sim[2] + sim[3]);
    return iabs(r) + iabs(i);
}

/* TDCRemoveEx<4> -> TCCA11a -> TDCEstimator -> drop   (power_clear path) */
static void carrier_sense(rx_t* rx, const vcs_t* in)
{
    cs_state* s = &rx->cs;
    const so_c16* pat = so_sts_pattern();
    vcs_t pi;
    for (int e = 0; e < 4; e++) pi.v[e] = so_c(so_w16((int32_t)in->v[e].re - rx->dc[e].re), so_w16((int32_t)in->v[e].im - rx->dc[e].im));  /* rep_sub (dc.hpp:68) */

    if (!s->sync_high) {
        vcs_t pii; for (int e = 0; e < 4; e++) pii.v[e] = so_sra(pi.v[e], 2);
        /* GetAutoCorrelation (cca.hpp:165-186) */
        int32_t sr = 0, si = 0;
        for (int e = 0; e < 4; e++) {
            int32_t re, im; so_conj_mul32(pii.v[e], s->his[s->his_idx].v[e], &re, &im);
            sr = so_w32((int64_t)sr + (re >> 4)); si = so_w32((int64_t)si + (im >> 4));
        }
        acc_push(&s->ac_re, sr); acc_push(&s->ac_im, si);
        int iAuto = iabs(s->ac_re.reg) + iabs(s->ac_im.reg);
        /* GetEnergy (cca.hpp:188-193) */
        int32_t se = 0;
        for (int e = 0; e < 4; e++) se = so_w32((int64_t)se + (so_sqnorm(pii.v[e]) >> 4));
        acc_push(&s->energy, se);
        int iEnergy = s->energy.reg;
        s->his[s->his_idx] = pii; s->his_idx = (s->his_idx + 1) & 3;
        s->sense_count += 4;
        if (iEnergy > (int)rx->cca_pwr_threshold && iAuto >= iEnergy - (iEnergy >> 3)) {
            s->auto_count++;
            s->sense_count = 0;
            if (s->auto_count >= 4) {
                /* establish_sync (cca.hpp:220-243) */
                int sindex = s->his_idx, sum_corr = 0;
                s->peak_corr = 0;
                for (int i = 0; i < 16; i++) {
                    int corr = cross_corr(s, sindex, pat + 16 * i);
                    if (corr > s->peak_corr) { s->peak_corr = corr; s->peak_index = i; }
                    sum_corr += corr;
                }
                if (s->peak_corr > (sum_corr >> 3)) {
                    s->sync_high = 1;
                    s->high_count = 0;
                    if (s->peak_index > 3) { s->high_count = (uint32_t)s->peak_index / 4; s->peak_index &= 3; }
                }
            }
        } else {
            s->auto_count = 0;
        }
    } else {
        for (int e = 0; e < 4; e++) s->his[s->his_idx].v[e] = so_sra(pi.v[e], 2);
        s->his_idx = (s->his_idx + 1) & 3;
        s->high_count++;
        if (s->high_count % 4 == 0) {
            /* check_sync (cca.hpp:245-265) */
            int corr = cross_corr(s, s->his_idx, pat + 16 * s->peak_index);
            int ok;
            if (corr < (s->peak_corr >> 1)) ok = 0; else { if (corr > s->peak_corr) s->peak_corr = corr; ok = 1; }
            if (!ok) {
                if (s->high_count > 8) { rx->cca_detected = 1; rx->frame_start = rx->pos20 + 4; }
                else { s->sync_high = 0; s->sense_count = 0; }
            }
        }
    }
    if (!s->sync_high) {
        /* energy gating -> TDCEstimator (dc.hpp:132-163) */
        int16_t hr = 0, hi = 0;
        for (int e = 0; e < 4; e++) { hr = so_w16((int32_t)hr + (pi.v[e].re >> 5)); hi = so_w16((int32_t)hi + (pi.v[e].im >> 5)); }
        for (int e = 0; e < 4; e++) { s->sum_dc[e].re = so_w16((int32_t)s->sum_dc[e].re + hr); s->sum_dc[e].im = so_w16((int32_t)s->sum_dc[e].im + hi); }
        if (s->dc_update_cnt == 0) {
            for (int e = 0; e < 4; e++) {
                rx->dc[e].re = so_w16((int32_t)rx->dc[e].re + (s->sum_dc[e].re >> 2));
                rx->dc[e].im = so_w16((int32_t)rx->dc[e].im + (s->sum_dc[e].im >> 2));
            }
            s->dc_update_cnt = 8; memset(s->sum_dc, 0, sizeof(s->sum_dc));
        }
        s->dc_update_cnt--;
    }
    /* cca.hpp:433-437 */
    if (s->sense_count >= 84 && !s->sync_high) rx->error_code = SO_E_CS_TIMEOUT;
}

static void soft_push(rx_t* rx, const uint8_t* p, uint32_t n)
{
    if (rx->soft_n + n > rx->soft_cap) { rx->soft_cap = (rx->soft_n + n) * 2 + 4096; rx->soft = (uint8_t*)realloc(rx->soft, rx->soft_cap); }
    memcpy(rx->soft + rx->soft_n, p, n); rx->soft_n += n;
}

static int nbpsc_of(uint32_t kbps)
{
    switch (kbps) { case 6000: case 9000: return 1; case 12000: case 18000: return 2; case 24000: case 36000: return 4; default: return 6; }
}

/* one OFDM symbol through T11aDataSymbol .. frame sink */
static void data_symbol(rx_t* rx, const so_c16 in80[80])
{
    so_c16 eq[64], trk[64]; uint8_t soft[288], dsoft[288];
    so_sym_front(&rx->fc, in80, eq);
    so_sym_track(&rx->fc, eq, trk);
    if (rx->trace && !rx->traced) {
        so_trace* t = rx->trace;
        if (t->eq && rx->sym_idx < t->cap_syms) memcpy(t->eq + 64 * rx->sym_idx, eq, sizeof(eq));
        if (t->tracked && rx->sym_idx < t->cap_syms) memcpy(t->tracked + 64 * rx->sym_idx, trk, sizeof(trk));
        t->n_syms = rx->sym_idx + 1;
    }
    rx->sym_idx++;
    if (!rx->plcp_is_data) {
        /* header chain: T11aDemapBPSK -> T11aDeinterleaveBPSK -> T11aViterbiSig -> T11aPLCPParser */
        so_demap(1, trk, soft); so_deinterleave(1, soft, dsoft);
        uint32_t sig = so_viterbi_sig(dsoft);
        uint32_t kbps; uint16_t len, cr, ns;
        if (so_parse_plcp(sig, &kbps, &len, &cr, &ns)) {
            rx->rate_kbps = kbps; rx->frame_length = len; rx->code_rate = cr; rx->nsym = ns;
            rx->total_symbols = (uint16_t)(ns + 1); rx->remain_symbols = rx->total_symbols;
            rx->plcp_is_data = 1;
        } else {
            rx->error_code = SO_E_PLCP_HEADER_FAIL;
        }
    } else if (rx->error_code == SO_E_SUCCESS) {
        int nb = nbpsc_of(rx->rate_kbps);
        so_demap(nb, trk, soft); so_deinterleave(nb, soft, dsoft);
        soft_push(rx, dsoft, (uint32_t)(48 * nb));
    }
    rx->remain_symbols--;                                                      /* PHY_11a.hpp:405 (ushort wrap as the reference) */
    if (rx->remain_symbols == 0) {
        /* Next()->Flush(): the Viterbi sub-graph has consumed everything by now */
        uint8_t* dec = (uint8_t*)malloc((size_t)rx->frame_length + 64);
        uint8_t* mpdu = rx->mpdu_buf + rx->mpdu_used;
        uint8_t* tmp = NULL;
        if (rx->mpdu_used + rx->frame_length > rx->mpdu_cap) { tmp = (uint8_t*)malloc((size_t)rx->frame_length + 8); mpdu = tmp; }
        int nd = so_viterbi_frame(rx->soft, rx->soft_n, rx->code_rate, rx->frame_length, dec);
        (void)nd;
        rx->error_code = so_desc_sink(dec, rx->frame_length, mpdu, &rx->frame_crc);
        if (rx->trace && !rx->traced) {
            so_trace* t = rx->trace;
            if (t->soft) { uint32_t n = rx->soft_n < t->cap_soft ? rx->soft_n : t->cap_soft; memcpy(t->soft, rx->soft, n); t->n_soft = n; }
            if (t->decoded) memcpy(t->decoded, dec, (size_t)rx->frame_length + 2);
        }
        free(dec); free(tmp);
        rx->frame_end = rx->pos20 + 4;
    }
}

/* TBB11bRxSwitch -> (carrier sense | T11aSymSelEx -> T11aLTS | T11aDataSymbol) for one 4-sample burst */
static void push_vcs(rx_t* rx, const vcs_t* v)
{
    if (!rx->cca_detected) { carrier_sense(rx, v); return; }
    if (!rx->symbol_is_data) {
        memcpy(rx->lts_q + rx->lts_n, v->v, sizeof(v->v)); rx->lts_n += 4;
        if (rx->lts_n == 144) {
            so_lts(&rx->fc, rx->lts_q);
            rx->cfo_est = rx->fc.CFO_est;
            if (rx->trace && !rx->traced && rx->trace->ctx_after_lts) *rx->trace->ctx_after_lts = rx->fc;
            rx->lts_n = 0; rx->symbol_is_data = 1;
        }
    } else {
        memcpy(rx->sym_q + rx->sym_n, v->v, sizeof(v->v)); rx->sym_n += 4;
        if (rx->sym_n == 80) { rx->sym_n = 0; data_symbol(rx, rx->sym_q); }
    }
}

static void frame_reset(rx_t* rx)                /* ssrc->Flush(); BB11aDemodCtx.Reset(); ssrc->Reset() (fb11a_demod.cpp:64-70) */
{
    rx->error_code = SO_E_SUCCESS; rx->cca_detected = 0;
    so_rx11a_ctx_reset(&rx->fc);
    rx->symbol_is_data = 0; rx->plcp_is_data = 0;
    rx->frame_length = 0; rx->total_symbols = 0; rx->remain_symbols = 0; rx->rate_kbps = 6000; rx->code_rate = SO_CR_12;
    cs_brick_reset(&rx->cs);
    rx->lts_n = 0; rx->sym_n = 0; rx->soft_n = 0; rx->sym_idx = 0; rx->nsym = 0;
}

static void emit_result(rx_t* rx, uint32_t err)
{
    if (rx->nres >= rx->max_res) return;
    so_frame_result* r = &rx->res[rx->nres++];
    memset(r, 0, sizeof(*r));
    r->start_sample = rx->frame_start; r->end_sample = (err == SO_E_PLCP_HEADER_FAIL) ? rx->pos20 : rx->frame_end;
    r->error_code = err; r->cfo_est = rx->cfo_est;
    if (err != SO_E_PLCP_HEADER_FAIL) {
        r->rate_kbps = rx->rate_kbps; r->length = rx->frame_length; r->nsym = rx->nsym; r->crc32 = rx->frame_crc;
        r->mpdu_offset = rx->mpdu_used;
        if (rx->mpdu_used + rx->frame_length <= rx->mpdu_cap) rx->mpdu_used += rx->frame_length;
    }
    if (rx->trace) rx->traced = 1;
}

int so_rx11a_capture(const so_c16* iq, uint32_t nsamples, int sample_rate_mhz,
                     so_frame_result* res, int max_res, uint8_t* mpdu_buf, uint32_t mpdu_cap, so_trace* trace)
{
    so_init();
    rx_t* rx = (rx_t*)calloc(1, sizeof(rx_t));
    rx->res = res; rx->max_res = max_res; rx->mpdu_buf = mpdu_buf; rx->mpdu_cap = mpdu_cap; rx->trace = trace;
    rx->cca_pwr_threshold = 1000 * 1000;                                   /* fb11ademod_config.hpp:107 */
    frame_reset(rx);
    if (trace) { trace->n_syms = 0; trace->n_soft = 0; }

    /* TMemSamples pumps 28 raw samples per Process() into its output pin queue (memsource.hpp:87-114,
     * queue = lcm(28,8) = 56 entries, pinqueue.h:104-150); at 40 MHz TDownSample2 (samples.hpp:27-45) pops
     * them in 8s and keeps the even ones.  A 20 MHz capture is DEFINED as the even samples of such a
     * stream, i.e. the same machine in units of raw pairs: queue 28, 14 per call, bursts of 4.
     * The queue is emulated literally because the LAST, partial source call still appends a full burst
     * whose tail is stale queue memory (memsource.hpp:99-107). */
    /* sample_rate_mhz == 44: the graph of CreateDemodGraph11a_44M (fb11ademod_config.hpp:236-300) over the 40 MHz stream
     * that TDownSample44_40 produces.  That brick declares no Reset/Flush of its own (sampling.hpp:35-66), so the base
     * TFilter::Reset/Flush (brick.h:310-311) neither clear nor pad ITS output queue: the raw samples waiting in front of
     * TDownSample2 survive the reset that follows a frame, where TMemSamples' queue (40 MHz graph) is flushed. */
    const int keep_queue = (sample_rate_mhz == 44);
    const uint32_t STR  = (sample_rate_mhz != 20) ? 2 : 1;                 /* raw samples per queue unit kept */
    const uint32_t APP  = 28 / (2 / STR);                                  /* 28 raw  | 14 */
    const uint32_t BUR  = 8 / (2 / STR);                                   /* 8 raw   | 4  */
    const uint32_t QSZ  = 56 / (2 / STR);
    so_c16 q[56]; memset(q, 0, sizeof(q));
    uint32_t w_cnt = 0, r_cnt = 0;
    uint32_t q_base = 0;                                                   /* source index (queue units) of q[0] */
    uint32_t src = 0, remain = nsamples;                                   /* in queue units (raw @40, samples @20) */
    uint32_t consumed = 0;                                                 /* 20 MHz index behind the last burst handed to the graph */
    int ret = 1;
    while (ret) {
        /* ---- ssrc->Process() */
        if (w_cnt == 0) q_base = src;
        if (remain > APP) {
            memcpy(q + w_cnt, iq + src, APP * sizeof(so_c16)); w_cnt += APP; src += APP; remain -= APP;
        } else if (remain == 0) {
            ret = 0;
        } else {
            memcpy(q + w_cnt, iq + src, remain * sizeof(so_c16)); w_cnt += APP; src += remain; remain = 0;
        }
        if (w_cnt > QSZ) w_cnt = QSZ;                                      /* cannot happen (assert in pinqueue.h:161) */
        if (ret) {
            while (w_cnt - r_cnt >= BUR) {
                vcs_t v;
                for (uint32_t e = 0; e < 4; e++) v.v[e] = q[r_cnt + e * STR];
                rx->pos20 = (q_base + r_cnt) / STR;
                r_cnt += BUR;
                consumed = (q_base + r_cnt) / STR;                         /* before the drained queue forgets its extent */
                if (r_cnt == w_cnt) r_cnt = w_cnt = 0;
                push_vcs(rx, &v);
            }
        }
        /* ---- RxThread bookkeeping (fb11a_demod.cpp:37-71) */
        uint32_t err = rx->error_code;
        if (err != SO_E_SUCCESS) {
            if (err == SO_E_CS_TIMEOUT) {
                /* ResetCarrierSense(); scs->Reset()  (nWaitCounter only paces re-entry of the thread routine) */
                rx->error_code = SO_E_SUCCESS; rx->cca_detected = 0;
                cs_brick_reset(&rx->cs);
            } else {
                rx->pos20 = consumed;                                      /* end_sample of a PLCP failure: the samples taken from the source so far */
                emit_result(rx, err);
                /* Flush + Reset: every pin queue is cleared, including the source's partial burst */
                if (!keep_queue) w_cnt = r_cnt = 0;
                frame_reset(rx);
            }
        }
    }
    int n = rx->nres;
    free(rx->soft); free(rx);
    return n;
}

/* ------------------------------------------------------------------ dump de-framing (brickutil.h:20-58) */
int so_load_dump(const uint8_t* file, uint32_t file_bytes, so_c16* out, uint32_t max_samples, int raw14)
{
    uint32_t n = 0, off = 0;
    while (off + 16 < file_bytes && n < max_samples) {
        off += 16;                                                         /* RX_BLOCK descriptor (_rx_manager.h:96-137) */
        uint32_t avail = (file_bytes - off) / 4; if (avail > 28) avail = 28;
        if (avail > max_samples - n) avail = max_samples - n;
        for (uint32_t i = 0; i < avail; i++) {
            int16_t re, im; memcpy(&re, file + off + 4 * i, 2); memcpy(&im, file + off + 4 * i + 2, 2);
            if (raw14) { re = (int16_t)(uint16_t)((uint16_t)re << 2); im = (int16_t)(uint16_t)((uint16_t)im << 2); }
            out[n + i] = so_c(re, im);
        }
        n += avail; off += 28 * 4;
    }
    return (int)n;
}
