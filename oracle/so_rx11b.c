/* so_rx11b.c -- TEST INFRASTRUCTURE: a literal, brick-by-brick C restatement of the reference's 802.11b receive graph
 *   src -> TDCRemove -> TBB11bRxSwitch -+-> TEnergyDetect -> TDCEstimator -> drop                      (carrier sense)
 *                                        +-> TSymTiming -> TBarkerSync -> TBB11bRxRateSel -+-> despread -> TSFDSync
 *                                                                                          +-> despread -> TDBPSKDemap -+
 *                                                                                          +-> despread -> TDQPSKDemap -+-> TDesc741 ->
 *                                                                                          +-> TCCK5P5Decoder ----------+
 *                                                                                          +-> TCCK11Decoder -----------+
 *   TBB11bPlcpSwitch -+-> TBB11bPlcpParser
 *                     +-> TBB11bFrameSink
 * (kernel/bb/demod11/fb11bdemod_config.hpp:122-172) driven by MAC11b_Receive (kernel/bb/demod11/fb11b_demod.cpp:27-76),
 * 44 MHz samples in, long preamble, 1 Mbps DBPSK, 2 Mbps DQPSK, 5.5 and 11 Mbps CCK payloads.  Every pin queue, Flush (pad + process) and
 * Reset of the brick framework is emulated as such, because what a frame leaves behind (CF_DifferentialDemap::last_symbol,
 * CF_Descramber::byte_reg, the DC estimate) is what the next frame starts from.
 * Pinned by the reference's own graph compiled from its sources (oracle/_ref/libsora_refgraph.so: ref_rx11b_capture),
 * tests/test_oracle_11b.py.
 * The 5.5 / 11 Mbps payloads go through TCCK5P5Decoder / TCCK11Decoder (kernel/bb/Brick11/src/cck.hpp:9-207, 209-763), restated
 * below hypothesis by hypothesis with the reference's 32-bit arithmetic, its comparison order (ties!) and its own DQPSK helper
 * (core/inc/soradsp.h:190-198 -- which, unlike the one in barkerspread.hpp, halves re/im first). */
#include <stdlib.h>
#include <string.h>
#include "so_internal.h"

enum { RATE_SYNC = 0, RATE_1M, RATE_2M, RATE_5P5M, RATE_11M };
enum { NO_PEAK_FOUND = 0, PEAK_FOUND, PEAK_VALID, PEAK_VALIDED, BARKER_SYNCED };

typedef struct {
    /* ---- context facades (ieee80211facade.hpp:21-135, stdfacade.h:14-57) */
    uint32_t error_code;
    int      power_detected;               /* CF_11CCA::cca_state */
    uint32_t cca_pwr_threshold;
    int      rxrate_state, plcp_data;      /* CF_11bRxMRSel, CF_11RxPLCPSwitch */
    so_c16   dc;                           /* CF_VecDC (all four lanes hold the same value) */
    so_c16   last_symbol;                  /* CF_DifferentialDemap -- never reset */
    uint8_t  byte_reg;                     /* CF_Descramber        -- never reset */
    uint16_t frame_length; uint32_t data_rate_kbps, frame_crc32;
    /* ---- TEnergyDetect (cca.hpp:11-98) */
    uint32_t average_energy, window[8], idx, count;
    /* ---- TDCEstimator (dc.hpp:92-166) */
    uint32_t update_cnt; so_c16 sum_dc;
    /* ---- TBB11bRxSwitch.opin1 -> TSymTiming (4 -> 28); NOT cleared by the switch's Reset (PHY_11b.hpp:336-339) */
    so_c16 stq[28]; int stq_n;
    int m_index, m_frag;                   /* TSymTiming (symtiming.hpp:10-170) */
    /* ---- TBarkerSync (symtiming.hpp:176-315) */
    int sync_flag, last_peak_cnt, m_max, search_count; so_c16 partial[11];
    /* ---- TBB11bRxRateSel.opin0..2 -> TBB11bDespread (1 -> 11) */
    so_c16 chipq[3][11]; int chipq_n[3];
    /* ---- TBB11bRxRateSel.opin3 -> TCCK5P5Decoder (1 -> 16), .opin4 -> TCCK11Decoder (1 -> 8; is_even is brick state, cleared by Reset) */
    so_c16 cckq[16]; int cckq_n; int cck_even;
    /* ---- TSFDSync (sfd_sync.hpp:10-134) */
    int bit_one_found; uint16_t word; int bit_err_cnt; uint32_t sync_cnt;
    /* ---- despread -> TDBPSKDemap (1 -> 8), despread -> TDQPSKDemap (1 -> 4) */
    so_c16 symq1[8]; int symq1_n; so_c16 symq2[4]; int symq2_n;
    /* ---- TBB11bPlcpSwitch.opin0 -> TBB11bPlcpParser (1 -> 6) */
    uint8_t hdrq[6]; int hdrq_n;
    /* ---- TBB11bFrameSink (PHY_11b.hpp:643-749) */
    uint8_t* frame_buf; uint32_t frame_buf_size; uint32_t byte_count, crc32;
} rx11b_t;

/* ------------------------------------------------------------------ sinks and the byte path */
static void frame_sink(rx11b_t* s, uint8_t b)
{
    if (s->byte_count < (uint32_t)(int32_t)(s->frame_length - 4)) {
        if (s->byte_count < s->frame_buf_size) s->frame_buf[s->byte_count] = b;
        s->byte_count++;
        s->crc32 = so_g_crc_lut[(s->crc32 ^ b) & 0xFF] ^ (s->crc32 >> 8);
    } else if (s->byte_count < (uint32_t)s->frame_length) {
        if (s->byte_count < s->frame_buf_size) s->frame_buf[s->byte_count] = b;
        s->byte_count++;
        if (s->byte_count == (uint32_t)s->frame_length - 1) {          /* "speculating ACK": only the first three FCS bytes */
            uint32_t p = 0;
            for (int k = 0; k < 4; k++)                                  /* uint* at byte_pointer - 3: the 4th byte is the stale buffer byte */
                p |= (uint32_t)(s->byte_count - 3 + k < s->frame_buf_size ? s->frame_buf[s->byte_count - 3 + k] : 0) << (8 * k);
            s->frame_crc32 = p;
            s->error_code = ((~s->crc32 & 0x00FFFFFFu) == (p & 0x00FFFFFFu)) ? SO_E_FRAME_OK : SO_E_CRC32_FAIL;
        }
    }
}

static uint16_t crc16_ccitt(const uint8_t* p, int n)                    /* CalcCRC16, core/inc/CRC16.h: reflected 0x8408, init 0xFFFF, ~ */
{
    uint16_t c = 0xFFFF;
    for (int i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (uint16_t)((c & 1) ? (c >> 1) ^ 0x8408 : c >> 1);
    }
    return (uint16_t)~c;
}

static void plcp_parser(rx11b_t* s, const uint8_t h[6])               /* TBB11bPlcpParser (PHY_11b.hpp:524-640) */
{
    const uint16_t crc = (uint16_t)(h[4] | (h[5] << 8));
    if (crc16_ccitt(h, 4) != crc) { s->error_code = SO_E_PLCP_HEADER_FAIL; return; }
    const uint8_t signal = h[0], service = h[1];
    uint16_t len = (uint16_t)(h[2] | (h[3] << 8));
    switch (signal) {
    case 0x0A: s->data_rate_kbps = 1000;  len = (uint16_t)(len >> 3); break;
    case 0x14: s->data_rate_kbps = 2000;  len = (uint16_t)(len >> 2); break;
    case 0x37: s->data_rate_kbps = 5500;  len = (uint16_t)(((len * 11) >> 4) - (service >> 7) - ((service >> 3) & 1)); break;
    case 0x6E: s->data_rate_kbps = 11000; len = (uint16_t)(((len * 11) >> 3) - (service >> 7) - ((service >> 3) & 1)); break;
    default:   s->data_rate_kbps = 0;     len = 0;
    }
    s->frame_length = len;
    switch (s->data_rate_kbps) {
    case 1000:  s->rxrate_state = RATE_1M; break;
    case 2000:  s->rxrate_state = RATE_2M; break;
    case 5500:  s->rxrate_state = RATE_5P5M; break;
    case 11000: s->rxrate_state = RATE_11M; break;
    }
    s->plcp_data = 1;
}

static void plcp_switch(rx11b_t* s, uint8_t b)                         /* TBB11bPlcpSwitch (PHY_11b.hpp:459-519) */
{
    if (!s->plcp_data) {
        s->hdrq[s->hdrq_n++] = b;
        if (s->hdrq_n == 6) { s->hdrq_n = 0; plcp_parser(s, s->hdrq); }
    } else frame_sink(s, b);
}

static void desc741(rx11b_t* s, uint8_t b)                             /* TDesc741 (scramble.hpp:93-170): z^-7 + z^-4 + 1, self-synchronising */
{
    uint8_t x = b, st = s->byte_reg & 0x7F, o = 0;
    for (int k = 0; k < 8; k++) {
        uint8_t o1 = (uint8_t)((x ^ st ^ (st >> 3)) & 1);
        st = (uint8_t)((st >> 1) | ((x & 1) << 6));
        o = (uint8_t)((o >> 1) | (o1 << 7));
        x >>= 1;
    }
    s->byte_reg = (uint8_t)(b >> 1);
    plcp_switch(s, o);
}

static inline uint32_t dot_sign(so_c16 ref, so_c16 x)                  /* (ulong)(ref.re*s.re + ref.im*s.im) >> 31 */
{ return (uint32_t)so_w32((int64_t)ref.re * x.re + (int64_t)ref.im * x.im) >> 31; }

static void dbpsk_demap(rx11b_t* s, const so_c16 in[8])               /* TDBPSKDemap::DemapDBPSK (barkerspread.hpp:312-390) */
{
    uint8_t r = 0; so_c16 ref = s->last_symbol;
    for (int i = 0; i < 8; i++) { r |= (uint8_t)(dot_sign(ref, in[i]) << i); ref = in[i]; }
    s->last_symbol = in[7];
    desc741(s, r);
}

static void dqpsk_demap(rx11b_t* s, const so_c16 in[4])               /* TDQPSKDemap::DemapDQPSK (barkerspread.hpp:396-454) */
{
    uint8_t r = 0; so_c16 ref = s->last_symbol;
    for (int i = 0; i < 4; i++) {
        const int32_t re = so_w32((int64_t)ref.re * in[i].re + (int64_t)ref.im * in[i].im);
        const int32_t im = so_w32((int64_t)ref.re * in[i].im - (int64_t)ref.im * in[i].re);
        r |= (uint8_t)(((uint32_t)so_w32((int64_t)re + im) >> 31) << (2 * i));
        r |= (uint8_t)(((uint32_t)so_w32((int64_t)re - im) >> 31) << (2 * i + 1));
        ref = in[i];
    }
    s->last_symbol = in[3];
    desc741(s, r);
}

static void sfd_sync(rx11b_t* s, so_c16 x)                             /* TSFDSync::Process (sfd_sync.hpp:76-126), one symbol */
{
    const uint16_t bit = (uint16_t)dot_sign(s->last_symbol, x);
    s->last_symbol = x;
    s->byte_reg &= 0x7F;
    const uint16_t sbit = (uint16_t)((bit ^ s->byte_reg ^ (s->byte_reg >> 3)) & 1);
    s->byte_reg = (uint8_t)((s->byte_reg >> 1) | (bit << 6));
    s->word = (uint16_t)((s->word >> 1) | (sbit << 15));
    s->sync_cnt++;
    if (!s->bit_one_found) {
        if (s->word == 0xFFFF) s->bit_one_found = 1;
    } else {
        if (s->word == 0xF3A0) s->rxrate_state = RATE_1M;              /* DOT11B_PLCP_LONG_PREAMBLE_SFD */
        else if (s->word != 0xFFFF) {
            if (s->bit_err_cnt++ > 32) { s->error_code = SO_E_SFD_FAIL; return; }
        }
    }
    if (s->sync_cnt > 128 + 16) s->error_code = SO_E_SFD_TIMEOUT;
}

/* TBB11bDespread::QuickBarkerDespread (barkerspread.hpp:277-303): chips 1 and 4 are negated BEFORE the >> 4, chips 8-10 are
 * subtracted after it; all in wrapping int16 */
static so_c16 despread(const so_c16 c[11])
{
    static const int pre_neg[8] = { 0, 1, 0, 0, 1, 0, 0, 0 };
    int16_t re[4] = { 0, 0, 0, 0 }, im[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < 4; k++) {                                       /* sum = chips 0..3 */
        so_c16 v = c[k]; if (pre_neg[k]) v = so_c(so_neg16(v.re), so_neg16(v.im));
        re[k] = (int16_t)(v.re >> 4); im[k] = (int16_t)(v.im >> 4);
    }
    for (int k = 0; k < 4; k++) {                                       /* + chips 4..7 */
        so_c16 v = c[4 + k]; if (pre_neg[4 + k]) v = so_c(so_neg16(v.re), so_neg16(v.im));
        re[k] = so_w16(re[k] + (v.re >> 4)); im[k] = so_w16(im[k] + (v.im >> 4));
    }
    for (int k = 0; k < 3; k++) {                                       /* - chips 8..10 (shift_element_right drops chip 7) */
        re[k] = so_w16(re[k] - (c[8 + k].re >> 4)); im[k] = so_w16(im[k] - (c[8 + k].im >> 4));
    }
    return so_c(so_w16(re[0] + re[1] + re[2] + re[3]), so_w16(im[0] + im[1] + im[2] + im[3]));
}


/* ------------------------------------------------------------------ CCK (cck.hpp) */
typedef struct { int32_t re, im; } c32_t;
static inline c32_t c32(int32_t re, int32_t im) { c32_t r; r.re = re; r.im = im; return r; }
static inline c32_t c32_add(c32_t a, c32_t b) { return c32(so_w32((int64_t)a.re + b.re), so_w32((int64_t)a.im + b.im)); }
static inline c32_t c32_sra2(c32_t a) { return c32(a.re >> 2, a.im >> 2); }                       /* shift_right(vci, 2) */
static inline c32_t c32_rot(c32_t a, int k)                              /* a * j^k, exact (components come from int16 sums) */
{ switch (k & 3) { case 0: return a; case 1: return c32(-a.im, a.re); case 2: return c32(-a.re, -a.im); default: return c32(a.im, -a.re); } }
static inline c32_t c32_of(so_c16 a) { return c32(a.re, a.im); }
static inline int32_t neg32(int32_t a) { return so_w32(-(int64_t)a); }

/* The four partial sums of one phi2 hypothesis, phi2 = m * pi/2 (r = (-j)^m):
 *   A1 = P1 + r P0, A2 = r P2 - P3, A3 = P5 + r P4, A4 = P7 - r P6
 * -- cck.hpp:268-275 (m = 0), 392-399 (m = 1), 514-521 (m = 2), 635-642 (m = 3), and the two hypotheses of the 5.5 Mbps decoder
 * (m = 1: cck.hpp:81-88, m = 3: 103-110). */
static void cck_partial(const so_c16 P[8], int m, c32_t A[4])
{
    const int k = (4 - m) & 3;                                           /* r = j^k */
    A[0] = c32_add(c32_of(P[1]), c32_rot(c32_of(P[0]), k));
    A[1] = c32_add(c32_rot(c32_of(P[2]), k), c32_rot(c32_of(P[3]), 2));
    A[2] = c32_add(c32_of(P[5]), c32_rot(c32_of(P[4]), k));
    A[3] = c32_add(c32_of(P[7]), c32_rot(c32_of(P[6]), k + 2));
}

/* One "Module" of CCK11_DECODER (e.g. cck.hpp:277-390): the four phi3 hypotheses L1..L4 = conj((A2 + q A1) >> 2) * ((A4 + q A3) >> 2),
 * q = 1, j, -1, -j with value bits 0x00, 0x30, 0x10, 0x20; per L the larger of |re|, |im| picks phi4; then the reference's comparison tree. */
static void cck11_module(const c32_t A[4], int32_t* max_out, uint8_t* val_out)
{
    static const int     q_of[4]   = { 0, 1, 2, 3 };
    static const uint8_t base_of[4] = { 0x00, 0x30, 0x10, 0x20 };
    int32_t M[4]; uint8_t V[4];
    for (int s = 0; s < 4; s++) {
        const c32_t bx = c32_sra2(c32_add(A[1], c32_rot(A[0], q_of[s]))), by = c32_sra2(c32_add(A[3], c32_rot(A[2], q_of[s])));
        const int32_t lre = so_w32((int64_t)so_w32((int64_t)bx.re * by.re) + so_w32((int64_t)bx.im * by.im));
        const int32_t lim = so_w32((int64_t)so_w32((int64_t)bx.re * by.im) - so_w32((int64_t)bx.im * by.re));
        const int32_t a1 = lre < 0 ? neg32(lre) : lre, a2 = lim < 0 ? neg32(lim) : lim;
        if (a1 > a2) { if (lre > 0) { M[s] = lre; V[s] = base_of[s] | 0x00; } else { M[s] = neg32(lre); V[s] = base_of[s] | 0x40; } }
        else         { if (lim > 0) { M[s] = lim; V[s] = base_of[s] | 0xC0; } else { M[s] = neg32(lim); V[s] = base_of[s] | 0x80; } }
    }
    int32_t mm; uint8_t vv;
    if (M[0] > M[1]) { mm = M[0]; vv = V[0]; } else { mm = M[1]; vv = V[1]; }
    if (M[2] > M[3]) { if (M[2] > mm) { mm = M[2]; vv = V[2]; } } else { if (M[3] > mm) { mm = M[3]; vv = V[3]; } }
    *max_out = mm; *val_out = vv;
}

static inline void cck_dqpsk_bits(uint8_t* r, int pos, so_c16 ref, so_c16 x)                    /* demap_dqpsk_bits, core/inc/soradsp.h:190-198 */
{
    int32_t re = so_w32((int64_t)ref.re * x.re + (int64_t)ref.im * x.im), im = so_w32((int64_t)ref.re * x.im - (int64_t)ref.im * x.re);
    re >>= 1; im >>= 1;
    *r |= (uint8_t)(((uint32_t)so_w32((int64_t)re + im) >> 31) << pos);
    *r |= (uint8_t)(((uint32_t)so_w32((int64_t)re - im) >> 31) << (pos + 1));
}

static uint8_t cck11_decode(rx11b_t* s, const so_c16 P[8])             /* TCCK11Decoder::CCK11_DECODER (cck.hpp:255-763) */
{
    c32_t A[4]; int32_t m1, m2, m34; uint8_t v1, v2, v34, out;
    cck_partial(P, 0, A); cck11_module(A, &m1, &v1);
    cck_partial(P, 1, A); cck11_module(A, &m2, &v2); v2 |= 0x08;
    if (m1 > m2) { cck_partial(P, 3, A); cck11_module(A, &m34, &v34); v34 |= 0x0C; out = m1 > m34 ? v1 : v34; }       /* "lable4" */
    else         { cck_partial(P, 2, A); cck11_module(A, &m34, &v34); v34 |= 0x04; out = m2 > m34 ? v2 : v34; }
    cck_dqpsk_bits(&out, 0, s->last_symbol, P[7]);
    out ^= (uint8_t)((s->cck_even << 1) | s->cck_even);
    s->cck_even ^= 1;
    s->last_symbol = P[7];
    return out;
}

/* One half byte of TCCK5P5Decoder (cck.hpp:70-206): phi2 = pi/2 against 3pi/2 with phi3 = 0; only Re of
 * ((-conj-ish B0) >> 2) * (B1 >> 2) is looked at -- the imaginary part of B0 is negated BEFORE the shift. */
static int32_t cck5_metric(const so_c16 P[8], int m, int* neg)
{
    c32_t A[4]; cck_partial(P, m, A);
    c32_t b0 = c32_add(A[0], A[1]), b1 = c32_add(A[2], A[3]);
    b0.im = neg32(b0.im);
    b0 = c32_sra2(b0); b1 = c32_sra2(b1);
    const int32_t lre = so_w32((int64_t)so_w32((int64_t)b0.re * b1.re) - so_w32((int64_t)b0.im * b1.im));
    *neg = !(lre > 0);
    return lre > 0 ? lre : neg32(lre);
}

static uint8_t cck5p5_decode(rx11b_t* s, const so_c16 P[16])           /* TCCK5P5Decoder::Process: b_isEven is a local, so every byte is even + odd */
{
    uint8_t out = 0;
    for (int half = 0; half < 2; half++) {
        const so_c16* Q = P + 8 * half; int n1, n2;
        const int32_t max1 = cck5_metric(Q, 1, &n1), max2 = cck5_metric(Q, 3, &n2);
        const uint8_t nib = max1 > max2 ? (uint8_t)(n1 ? 0x08 : 0x00) : (uint8_t)(n2 ? 0x0C : 0x04);
        if (half == 0) out = nib; else out |= (uint8_t)(nib << 4);
        cck_dqpsk_bits(&out, 4 * half, s->last_symbol, Q[7]);
        if (half == 1) out ^= 0x30;
        s->last_symbol = Q[7];
    }
    return out;
}

static void symbol_out(rx11b_t* s, int port, so_c16 sym)               /* what follows each despreader */
{
    if (port == 0) sfd_sync(s, sym);
    else if (port == 1) { s->symq1[s->symq1_n++] = sym; if (s->symq1_n == 8) { s->symq1_n = 0; dbpsk_demap(s, s->symq1); } }
    else { s->symq2[s->symq2_n++] = sym; if (s->symq2_n == 4) { s->symq2_n = 0; dqpsk_demap(s, s->symq2); } }
}

static void rate_sel(rx11b_t* s, so_c16 chip)                          /* TBB11bRxRateSel::Process (PHY_11b.hpp:421-452) */
{
    const int port = s->rxrate_state;
    if (port > RATE_2M) {                                               /* opin3 / opin4 -> the CCK decoders -> TDesc741 */
        const int need = port == RATE_5P5M ? 16 : 8;
        s->cckq[s->cckq_n++] = chip;
        if (s->cckq_n == need) { s->cckq_n = 0; desc741(s, port == RATE_5P5M ? cck5p5_decode(s, s->cckq) : cck11_decode(s, s->cckq)); }
        return;
    }
    s->chipq[port][s->chipq_n[port]++] = chip;
    if (s->chipq_n[port] == 11) { s->chipq_n[port] = 0; symbol_out(s, port, despread(s->chipq[port])); }
}

static void barker_sync(rx11b_t* s, so_c16 in)                         /* TBarkerSync::Process (symtiming.hpp:229-291), one chip */
{
    if (s->sync_flag == BARKER_SYNCED) { rate_sel(s, in); return; }
    s->search_count++;
    if (s->search_count >= 11 * 4) { s->error_code = SO_E_SYNC_TIMEOUT; return; }
    /* UpdateBarkerCorrelation (:296-313) */
    const so_c16 ss = so_sra(in, 4);
    so_c16* p = s->partial;
#define SUB(a) so_c(so_w16((a).re - ss.re), so_w16((a).im - ss.im))
#define ADD(a) so_c(so_w16((a).re + ss.re), so_w16((a).im + ss.im))
    const so_c16 o = SUB(p[0]);
    p[0] = SUB(p[1]); p[1] = SUB(p[2]); p[2] = ADD(p[3]); p[3] = ADD(p[4]); p[4] = ADD(p[5]); p[5] = SUB(p[6]);
    p[6] = ADD(p[7]); p[7] = ADD(p[8]); p[8] = SUB(p[9]); p[9] = ss;
#undef SUB
#undef ADD
    const int corr = so_sqnorm(o);
    switch (s->sync_flag) {
    case NO_PEAK_FOUND:
        if (corr > s->m_max) { s->m_max = corr; s->last_peak_cnt = 1; }
        else if (++s->last_peak_cnt == 11) s->sync_flag = PEAK_FOUND;
        break;
    case PEAK_FOUND:
        s->m_max = corr / 2; s->last_peak_cnt = 1; s->sync_flag = PEAK_VALID;
        break;
    case PEAK_VALID:
        if (corr > s->m_max) { s->m_max = corr; s->last_peak_cnt = 0; s->sync_flag = NO_PEAK_FOUND; }
        else if (++s->last_peak_cnt == 11) s->sync_flag = PEAK_VALIDED;
        break;
    default:
        s->sync_flag = BARKER_SYNCED;                                   /* "just skip one more symbol" */
    }
}

static void sym_timing(rx11b_t* s, so_c16 blk[28])                     /* TSymTiming::Process (symtiming.hpp:42-64) on one 28-sample block */
{
    /* Decimation (:66-83) */
    int idx = s->m_index;
    while (idx < 28) {
        so_c16 out;
        if (idx < 0) { out = blk[0]; s->m_index += 4; } else out = blk[idx];
        idx += 4;
        barker_sync(s, out);
    }
    if (s->m_index >= 4) s->m_index = 0;
    /* AdjustTiming (:118-170): early-late detector on the energies of the four sampling phases */
    int32_t sum[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 28; i++) sum[i & 3] = so_w32((int64_t)sum[i & 3] + so_sqnorm(so_sra(blk[i], 3)));
    const int mi = s->m_index;
    const int early = (mi == 0) ? 3 : mi - 1, late = (mi == 3) ? 0 : mi + 1;
    if (sum[early] < sum[late]) {
        if (sum[mi] < sum[early]) { s->m_index++; s->m_frag = 0; }
        else if (sum[mi] < sum[late]) s->m_frag++;
    } else {
        if (sum[mi] < sum[late]) { s->m_index--; s->m_frag = 0; }
        else if (sum[mi] < sum[early]) s->m_frag--;
    }
    if (s->m_frag >= 4) { s->m_index++; s->m_frag = -3; }
    else if (s->m_frag <= -4) { s->m_index--; s->m_frag = 3; }
}

static void energy_detect(rx11b_t* s, const so_c16 v[4])              /* TEnergyDetect::Process (cca.hpp:54-96) + TDCEstimator (dc.hpp:131-163) */
{
    uint32_t ave = 0;
    for (int k = 0; k < 4; k++) ave = (uint32_t)so_w32((int64_t)(int32_t)ave + (so_sqnorm(v[k]) >> 5));
    s->average_energy = s->average_energy - s->window[s->idx] + ave;
    s->window[s->idx] = ave;
    if (++s->idx >= 8) s->idx = 0;
    s->count++;
    if (s->count >= 32) {
        if (s->count >= 100) { s->error_code = SO_E_CS_TIMEOUT; return; }   /* ipin.clear(); return */
        if (s->average_energy >= s->cca_pwr_threshold) s->power_detected = 1;
    }
    if (!s->power_detected) {                                          /* energy gating: only low-power samples reach the DC estimator */
        so_c16 h = so_c(0, 0);
        for (int k = 0; k < 4; k++) h = so_c(so_w16(h.re + (v[k].re >> 5)), so_w16(h.im + (v[k].im >> 5)));
        s->sum_dc = so_c(so_w16(s->sum_dc.re + h.re), so_w16(s->sum_dc.im + h.im));
        if (s->update_cnt == 0) {
            s->dc = so_c(so_w16(s->dc.re + (s->sum_dc.re >> 2)), so_w16(s->dc.im + (s->sum_dc.im >> 2)));
            s->update_cnt = 8; s->sum_dc = so_c(0, 0);
        }
        s->update_cnt--;
    }
}

static void rx_switch(rx11b_t* s, const so_c16 v[4])                  /* TBB11bRxSwitch::Process (PHY_11b.hpp:354-372) */
{
    if (!s->power_detected) energy_detect(s, v);
    else {
        memcpy(s->stq + s->stq_n, v, 4 * sizeof(so_c16)); s->stq_n += 4;
        if (s->stq_n == 28) { s->stq_n = 0; sym_timing(s, s->stq); }
    }
}

/* pRxSource->Flush(): what is queued is padded with zero samples and pushed through (brick.h: FlushPort; only the ports
 * below hold partial bursts).  The switch flushes the branch its state selects (PHY_11b.hpp:341-352); the rate selector
 * pads and processes its current port but does not flush what follows it (:398-419), so the despreaders' own output
 * queues -- cleared by Reset -- never see their Flush. */
static void graph_flush(rx11b_t* s)
{
    if (!s->power_detected) return;                                     /* port 0: every queue on that branch is 4 -> 4 */
    if (s->stq_n > 0) {                                                 /* pad() fills up to the next multiple of 28 */
        memset(s->stq + s->stq_n, 0, (size_t)(28 - s->stq_n) * sizeof(so_c16)); s->stq_n = 0;
        sym_timing(s, s->stq);
    }
    const int port = s->rxrate_state;                                   /* TSymTiming / TBarkerSync: 1 -> 1, nothing queued */
    if (port <= RATE_2M && s->chipq_n[port] > 0) {
        memset(s->chipq[port] + s->chipq_n[port], 0, (size_t)(11 - s->chipq_n[port]) * sizeof(so_c16)); s->chipq_n[port] = 0;
        symbol_out(s, port, despread(s->chipq[port]));
    }
    if (port > RATE_2M && s->cckq_n > 0) {                              /* opin3/4().pad(); Next3/4()->Process(): one more byte out of zero chips */
        const int need = port == RATE_5P5M ? 16 : 8;
        memset(s->cckq + s->cckq_n, 0, (size_t)(need - s->cckq_n) * sizeof(so_c16)); s->cckq_n = 0;
        desc741(s, port == RATE_5P5M ? cck5p5_decode(s, s->cckq) : cck11_decode(s, s->cckq));
    }
}

static void graph_reset(rx11b_t* s)                                    /* BB11bDemodCtx.reset() + pRxSource->Reset() (fb11b_demod.cpp:68-70) */
{
    s->error_code = SO_E_SUCCESS; s->power_detected = 0; s->rxrate_state = RATE_SYNC; s->plcp_data = 0;
    s->average_energy = 0; memset(s->window, 0, sizeof(s->window)); s->idx = 0; s->count = 0;      /* TEnergyDetect::__init */
    s->update_cnt = 8; s->sum_dc = so_c(0, 0);                                                     /* TDCEstimator::__init */
    /* stq (the switch's port towards TSymTiming) is NOT cleared: the switch's Reset only forwards (PHY_11b.hpp:336-339) */
    s->m_index = 2; s->m_frag = 0;                                                                 /* TSymTiming::_init */
    s->sync_flag = NO_PEAK_FOUND; s->last_peak_cnt = -1; s->m_max = 0; s->search_count = 0; memset(s->partial, 0, sizeof(s->partial));
    memset(s->chipq_n, 0, sizeof(s->chipq_n)); s->cckq_n = 0; s->cck_even = 0;
    s->bit_one_found = 0; s->word = 0; s->bit_err_cnt = 0; s->sync_cnt = 0;
    s->symq1_n = s->symq2_n = 0; s->hdrq_n = 0;
    s->crc32 = 0xFFFFFFFFu; s->byte_count = 0;
}

/* Test11B_FB_Demod / MAC11b_Receive over one 44 MHz capture.  end_sample = CF_MemSamples::mem_sample_index() when the
 * event is seen (44 MHz samples); start_sample is not defined by the reference and reported as 0. */
int so_rx11b_capture(const so_c16* iq, uint32_t nsamples, so_frame_result* res, int max_res, uint8_t* mpdu_buf, uint32_t mpdu_cap)
{
    so_init();
    rx11b_t* s = (rx11b_t*)calloc(1, sizeof(rx11b_t));
    uint8_t out[4096];
    s->frame_buf = out; s->frame_buf_size = sizeof(out); s->cca_pwr_threshold = 1000 * 1000;
    memset(out, 0, sizeof(out));
    graph_reset(s);
    uint32_t pos = 0, remain = nsamples, used = 0; int n = 0;
    so_c16 blk[28];
    for (;;) {
        /* ---- TMemSamples::Process (memsource.hpp:87-114): 28 samples per call; the last, partial call pads with stale queue memory */
        int ret = 1;
        if (remain > 28) { memcpy(blk, iq + pos, sizeof(blk)); pos += 28; remain -= 28; }
        else if (remain == 0) ret = 0;
        else { memcpy(blk, iq + pos, remain * sizeof(so_c16)); pos += remain; remain = 0; }
        if (ret)
            for (int i = 0; i < 7; i++) {                               /* TDCRemove (dc.hpp:6-38): wrapping subtraction */
                so_c16 v[4];
                for (int k = 0; k < 4; k++) v[k] = so_c(so_w16(blk[4 * i + k].re - s->dc.re), so_w16(blk[4 * i + k].im - s->dc.im));
                rx_switch(s, v);
                if (s->error_code == SO_E_CS_TIMEOUT && !s->power_detected) break;   /* TEnergyDetect: ipin.clear(); return 0 -- the rest of this call is dropped */
            }
        const uint32_t err = s->error_code;
        if (err != SO_E_SUCCESS) {
            if (err != SO_E_CS_TIMEOUT && n < max_res) {
                so_frame_result* r = &res[n++];
                memset(r, 0, sizeof(*r));
                r->error_code = err; r->end_sample = pos; r->rate_kbps = s->data_rate_kbps; r->length = s->frame_length;
                r->crc32 = s->frame_crc32; r->mpdu_offset = used;
                if ((err == SO_E_FRAME_OK || err == SO_E_CRC32_FAIL) && used + s->frame_length <= mpdu_cap) {
                    memcpy(mpdu_buf + used, out, s->frame_length); used += s->frame_length;
                }
            }
            if (err == SO_E_FRAME_OK || err == SO_E_CRC32_FAIL) {       /* "jump advance of the last CRC byte" (fb11b_demod.cpp:47-63) */
                uint32_t off = s->data_rate_kbps == 1000 ? 8 * 11 * 4 : s->data_rate_kbps == 2000 ? 4 * 11 * 4 : s->data_rate_kbps == 5500 ? 8 * 2 * 4 : s->data_rate_kbps == 11000 ? 8 * 1 * 4 : 0;
                off = (off + 3) / 4 * 4; if (off > remain) off = remain;
                pos += off; remain -= off;
            }
            graph_flush(s);
            graph_reset(s);
            continue;                                                   /* MAC11b_Receive returns and is called again; rc is not looked at */
        }
        if (!ret) break;
    }
    free(s);
    return n;
}
