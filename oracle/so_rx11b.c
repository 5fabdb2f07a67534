/* so_rx11b.c -- TEST INFRASTRUCTURE: a literal, brick-by-brick C restatement of the reference's 802.11b receive graph
 *   src -> TDCRemove -> TBB11bRxSwitch -+-> TEnergyDetect -> TDCEstimator -> drop                      (carrier sense)
 *                                        +-> TSymTiming -> TBarkerSync -> TBB11bRxRateSel -+-> despread -> TSFDSync
 *                                                                                          +-> despread -> TDBPSKDemap -+
 *                                                                                          +-> despread -> TDQPSKDemap -+-> TDesc741 ->
 *   TBB11bPlcpSwitch -+-> TBB11bPlcpParser
 *                     +-> TBB11bFrameSink
 * (kernel/bb/demod11/fb11bdemod_config.hpp:122-172) driven by MAC11b_Receive (kernel/bb/demod11/fb11b_demod.cpp:27-76),
 * 44 MHz samples in, long preamble, 1 Mbps DBPSK and 2 Mbps DQPSK payloads.  Every pin queue, Flush (pad + process) and
 * Reset of the brick framework is emulated as such, because what a frame leaves behind (CF_DifferentialDemap::last_symbol,
 * CF_Descramber::byte_reg, the DC estimate) is what the next frame starts from.
 * Pinned by the reference's own graph compiled from its sources (oracle/_ref/libsora_refgraph.so: ref_rx11b_capture),
 * tests/test_oracle_11b.py.
 * NOT restated: the 5.5 / 11 Mbps CCK decoders (cck.hpp).  A header that announces one of those rates ends the frame here
 * with SO_E_NOT_SUPPORTED; the reference goes on into its CCK branch (which does not decode its own modulator's output in
 * this build).  Such frames are outside the parity claim. */
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
    if (s->rxrate_state == RATE_5P5M || s->rxrate_state == RATE_11M) s->error_code = SO_E_NOT_SUPPORTED;   /* see the header of this file */
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

static void symbol_out(rx11b_t* s, int port, so_c16 sym)               /* what follows each despreader */
{
    if (port == 0) sfd_sync(s, sym);
    else if (port == 1) { s->symq1[s->symq1_n++] = sym; if (s->symq1_n == 8) { s->symq1_n = 0; dbpsk_demap(s, s->symq1); } }
    else { s->symq2[s->symq2_n++] = sym; if (s->symq2_n == 4) { s->symq2_n = 0; dqpsk_demap(s, s->symq2); } }
}

static void rate_sel(rx11b_t* s, so_c16 chip)                          /* TBB11bRxRateSel::Process (PHY_11b.hpp:421-452) */
{
    const int port = s->rxrate_state;
    if (port > RATE_2M) return;                                         /* CCK branches: not restated */
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
}

static void graph_reset(rx11b_t* s)                                    /* BB11bDemodCtx.reset() + pRxSource->Reset() (fb11b_demod.cpp:68-70) */
{
    s->error_code = SO_E_SUCCESS; s->power_detected = 0; s->rxrate_state = RATE_SYNC; s->plcp_data = 0;
    s->average_energy = 0; memset(s->window, 0, sizeof(s->window)); s->idx = 0; s->count = 0;      /* TEnergyDetect::__init */
    s->update_cnt = 8; s->sum_dc = so_c(0, 0);                                                     /* TDCEstimator::__init */
    /* stq (the switch's port towards TSymTiming) is NOT cleared: the switch's Reset only forwards (PHY_11b.hpp:336-339) */
    s->m_index = 2; s->m_frag = 0;                                                                 /* TSymTiming::_init */
    s->sync_flag = NO_PEAK_FOUND; s->last_peak_cnt = -1; s->m_max = 0; s->search_count = 0; memset(s->partial, 0, sizeof(s->partial));
    memset(s->chipq_n, 0, sizeof(s->chipq_n));
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
                uint32_t off = s->data_rate_kbps == 1000 ? 8 * 11 * 4 : s->data_rate_kbps == 2000 ? 4 * 11 * 4 : 0;
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
