/* so_tx11a.c -- 802.11a transmitter of the oracle: the test-vector generator (TEST INFRASTRUCTURE).
 *
 * Restates the modulation graph kernel/bb/demod11/fb11amod_config.hpp:74-110
 *   TBB11aSrc -> T11aSc -> TBB11aMRSelect -> TConvEncode_{12,23,34} -> T11aInterleave* -> TMap11a*
 *   -> T11aAddPilot -> TIFFTx -> TPackSample16to8 -> TModSink
 * and the preamble source kernel/bb/Brick11/src/preamble11a.hpp:19-140.  Output: COMPLEX8 @40 MHz,
 * exactly what `demod11 -m` writes; `demod11 -c` then expands each component <<8 into an RX_BLOCK dump
 * (kernel/bb/demod11/modulate11a.cpp:131-190).
 */
#include <string.h>
#include <stdlib.h>
#include "so_oracle.h"
#include "so_internal.h"

static const char LTS_Positive_table[64] = {       /* ieee80211const.h:23-28 */
    0,1,0,0,1,1,0,1,0,1,0,0,0,0,0,1, 1,0,0,1,0,1,0,1,1,1,1,0,0,0,0,0,
    0,0,0,0,0,0,1,1,0,0,1,1,0,1,0,1, 1,1,1,1,1,0,0,1,1,0,1,0,1,1,1,1 };
static const char PilotSgnTx[128] = {              /* pilot.hpp:10-28 */
     0, 0, 0,-1,-1,-1, 0,-1, -1,-1,-1, 0, 0,-1, 0,-1, -1, 0, 0,-1, 0, 0,-1, 0,  0, 0, 0, 0, 0,-1, 0, 0,
     0,-1, 0, 0,-1,-1, 0, 0,  0,-1, 0,-1,-1,-1, 0,-1,  0,-1,-1, 0,-1,-1, 0, 0,  0, 0, 0,-1,-1, 0, 0,-1,
    -1, 0,-1, 0,-1, 0, 0,-1, -1,-1, 0, 0,-1,-1,-1,-1,  0,-1,-1, 0,-1, 0, 0, 0,  0,-1, 0,-1, 0,-1, 0,-1,
    -1,-1,-1,-1, 0,-1, 0, 0, -1, 0,-1, 0, 0, 0,-1,-1,  0,-1,-1,-1, 0, 0, 0,-1, -1,-1,-1,-1,-1,-1, 0, 0 };

#define BPSK_MOD 10720                               /* mapper11a.hpp:8-11 */
static const int16_t kmod_of[7] = { 0, BPSK_MOD, (int16_t)(BPSK_MOD / 1.414), 0, (int16_t)(BPSK_MOD / 3.162), 0, (int16_t)(BPSK_MOD / 6.481) };

static int8_t sat8(int16_t v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }    /* _mm_packs_epi16 (stdbrick.hpp:430) */

/* preamble11a.hpp:19-100: 640 samples @40 MHz */
static void build_preamble(so_c16 lut[640])
{
    so_c16 f[128], t[128];
    const int16_t sts_mod = (int16_t)(uint16_t)(1.0 * BPSK_MOD * 1.472), lts_mod = BPSK_MOD;
    memset(f, 0, sizeof(f));
    f[4].re = f[4].im = (int16_t)-sts_mod;   f[8].re = f[8].im = (int16_t)-sts_mod; f[12].re = f[12].im = sts_mod;
    f[16].re = f[16].im = sts_mod;  f[20].re = f[20].im = sts_mod;  f[24].re = f[24].im = sts_mod;
    f[104].re = f[104].im = sts_mod; f[108].re = f[108].im = (int16_t)-sts_mod; f[112].re = f[112].im = sts_mod;
    f[116].re = f[116].im = (int16_t)-sts_mod; f[120].re = f[120].im = (int16_t)-sts_mod; f[124].re = f[124].im = sts_mod;
    so_ifft128(f, t);                                /* IFFTSSEEx<128> then FFTLUTMapTable reorder; >>4 commutes with the reorder */
    for (int i = 0; i < 128; i++) lut[i] = so_sra(t[i], 4);
    for (int i = 0; i < 192; i++) lut[128 + i] = lut[i];                  /* forward, overlapping copy: periodic extension */
    lut[0] = so_sra(lut[0], 1); lut[1] = so_sra(lut[1], 1); lut[318] = so_sra(lut[318], 1); lut[319] = so_sra(lut[319], 1);

    memset(f, 0, sizeof(f));
    for (int i = 1; i <= 26; i++)      f[i].re = LTS_Positive_table[i] ? lts_mod : (int16_t)-lts_mod;
    for (int i = 64 - 26; i < 64; i++) f[i + 64].re = LTS_Positive_table[i] ? lts_mod : (int16_t)-lts_mod;
    so_ifft128(f, t);
    for (int i = 0; i < 128; i++) lut[320 + 64 + i] = so_sra(t[i], 4);
    for (int i = 0; i < 128; i++) lut[320 + 64 + 128 + i] = lut[320 + 64 + i];
    for (int i = 0; i < 64; i++)  lut[320 + i] = lut[640 - 64 + i];      /* GI2 */
    lut[320] = so_sra(lut[320], 1); lut[321] = so_sra(lut[321], 1); lut[638] = so_sra(lut[638], 1); lut[639] = so_sra(lut[639], 1);
}

/* conv_enc.hpp:6-14: generator polynomials on (newest bit x, state s with the previous bit at bit 5) */
static int G0(uint32_t x, uint32_t s) { return (int)((x ^ (s >> 4) ^ (s >> 3) ^ (s >> 1) ^ s) & 1); }
static int G1(uint32_t x, uint32_t s) { return (int)((x ^ s ^ (s >> 3) ^ (s >> 4) ^ (s >> 5)) & 1); }

typedef struct { uint8_t* b; uint32_t n, cap; } bitbuf;                 /* one bit per byte */
static void bb_push(bitbuf* q, int bit) { if (q->n == q->cap) { q->cap = q->cap * 2 + 1024; q->b = (uint8_t*)realloc(q->b, q->cap); } q->b[q->n++] = (uint8_t)bit; }

/* Encode a bit stream (LSB-first bytes) with puncturing pattern of the given code rate.
 * TConvEncode_12/_23/_34 (conv_enc.hpp:18-330): 1/2 -> A B; 2/3 -> A B A per 2 bits; 3/4 -> A B A B' (A1 B1 A2 B3). */
static void conv_encode(const uint8_t* bytes, uint32_t nbytes, int code_rate, bitbuf* out)
{
    uint32_t s = 0; uint32_t phase = 0;
    for (uint32_t i = 0; i < nbytes * 8; i++) {
        uint32_t x = (bytes[i >> 3] >> (i & 7)) & 1;
        int a = G0(x, s), b = G1(x, s);
        if (code_rate == SO_CR_12) { bb_push(out, a); bb_push(out, b); }
        else if (code_rate == SO_CR_23) { if (phase == 0) { bb_push(out, a); bb_push(out, b); } else bb_push(out, a); phase = (phase + 1) % 2; }
        else { if (phase == 0) { bb_push(out, a); bb_push(out, b); } else if (phase == 1) bb_push(out, a); else bb_push(out, b); phase = (phase + 1) % 3; }
        s = (s >> 1) | (x << 5);
    }
}

static uint32_t gray2bin(uint32_t g) { for (uint32_t sh = 1; sh < 32; sh <<= 1) g ^= g >> sh; return g; }
static uint32_t bitrev_n(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; i++) r |= ((v >> i) & 1) << (n - 1 - i); return r; }
/* InitQamMapLut (mapper11a.hpp:16-43): M bits (first-transmitted = MSB after reversal), Gray -> level */
static int16_t qam_level(const uint8_t* bits, int M, int16_t kmod)
{
    uint32_t rg = 0; for (int i = 0; i < M; i++) rg |= (uint32_t)bits[i] << i;
    uint32_t b = gray2bin(bitrev_n(rg, M));
    int l = (int)b * 2 - ((1 << M) - 1);
    return (int16_t)(l * kmod);
}

/* one OFDM symbol: ncbps coded bits -> interleave -> map -> pilots -> 128-pt IFFT + GI -> 160 COMPLEX8 */
static void emit_symbol(const uint8_t* cbits, int nbpsc, int pilot_index, int8_t* out8)
{
    const int N = 48 * nbpsc, s = nbpsc / 2 > 1 ? nbpsc / 2 : 1;
    uint8_t ib[288];
    for (int k = 0; k < N; k++) {                                         /* interleave.hpp:43-58 (I_SS = 1) */
        int i = (N / 16) * (k % 16) + k / 16;
        int j = s * (i / s) + (i + N - (16 * i) / N) % s;
        ib[j] = cbits[k];
    }
    so_c16 car[48];
    const int16_t kmod = kmod_of[nbpsc];
    for (int c = 0; c < 48; c++) {
        const uint8_t* b = ib + c * nbpsc;
        if (nbpsc == 1) car[c] = so_c(b[0] ? BPSK_MOD : -BPSK_MOD, 0);
        else car[c] = so_c(qam_level(b, nbpsc / 2, kmod), qam_level(b + nbpsc / 2, nbpsc / 2, kmod));
    }
    so_c16 f64[64]; memset(f64, 0, sizeof(f64));                          /* T11aAddPilot (pilot.hpp:76-118) */
    const so_c16* in = car;
    for (int i = 64 - 26; i < 64; i++) { if (i == 64 - 7 || i == 64 - 21) continue; f64[i] = *in++; }
    for (int i = 1; i <= 26; i++)      { if (i == 7 || i == 21) continue; f64[i] = *in++; }
    int16_t p = PilotSgnTx[pilot_index] ? (int16_t)-BPSK_MOD : BPSK_MOD;
    f64[7].re = p; f64[21].re = (int16_t)-p; f64[64 - 7].re = p; f64[64 - 21].re = p;
    f64[7].im = f64[21].im = f64[64 - 7].im = f64[64 - 21].im = 0;

    so_c16 f[128], t[128], sym[160];                                     /* TIFFTx (fft.hpp:21-59) */
    memset(f, 0, sizeof(f));
    memcpy(f, f64, 32 * sizeof(so_c16)); memcpy(f + 96, f64 + 32, 32 * sizeof(so_c16));
    so_ifft128(f, t);
    for (int i = 0; i < 128; i++) sym[32 + i] = so_sra(t[i], 4);
    for (int i = 0; i < 32; i++) sym[i] = sym[128 + i];                   /* GI = last 32 */
    sym[0] = so_sra(sym[0], 1); sym[1] = so_sra(sym[1], 1); sym[158] = so_sra(sym[158], 1); sym[159] = so_sra(sym[159], 1);
    for (int i = 0; i < 160; i++) { out8[2 * i] = sat8(sym[i].re); out8[2 * i + 1] = sat8(sym[i].im); }
}

static int ndbps_tx(uint32_t kbps)
{
    switch (kbps) { case 6000: return 24; case 9000: return 36; case 12000: return 48; case 18000: return 72;
                    case 24000: return 96; case 36000: return 144; case 48000: return 192; case 54000: return 216; }
    return 0;
}
static int rate_code(uint32_t kbps)                 /* ieee80211const.h:3-10 */
{
    switch (kbps) { case 6000: return 0xB; case 9000: return 0xF; case 12000: return 0xA; case 18000: return 0xE;
                    case 24000: return 0x9; case 36000: return 0xD; case 48000: return 0x8; case 54000: return 0xC; }
    return 0;
}

int so_tx11a(const uint8_t* mpdu_nofcs, uint32_t len, uint32_t rate_kbps, uint8_t scramble_seed,
             int8_t* out8, uint32_t max_samples)
{
    so_init();
    const int nd = ndbps_tx(rate_kbps), rc = rate_code(rate_kbps);
    if (!nd || !rc) return -1;
    int nbpsc, cr;
    switch (rate_kbps) { case 6000: nbpsc = 1; cr = SO_CR_12; break; case 9000: nbpsc = 1; cr = SO_CR_34; break;
        case 12000: nbpsc = 2; cr = SO_CR_12; break; case 18000: nbpsc = 2; cr = SO_CR_34; break;
        case 24000: nbpsc = 4; cr = SO_CR_12; break; case 36000: nbpsc = 4; cr = SO_CR_34; break;
        case 48000: nbpsc = 6; cr = SO_CR_23; break; default: nbpsc = 6; cr = SO_CR_34; break; }

    /* PLCP SIGNAL (ieee80211a_cmn.h:8-26); LENGTH counts the FCS (PHY_11a.hpp:87) */
    uint32_t sig = (uint32_t)rc | ((len + 4) << 5);
    uint32_t par = sig ^ (sig >> 16); par ^= par >> 8; par ^= par >> 4; par ^= par >> 2; par ^= par >> 1;
    sig |= (par & 1) << 17;

    /* TBB11aSrc::Process (PHY_11a.hpp:132-202): SERVICE(2) + MPDU + FCS(4) + tail(1) + pad */
    int ndp = (rate_kbps == 9000) ? nd * 2 : nd;
    uint32_t dbytes = 2 + (len + 4) + 1;
    uint32_t rem = (dbytes * 8) % (uint32_t)ndp;
    uint32_t pad_bits = rem ? (uint32_t)ndp - rem : 0;
    uint32_t npad = (pad_bits + 7) / 8;
    uint32_t nbytes = dbytes + npad;
    uint8_t* data = (uint8_t*)calloc(nbytes + 8, 1);
    uint32_t fcs = so_crc32(mpdu_nofcs, len);
    memcpy(data + 2, mpdu_nofcs, len); memcpy(data + 2 + len, &fcs, 4);
    /* T11aSc (scramble.hpp:237-251): register = previous 8 output bits; tail byte keeps only its two pad bits */
    uint8_t reg = scramble_seed;
    for (uint32_t i = 0; i < nbytes; i++) {
        reg = so_g_scr_lut[reg >> 1];
        uint8_t code = data[i] ^ reg;
        if (i == dbytes - 1) code &= 0xC0;
        data[i] = code;
    }

    uint32_t nsym_data = nbytes * 8 / (uint32_t)nd;
    uint32_t total = 640 + 160 * (1 + nsym_data);
    if (total > max_samples) { free(data); return -2; }

    so_c16 pre[640]; build_preamble(pre);
    for (int i = 0; i < 640; i++) { out8[2 * i] = sat8(pre[i].re); out8[2 * i + 1] = sat8(pre[i].im); }

    /* SIGNAL: 3 bytes, unscrambled, 6 Mbps path (enc6 -> BPSK), pilot index 127 */
    bitbuf cb = { NULL, 0, 0 };
    uint8_t sb[3] = { (uint8_t)sig, (uint8_t)(sig >> 8), (uint8_t)(sig >> 16) };
    conv_encode(sb, 3, SO_CR_12, &cb);
    emit_symbol(cb.b, 1, 127, out8 + 2 * 640);
    cb.n = 0;
    conv_encode(data, nbytes, cr, &cb);                                    /* a fresh encoder register (or enc6 after 6 zero tail bits): state 0 */
    int pidx = 0;                                                          /* m_PilotIndex 127 -> 0 after SIGNAL (pilot.hpp:66-69) */
    for (uint32_t sidx = 0; sidx < nsym_data; sidx++) {
        emit_symbol(cb.b + (size_t)sidx * 48 * (uint32_t)nbpsc, nbpsc, pidx, out8 + 2 * (640 + 160 * (1 + sidx)));
        pidx++; if (pidx >= 127) pidx = 0;
    }
    free(cb.b); free(data);
    return (int)total;
}
