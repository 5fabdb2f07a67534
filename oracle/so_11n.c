/* so_11n.c -- TEST INFRASTRUCTURE: 802.11n (row f1) bricks restated so far, stage level.
 *   so_demap11n        T11nDemap{BPSK,QPSK,QAM16,QAM64}  (kernel/bb/Brick11/src/demapper11n.hpp:89-309 over dsp_demap.h)
 *   so_deinterleave11n T11nDeinterleave{BPSK,QPSK,QAM16,QAM64}_S{0,1}  (deinterleaver_11n.hpp:4-1618)
 * Pinned by the reference's own bricks compiled from its sources (oracle/_ref/libsora_refgraph.so: ref_11n_demap,
 * ref_11n_deinterleave) -- live and through tests/golden/ref_vectors_11n.npz.
 * The soft-value tables of dsp_demap.h ("constructed at Eb/N0 about 4 dB") are step functions of the limited coordinate
 * v in [-128,127]; they are regenerated here from their run lengths (value, count from v = -128 upward), checked
 * entry for entry against the reference bricks.  The de-interleavers are the standard HT interleaver
 * (N_COL 13, N_ROW 4 N_BPSC, N_ROT 11) inverted: out[k] = in[r(k)]. */
#define _USE_MATH_DEFINES
#define _DEFAULT_SOURCE              /* M_PI under -std=c11 */
#include <math.h>
#include <string.h>
#include "so_internal.h"
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { uint8_t v, n; } run_t;
static const run_t RL_BPSK[]   = {{0,97},{1,10},{2,10},{3,11},{4,11},{5,10},{6,10},{7,97}};                       /* also QPSK */
static const run_t RL_16_0[]   = {{0,113},{1,7},{2,4},{3,4},{4,5},{5,4},{6,7},{7,112}};
static const run_t RL_16_1[]   = {{0,58},{1,3},{2,2},{3,2},{4,2},{5,3},{6,3},{7,111},{6,3},{5,3},{4,2},{3,2},{2,2},{1,3},{0,57}};
static const run_t RL_64_0[]   = {{0,122},{1,3},{2,2},{3,1},{4,2},{5,2},{6,3},{7,121}};
static const run_t RL_64_1[]   = {{0,52},{1,3},{2,2},{3,2},{4,1},{5,2},{6,3},{7,127},{6,3},{5,2},{4,1},{3,2},{2,2},{1,3},{0,51}};
static const run_t RL_64_2[]   = {{0,18},{1,2},{2,2},{3,2},{4,2},{5,1},{6,3},{7,57},{6,3},{5,2},{4,2},{3,1},{2,2},{1,3},{0,57},
                                  {1,3},{2,2},{3,1},{4,2},{5,2},{6,3},{7,57},{6,3},{5,1},{4,2},{3,2},{2,2},{1,2},{0,17}};
static uint8_t g_lut[6][256]; static int g_ready;
static void expand(uint8_t* lut, const run_t* r, int nr) { int k = 0; for (int i = 0; i < nr; i++) for (int j = 0; j < r[i].n; j++) lut[k++] = r[i].v; }
static void init11n(void)
{
    if (g_ready) return;
    expand(g_lut[0], RL_BPSK, 8); expand(g_lut[1], RL_16_0, 8); expand(g_lut[2], RL_16_1, 15);
    expand(g_lut[3], RL_64_0, 8); expand(g_lut[4], RL_64_1, 15); expand(g_lut[5], RL_64_2, 29);
    g_ready = 1;
}
const uint8_t* so_demap11n_lut(int which) { init11n(); return g_lut[which]; }   /* [v + 128]; 0 bpsk/qpsk, 1-2 16-QAM, 3-5 64-QAM */

static inline int lim(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }   /* demap_limit (dsp_demap.h) */

int so_demap11n(int nbpsc, const so_c16 in[64], uint8_t* out)
{
    init11n();
    int j = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int lo = pass ? 1 : 64 - 28, hi = pass ? 28 : 63;
        for (int i = lo; i <= hi; i++) {
            if (i == 64 - 21 || i == 64 - 7 || i == 7 || i == 21) continue;         /* pilots */
            const int re = lim(in[i].re) + 128, im = lim(in[i].im) + 128;
            switch (nbpsc) {
            case 1: out[j++] = g_lut[0][re]; break;                                 /* demap_bpsk_i */
            case 2: out[j++] = g_lut[0][re]; out[j++] = g_lut[0][im]; break;
            case 4: out[j++] = g_lut[1][re]; out[j++] = g_lut[2][re]; out[j++] = g_lut[1][im]; out[j++] = g_lut[2][im]; break;
            case 6: out[j++] = g_lut[3][re]; out[j++] = g_lut[4][re]; out[j++] = g_lut[5][re];
                    out[j++] = g_lut[3][im]; out[j++] = g_lut[4][im]; out[j++] = g_lut[5][im]; break;
            default: return -1;
            }
        }
    }
    return j;
}

int so_deinterleave11n_index(int nbpsc, int stream, int k)     /* r(k): where output bit k of the de-interleaver comes from */
{
    const int n = 52 * nbpsc, s = nbpsc / 2 > 1 ? nbpsc / 2 : 1, nrow = 4 * nbpsc;
    const int i = nrow * (k % 13) + k / 13;
    int j = s * (i / s) + (i + n - (13 * i) / n) % s;
    if (stream > 0) j = ((j - ((stream * 2) % 3 + 3 * (stream / 3)) * 11 * nbpsc) % n + n) % n;
    return j;
}

int so_deinterleave11n(int nbpsc, int stream, const uint8_t* in, uint8_t* out)
{
    if (nbpsc != 1 && nbpsc != 2 && nbpsc != 4 && nbpsc != 6) return -1;
    const int n = 52 * nbpsc;
    for (int k = 0; k < n; k++) out[k] = in[so_deinterleave11n_index(nbpsc, stream, k)];
    return n;
}

/* ------------------------------------------------------------------ MIMO 2x2 channel estimation / zero-forcing detection
 *   so_mimo_est11n   TMimoChannelEst  (channel_11n.hpp:329-443): the two HT-LTF symbols of each RX chain (after the FFT) ->
 *                    H (P-matrix combination, HT-LTF signs removed) and its inverse x 2^16, computed in single-precision
 *                    floats exactly as the SSE code does (brick/inc/sora_matrix.h:134-148,305-313; vector128.h:1107-1116):
 *                    every product and every sum is rounded on its own (no fused multiply-add), division is IEEE,
 *                    conversion back is round-to-nearest-even with 0x80000000 for anything out of range (cvtps2dq),
 *                    then a saturating pack.
 *   so_mimo_comp11n  TMimoChannelComp (channel_11n.hpp:445-521): x = (Hinv y) >> 9, saturating pack.
 * Pinned by the reference's own bricks (ref_11n_mimo_est / ref_11n_mimo_comp). */
#include <math.h>

static const int8_t HTLTF_K[57] = {  /* HT-LTF, carriers -28..28 (IEEE 802.11n 20 MHz) */
    1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
    1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1 };
static inline int htltf_negate(int bin)      /* _80211n_HTLTFMask: every bin whose HT-LTF value is not +1 is negated (unused bins too) */
{
    const int k = bin < 32 ? bin : bin - 64;
    return !(k >= -28 && k <= 28 && HTLTF_K[k + 28] == 1);
}
typedef struct { float re, im; } cf_t;
static inline cf_t cf_mul(cf_t a, cf_t b)    /* vcf mul: moveldup/movehdup, two mul_ps, flip, addsub_ps */
{
    const float acr = a.re * b.re, aci = a.im * b.re, adr = a.re * b.im, adi = a.im * b.im;
    cf_t r; r.re = acr - adi; r.im = aci + adr; return r;
}
static inline int32_t cvtps(float x) { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int32_t)lrintf(x) : INT32_MIN; }

void so_mimo_est11n(const so_c16 ltf0[128], const so_c16 ltf1[128], so_c16 h[2][128], so_c16 hinv[2][128])
{
    const so_c16* ltf[2] = { ltf0, ltf1 };
    for (int r = 0; r < 2; r++)
        for (int i = 0; i < 64; i++) {
            so_c16 d = so_sra(so_csubs(ltf[r][i], ltf[r][i + 64]), 1), s = so_sra(so_cadds(ltf[r][i], ltf[r][i + 64]), 1);
            if (htltf_negate(i)) { d = so_c(so_neg16(d.re), so_neg16(d.im)); s = so_c(so_neg16(s.re), so_neg16(s.im)); }
            h[r][i] = d; h[r][i + 64] = s;
        }
    for (int i = 0; i < 64; i++) {
        const cf_t a00 = { (float)h[0][i].re, (float)h[0][i].im }, a01 = { (float)h[0][i + 64].re, (float)h[0][i + 64].im };
        const cf_t a10 = { (float)h[1][i].re, (float)h[1][i].im }, a11 = { (float)h[1][i + 64].re, (float)h[1][i + 64].im };
        const cf_t ad = cf_mul(a00, a11), bc = cf_mul(a01, a10);
        const cf_t det = { ad.re - bc.re, ad.im - bc.im };
        const float t0 = det.re * det.re, t1 = det.im * det.im;
        const float n = (t0 + t1) / 65536.0f;                                     /* SquaredNorm, then div(n, scale) */
        const cf_t ds = { det.re, -det.im };
        const cf_t m01 = { -a01.re, -a01.im }, m10 = { -a10.re, -a10.im };
        const cf_t r00 = cf_mul(a11, ds), r01 = cf_mul(m01, ds), r10 = cf_mul(m10, ds), r11 = cf_mul(a00, ds);
        const cf_t* rr[4] = { &r00, &r01, &r10, &r11 };
        so_c16* dst[4] = { &hinv[0][i], &hinv[0][i + 64], &hinv[1][i], &hinv[1][i + 64] };
        for (int k = 0; k < 4; k++) {
            const float qre = rr[k]->re / n, qim = rr[k]->im / n;
            int32_t ire = cvtps(qre), iim = cvtps(qim);
            dst[k]->re = so_sat16(ire); dst[k]->im = so_sat16(iim);
        }
    }
}

void so_mimo_comp11n(const so_c16 hinv[2][128], const so_c16 y0[64], const so_c16 y1[64], so_c16 x0[64], so_c16 x1[64])
{
    for (int i = 0; i < 64; i++) {
        int32_t ar, ai, br, bi;
        so_mul32(hinv[0][i], y0[i], &ar, &ai); so_mul32(hinv[0][i + 64], y1[i], &br, &bi);
        x0[i] = so_c(so_sat16(so_w32((int64_t)ar + br) >> 9), so_sat16(so_w32((int64_t)ai + bi) >> 9));
        so_mul32(hinv[1][i], y0[i], &ar, &ai); so_mul32(hinv[1][i + 64], y1[i], &br, &bi);
        x1[i] = so_c(so_sat16(so_w32((int64_t)ar + br) >> 9), so_sat16(so_w32((int64_t)ai + bi) >> 9));
    }
}

/* ------------------------------------------------------------------ dsp_math (Brick11/src/dsp_math.h) and the bricks built on it
 * The 11n path does its trigonometry with two tables generated at start-up with the C library's double-precision cos / sin /
 * atan (dsp_math.h:215-245); they are generated here the same way (the parity claim is for the same libm -- the reference
 * compiled in this image uses the very same one).
 *   so_cfo_est11n       TFreqEstimator_11n (freqoffset_11n.hpp:42-160): joint estimate over both RX chains' L-LTF
 *   so_freq_comp11n     TFreqComp_11n      (freqoffset_11n.hpp:162-280): 8-sample bursts, phase ramp minus tracked phase
 *   so_pilot_track11n   TPilotTrack_11n    (pilot_11n.hpp:84-141): mean pilot phase of both streams into vfo_theta_i */
static so_c16 g_sincos[65536]; static int16_t g_atan[4097]; static int g_dsp_ready;
static void dsp_init(void)
{
    if (g_dsp_ready) return;
    for (unsigned i = 0; i < 65536; i++) {
        const double r = (double)i * 2.0 * M_PI / 65535.0;
        g_sincos[i].re = (int16_t)(cos(r) * 32767.5); g_sincos[i].im = (int16_t)(sin(r) * 32767.5);
    }
    for (int i = 0; i <= 4096; i++) g_atan[i] = (int16_t)(atan((double)i / 4096.0) / (M_PI / 4.0) * 8192);
    g_dsp_ready = 1;
}
const so_c16* so_dsp_sincos_table(void) { dsp_init(); return g_sincos; }
const int16_t* so_dsp_atan_table(void) { dsp_init(); return g_atan; }

static inline int32_t iabs32(int32_t x) { const int32_t t = x >> 31; return so_w32((int64_t)(x ^ t) - t); }
static inline int32_t imax32(int32_t x, int32_t y) { return (int32_t)(so_w32((int64_t)so_w32((int64_t)x + y) + iabs32(so_w32((int64_t)x - y))) >> 1); }
int16_t so_dsp_atan16(int16_t x, int16_t y)                               /* dsp_math::atan(short, short) */
{
    dsp_init();
    const int16_t sign = (int16_t)((x ^ y) >> 15);
    const int16_t absx = (int16_t)((x ^ (x >> 15)) - (x >> 15)), absy = (int16_t)((y ^ (y >> 15)) - (y >> 15));
    const int16_t tsign = (int16_t)(((int)absx - (int)absy) >> 15);
    int tmax = absx, tmin = absy; const int tsum = tmax + tmin;
    tmax = imax32(tmax, tmin); tmin = tsum - tmax;
    if (tmax == 0) return 0;
    int idx = so_w32(((int64_t)so_w32((int64_t)tmin << 16) + (tmax >> 1))) / tmax;
    idx >>= 4;
    if (idx < 0 || idx >= 4097) return 0;
    int16_t srad = g_atan[idx];
    srad = (int16_t)((16384 & tsign) + ((srad ^ tsign) - tsign));
    srad ^= sign; srad = (int16_t)(srad - sign);
    return srad;
}
int16_t so_dsp_atan32(int32_t x, int32_t y)                               /* dsp_math::atan(int, int) */
{
    dsp_init();
    const int16_t sign = (int16_t)(((x ^ y) >> 31) & 0xFFFF);
    const int32_t absx = iabs32(x), absy = iabs32(y);
    const int16_t tsign = (int16_t)((so_w32((int64_t)absx - absy) >> 31) & 0xFFFF);
    int32_t tmax = absx, tmin = absy; const int32_t tsum = so_w32((int64_t)tmax + tmin);
    tmax = imax32(tmax, tmin); tmin = so_w32((int64_t)tsum - tmax);
    int64_t i64x = (int64_t)tmin << 16, i64y = tmax;
    if (i64y == 0) i64y = 1;
    int idx = (int)((i64x + (i64y >> 1)) / i64y);
    idx >>= 4;
    if (idx < 0 || idx >= 4097) return 0;
    int16_t srad = g_atan[idx];
    srad = (int16_t)((16384 & tsign) + ((srad ^ tsign) - tsign));
    srad ^= sign; srad = (int16_t)(srad - sign);
    return srad;
}

/* state[0..7] = vfo_delta_i, [8..15] = vfo_step_i, [16..23] = vfo_theta_i (CF_FreqOffset_11n); returns CFO_est */
int16_t so_cfo_est11n(const so_c16 l0[128], const so_c16 l1[128], int16_t state[24])
{
    int32_t sre = 0, sim = 0;
    const so_c16* l[2] = { l0, l1 };
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 64; i++) {
            int32_t re, im; so_conj_mul32(l[c][i], l[c][i + 64], &re, &im);
            sre = so_w32((int64_t)sre + (re >> 7)); sim = so_w32((int64_t)sim + (im >> 7));
        }
    int16_t d = so_dsp_atan32(sre, sim);
    d = (int16_t)(d >> 6);
    for (int k = 0; k < 8; k++) { state[k] = so_w16(k * d); state[8 + k] = so_w16(d << 3); state[16 + k] = 0; }
    return d;
}

void so_freq_comp11n(int16_t state[24], const so_c16* in0, const so_c16* in1, so_c16* out0, so_c16* out1, int nbursts)
{
    dsp_init();
    for (int b = 0; b < nbursts; b++) {
        for (int k = 0; k < 8; k++) {
            const so_c16 cof = g_sincos[(uint16_t)so_w16(state[k] - state[16 + k])];
            int32_t re, im;
            so_mul32(in0[8 * b + k], cof, &re, &im); out0[8 * b + k] = so_c(so_sat16(re >> 15), so_sat16(im >> 15));
            so_mul32(in1[8 * b + k], cof, &re, &im); out1[8 * b + k] = so_c(so_sat16(re >> 15), so_sat16(im >> 15));
        }
        for (int k = 0; k < 8; k++) state[k] = so_w16(state[k] + state[8 + k]);
    }
}

void so_pilot_track11n(int16_t theta[8], const so_c16 x0[64], const so_c16 x1[64])
{
    const so_c16* x[2] = { x0, x1 }; int t[2];
    for (int s = 0; s < 2; s++) {
        const int th = so_dsp_atan16(x[s][64 - 21].re, x[s][64 - 21].im) + so_dsp_atan16(x[s][64 - 7].re, x[s][64 - 7].im)
                     + so_dsp_atan16(x[s][7].re, x[s][7].im) + so_dsp_atan16(x[s][21].re, x[s][21].im);
        t[s] = (int16_t)(th >> 2);
    }
    const int16_t th = (int16_t)((t[0] + t[1]) >> 1);
    for (int k = 0; k < 8; k++) theta[k] = so_w16(theta[k] + th);
}

/* ------------------------------------------------------------------ the legacy (SISO) part of the 11n preamble
 *   so_siso_est11n   TSisoChannelEst  (channel_11n.hpp:33-231): per RX chain and L-LTF half c = trunc((x << 16 + |x|^2 / 2) / |x|^2),
 *                    saturating pack, conjugate and L-LTF sign folded into one masked negation (_80211_LLTFMask: im negated where
 *                    the L-LTF is +1, re elsewhere), the two halves averaged with a wrapping add; bins 28..35 are never written
 *                    by the brick (0 here)
 *   so_siso_comp11n  TSisoChannelComp (channel_11n.hpp:233-297): sat((y * c) >> 9) per chain
 *   so_mrc11n        TMrcCombine      (PHY_11n.hpp:362-398): (a + b) >> 1, wrapping add
 *   so_sig_demap11n  T11nSigDemap     (demapper11n.hpp:6-87): L-SIG on I, the two HT-SIG symbols on Q (rotated BPSK), 48 carriers each */
static const int8_t LLTF_K[53] = {   /* L-LTF, carriers -26..26 */
    1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
    1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1 };
static inline int lltf_plus(int bin) { const int k = bin < 32 ? bin : bin - 64; return k >= -26 && k <= 26 && LLTF_K[k + 26] == 1; }

/* The rounding term is the reference's, lane for lane: the vector of four |x|^2 >> 1 (one per carrier) is added to vectors that hold
 * (re, im) pairs, so component c of carrier j of a group of four gets |x[(2j + c) mod 4]|^2 >> 1 (channel_11n.hpp:47-56). */
static so_c16 siso_one(const so_c16* x4, int j, int bin)
{
    int32_t sq = so_sqnorm(x4[j]);
    if (sq == 0) sq = 1;
    const int32_t hr = so_sqnorm(x4[(2 * j) & 3]) >> 1, hi = so_sqnorm(x4[(2 * j + 1) & 3]) >> 1;
    const int32_t re = so_w32(((int64_t)x4[j].re << 16) + hr) / sq, im = so_w32(((int64_t)x4[j].im << 16) + hi) / sq;
    so_c16 c = so_c(so_sat16(re), so_sat16(im));
    if (lltf_plus(bin)) c.im = so_neg16(c.im); else c.re = so_neg16(c.re);
    return c;
}
void so_siso_est11n(const so_c16 l0[128], const so_c16 l1[128], so_c16 ch[2][64])
{
    const so_c16* l[2] = { l0, l1 };
    for (int r = 0; r < 2; r++)
        for (int i = 0; i < 64; i++) {
            if (i >= 28 && i < 36) { ch[r][i] = so_c(0, 0); continue; }
            const so_c16 a = siso_one(l[r] + (i & ~3), i & 3, i), b = siso_one(l[r] + 64 + (i & ~3), i & 3, i);
            ch[r][i] = so_c((int16_t)(so_w16(a.re + b.re) >> 1), (int16_t)(so_w16(a.im + b.im) >> 1));
        }
}
void so_siso_comp11n(const so_c16 ch[2][64], const so_c16 y0[64], const so_c16 y1[64], so_c16 x0[64], so_c16 x1[64])
{
    for (int i = 0; i < 64; i++) {
        int32_t re, im;
        so_mul32(y0[i], ch[0][i], &re, &im); x0[i] = so_c(so_sat16(re >> 9), so_sat16(im >> 9));
        so_mul32(y1[i], ch[1][i], &re, &im); x1[i] = so_c(so_sat16(re >> 9), so_sat16(im >> 9));
    }
}
void so_mrc11n(const so_c16 a[64], const so_c16 b[64], so_c16 out[64])
{
    for (int i = 0; i < 64; i++) out[i] = so_c((int16_t)(so_w16(a[i].re + b[i].re) >> 1), (int16_t)(so_w16(a[i].im + b[i].im) >> 1));
}
void so_sig_demap11n(const so_c16 sym[192], uint8_t soft[144])
{
    init11n();
    int j = 0;
    for (int s = 0; s < 3; s++)
        for (int pass = 0; pass < 2; pass++) {
            const int lo = pass ? 1 : 64 - 26, hi = pass ? 26 : 63;
            for (int i = lo; i <= hi; i++) {
                if (i == 64 - 21 || i == 64 - 7 || i == 7 || i == 21) continue;
                const so_c16 v = sym[64 * s + i];
                soft[j++] = g_lut[0][lim(s == 0 ? v.re : v.im) + 128];
            }
        }
}

/* ------------------------------------------------------------------ L-SIG / HT-SIG decoding
 * T11aDeinterleaveBPSK on each of the three symbols -> T11nViterbiSig (viterbi.hpp:51-99: Viterbi_sig11 over 24 and over 48 trellis
 * steps, each >> 6 to drop the zero prefix; 3 + 6 output bytes) -> T11nSigParser (PHY_11n.hpp:432-513).  The parser's context fields
 * start at 0 and are written exactly where the reference writes them, so a failed parse reports the same partial state. */
static uint8_t crc8_htsig(const uint8_t* p)                                 /* CalcCRC8(p, 4, 2) (core/inc/CRC8.h): reflected, poly 0xE0 */
{
    uint8_t crc = 0xFF;
    for (int i = 0; i < 4; i++) { crc ^= p[i]; for (int b = 0; b < 8; b++) crc = (crc & 1) ? (uint8_t)((crc >> 1) ^ 0xE0) : (uint8_t)(crc >> 1); }
    crc ^= p[4] & 3;
    for (int b = 0; b < 2; b++) crc = (crc & 1) ? (uint8_t)((crc >> 1) ^ 0xE0) : (uint8_t)(crc >> 1);
    return (uint8_t)~crc;
}
int so_sig_decode11n(const uint8_t soft[144], uint8_t out9[9], uint32_t f[9])
{
    static const uint32_t rate[16] = { 0,0,0,0,0,0,0,0, 48000,24000,12000,6000,54000,36000,18000,9000 };   /* ieee80211a_cmn.h:97-107 */
    static const int ndbps[3] = { 52, 104, 156 };                            /* DOT11N_RATE_PARAMS[8..10] (ieee80211const.h:46-49) */
    uint8_t di[144], b[8];
    for (int s = 0; s < 3; s++) so_deinterleave(1, soft + 48 * s, di + 48 * s);
    so_viterbi_sig_bits(di, 24, b);
    const uint32_t lsig = ((uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16) >> 6;
    so_viterbi_sig_bits(di + 48, 48, b);
    uint64_t ht = 0; for (int i = 0; i < 6; i++) ht |= (uint64_t)b[i] << (8 * i);
    ht >>= 6;
    out9[0] = (uint8_t)lsig; out9[1] = (uint8_t)(lsig >> 8); out9[2] = (uint8_t)(lsig >> 16);
    for (int i = 0; i < 6; i++) out9[3 + i] = (uint8_t)(ht >> (8 * i));
    memset(f, 0, 9 * sizeof(uint32_t));
    const uint8_t* ip = out9 + 3;
    int ok = 0;
    do {
        /* _parse_plcp on the 32-bit read of bytes 0..3 (byte 3 = HT-SIG byte 0 is masked off) */
        const uint32_t sig = lsig & 0xFFFFFF;
        if (sig & 0xFC0010) break;
        uint32_t par = (sig >> 16) ^ sig; par ^= par >> 8; par ^= par >> 4; par ^= par >> 2; par ^= par >> 1;
        if (par & 1) break;
        f[1] = rate[sig & 0xF];
        if (f[1] == 0) break;
        f[2] = ((sig >> 5) & 0xFFF) * 2;
        if (f[2] > 1500) break;
        /* _parse_htsig: the comparison is made in int, so any of the six tail bits set (ip[5] >> 2) fails it */
        if ((int)crc8_htsig(ip) != ((ip[4] >> 2) | (ip[5] << 6))) break;        /* mcs and length stay 0 */
        f[3] = ip[0] & 0x7F;
        if (f[3] < 8 || f[3] >= 11) break;
        f[4] = (uint32_t)ip[1] | (uint32_t)ip[2] << 8;
        if (f[4] > 1500) break;
        f[5] = (f[3] % 8 == 2) ? SO_CR_34 : SO_CR_12;                        /* BB11nGetCodingRateFromMcsIndex for MCS 8..10 */
        const int bits = (int)f[4] * 8 + 16 + 6, nd = ndbps[f[3] - 8];
        f[6] = f[7] = (uint32_t)((bits + nd - 1) / nd + 4);
        f[2] = f[4];                                                         /* frame_length = ht_frame_length */
        f[8] = 3;                                                            /* SYMBOL_HT_STF, checked against the brick below */
        ok = 1;
    } while (0);
    if (!ok) f[0] = SO_E_PLCP_HEADER_FAIL;
    return ok;
}
