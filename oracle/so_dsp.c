/* so_dsp.c -- fixed-point FFT/IFFT and small primitives of the oracle (TEST INFRASTRUCTURE).
 * Scalar restatement of kernel/core/inc/fft_r4dif.h and ifft_r4dif.h: radix-4 DIF, every stage shifts
 * its inputs right by 2 (terminal 8-point stage: 3), saturating add/sub, xor-approximated negation,
 * twiddle product with wrapping pack, bit-reversed output reorder. */
#include <string.h>
#include "so_oracle.h"
#include "so_internal.h"

static inline so_c16 tw(const int16_t* t, int j) { return so_c(t[2 * j], t[2 * j + 1]); }

/* FFTSSE<N> / IFFTSSE<N>: one radix-4 DIF stage over x[0..n) (fft_r4dif.h:11-47, ifft_r4dif.h:11-47) */
static void r4_stage(so_c16* x, int n, int inverse)
{
    const int q = n / 4;
    const int16_t *t1 = so_twiddle(n, 1), *t2 = so_twiddle(n, 2), *t3 = so_twiddle(n, 3);
    for (int e = 0; e < q; e++) {
        so_c16 a = so_sra(x[e], 2), b = so_sra(x[e + q], 2), c = so_sra(x[e + 2 * q], 2), d = so_sra(x[e + 3 * q], 2);
        so_c16 ac = so_cadds(a, c), bd = so_cadds(b, d), a_c = so_csubs(a, c), b_d = so_csubs(b, d);
        x[e] = so_cadds(ac, bd);
        so_c16 x2 = so_csubs(ac, bd);
        so_c16 jb_d = so_mul_j(b_d);
        if (!inverse) {
            x[e + q]     = so_mul_shift(x2, tw(t2, e), 15);
            x[e + 2 * q] = so_mul_shift(so_csubs(a_c, jb_d), tw(t1, e), 15);
            x[e + 3 * q] = so_mul_shift(so_cadds(a_c, jb_d), tw(t3, e), 15);
        } else {
            x[e + q]     = so_conj_mul_shift(x2, tw(t2, e), 15);
            x[e + 2 * q] = so_conj_mul_shift(so_cadds(a_c, jb_d), tw(t1, e), 15);
            x[e + 3 * q] = so_conj_mul_shift(so_csubs(a_c, jb_d), tw(t3, e), 15);
        }
    }
}

/* the shared tail of FFTSSEEx<4>/<8>: (s0..s3) -> 4-point DFT without input shift
 * (fft_r4dif.h:60-83, ifft_r4dif.h:60-83) */
static void dft4_core(so_c16 s[4], int inverse)
{
    so_c16 A0 = so_cadds(s[0], s[2]), A1 = so_cadds(s[1], s[3]);
    so_c16 B0 = so_cadds(so_cnot(s[2]), s[0]), B1 = so_cadds(so_cnot(s[3]), s[1]);
    so_c16 B1r = inverse ? so_c((int16_t)~B1.im, B1.re)      /* ~ +j*B1 */
                         : so_c(B1.im, (int16_t)~B1.re);      /* ~ -j*B1 */
    s[0] = so_cadds(A0, A1);
    s[1] = so_cadds(so_cnot(A1), A0);
    s[2] = so_cadds(B0, B1r);
    s[3] = so_cadds(so_cnot(B1r), B0);
}

static void t4(so_c16* x, int inverse)           /* FFTSSEEx<4> */
{
    so_c16 s[4];
    for (int i = 0; i < 4; i++) s[i] = so_sra(x[i], 2);
    dft4_core(s, inverse);
    memcpy(x, s, sizeof(s));
}

static void t8(so_c16* x, int inverse)           /* FFTSSEEx<8> (fft_r4dif.h:86-130, ifft_r4dif.h:86-130) */
{
    so_c16 a[4], b[4], d[4], s[4], e[4], g[4], f[4];
    const int16_t* w = so_twiddle(8, 1);
    for (int i = 0; i < 4; i++) { a[i] = so_sra(x[i], 3); b[i] = so_sra(x[4 + i], 3); }
    for (int i = 0; i < 4; i++) { d[i] = so_csubs(a[i], b[i]); s[i] = so_cadds(a[i], b[i]); }
    e[0] = d[0]; e[1] = d[1];
    for (int i = 2; i < 4; i++)
        e[i] = inverse ? so_c((int16_t)~d[i].im, d[i].re) : so_c(d[i].im, (int16_t)~d[i].re);
    g[0] = so_cadds(e[0], e[2]);           g[1] = so_cadds(e[1], e[3]);
    g[2] = so_cadds(so_cnot(e[2]), e[0]);  g[3] = so_cadds(so_cnot(e[3]), e[1]);
    for (int i = 0; i < 4; i++)
        f[i] = inverse ? so_conj_mul_shift(g[i], tw(w, i), 15) : so_mul_shift(g[i], tw(w, i), 15);
    x[4] = so_cadds(f[0], f[1]);  x[5] = so_cadds(so_cnot(f[1]), f[0]);
    x[6] = so_cadds(f[2], f[3]);  x[7] = so_cadds(so_cnot(f[3]), f[2]);
    dft4_core(s, inverse);
    memcpy(x, s, sizeof(s));
}

static void r4_rec(so_c16* x, int n, int inverse)   /* FFTSSEEx<N> (fft_r4dif.h:49-58) */
{
    if (n == 4) { t4(x, inverse); return; }
    if (n == 8) { t8(x, inverse); return; }
    r4_stage(x, n, inverse);
    for (int k = 0; k < 4; k++) r4_rec(x + k * (n / 4), n / 4, inverse);
}

static void fft_any(const so_c16* in, so_c16* out, int n, int inverse)
{
    so_c16 t[256];
    int lg = 0; while ((1 << lg) < n) lg++;
    memcpy(t, in, (size_t)n * sizeof(so_c16));
    r4_rec(t, n, inverse);
    for (int i = 0; i < n; i++) {                    /* FFT<N>LUTMap = bit reversal (fft_r4dif.h:137-139) */
        int r = 0; for (int k = 0; k < lg; k++) r |= ((i >> k) & 1) << (lg - 1 - k);
        out[i] = t[r];
    }
}

void so_fft64(const so_c16* in, so_c16* out)   { so_init(); fft_any(in, out, 64, 0); }
void so_ifft64(const so_c16* in, so_c16* out)  { so_init(); fft_any(in, out, 64, 1); }
void so_fft128(const so_c16* in, so_c16* out)  { so_init(); fft_any(in, out, 128, 0); }
void so_ifft128(const so_c16* in, so_c16* out) { so_init(); fft_any(in, out, 128, 1); }

so_c16 so_mul_q15(so_c16 a, so_c16 b)               /* vcs mul(a,b): vector128.h:1201-1211 */
{
    int32_t re, im; so_mul32(a, b, &re, &im);
    return so_c(so_w16(re >> 15), so_w16(im >> 15));
}

uint32_t so_crc32(const uint8_t* p, uint32_t n)     /* CalcCRC32: CRC32.h:81-93 */
{
    so_init();
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; i++) c = (c >> 8) ^ so_g_crc_lut[(p[i] ^ c) & 0xFF];
    return ~c;
}
