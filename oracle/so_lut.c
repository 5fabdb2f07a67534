/* so_lut.c -- oracle look-up tables, restated as closed forms (TEST INFRASTRUCTURE).
 *
 * The reference ships these as literal tables; each one is regenerated here from a formula that was
 * checked entry-for-entry against the reference header (tests/test_oracle_vs_reference.py re-checks
 * whenever oracle/_ref is available, tests/test_oracle_luts.py pins their sha256 everywhere):
 *   usin_lut/ucos_lut [65536]   core/inc/intalglut.h:4,3648   = round(32767*sin|cos(i*3.141593/32768))
 *   uatan2_lut [256][256]       core/inc/intalglut.h:7332     = trunc(atan2(y,x)*32768/3.141593), y,x int8
 *        (the generator used PI = 3.141593; 32768 saturates to 32767)
 *   wFFTLUT<N>_<k>              core/inc/fft_lut_twiddle.h    = trunc(32767*cos), trunc(-32767*sin) of 2*pi*k*j/N
 *   FFT<N>LUTMap                core/inc/fft_lut_bitreversal.h = bit reversal
 *   DemapperCore LUTs           bb/Brick11/src/demapper.h:55-130 = step functions (break points below)
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <string.h>
#include <stdlib.h>
#include "so_oracle.h"
#include "so_internal.h"

static int16_t g_usin[65536], g_ucos[65536], g_uatan2[256 * 256];
static uint8_t g_demap[4][256];
static int16_t g_tw[8][3][2 * 64];       /* log2(n) index (3..8) -> k-1 -> pairs */
static so_c16  g_sts[16][16];
uint32_t so_g_crc_lut[256];
uint8_t  so_g_scr_lut[128];
static int g_inited;

#define SO_GEN_PI 3.141593               /* the constant the reference's table generator used */

static void gen_trig(void)
{
    for (int i = 0; i < 65536; i++) {
        double x = (double)i * SO_GEN_PI / 32768.0;
        g_usin[i] = (int16_t)floor(32767.0 * sin(x) + 0.5);
        g_ucos[i] = (int16_t)floor(32767.0 * cos(x) + 0.5);
    }
    for (int yi = 0; yi < 256; yi++)
        for (int xi = 0; xi < 256; xi++) {
            int y = (int8_t)yi, x = (int8_t)xi;
            double v = atan2((double)y, (double)x) * 32768.0 / SO_GEN_PI;
            int t = (int)v;                           /* truncation toward zero */
            if (t > 32767) t = 32767;
            g_uatan2[yi * 256 + xi] = (int16_t)t;
        }
}

/* step functions: value changes at the listed (signed) inputs; input is the clamped (x>>4) in [-128,127] */
typedef struct { int at; int val; } so_step;
static const so_step st_bpsk[]   = {{-128,0},{-30,1},{-17,2},{-8,3},{0,4},{9,5},{18,6},{31,7}};
static const so_step st_q16_2[]  = {{-128,0},{-70,1},{-67,2},{-65,3},{-63,4},{-61,5},{-58,6},{-55,7},{56,6},{59,5},{62,4},{64,3},{66,2},{68,1},{71,0}};
static const so_step st_q64_2[]  = {{-128,0},{-68,1},{-65,2},{-63,3},{-61,4},{-60,5},{-58,6},{-55,7},{56,6},{59,5},{61,4},{62,3},{64,2},{66,1},{69,0}};
static const so_step st_q64_3[]  = {{-128,0},{-98,1},{-96,2},{-94,3},{-92,4},{-90,5},{-89,6},{-86,7},{-37,6},{-34,5},{-32,4},{-30,3},{-29,2},{-27,1},{-24,0},
                                    {25,1},{28,2},{30,3},{31,4},{33,5},{35,6},{38,7},{87,6},{90,5},{91,4},{93,3},{95,2},{97,1},{99,0}};
static void gen_step(uint8_t* lut, const so_step* st, int n)
{
    for (int v = -128; v < 128; v++) {
        int val = 0;
        for (int k = 0; k < n; k++) if (v >= st[k].at) val = st[k].val;
        lut[(uint8_t)v] = (uint8_t)val;               /* table is indexed by (uchar)value, demapper.h:19-25 */
    }
}

static void gen_twiddle(void)
{
    for (int lg = 3; lg <= 8; lg++) {
        int n = 1 << lg;
        for (int k = 1; k <= 3; k++)
            for (int j = 0; j < n / 4; j++) {
                double a = 2.0 * M_PI * (double)(k * j) / (double)n;
                g_tw[lg - 1][k - 1][2 * j]     = (int16_t)(32767.0 * cos(a));    /* C cast = trunc */
                g_tw[lg - 1][k - 1][2 * j + 1] = (int16_t)(-32767.0 * sin(a));
            }
    }
    /* wFFTLUT8 (fft_lut_twiddle.h:61575-61581) is {W8^0, W8^1, W8^0, W8^3} */
    int16_t* t8 = g_tw[3 - 1][0];
    int16_t w1r = (int16_t)(32767.0 * cos(M_PI / 4)), w1i = (int16_t)(-32767.0 * sin(M_PI / 4));
    int16_t w3r = (int16_t)(32767.0 * cos(3 * M_PI / 4)), w3i = (int16_t)(-32767.0 * sin(3 * M_PI / 4));
    t8[0] = 32767; t8[1] = 0; t8[2] = w1r; t8[3] = w1i; t8[4] = 32767; t8[5] = 0; t8[6] = w3r; t8[7] = w3i;
}

static void gen_misc(void)
{
    for (uint32_t i = 0; i < 256; i++) {              /* reflected CRC-32, poly 0xEDB88320 (CRC32.h:5-74) */
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        so_g_crc_lut[i] = c;
    }
    for (int i = 0; i < 128; i++) {                   /* scramble.hpp:188-203 / :279-295 */
        uint8_t x = (uint8_t)(i << 1);
        for (int k = 0; k < 8; k++) {
            uint8_t o1 = ((x >> 1) ^ (x >> 4)) & 1;
            x = (uint8_t)((x >> 1) | (o1 << 7));
        }
        so_g_scr_lut[i] = x;
    }
    /* STS correlation patterns: brick/inc/sequence.h:5-33 + cca.hpp:268-277 */
    so_c16 f[64], t[64];
    memset(f, 0, sizeof(f));
    const int16_t M = 10000;
    f[4].re = f[4].im = -M;   f[8].re = f[8].im = -M;   f[12].re = f[12].im = M;
    f[16].re = f[16].im = M;  f[20].re = f[20].im = M;  f[24].re = f[24].im = M;
    f[64 - 24].re = f[64 - 24].im = M;  f[64 - 20].re = f[64 - 20].im = -M; f[64 - 16].re = f[64 - 16].im = M;
    f[64 - 12].re = f[64 - 12].im = -M; f[64 - 8].re = f[64 - 8].im = -M;   f[64 - 4].re = f[64 - 4].im = M;
    so_ifft64(f, t);
    for (int i = 0; i < 16; i++) memcpy(g_sts[i], &t[i], 16 * sizeof(so_c16));
}

void so_init(void)
{
    if (g_inited) return;
    gen_trig();
    gen_step(g_demap[0], st_bpsk,  (int)(sizeof(st_bpsk) / sizeof(so_step)));
    gen_step(g_demap[1], st_q16_2, (int)(sizeof(st_q16_2) / sizeof(so_step)));
    gen_step(g_demap[2], st_q64_2, (int)(sizeof(st_q64_2) / sizeof(so_step)));
    gen_step(g_demap[3], st_q64_3, (int)(sizeof(st_q64_3) / sizeof(so_step)));
    gen_twiddle();
    g_inited = 1;        /* gen_misc uses so_ifft64 -> so_twiddle */
    gen_misc();
}

const int16_t* so_usin_lut(void)   { so_init(); return g_usin; }
const int16_t* so_ucos_lut(void)   { so_init(); return g_ucos; }
const int16_t* so_uatan2_lut(void) { so_init(); return g_uatan2; }
const uint8_t* so_demap_lut(int w) { so_init(); return g_demap[w & 3]; }
const so_c16*  so_sts_pattern(void){ so_init(); return &g_sts[0][0]; }
const int16_t* so_twiddle(int n, int k)
{
    so_init();
    int lg = 0; while ((1 << lg) < n) lg++;
    if (lg < 3 || lg > 8 || k < 1 || k > 3) return NULL;
    return g_tw[lg - 1][k - 1];
}

/* core/inc/intalg.h:58-113 */
static int bit_scope_u(uint32_t x) { int p = 0; while (x >>= 1) p++; return p; }   /* highest set bit position; 0 for 0 and 1 */
int16_t so_uatan2(int y, int x)
{
    so_init();
    /* bit_scope: position of the highest set bit of |v| (bit_high_pos_lut[0] == 0) */
    int ys = bit_scope_u((uint32_t)(y > 0 ? y : -y));
    int xs = bit_scope_u((uint32_t)(x > 0 ? x : -x));
    int shift = (xs > ys ? xs : ys) - 6;
    if (shift > 0) return g_uatan2[(uint8_t)(y >> shift) * 256 + (uint8_t)(x >> shift)];
    return g_uatan2[(uint8_t)y * 256 + (uint8_t)x];
}
