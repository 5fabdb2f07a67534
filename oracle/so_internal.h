/* so_internal.h -- shared scalar helpers of the oracle (TEST INFRASTRUCTURE).
 * Each helper restates one operator of kernel/core/inc/vector128.h on ONE element. */
#ifndef SO_INTERNAL_H
#define SO_INTERNAL_H
#include <stdint.h>
#include "so_oracle.h"

extern uint32_t so_g_crc_lut[256];
extern uint8_t  so_g_scr_lut[128];

static inline int16_t so_w16(int32_t v) { return (int16_t)(uint16_t)(uint32_t)v; }           /* wrapping pack (vector128.h:876-883) */
static inline int16_t so_sat16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }
static inline int16_t so_adds(int16_t a, int16_t b) { return so_sat16((int32_t)a + b); }      /* _mm_adds_epi16 */
static inline int16_t so_subs(int16_t a, int16_t b) { return so_sat16((int32_t)a - b); }      /* _mm_subs_epi16 */
static inline int16_t so_neg16(int16_t a) { return so_w16(-(int32_t)a); }                     /* _mm_sign_epi16(a, <0): -(-32768) wraps */
static inline int32_t so_w32(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; }            /* _mm_madd_epi16 / _mm_add_epi32 wrap */

static inline so_c16 so_c(int16_t re, int16_t im) { so_c16 r; r.re = re; r.im = im; return r; }
static inline so_c16 so_sra(so_c16 a, int n) { return so_c((int16_t)(a.re >> n), (int16_t)(a.im >> n)); }   /* shift_right(vcs) */
static inline so_c16 so_cadds(so_c16 a, so_c16 b) { return so_c(so_adds(a.re, b.re), so_adds(a.im, b.im)); }
static inline so_c16 so_csubs(so_c16 a, so_c16 b) { return so_c(so_subs(a.re, b.re), so_subs(a.im, b.im)); }
static inline so_c16 so_cnot(so_c16 a) { return so_c((int16_t)~a.re, (int16_t)~a.im); }                      /* xor all-ones: -x-1 */
static inline so_c16 so_mul_j(so_c16 a) { return so_c((int16_t)~a.im, a.re); }                               /* mul_j, vector128.h:1258-1261 */

/* mul(vi& re, vi& im, a, b): a*b, 32-bit (vector128.h:1075-1081) */
static inline void so_mul32(so_c16 a, so_c16 b, int32_t* re, int32_t* im)
{
    *re = so_w32((int64_t)a.re * b.re + (int64_t)a.im * so_neg16(b.im));
    *im = so_w32((int64_t)a.re * b.im + (int64_t)a.im * b.re);
}
/* conj_mul(vi& re, vi& im, a, b): a*conj(b), 32-bit (vector128.h:1038-1044) */
static inline void so_conj_mul32(so_c16 a, so_c16 b, int32_t* re, int32_t* im)
{
    *re = so_w32((int64_t)a.re * b.re + (int64_t)a.im * b.im);
    *im = so_w32((int64_t)so_neg16(b.im) * a.re + (int64_t)b.re * a.im);
}
/* mul_shift(a,b,n): approximate-conjugate product used by the FFT (vector128.h:1235-1246) */
static inline so_c16 so_mul_shift(so_c16 a, so_c16 b, int n)
{
    int32_t v0 = so_w32((int64_t)a.re * b.re + (int64_t)a.im * (int16_t)~b.im);
    int32_t v1 = so_w32((int64_t)a.re * b.im + (int64_t)a.im * b.re);
    return so_c(so_w16(v0 >> n), so_w16(v1 >> n));
}
/* conj_mul_shift(a,b,n): a*conj(b)>>n used by the IFFT (vector128.h:1215-1231) */
static inline so_c16 so_conj_mul_shift(so_c16 a, so_c16 b, int n)
{
    int32_t v0 = so_w32((int64_t)a.re * b.re + (int64_t)a.im * b.im);
    int32_t v1 = so_w32((int64_t)a.im * b.re + (int64_t)so_neg16(a.re) * b.im);
    return so_c(so_w16(v0 >> n), so_w16(v1 >> n));
}
static inline int32_t so_sqnorm(so_c16 a) { return so_w32((int64_t)a.re * a.re + (int64_t)a.im * a.im); }     /* SquaredNorm */

#endif
