// dev_arith.h -- device-side fixed-point primitives of the 802.11a RX path (gfx950).
//
// Each function is the single-element meaning of one SSE operator the reference's bricks are built
// from (kernel/core/inc/vector128.h); the roles and line numbers are cited so parity can be audited.
// A COMPLEX16 lives in two sign-extended 32-bit VGPRs (re, im): int16 wrap = v_bfe_i32,
// int16 saturation = v_med3_i32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sora {

struct cpx { int re, im; };                                   // values always within int16 range

__device__ __forceinline__ int w16(int v) { return (int)(short)v; }                    // wrapping pack (vector128.h:876-883)
__device__ __forceinline__ int sat16(int v) { return min(max(v, -32768), 32767); }     // _mm_adds/subs_epi16, packs
__device__ __forceinline__ int neg16(int v) { return w16(-v); }                         // _mm_sign_epi16 by a negative: -(-32768) wraps
__device__ __forceinline__ cpx mk(int re, int im) { cpx r; r.re = re; r.im = im; return r; }
__device__ __forceinline__ cpx unpack(uint32_t u) { return mk((int)(short)(u & 0xFFFF), (int)u >> 16); }
__device__ __forceinline__ uint32_t pack(cpx a) { return ((uint32_t)a.re & 0xFFFFu) | ((uint32_t)a.im << 16); }
__device__ __forceinline__ cpx sra(cpx a, int n) { return mk(a.re >> n, a.im >> n); }  // shift_right(vcs)
__device__ __forceinline__ cpx cadds(cpx a, cpx b) { return mk(sat16(a.re + b.re), sat16(a.im + b.im)); }
__device__ __forceinline__ cpx csubs(cpx a, cpx b) { return mk(sat16(a.re - b.re), sat16(a.im - b.im)); }
__device__ __forceinline__ cpx cnot(cpx a) { return mk(~a.re, ~a.im); }                // xor all-ones: -x-1
__device__ __forceinline__ cpx mul_j(cpx a) { return mk(~a.im, a.re); }                // vector128.h:1258-1261 (approximate)

// a*b, 32-bit parts: mul(vi&,vi&,a,b) vector128.h:1075-1081
__device__ __forceinline__ void mul32(cpx a, cpx b, int& re, int& im)
{
    re = (int)((unsigned)(a.re * b.re) + (unsigned)(a.im * neg16(b.im)));
    im = (int)((unsigned)(a.re * b.im) + (unsigned)(a.im * b.re));
}
// a*conj(b), 32-bit parts: conj_mul(vi&,vi&,a,b) vector128.h:1038-1044
__device__ __forceinline__ void conj_mul32(cpx a, cpx b, int& re, int& im)
{
    re = (int)((unsigned)(a.re * b.re) + (unsigned)(a.im * b.im));
    im = (int)((unsigned)(neg16(b.im) * a.re) + (unsigned)(b.re * a.im));
}
// vcs mul(a,b): Q15 product, wrapping pack (vector128.h:1201-1211) -- TFreqCompensation, TPhaseCompensate, TPilotTrack
__device__ __forceinline__ cpx mul_q15(cpx a, cpx b)
{
    int re, im; mul32(a, b, re, im);
    return mk(w16(re >> 15), w16(im >> 15));
}
// mul_shift(a,b,15): FFT twiddle product with the xor-approximated conjugate (vector128.h:1235-1246)
__device__ __forceinline__ cpx mul_shift15(cpx a, cpx b)
{
    int v0 = (int)((unsigned)(a.re * b.re) + (unsigned)(a.im * (int)(short)~b.im));
    int v1 = (int)((unsigned)(a.re * b.im) + (unsigned)(a.im * b.re));
    return mk(w16(v0 >> 15), w16(v1 >> 15));
}
__device__ __forceinline__ int sqnorm(cpx a) { return (int)((unsigned)(a.re * a.re) + (unsigned)(a.im * a.im)); }   // SquaredNorm

// ---------------------------------------------------------------------------------------------
// Tables living in HBM/L2 (uploaded once per handle by the host side).
struct Tables {
    const short*    usin;        // [65536]  core/inc/intalglut.h usin_lut
    const short*    ucos;        // [65536]
    const uint32_t* rot;         // [65536]  {ucos, -usin} packed: the rotation coefficient of an FP_RAD angle in one read
    const short*    uatan2;      // [256*256]
    const uint8_t*  demap;       // [4][256]  bpsk, qam16_2, qam64_2, qam64_3 (demapper.h:55-130)
    const uint32_t* tw64;        // [3][16] packed W64^{k j}, k=1,2,3   (fft_lut_twiddle.h:61433-61504)
    const uint32_t* tw16;        // [3][4]  packed W16^{k j}
    const uint32_t* sts;         // [16][16] packed STS correlation patterns (cca.hpp:268-277)
    const uint16_t* deint;       // [4][288] de-interleaver source index per output position (BPSK,QPSK,QAM16,QAM64)
    const uint32_t* crc;         // [256]
    const uint8_t*  scr;         // [128]
    const uint8_t*  scr_seq;     // [127] scrambler byte starting at cycle position q
    const uint8_t*  scr_phase;   // [128] cycle position of a 7-bit descrambler seed (255 for seed 0)
    const uint32_t* tw128;       // [3][32] packed W128^{k j}
    const uint32_t* tw32;        // [3][8]  packed W32^{k j}
    const uint32_t* tw8;         // [4]     {W8^0, W8^1, W8^0, W8^3} (fft_lut_twiddle.h:61575-61581)
    const uint32_t* crcz;        // [6][8][16] CRC-32 register after 40 * 2^k zero bytes, per nibble of the start value (parallel CRC)
    const uint32_t* trk;         // TrkTables (below): usin / ucos / uatan2 once more, folded to 113 KB so that a workgroup can hold them in LDS (k_track_lds)
};

// The three trigonometric tables of the pilot tracker's chain, small enough for one workgroup's LDS (round 5).  usin / ucos (65536 entries each) are a quarter
// wave plus a two-bit signed correction per entry: the reference's generator used pi = 3.141593, so its tables are not exactly symmetric -- 481 / 474 entries differ
// by one from the mirrored quarter wave -- and uatan2 (256 x 256) is odd in y for every y but -128: rows 0 .. 127 and the negated row -128.  The host builds these
// from the full tables and checks that they reproduce EVERY entry before it uploads them (sora_hip.cpp: trk_tables_exact).
struct TrkTables {
    int16_t  q[16392];           // usin[0 .. 16384] (padded to a multiple of 16 bytes)
    // two bits per angle a, at bit 2 (a & 15) of word a >> 4: usin[a] minus the mirrored quarter wave, as a signed two-bit number (0, +1, -1)
    uint32_t e2s[4096];
    uint32_t e2c[4096];          // the same for ucos[a] against the quarter wave mirrored for a + 16384
    int16_t  h[129 * 256];       // uatan2[y][x & 0xFF] for y = 0 .. 127; row 128 = -uatan2[-128][..]: uatan2(y < 0, x) = -h[-y][x & 0xFF]
};
// (index arithmetic shared by the host's check and the kernel) the quarter-wave index of angle a: q in even quadrants, 16384 - q in odd ones, q = a & 0x3FFF
__host__ __device__ inline int trk_quarter_index(unsigned a) { const int m = -(int)((a >> 14) & 1u); return (int)((((int)a ^ m) & 0x3FFF) - m); }
__host__ __device__ inline int trk_sext2(uint32_t word, unsigned a) { return ((int)(word << (30u - 2u * (a & 15u)))) >> 30; }
template <typename TBL> __host__ __device__ inline int trk_usin(const TBL& t, unsigned a)     // usin[a], a = FP_RAD angle & 0xFFFF
{
    const int ms = -(int)((a >> 15) & 1u);
    return ((t.q[trk_quarter_index(a)] ^ ms) - ms) + trk_sext2(t.e2s[a >> 4], a);
}
template <typename TBL> __host__ __device__ inline int trk_ucos(const TBL& t, unsigned a)     // ucos[a]
{
    const int mc = -(int)(((a >> 15) ^ (a >> 14)) & 1u);
    return ((t.q[16384 - trk_quarter_index(a)] ^ mc) - mc) + trk_sext2(t.e2c[a >> 4], a);
}
// uatan2_lut[(ys & 0xFF) * 256 + (xs & 0xFF)], ys in -128 .. 127
template <typename TBL> __host__ __device__ inline int trk_uatan2_entry(const TBL& t, int ys, int xs)
{
    const int sy = ys >> 31, r = (ys ^ sy) - sy, v = t.h[r * 256 + (xs & 0xFF)];
    return (v ^ sy) - sy;
}

// uatan2 (core/inc/intalg.h:100-113): highest set bit of |y|,|x| -> common shift -> 256x256 LUT
__device__ __forceinline__ int bit_scope(int v) { unsigned a = (unsigned)(v > 0 ? v : -v); return a ? 31 - __clz(a) : 0; }
__device__ __forceinline__ int uatan2(const Tables& T, int y, int x)
{
    int ys = bit_scope(y), xs = bit_scope(x);
    int shift = max(xs, ys) - 6;
    if (shift > 0) { y >>= shift; x >>= shift; }
    return (int)T.uatan2[((unsigned)y & 0xFF) * 256 + ((unsigned)x & 0xFF)];
}
__device__ __forceinline__ cpx rot_coeff(const Tables& T, int th)        // (ucos(th), -usin(th)) with FP_RAD th
{
    return unpack(T.rot[(unsigned)th & 0xFFFFu]);
}

// data carrier k (0..47) -> FFT bin, in demap order -26..-1, +1..+26 without the pilots (demapper11a.hpp:20-37)
__device__ __forceinline__ int carrier_bin48(int k)
{
    if (k < 24) { int b = 38 + k; if (b >= 43) b++; if (b >= 57) b++; return b; }
    int b = 1 + (k - 24); if (b >= 7) b++; if (b >= 21) b++; return b;
}
// wave-level barrier for LDS that only one wave touches
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// 64-point radix-4 DIF FFT (core/inc/fft_r4dif.h) for a group of 16 lanes, 4 points per lane, staged
// through a 64-entry LDS slice `s` private to the group.  `e` = lane index within the group (0..15).
// In: x[m] = point e+16m.  Out: y[q] = bin e+16q (natural order).  All lanes of the group must call.
// SYNC is a barrier covering the group (block barrier, or nothing inside a single wave after a waitcnt).
struct Fft64Tw { uint32_t w64_1, w64_2, w64_3, w16_1, w16_2, w16_3; };   // the six twiddles lane e of a group needs (loop-invariant)
__device__ __forceinline__ Fft64Tw fft64_twiddles(const Tables& T, int e)
{
    const int i = e & 3;
    return Fft64Tw{ T.tw64[e], T.tw64[16 + e], T.tw64[32 + e], T.tw16[i], T.tw16[4 + i], T.tw16[8 + i] };
}
template <typename SYNC>
__device__ __forceinline__ void fft64_core(cpx x[4], uint32_t* s, int e, const Fft64Tw& W, SYNC sync)   // result left in s[]: bin j at slot bitrev6(j)
{
    sync();                                                                           // previous users of s[] are done
    // stage N=64: butterfly e on points e, e+16, e+32, e+48   (FFTSSE<64>, fft_r4dif.h:11-47)
    {
        cpx a = sra(x[0], 2), b = sra(x[1], 2), c = sra(x[2], 2), d = sra(x[3], 2);
        cpx ac = cadds(a, c), bd = cadds(b, d), a_c = csubs(a, c), b_d = csubs(b, d);
        cpx jb = mul_j(b_d);
        s[e]      = pack(cadds(ac, bd));
        s[e + 16] = pack(mul_shift15(csubs(ac, bd),  unpack(W.w64_2)));              // W64^{2e}
        s[e + 32] = pack(mul_shift15(csubs(a_c, jb), unpack(W.w64_1)));              // W64^{e}
        s[e + 48] = pack(mul_shift15(cadds(a_c, jb), unpack(W.w64_3)));              // W64^{3e}
    }
    sync();
    // stage N=16 on quarter k: butterfly i on points 16k+i+{0,4,8,12}   (FFTSSE<16>)
    {
        const int k = e >> 2, i = e & 3, base = 16 * k + i;
        cpx a = sra(unpack(s[base]), 2), b = sra(unpack(s[base + 4]), 2), c = sra(unpack(s[base + 8]), 2), d = sra(unpack(s[base + 12]), 2);
        cpx ac = cadds(a, c), bd = cadds(b, d), a_c = csubs(a, c), b_d = csubs(b, d);
        cpx jb = mul_j(b_d);
        s[base]      = pack(cadds(ac, bd));                                           // in place: same lane, same slots
        s[base + 4]  = pack(mul_shift15(csubs(ac, bd),  unpack(W.w16_2)));
        s[base + 8]  = pack(mul_shift15(csubs(a_c, jb), unpack(W.w16_1)));
        s[base + 12] = pack(mul_shift15(cadds(a_c, jb), unpack(W.w16_3)));
    }
    sync();
    // terminal 4-point stage on points 4e..4e+3   (FFTSSEEx<4>, fft_r4dif.h:60-83)
    {
        cpx c0 = sra(unpack(s[4 * e]), 2), c1 = sra(unpack(s[4 * e + 1]), 2), c2 = sra(unpack(s[4 * e + 2]), 2), c3 = sra(unpack(s[4 * e + 3]), 2);
        cpx A0 = cadds(c0, c2), A1 = cadds(c1, c3);
        cpx B0 = cadds(cnot(c2), c0), B1 = cadds(cnot(c3), c1);
        cpx B1r = mk(B1.im, ~B1.re);                                                   // ~ -j*B1
        s[4 * e]     = pack(cadds(A0, A1));
        s[4 * e + 1] = pack(cadds(cnot(A1), A0));
        s[4 * e + 2] = pack(cadds(B0, B1r));
        s[4 * e + 3] = pack(cadds(cnot(B1r), B0));
    }
    sync();
}
template <typename SYNC>
__device__ __forceinline__ void fft64_group(cpx x[4], cpx y[4], uint32_t* s, int e, const Fft64Tw& W, SYNC sync)
{
    fft64_core(x, s, e, W, sync);
    // bit-reversed reorder (FFT64LUTMap, fft_lut_bitreversal.h:76-142): bin j <- slot bitrev6(j)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const unsigned j = (unsigned)(e + 16 * q);
        y[q] = unpack(s[__brev(j) >> 26]);
    }
}

template <typename SYNC>
__device__ __forceinline__ void fft64_group(cpx x[4], cpx y[4], uint32_t* s, int e, const Tables& T, SYNC sync)
{
    fft64_group(x, y, s, e, fft64_twiddles(T, e), sync);
}

// a*conj(b) >> 15 with the xor-approximated negation of the reference (conj_mul_shift, vector128.h:1248-1256): IFFT twiddles
__device__ __forceinline__ cpx conj_mul_shift15(cpx a, cpx b)
{
    int v0 = (int)((unsigned)(a.re * b.re) + (unsigned)(a.im * b.im));
    int v1 = (int)((unsigned)(a.im * b.re) + (unsigned)(neg16(a.re) * b.im));
    return mk(w16(v0 >> 15), w16(v1 >> 15));
}

// ---------------------------------------------------------------------------------------------
// 128-point radix-4 DIF FFT / IFFT (core/inc/fft_r4dif.h, ifft_r4dif.h: 128 = 4 x 32, 32 = 4 x 8, 8-point terminal
// stage with input shift 3) for a group of 32 lanes, 4 points per lane, staged through a 128-entry LDS slice `s`.
// In: x[m] = point e + 32 m.  Out: y[q] = point e + 32 q (natural order).  All lanes of the group must call; SYNC is a
// barrier covering the group.
template <bool INV>
__device__ __forceinline__ void r4_bfly128(cpx a, cpx b, cpx c, cpx d, cpx w1, cpx w2, cpx w3, cpx& y0, cpx& y1, cpx& y2, cpx& y3)
{
    a = sra(a, 2); b = sra(b, 2); c = sra(c, 2); d = sra(d, 2);                 // FFTSSE<N> / IFFTSSE<N> (:11-47)
    cpx ac = cadds(a, c), bd = cadds(b, d), a_c = csubs(a, c), b_d = csubs(b, d);
    cpx jb = mul_j(b_d);
    y0 = cadds(ac, bd);
    if (!INV) { y1 = mul_shift15(csubs(ac, bd), w2); y2 = mul_shift15(csubs(a_c, jb), w1); y3 = mul_shift15(cadds(a_c, jb), w3); }
    else      { y1 = conj_mul_shift15(csubs(ac, bd), w2); y2 = conj_mul_shift15(cadds(a_c, jb), w1); y3 = conj_mul_shift15(csubs(a_c, jb), w3); }
}

struct Fft128Tw { uint32_t w128[3], w32[3], w8[4]; };               // the twiddles lane e of a 32-lane group needs (loop-invariant)
__device__ __forceinline__ Fft128Tw fft128_twiddles(const Tables& T, int e)
{
    const int j = e & 7;
    return Fft128Tw{ { T.tw128[e], T.tw128[32 + e], T.tw128[64 + e] }, { T.tw32[j], T.tw32[8 + j], T.tw32[16 + j] }, { T.tw8[0], T.tw8[1], T.tw8[2], T.tw8[3] } };
}
template <bool INV, typename SYNC>
// result left in s[]: point j at slot bitrev7(j)
__device__ __forceinline__ void fft128_core(const cpx x[4], uint32_t* s, int e, const Fft128Tw& T, SYNC sync)
{
    sync();
    {   // stage N=128: butterfly e on points e, e+32, e+64, e+96
        cpx y0, y1, y2, y3;
        r4_bfly128<INV>(x[0], x[1], x[2], x[3], unpack(T.w128[0]), unpack(T.w128[1]), unpack(T.w128[2]), y0, y1, y2, y3);
        s[e] = pack(y0); s[e + 32] = pack(y1); s[e + 64] = pack(y2); s[e + 96] = pack(y3);
    }
    sync();
    {   // stage N=32 on quarter k = e>>3: butterfly j = e&7 on points 32k + j + {0,8,16,24}
        const int k = e >> 3, j = e & 7, base = 32 * k + j;
        cpx y0, y1, y2, y3;
        r4_bfly128<INV>(unpack(s[base]), unpack(s[base + 8]), unpack(s[base + 16]), unpack(s[base + 24]),
                        unpack(T.w32[0]), unpack(T.w32[1]), unpack(T.w32[2]), y0, y1, y2, y3);
        s[base] = pack(y0); s[base + 8] = pack(y1); s[base + 16] = pack(y2); s[base + 24] = pack(y3);
    }
    sync();
    if (e < 16) {   // terminal 8-point stage (FFTSSEEx<8> / IFFTSSEEx<8>, :86-130) on points 8m..8m+7, m = lane
        uint32_t* p = s + 8 * e;
        cpx a[4], b[4], d[4], sm[4], ee[4], gg[4], ff[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { a[q] = sra(unpack(p[q]), 3); b[q] = sra(unpack(p[4 + q]), 3); }
#pragma unroll
        for (int q = 0; q < 4; q++) { d[q] = csubs(a[q], b[q]); sm[q] = cadds(a[q], b[q]); }
        ee[0] = d[0]; ee[1] = d[1];
        ee[2] = INV ? mk(~d[2].im, d[2].re) : mk(d[2].im, ~d[2].re);
        ee[3] = INV ? mk(~d[3].im, d[3].re) : mk(d[3].im, ~d[3].re);
        gg[0] = cadds(ee[0], ee[2]); gg[1] = cadds(ee[1], ee[3]); gg[2] = cadds(cnot(ee[2]), ee[0]); gg[3] = cadds(cnot(ee[3]), ee[1]);
#pragma unroll
        for (int q = 0; q < 4; q++) ff[q] = INV ? conj_mul_shift15(gg[q], unpack(T.w8[q])) : mul_shift15(gg[q], unpack(T.w8[q]));
        p[4] = pack(cadds(ff[0], ff[1])); p[5] = pack(cadds(cnot(ff[1]), ff[0]));
        p[6] = pack(cadds(ff[2], ff[3])); p[7] = pack(cadds(cnot(ff[3]), ff[2]));
        cpx A0 = cadds(sm[0], sm[2]), A1 = cadds(sm[1], sm[3]);
        cpx B0 = cadds(cnot(sm[2]), sm[0]), B1 = cadds(cnot(sm[3]), sm[1]);
        cpx B1r = INV ? mk(~B1.im, B1.re) : mk(B1.im, ~B1.re);
        p[0] = pack(cadds(A0, A1)); p[1] = pack(cadds(cnot(A1), A0)); p[2] = pack(cadds(B0, B1r)); p[3] = pack(cadds(cnot(B1r), B0));
    }
    sync();
}
template <bool INV, typename SYNC>
__device__ __forceinline__ void fft128_group(const cpx x[4], cpx y[4], uint32_t* s, int e, const Tables& T, SYNC sync)
{
    fft128_core<INV>(x, s, e, fft128_twiddles(T, e), sync);
#pragma unroll
    for (int q = 0; q < 4; q++) y[q] = unpack(s[__brev((unsigned)(e + 32 * q)) >> 25]);   // FFT128LUTMap = 7-bit bit reversal
}

// ---------------------------------------------------------------------------------------------
// The same arithmetic on PACKED COMPLEX16 (re in the low, im in the high half of one VGPR), for the streaming kernels.
// On gfx950 the simple VOP2 integer ops issue at full rate and everything VOP3-encoded (v_med3, v_bfe, v_mad ...) at half
// rate (profiles/r01_issue_probe_table.txt); one v_pk_add_i16 ... clamp replaces the two adds and two v_med3 of a
// saturating complex add, v_pk_ashrrev_i16 the two shifts, and a twiddle product is two v_dot2c_i32_i16 (a 16x16+16x16
// multiply-add that wraps like pmaddwd) and a bit-field insert.  Same results, bit for bit (tests/test_gpu_stages.py).
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t pcx;                                          // packed COMPLEX16
__device__ __forceinline__ pcx pk_sra(pcx a, int n) { return __builtin_bit_cast(pcx, (s16x2_t)(__builtin_bit_cast(s16x2_t, a) >> (short)n)); }
__device__ __forceinline__ pcx pk_adds(pcx a, pcx b) { return __builtin_bit_cast(pcx, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2_t, a),
        __builtin_bit_cast(s16x2_t, b))); }
__device__ __forceinline__ pcx pk_subs(pcx a, pcx b) { return __builtin_bit_cast(pcx, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_t, a),
        __builtin_bit_cast(s16x2_t, b))); }
__device__ __forceinline__ pcx pk_swap(pcx a) { return (a >> 16) | (a << 16); }
__device__ __forceinline__ pcx pk_mul_j(pcx a) { return pk_swap(a) ^ 0x0000FFFFu; }           // (~im, re)   mul_j
__device__ __forceinline__ pcx pk_neg_j(pcx a) { return pk_swap(a) ^ 0xFFFF0000u; }           // (im, ~re)   the 4-point terminal stage's -j
__device__ __forceinline__ pcx pk_neg16_hi(pcx b) { return (b & 0xFFFFu) | ((0u - (b >> 16)) << 16); }   // (re, neg16(im))
struct PkTw { pcx a, b; };                                     // second operands of the two dot products of a complex product
__device__ __forceinline__ PkTw pk_tw_fft(pcx w) { return PkTw{ w ^ 0xFFFF0000u, pk_swap(w) }; }         // mul_shift15: (re, ~im), (im, re)
__device__ __forceinline__ PkTw pk_tw_mul(pcx w) { return PkTw{ pk_neg16_hi(w), pk_swap(w) }; }           // mul / mul_q15: (re, neg16(im)), (im, re)
template <int SHIFT>
__device__ __forceinline__ pcx pk_cmul(pcx x, PkTw t)         // ((x.re t.a.lo + x.im t.a.hi) >> SHIFT, (x.re t.b.lo + x.im t.b.hi) >> SHIFT), wrapping packs
{
    const int v0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, x), __builtin_bit_cast(s16x2_t, t.a), 0, false);
    const int v1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, x), __builtin_bit_cast(s16x2_t, t.b), 0, false);
    return (((uint32_t)v0 >> SHIFT) & 0xFFFFu) | (((uint32_t)v1 << (16 - SHIFT)) & 0xFFFF0000u);
}

struct Fft64TwPk { PkTw w64[3], w16[3]; };
__device__ __forceinline__ Fft64TwPk fft64_twiddles_pk(const Tables& T, int e)
{
    const Fft64Tw W = fft64_twiddles(T, e);
    return Fft64TwPk{ { pk_tw_fft(W.w64_1), pk_tw_fft(W.w64_2), pk_tw_fft(W.w64_3) }, { pk_tw_fft(W.w16_1), pk_tw_fft(W.w16_2), pk_tw_fft(W.w16_3) } };
}
// radix-4 DIF butterfly of FFTSSE<N> (fft_r4dif.h:11-47) on packed values: y0 = sum, y1 (x W^2), y2 (x W^1), y3 (x W^3)
__device__ __forceinline__ void pk_r4(pcx x0, pcx x1, pcx x2, pcx x3, const PkTw (&w)[3], pcx& y0, pcx& y1, pcx& y2, pcx& y3)
{
    const pcx a = pk_sra(x0, 2), b = pk_sra(x1, 2), c = pk_sra(x2, 2), d = pk_sra(x3, 2);
    const pcx ac = pk_adds(a, c), bd = pk_adds(b, d), a_c = pk_subs(a, c), b_d = pk_subs(b, d);
    const pcx jb = pk_mul_j(b_d);
    y0 = pk_adds(ac, bd);
    y1 = pk_cmul<15>(pk_subs(ac, bd), w[1]);
    y2 = pk_cmul<15>(pk_subs(a_c, jb), w[0]);
    y3 = pk_cmul<15>(pk_adds(a_c, jb), w[2]);
}
// 4-point terminal stage (FFTSSEEx<4>, fft_r4dif.h:60-83), input shift 2
__device__ __forceinline__ void pk_t4(pcx c0, pcx c1, pcx c2, pcx c3, pcx& y0, pcx& y1, pcx& y2, pcx& y3)
{
    c0 = pk_sra(c0, 2); c1 = pk_sra(c1, 2); c2 = pk_sra(c2, 2); c3 = pk_sra(c3, 2);
    const pcx A0 = pk_adds(c0, c2), A1 = pk_adds(c1, c3), B0 = pk_adds(~c2, c0), B1 = pk_adds(~c3, c1);
    const pcx B1r = pk_neg_j(B1);
    y0 = pk_adds(A0, A1); y1 = pk_adds(~A1, A0); y2 = pk_adds(B0, B1r); y3 = pk_adds(~B1r, B0);
}
// fft64_core on packed values: x[m] = point e + 16 m in, result in s[] (bin j at slot bitrev6(j))
template <typename SYNC>
__device__ __forceinline__ void fft64_core_pk(const pcx x[4], uint32_t* s, int e, const Fft64TwPk& W, SYNC sync)
{
    sync();
    pk_r4(x[0], x[1], x[2], x[3], W.w64, s[e], s[e + 16], s[e + 32], s[e + 48]);
    sync();
    {
        const int base = 16 * (e >> 2) + (e & 3);
        pcx y0, y1, y2, y3;
        pk_r4(s[base], s[base + 4], s[base + 8], s[base + 12], W.w16, y0, y1, y2, y3);
        s[base] = y0; s[base + 4] = y1; s[base + 8] = y2; s[base + 12] = y3;
    }
    sync();
    {
        const uint4 c = reinterpret_cast<const uint4*>(s)[e];
        uint4 y;
        pk_t4(c.x, c.y, c.z, c.w, y.x, y.y, y.z, y.w);
        reinterpret_cast<uint4*>(s)[e] = y;
    }
    sync();
}

struct Fft128TwPk { PkTw w128[3], w32[3], w8[4]; };
__device__ __forceinline__ Fft128TwPk fft128_twiddles_pk(const Tables& T, int e)
{
    const Fft128Tw W = fft128_twiddles(T, e);
    return Fft128TwPk{ { pk_tw_fft(W.w128[0]), pk_tw_fft(W.w128[1]), pk_tw_fft(W.w128[2]) }, { pk_tw_fft(W.w32[0]), pk_tw_fft(W.w32[1]), pk_tw_fft(W.w32[2]) },
                       { pk_tw_fft(W.w8[0]), pk_tw_fft(W.w8[1]), pk_tw_fft(W.w8[2]), pk_tw_fft(W.w8[3]) } };
}
// forward FFT<128> on packed values (fft128_core<false>): the 8-point terminal stage (FFTSSEEx<8>, fft_r4dif.h:86-130) of block
// e >> 1 is shared by a lane pair -- lane 2m takes the sum half (outputs 0..3), lane 2m+1 the difference half (outputs 4..7),
// expressed without a branch: v = sat(a +- b), the -j rotation and the W8 products selected per lane.
template <typename SYNC>
__device__ __forceinline__ void fft128_core_pk(const pcx x[4], uint32_t* s, int e, const Fft128TwPk& W, SYNC sync)
{
    sync();
    pk_r4(x[0], x[1], x[2], x[3], W.w128, s[e], s[e + 32], s[e + 64], s[e + 96]);
    sync();
    {
        const int base = 32 * (e >> 3) + (e & 7);
        pcx y0, y1, y2, y3;
        pk_r4(s[base], s[base + 8], s[base + 16], s[base + 24], W.w32, y0, y1, y2, y3);
        s[base] = y0; s[base + 8] = y1; s[base + 16] = y2; s[base + 24] = y3;
    }
    sync();
    {
        const bool hi = e & 1;                                                    // difference half
        uint32_t* p = s + 8 * (e >> 1);
        const uint4 A = reinterpret_cast<const uint4*>(p)[0], B = reinterpret_cast<const uint4*>(p)[1];
        pcx v[4];
        const pcx a[4] = { pk_sra(A.x, 3), pk_sra(A.y, 3), pk_sra(A.z, 3), pk_sra(A.w, 3) }, b[4] = { pk_sra(B.x, 3), pk_sra(B.y, 3), pk_sra(B.z, 3), pk_sra(B.w, 3) };
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = hi ? pk_subs(a[q], b[q]) : pk_adds(a[q], b[q]);
        if (hi) { v[2] = pk_neg_j(v[2]); v[3] = pk_neg_j(v[3]); }                 // ee[2], ee[3] = -j d
        pcx g0 = pk_adds(v[0], v[2]), g1 = pk_adds(v[1], v[3]), g2 = pk_adds(~v[2], v[0]), g3 = pk_adds(~v[3], v[1]);
        if (hi) { g0 = pk_cmul<15>(g0, W.w8[0]); g1 = pk_cmul<15>(g1, W.w8[1]); g2 = pk_cmul<15>(g2, W.w8[2]); g3 = pk_cmul<15>(g3, W.w8[3]); }
        else g3 = pk_neg_j(g3);                                                   // B1r
        uint4 y;
        y.x = pk_adds(g0, g1); y.y = pk_adds(~g1, g0); y.z = pk_adds(g2, g3); y.w = pk_adds(~g3, g2);
        sync();                                                                   // both lanes of a pair have read the block before either half is rewritten
        reinterpret_cast<uint4*>(p)[hi ? 1 : 0] = y;
    }
    sync();
}

// ---- the inverse transform on packed values (IFFT<128>, core/inc/ifft_r4dif.h): conj_mul_shift negates the DATA's real part (neg16, wrapping: -32768
// stays -32768), which on packed values is one v_pk_mul_lo_u16 by (0xFFFF, 1); the twiddles go in as they are
__device__ __forceinline__ pcx pk_neg16_lo(pcx a)
{
    typedef unsigned short u16x2p_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(pcx, (u16x2p_t)(__builtin_bit_cast(u16x2p_t, a) * u16x2p_t{ (unsigned short)0xFFFFu, (unsigned short)1u }));
}
// conj_mul_shift15(x, w): ((x.re w.re + x.im w.im) >> 15, (x.im w.re + neg16(x.re) w.im) >> 15)
__device__ __forceinline__ pcx pk_conj_cmul15(pcx x, pcx w)
{
    const int v0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, x), __builtin_bit_cast(s16x2_t, w), 0, false);
    const int v1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, pk_neg16_lo(x)), __builtin_bit_cast(s16x2_t, pk_swap(w)), 0, false);
    return (((uint32_t)v0 >> 15) & 0xFFFFu) | (((uint32_t)v1 << 1) & 0xFFFF0000u);
}
__device__ __forceinline__ void pk_r4_inv(pcx x0, pcx x1, pcx x2, pcx x3, const uint32_t (&w)[3], pcx& y0, pcx& y1, pcx& y2, pcx& y3)
{
    const pcx a = pk_sra(x0, 2), b = pk_sra(x1, 2), c = pk_sra(x2, 2), d = pk_sra(x3, 2);
    const pcx ac = pk_adds(a, c), bd = pk_adds(b, d), a_c = pk_subs(a, c), b_d = pk_subs(b, d);
    const pcx jb = pk_mul_j(b_d);
    y0 = pk_adds(ac, bd);
    y1 = pk_conj_cmul15(pk_subs(ac, bd), w[1]);
    y2 = pk_conj_cmul15(pk_adds(a_c, jb), w[0]);
    y3 = pk_conj_cmul15(pk_subs(a_c, jb), w[2]);
}
// fft128_core<true> on packed values; the 8-point terminal stage shared by a lane pair as in fft128_core_pk
template <typename SYNC>
__device__ __forceinline__ void ifft128_core_pk(const pcx x[4], uint32_t* s, int e, const Fft128Tw& W, SYNC sync)
{
    sync();
    pk_r4_inv(x[0], x[1], x[2], x[3], W.w128, s[e], s[e + 32], s[e + 64], s[e + 96]);
    sync();
    {
        const int base = 32 * (e >> 3) + (e & 7);
        pcx y0, y1, y2, y3;
        pk_r4_inv(s[base], s[base + 8], s[base + 16], s[base + 24], W.w32, y0, y1, y2, y3);
        s[base] = y0; s[base + 8] = y1; s[base + 16] = y2; s[base + 24] = y3;
    }
    sync();
    {
        const bool hi = e & 1;                                                    // difference half
        uint32_t* p = s + 8 * (e >> 1);
        const uint4 A = reinterpret_cast<const uint4*>(p)[0], B = reinterpret_cast<const uint4*>(p)[1];
        pcx v[4];
        const pcx a[4] = { pk_sra(A.x, 3), pk_sra(A.y, 3), pk_sra(A.z, 3), pk_sra(A.w, 3) }, b[4] = { pk_sra(B.x, 3), pk_sra(B.y, 3), pk_sra(B.z, 3), pk_sra(B.w, 3) };
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = hi ? pk_subs(a[q], b[q]) : pk_adds(a[q], b[q]);
        if (hi) { v[2] = pk_mul_j(v[2]); v[3] = pk_mul_j(v[3]); }                 // ee[2], ee[3] = (~im, re): +j d
        pcx g0 = pk_adds(v[0], v[2]), g1 = pk_adds(v[1], v[3]), g2 = pk_adds(~v[2], v[0]), g3 = pk_adds(~v[3], v[1]);
        if (hi) { g0 = pk_conj_cmul15(g0, W.w8[0]); g1 = pk_conj_cmul15(g1, W.w8[1]); g2 = pk_conj_cmul15(g2, W.w8[2]); g3 = pk_conj_cmul15(g3, W.w8[3]); }
        else g3 = pk_mul_j(g3);                                                   // B1r = (~B1.im, B1.re)
        uint4 y;
        y.x = pk_adds(g0, g1); y.y = pk_adds(~g1, g0); y.z = pk_adds(g2, g3); y.w = pk_adds(~g3, g2);
        sync();
        reinterpret_cast<uint4*>(p)[hi ? 1 : 0] = y;
    }
    sync();
}

// ---------------------------------------------------------------------------------------------
// CRC-32 (reflected, init 0xFFFFFFFF, no final xor here) of n >= 4 bytes in LDS by one wave.  The register update is
// linear over GF(2): CRC(init, M) = CRC(0, M') with the first four bytes complemented, and
// CRC(0, M1 | M2) = Z_|M2|(CRC(0, M1)) ^ CRC(0, M2) with Z_m = "m zero bytes".  Lane l takes the 40 bytes that END 40 l
// bytes before the end (byte table s_crc), then six tree levels fold lane l + 2^k into lane l through Z_(40 * 2^k)
// (8 nibble look-ups in s_z each).  Lane 0 returns the register after the whole message.
__device__ __forceinline__ uint32_t crc32_wave(const uint8_t* bytes, int n, const uint32_t* s_crc, const uint32_t* s_z, int lane)
{
    uint32_t c = 0;
    const int i0 = n - 40 * (lane + 1);
#pragma unroll 8
    for (int q = 0; q < 40; q++) {
        const int i = i0 + q;
        if (i >= 0) c = (c >> 8) ^ s_crc[(c ^ bytes[i] ^ (i < 4 ? 0xFFu : 0u)) & 0xFFu];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint32_t o = (uint32_t)__shfl_down((int)c, 1 << k);
        uint32_t z = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) z ^= s_z[(k * 8 + q) * 16 + ((o >> (4 * q)) & 15u)];
        c ^= z;
    }
    return c;
}

}  // namespace sora
