// dev_11n.h -- device-side pieces of the 802.11n bricks shared by the stage kernels (k_11n.hip) and the whole-path receive kernel
// (k_rx11n.hip).  Reference locations are cited at each function.
#pragma once
#include "dev_arith.h"

namespace sora {
namespace {
struct Run { uint8_t v, n; };
static __constant__ Run kRuns[83] = {
    {0,97},{1,10},{2,10},{3,11},{4,11},{5,10},{6,10},{7,97},                                                              // [0,8)   BPSK / QPSK
    {0,113},{1,7},{2,4},{3,4},{4,5},{5,4},{6,7},{7,112},                                                                   // [8,16)  16-QAM bit 0
    {0,58},{1,3},{2,2},{3,2},{4,2},{5,3},{6,3},{7,111},{6,3},{5,3},{4,2},{3,2},{2,2},{1,3},{0,57},                         // [16,31) 16-QAM bit 1
    {0,122},{1,3},{2,2},{3,1},{4,2},{5,2},{6,3},{7,121},                                                                   // [31,39) 64-QAM bit 0
    {0,52},{1,3},{2,2},{3,2},{4,1},{5,2},{6,3},{7,127},{6,3},{5,2},{4,1},{3,2},{2,2},{1,3},{0,51},                         // [39,54) 64-QAM bit 1
    {0,18},{1,2},{2,2},{3,2},{4,2},{5,1},{6,3},{7,57},{6,3},{5,2},{4,2},{3,1},{2,2},{1,3},{0,57},
    {1,3},{2,2},{3,1},{4,2},{5,2},{6,3},{7,57},{6,3},{5,1},{4,2},{3,2},{2,2},{1,2},{0,17} };                               // [54,83) 64-QAM bit 2
static __constant__ int kRunFirst[7] = { 0, 8, 16, 31, 39, 54, 83 };

__device__ __forceinline__ int data_bin(int l)               // carrier walk of the 11n demappers: -28..-1 then 1..28, pilots at +-7, +-21 skipped
{
    if (l < 26) return l < 7 ? 36 + l : (l < 20 ? 37 + l : 38 + l);
    const int m = l - 26;
    return m < 6 ? 1 + m : (m < 19 ? 2 + m : 3 + m);
}

__device__ __forceinline__ void fill_demap_luts(uint8_t (*lut)[256])      // the six step tables of dsp_demap.h, index v + 128; blockDim.x == 256
{
    const int t = threadIdx.x;
    for (int w = 0; w < 6; w++) {
        int acc = 0; uint8_t val = 0;
        for (int r = kRunFirst[w]; r < kRunFirst[w + 1]; r++) { if (t >= acc && t < acc + kRuns[r].n) val = kRuns[r].v; acc += kRuns[r].n; }
        lut[w][t] = val;
    }
}

static __constant__ int8_t kHtLtf[57] = {   // HT-LTF, carriers -28..28 (IEEE 802.11n, 20 MHz)
    1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
    1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1 };
struct cf { float re, im; };
// vcf mul (vector128.h:1107-1116): every product and every sum rounded on its own -- no fused multiply-add
// (the default -ffp-contract=fast-honor-pragmas would fuse a*b - c*d into an fma: one rounding less than the reference's
//  mulps / addsubps.  The pragma keeps every operation on its own; plain operators are IEEE single precision on gfx950.)
__device__ __forceinline__ cf cf_mul(cf a, cf b)
{
#pragma clang fp contract(off)
    cf r; r.re = (a.re * b.re) - (a.im * b.im); r.im = (a.im * b.re) + (a.re * b.im); return r;
}
__device__ __forceinline__ int cvtps_sat16(float x)    // cvtps2dq (nearest even; 0x80000000 when out of range or NaN), then packssdw
{
    const int v = (x >= -2147483648.0f && x < 2147483648.0f) ? (int)rintf(x) : (int)0x80000000;
    return sat16(v);
}

static __constant__ unsigned long long kLLtfPlus = 0xF59FACC007A982B2ull;     // bit i: the L-LTF is +1 on FFT bin i (_80211_LLTFMask 0xFFFF0000 lanes)
__device__ __forceinline__ int sqn_wrap(cpx v) { return (int)((unsigned)(v.re * v.re) + (unsigned)(v.im * v.im)); }
__device__ __forceinline__ cpx siso_one(const uint32_t* x4, int j, int bin)
{
    const cpx x = unpack(x4[j]), xr = unpack(x4[(2 * j) & 3]), xi = unpack(x4[(2 * j + 1) & 3]);
    int sq = sqn_wrap(x);                                                  // pmaddwd: (-32768, -32768) wraps to INT_MIN
    if (sq == 0) sq = 1;
    const int hr = sqn_wrap(xr) >> 1, hi = sqn_wrap(xi) >> 1;
    const int re = (int)(((unsigned)x.re << 16) + (unsigned)hr) / sq, im = (int)(((unsigned)x.im << 16) + (unsigned)hi) / sq;
    cpx c = mk(sat16(re), sat16(im));
    if ((kLLtfPlus >> bin) & 1) c.im = (short)-c.im; else c.re = (short)-c.re;
    return c;
}

template <int NB>
__device__ __forceinline__ uint64_t viterbi_sig_wave(const uint8_t* soft, uint64_t* dec, int lane)     // soft[2 * NB] de-interleaved, dec[NB + 1] in LDS
{
    const int n = lane, r0 = n, r1 = 64 | n;
    const int cA0 = __popc(r0 & 0155) & 1, cB0 = __popc(r0 & 0117) & 1, cA1 = __popc(r1 & 0155) & 1, cB1 = __popc(r1 & 0117) & 1;
    unsigned m = (n == 0) ? 0u : 0x30u;
    if (lane == 0) dec[0] = 0;
#pragma unroll 8
    for (int t = 1; t <= NB; t++) {
        const int va = soft[2 * (t - 1)], vb = soft[2 * (t - 1) + 1];
        const unsigned m0 = (unsigned)__shfl((int)m, n >> 1), m1 = (unsigned)__shfl((int)m, 32 + (n >> 1));
        const unsigned b0 = (cA0 ? 2 * (7 - va) : 2 * va) + (cB0 ? 2 * (7 - vb) : 2 * vb);
        const unsigned b1 = (cA1 ? 2 * (7 - va) : 2 * va) + (cB1 ? 2 * (7 - vb) : 2 * vb);
        const unsigned c0 = (m0 + b0) & 0xFE, c1 = ((m1 + b1) & 0xFF) | 1;
        m = min(c0, c1);
        { const uint64_t d = __ballot(m & 1); if (lane == 0) dec[t] = d; }
        if ((t & 7) == 0) {
            unsigned mn = m;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
            m = (m - (mn & 0xFE)) & 0xFF;
        }
    }
    unsigned kmin = (m << 8) | ((unsigned)n << 2);                           // smallest metric, then smallest state (INDEXES, hmin)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o));
    int pos = (int)((kmin >> 2) & 0x3F) | (int)(((kmin >> 8) & 1) << 6);
    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
    uint64_t out = 0;
    for (int b = 0; b < NB; b++) {                                           // bit b of the walk is output bit NB - 1 - b
        out |= (uint64_t)((pos >> 6) & 1) << (NB - 1 - b);
        pos = (pos >> 1) & 0x3F;
        pos |= (int)((dec[NB - 1 - b] >> pos) & 1) << 6;
    }
    return out;
}

__device__ __forceinline__ int atan_tail(const short* tab, int idx, int tsign, int sign)
{
    if (idx < 0 || idx >= 4097) return 0;
    int srad = tab[idx];
    srad = (int)(short)((16384 & tsign) + ((srad ^ tsign) - tsign));
    srad ^= sign; return (int)(short)(srad - sign);
}
__device__ __forceinline__ int dsp_atan16(const short* tab, int x, int y)      // dsp_math::atan(short, short); x, y already int16 values
{
    const int sign = (x ^ y) >> 15;                                            // -1 / 0
    const int absx = (int)(short)((x ^ (x >> 15)) - (x >> 15)), absy = (int)(short)((y ^ (y >> 15)) - (y >> 15));
    const int tsign = (int)(short)((absx - absy) >> 15);
    const int tsum = absx + absy, d = absx - absy;
    const int tmax = (tsum + ((d ^ (d >> 31)) - (d >> 31))) >> 1, tmin = tsum - tmax;
    if (tmax == 0) return 0;
    int idx;
    if (tmin >= 0 && tmin <= tmax && tmax < 32768) {
        // The usual case (everything but an input of -32768): numerator < 2^31 + 2^14, divisor < 2^15, quotient <= 65536.  One reciprocal and a
        // correction in each direction instead of the 30-instruction integer division: the float estimate is off by less than 0.02.
        const unsigned num = ((unsigned)tmin << 16) + ((unsigned)tmax >> 1);
        unsigned q = (unsigned)((float)num * __builtin_amdgcn_rcpf((float)tmax));
        int r = (int)(num - __umul24(q, (unsigned)tmax));
        if (r < 0) { q--; r += tmax; }
        if (r >= tmax) q++;
        idx = (int)q;
    } else idx = (int)(((unsigned)tmin << 16) + (unsigned)(tmax >> 1)) / tmax;
    return atan_tail(tab, idx >> 4, tsign, sign);
}
__device__ __forceinline__ int dsp_atan32(const short* tab, int x, int y)      // dsp_math::atan(int, int), for |x| + |y| < 2^31
{
    const int sign = (int)(short)(((x ^ y) >> 31) & 0xFFFF);
    const int absx = (x ^ (x >> 31)) - (x >> 31), absy = (y ^ (y >> 31)) - (y >> 31);
    const int tsign = (int)(short)(((absx - absy) >> 31) & 0xFFFF);
    const int tsum = absx + absy, d = absx - absy;
    const int tmax = (tsum + ((d ^ (d >> 31)) - (d >> 31))) >> 1, tmin = tsum - tmax;
    const long long i64y = tmax == 0 ? 1 : tmax;
    const int idx = (int)((((long long)tmin << 16) + (i64y >> 1)) / i64y);
    return atan_tail(tab, idx >> 4, tsign, sign);
}

// T11nDeinterleave*_S{0,1} (deinterleaver_11n.hpp): source position of de-interleaved position k -- the HT interleaver (N_COL 13,
// N_ROW 4 N_BPSC, N_ROT 11) inverted
__device__ __forceinline__ int deint11n_index(int nb, int iss, int k)
{
    const int s = nb / 2 > 1 ? nb / 2 : 1, nrow = 4 * nb, np = 52 * nb;
    const int i = nrow * (k % 13) + k / 13;
    int j = s * (i / s) + (i + np - (13 * i) / np) % s;
    if (iss > 0) j = ((j - ((iss * 2) % 3 + 3 * (iss / 3)) * 11 * nb) % np + np) % np;
    return j;
}
}  // namespace
}  // namespace sora
