// k_scan.hip -- per-capture front end: carrier sense + frame synchronisation + LTS + SIGNAL decode.
//
// One wave64 per capture reproduces what the reference's source thread reports (kernel/bb/demod11/fb11a_demod.cpp:29-81 driving
// CreateDemodGraph11a_40M, fb11ademod_config.hpp:168-233): 28 raw samples per TMemSamples::Process() call, 4-sample bursts through
//   TDownSample2 -> TBB11bRxSwitch -> [TDCRemoveEx -> TCCA11a -> TDCEstimator] | [T11aLTS | T11aDataSymbol ...]
// with the error_code test after every source call.  Round 6: the walk is no longer a loop over source calls.  Positions are kept in
// 20 MHz-rate samples s (a burst = 4, a source call = 14 at either input rate; a 40 MHz capture only ever uses its even samples), and the
// graph's behaviour between two observable events is evaluated in PASSES:
//   * carrier sense (cca.hpp:386-437) up to eight bursts at a time, one sample per lane: sliding sums by prefix sums, the carrier test of
//     every burst, and the detection counters (auto_count / sense_count / the time-out) from the ballot of those tests with bit arithmetic;
//     a pass ends with the burst that calls establish_sync (cca.hpp:220-243: sixteen patterns x sixteen taps, four taps per lane);
//   * the short training symbols (check_sync, cca.hpp:245-265) sixteen bursts at a time: a row of sixteen lanes per check;
//   * T11aLTS, the SIGNAL symbol and the frame's end are positions computed from the detection point; the source-call grid only matters
//     where the reference looks at it: a carrier-sense time-out resets the bricks at the END of the call that raised it, and a frame event
//     drops the rest of its call's queue (Flush + Reset, fb11a_demod.cpp:64-70).
// The samples come from an LDS ring that is filled 256 samples ahead of the walk (four coalesced loads in flight while the passes run), so
// a pass waits for LDS, not for HBM.  Output: the frame table, per-frame contexts and the symbol-slot map that the batched per-symbol
// kernels consume.  Data symbols are NOT decoded here.
//
// Capture contract: nsamples is a whole number of source bursts (28 raw samples @40 MHz / 14 @20 MHz) --
// true of every Sora dump (RX_BLOCK = 28 samples, core/inc/_rx_manager.h:96-137); a trailing partial burst
// is ignored (the reference would run it padded with stale pin-queue memory, memsource.hpp:99-107).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace sora {

// The capture's samples around the walk, one packed COMPLEX16 per word: sample i (20 MHz-rate index) at s_ring[i & 511] (k_scan keeps [lo, hi) valid).
constexpr uint32_t kRing = 512;
__shared__ uint32_t s_ring[kRing];

#ifdef SORA_SCAN_PROBE
__device__ unsigned long long g_scan_probe[16];              // [0..7] ticks per region, [8..15] how often (capture 0 only; tools/probe_scan.py)
#define PROBE_DECL() long long _pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned _pn[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROBE_T0() const long long _t0 = clock64()
#define PROBE_ADD(i) do { _pa[i] += clock64() - _t0; _pn[i]++; } while (0)
#define PROBE_T(n) const long long n = clock64()
#define PROBE_A(n, i) do { _pa[i] += clock64() - n; _pn[i]++; } while (0)
#define PROBE_TK() const long long _tk = clock64()
#define PROBE_ADDK(i) do { _pa[i] += clock64() - _tk; _pn[i]++; \
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int q = 0; q < 8; q++) { g_scan_probe[q] += (unsigned long long)_pa[q]; g_scan_probe[8 + q] += _pn[q]; } } while (0)
#else
#define PROBE_DECL()
#define PROBE_T0()
#define PROBE_ADD(i)
#define PROBE_T(n)
#define PROBE_A(n, i)
#define PROBE_TK()
#define PROBE_ADDK(i)
#endif

// CMovingWindow<int,4> + CAccumulator (dspalg.hpp:5-98).  The sum is a wave-uniform scalar; the window lives in a vector register, element g
// (oldest first) in lanes 4 g .. 4 g + 3 of every 16-lane row, so that the idle fast path can run eight bursts' sliding sums as one prefix sum.
struct Acc4 { int reg; uint32_t Z; };
__device__ __forceinline__ void acc_clear(Acc4& a) { a.reg = 0; a.Z = 0; }
__device__ __forceinline__ void acc_push(Acc4& a, int d)                    // d wave-uniform
{
    const int e0 = __builtin_amdgcn_readfirstlane((int)a.Z);
    a.reg = (int)((unsigned)a.reg + (unsigned)d - (unsigned)e0);
    const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.Z, 0x104, 0xF, 0xF, true);     // row_shl:4: element g <- element g + 1
    a.Z = (threadIdx.x & 12u) == 12u ? (uint32_t)d : sh;
}
// inclusive prefix sum over the eight 4-lane groups of lanes 0..31 (lanes 32..63 mirror them): row_shr:4, row_shr:8, row_bcast:15 into rows 1 / 3
__device__ __forceinline__ uint32_t group_scan(uint32_t t)
{
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xF, 0xF, true);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xF, 0xF, true);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xA, 0xF, false);
    return t;
}

// inclusive prefix sum over the sixteen 4-lane groups of the wave: row_shr:4, row_shr:8, row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 and 3
__device__ __forceinline__ uint32_t group_scan64(uint32_t t)
{
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xF, 0xF, true);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xF, 0xF, true);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xA, 0xF, false);
    t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xC, 0xF, false);
    return t;
}
// the value of the lane 16 below; row 0 takes `first` (its own lanes')
__device__ __forceinline__ uint32_t shift16(uint32_t v, uint32_t first)
{
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);             // [0] = rows {0, 0, 2, 2}, [1] = rows {1, 1, 3, 3}
    auto r32 = __builtin_amdgcn_permlane32_swap(r16[1], r16[1], false, false);    // [0] = rows {1, 1, 1, 1}
    const uint32_t l = threadIdx.x;
    return l < 16u ? first : (l & 48u) == 32u ? r32[0] : r16[0];
}
// bits 0, 4, 8 .. 60 of x (nothing else set) -> bits 0 .. 15
__device__ __forceinline__ uint32_t compress4(uint64_t x)
{
    x = (x | (x >> 3)) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    return (uint32_t)((x | (x >> 24)) & 0xFFFFull);
}

// Cross-lane sums and minima as DPP moves and row swaps (VALU latency) instead of ds_bpermute round trips through the LDS pipeline: this kernel is one wave per capture
// walking a serial state machine, and what it waits for is mostly these (round 5: a lone capture's, or a sixteen-frame capture's, k_scan is 35 us per frame).
template <int CTRL> __device__ __forceinline__ unsigned sdpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned quad_sum(unsigned v) { v += sdpp<0xB1>(v); v += sdpp<0x4E>(v); return v; }                // over lanes 4q .. 4q + 3, in all four
__device__ __forceinline__ unsigned row_sum(unsigned v) { v = quad_sum(v); v += sdpp<0x141>(v); v += sdpp<0x140>(v); return v; }   // over the 16 lanes of a row (^7 then ^15 pair up what ^1, ^2 left)
__device__ __forceinline__ unsigned other_row_even(unsigned v)                   // for the lanes of rows 1 and 3: the value of the lane 16 below (rows 0 and 2 get their own)
{
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return r[0];
}
__device__ __forceinline__ int wave_sum(int v)
{
    unsigned u = row_sum((unsigned)v);
    auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u = r16[0] + r16[1];
    auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return (int)(r32[0] + r32[1]);
}
__device__ __forceinline__ unsigned wave_min(unsigned v)
{
    v = min(v, sdpp<0xB1>(v)); v = min(v, sdpp<0x4E>(v)); v = min(v, sdpp<0x141>(v)); v = min(v, sdpp<0x140>(v));
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = min(r16[0], r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return min(r32[0], r32[1]);
}

// LTS_Sequence_11a (channel_11a.hpp:13-18): 1 -> +norm_one, 0 -> -norm_one
__device__ __constant__ uint8_t kLtsSeq[64] = {
    0,1,0,0,1,1,0,1,0,1,0,0,0,0,0,1, 1,0,0,1,0,1,0,1,1,1,1,0,0,0,0,0,
    0,0,0,0,0,0,1,1,0,0,1,1,0,1,0,1, 1,1,1,1,1,0,0,1,1,0,1,0,1,1,1,1 };

// data carrier k (0..47) -> FFT bin, in demap order -26..-1, +1..+26 without pilots (demapper11a.hpp:20-37)
__device__ __forceinline__ int carrier_bin(int k)
{
    // negative half: bins 38..63 without 43, 57 ; positive half: bins 1..26 without 7, 21
    if (k < 24) { int b = 38 + k; if (b >= 43) b++; if (b >= 57) b++; return b; }
    int b = 1 + (k - 24); if (b >= 7) b++; if (b >= 21) b++; return b;
}

// The tables the two out-of-line sections use, handed over BY VALUE: a reference to the kernel's argument block would force that block
// into private memory, and everything read from it (thresholds, strides) would count as lane-varying -- the whole carrier-sense state machine
// would then run in vector registers under exec masks.
struct ScanTabs { const short* uatan2; const uint32_t* rot; const uint8_t* demap; const uint16_t* deint; const uint32_t* tw64; const uint32_t* tw16; };
template <typename P> __device__ __forceinline__ P* uni_ptr(P* p)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (P*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ Tables tables_of(const ScanTabs& b)
{
    Tables T = {};
    T.uatan2 = uni_ptr(b.uatan2); T.rot = uni_ptr(b.rot); T.demap = uni_ptr(b.demap); T.deint = uni_ptr(b.deint); T.tw64 = uni_ptr(b.tw64); T.tw16 = uni_ptr(b.tw16);
    return T;
}

// ---- the frame's header, out of line (once per frame: its temporaries and table pointers stay out of the carrier-sense loop's register allocation).
// T11aLTS on the 144 samples at lts_start (channel_11a.hpp:206-330): CFO estimate, frequency shift, FFT<64>, channel inverse -> *fx; and the SIGNAL symbol
// behind them: T11aDataSymbol .. T11aViterbiSig .. T11aPLCPParser (PHY_11a.hpp:389-580).  Nothing observable happens between the two bricks' firings, so
// they run as one section: the two FFTs side by side in two groups of sixteen lanes (packed arithmetic, dev_arith.h), the frequency and channel
// coefficients handed on in registers (and stored for the per-symbol kernels).  Every field of the result is wave-uniform.
struct SigOut { uint32_t ok, kbps, len, nsym, cr, nb; int cfo_comp, sfo_comp, cfo_tr, sfo_tr, cfo; };
__device__ __noinline__ SigOut header_section(ScanTabs tabs, uint32_t lts_start_, FrameCtx* fx_)
{
    const Tables T = tables_of(tabs);
    FrameCtx* fx = uni_ptr(fx_);
    const uint32_t lts_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)lts_start_), sym_start = lts_start + 144u;
    __shared__ uint32_t s_a[64], s_b[64];
    __shared__ uint8_t  s_soft[64];
    const int lane = threadIdx.x, e = lane & 15;
    auto sync = []() { wave_lds_sync(); };
    // table reads that depend on no sample: issued first
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    const int t2 = lane < 24 ? 2 * lane : 0;
    const int d0 = T.deint[t2], d1 = T.deint[t2 + 1];                          // T11aDeinterleaveBPSK: the two carriers of trellis step t = lane
    // the samples sit in the ring.  LTS: x[n] = sample 8 + n, the first 64 are >>1 (rep_shift_right<16>, :216); SIGNAL: skip CP 8, >>1 (channel_11a.hpp:643-644)
    const cpx x1 = sra(unpack(s_ring[(lts_start + 8u + (uint32_t)lane) & (kRing - 1u)]), 1);
    const cpx x2 = unpack(s_ring[(lts_start + 72u + (uint32_t)lane) & (kRing - 1u)]);
    const cpx sg = sra(unpack(s_ring[(sym_start + 8u + (uint32_t)lane) & (kRing - 1u)]), 1);
    int re, im; conj_mul32(x2, x1, re, im);                                    // FreqOffsetEstimate<16> (dspalg.hpp:226-243)
    const int sum_re = __builtin_amdgcn_readfirstlane(wave_sum(re >> 5)), sum_im = __builtin_amdgcn_readfirstlane(wave_sum(im >> 5));
    const int arg = __builtin_amdgcn_readfirstlane(uatan2(T, sum_im, sum_re));
    const int cfo = arg >> 6;                                                  // size_t divisor: unsigned division = floor (dspalg.hpp:242)
    // BuildFrequencyShiftCoeffs<64>(.., 0, CFO_est): ph = lane*cfo (mod 2^16)   (dspalg.hpp:200-208)
    const cpx fc = rot_coeff(T, w16(lane * cfo));
    fx->freq[lane] = pack(fc);
    s_a[lane] = pack(mul_q15(x1, fc));                                         // FrequencyShift (:120)
    s_b[lane] = pack(mul_q15(sg, fc));                                         // TFreqCompensation
    // FFT<64> twice: lanes 0..15 the LTS, lanes 16..31 the SIGNAL symbol (lanes 32..63 mirror them: same values to the same words)
    {
        uint32_t* const sf = (lane & 16) ? s_b : s_a;
        sync();
        const pcx xin[4] = { sf[e], sf[e + 16], sf[e + 32], sf[e + 48] };
        fft64_core_pk(xin, sf, e, W, sync);                                    // bin j at slot bitrev6(j)
    }
    const unsigned slot = __brev((unsigned)lane) >> 26;                        // lane L takes bin L
    const cpx Yl = unpack(s_a[slot]), Ys = unpack(s_b[slot]);
    cpx eq = mk(0, 0);
    {
        uint32_t coef = 0;
        if (!(lane >= 28 && lane < 36)) {                                      // _channel_estimation (:125-178)
            const int en = sqnorm(Yl) >> 8;
            const cpx L = mk(kLtsSeq[lane] ? 1600 : -1600, 0);
            int cre, cim; conj_mul32(L, Yl, cre, cim);
            int rre = 0, rim = 0;
            if (en != 0) { rre = cre / en; rim = cim / en; }
            coef = pack(mk(w16(rre), w16(rim)));
            int qr, qi; mul32(Ys, unpack(coef), qr, qi);                       // TChannelEqualization (channel_11a.hpp:548-574)
            eq = mk(w16(qr >> 8), w16(qi >> 8));
        }
        fx->chan[lane] = coef;
    }
    // TPhaseCompensate with the reset CompCoeffs (0x7fff, 0) (ieee80211facade.hpp:198-206)
    const cpx pc = mul_q15(eq, mk(0x7fff, 0));
    const uint32_t ppc = pack(pc);
    // _pilot_track (pilot.hpp:166-233), symbol_count = 127 -> PilotSgn[127] = 0
    const cpx p43 = unpack((uint32_t)__builtin_amdgcn_readlane((int)ppc, 43)), p57 = unpack((uint32_t)__builtin_amdgcn_readlane((int)ppc, 57));
    const cpx p7 = unpack((uint32_t)__builtin_amdgcn_readlane((int)ppc, 7)), p21 = unpack((uint32_t)__builtin_amdgcn_readlane((int)ppc, 21));
    const int th1 = __builtin_amdgcn_readfirstlane(uatan2(T, p43.im, p43.re)), th2 = __builtin_amdgcn_readfirstlane(uatan2(T, p57.im, p57.re));
    const int th3 = __builtin_amdgcn_readfirstlane(uatan2(T, p7.im, p7.re)),   th4 = __builtin_amdgcn_readfirstlane(uatan2(T, -p21.im, -p21.re));
    const int avg = w16((th1 + th2 + th3 + th4) / 4);
    const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
    const int cfo_tracker = w16(avg >> 2), sfo_tracker = w16(del >> 2);
    const int cfo_comp = w16(avg + cfo_tracker), sfo_comp = w16(del + sfo_tracker);
    // T11aDemapBPSK on the rotated carriers, every bin in its own lane -> T11aDeinterleaveBPSK
    {
        const int cidx = lane < 32 ? lane : lane - 64;                         // signed carrier number
        const cpx r = mul_q15(pc, rot_coeff(T, w16(avg + cidx * del)));
        int v = r.re >> 4; v = min(max(v, -128), 127);                         // demap_limit (demapper.h:141-151)
        // DemapperCore's BPSK step function (demapper.h:55-130): the soft value is the number of steps at or below v
        s_soft[lane] = (uint8_t)((v >= -30) + (v >= -17) + (v >= -8) + (v >= 0) + (v >= 9) + (v >= 18) + (v >= 31));
    }
    sync();
    uint8_t sa = 0, sb = 0;                                                     // de-interleaved soft pair of trellis step t = lane (t < 24)
    if (lane < 24) { sa = s_soft[carrier_bin(d0)]; sb = s_soft[carrier_bin(d1)]; }
    // ---- Viterbi_sig11 (viterbicore.h:35-261), the 64 states in the 64 lanes, IN PLACE: the butterfly (j, j + 32) -> (2 j, 2 j + 1) leaves new state rol6(s)
    // in the lane that held s, so after u steps lane p holds state rol6^u(p) and a step's partner is lane p ^ (32 >> (u - 1) % 6): a permlane swap, a row
    // rotation or a quad permutation -- no LDS round trip in the 24-step chain (round 5: two ds_bpermute per step, 5 k of the section's 9 k cycles).
    // Both generators hold the register's oldest and newest bit (0155, 0117), so the branch to (64 | n) costs 28 - (the branch to n): one metric per step.
    auto rol6c = [](unsigned x, unsigned k) { return ((x << k) | (x >> (6u - k))) & 63u; };
    int selV[6], negV[6], n29V[6];                                              // per step phase k = u % 6: the lane's state is rol6(lane, k)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const unsigned n = k ? rol6c((unsigned)lane, (unsigned)k) : (unsigned)lane;
        const int cA = __popc(n & 0155) & 1, cB = __popc(n & 0117) & 1;
        selV[k] = (cA ^ cB) ? -1 : 0; negV[k] = cB ? -1 : 0; n29V[k] = cB ? 29 : 0;
    }
    unsigned m = (lane == 0) ? 0u : 0x30u;
    uint64_t dec[25];                                                           // decision of the state in lane p at step u: bit p of dec[u] (scalars: the loops are unrolled)
    dec[0] = 0;
#pragma unroll
    for (int u = 1; u <= 24; u++) {
        const int ph = (u - 1) % 6, k = u % 6;
        const unsigned X = 32u >> ph;
        // the partner's metric (lane ^ X)
        unsigned pm;
        if (ph == 0) { auto r = __builtin_amdgcn_permlane32_swap(m, m, false, false); pm = lane < 32 ? r[1] : r[0]; }
        else if (ph == 1) { auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false); pm = (lane & 16) ? r[0] : r[1]; }
        else if (ph == 2) pm = sdpp<0x128>(m);                                  // row_ror:8
        else if (ph == 3) {                                                     // lanes with bit 2 set take lane - 4 (row_shr:4, banks 1 and 3), the others lane + 4 (row_shl:4, banks 0 and 2)
            const int t4 = __builtin_amdgcn_update_dpp(0, (int)m, 0x114, 0xF, 0xA, false);
            pm = (unsigned)__builtin_amdgcn_update_dpp(t4, (int)m, 0x104, 0xF, 0x5, false);
        }
        else if (ph == 4) pm = sdpp<0x4E>(m);                                   // quad_perm [2,3,0,1]
        else pm = sdpp<0xB1>(m);                                                // quad_perm [1,0,3,2]
        asm volatile("" : "+v"(pm));                                            // (the exchange stays a move: see k_rx.hip on DPP operands folded into subtractions)
        const bool own_hi = ((unsigned)lane & X) != 0u;                         // the lane's old state has bit 5 set: it is the (j + 32) of its butterfly
        const unsigned m0 = own_hi ? pm : m, m1 = own_hi ? m : pm;              // metrics of the predecessors n >> 1 and 32 + (n >> 1) of the new state n
        // branch metrics (VIT_MA / VIT_MB, viterbilut.h): soft values of step u, wave-uniform
        const int va = __builtin_amdgcn_readlane((int)sa, u - 1), vb = __builtin_amdgcn_readlane((int)sb, u - 1);
        const int P = 2 * va + 2 * vb, dQP = 14 - 4 * va;                       // (cA, cB) = (0, 0): P; (1, 0): Q = 14 - 2 va + 2 vb = P + dQP; (., 1): 28 - that
        const int x = P + (selV[k] & dQP);
        const unsigned b0 = (unsigned)((x ^ negV[k]) + n29V[k]), b1 = 28u - b0;
        const unsigned c0 = (m0 + b0) & 0xFE, c1 = ((m1 + b1) & 0xFF) | 1;
        m = min(c0, c1);
        dec[u] = __ballot(m & 1);
        if ((u & 7) == 0) m = (m - (wave_min(m) & 0xFE)) & 0xFF;
    }
    // (24 steps = four turns of the rotation: lane p holds state p again; the extra normalisation before the trace-back does not change LSBs or the arg-min order)
    const unsigned key = (m << 8) | ((unsigned)lane << 2);
    const unsigned kmin = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min(key));
    int pos = (int)((kmin >> 2) & 0x3F) | (int)(((kmin >> 8) & 1) << 6);
    uint32_t sig = 0;
#pragma unroll
    for (int b = 0; b < 24; b++) {
        // reference emits MSB-first per byte while walking back: bit b of the walk is output bit (23-b)
        sig |= (uint32_t)((pos >> 6) & 1) << (23 - b);
        pos = (pos >> 1) & 0x3F;
        // the decision of state pos at step 23 - b sits in the lane that held it then: ror6^(23 - b)(pos)
        const unsigned r = (unsigned)(23 - b) % 6u;
        const unsigned ln = r ? rol6c((unsigned)pos, 6u - r) : (unsigned)pos;
        pos |= (int)((dec[23 - b] >> ln) & 1) << 6;
    }
    sig = (uint32_t)__builtin_amdgcn_readfirstlane((int)sig) >> 6;             // viterbi.hpp:39 (wave-uniform)
    // ---- T11aPLCPParser::_parse_plcp (PHY_11a.hpp:548-580)
    bool ok = true;
    sig &= 0xFFFFFF;
    if (sig & 0xFC0010) ok = false;
    uint32_t par = (sig >> 16) ^ sig; par = (par >> 8) ^ par; par = (par >> 4) ^ par; par = (par >> 2) ^ par; par = (par >> 1) ^ par;
    if (par & 1) ok = false;
    uint32_t kbps = 0; int nd = 0, nb = 0, cr = 0;
    switch (sig & 0xF) {                                                        // ieee80211a_cmn.h:97-107, :65-94, :114-149
    case 0x8: kbps = 48000; nd = 192; nb = 6; cr = 1; break;  case 0x9: kbps = 24000; nd = 96;  nb = 4; cr = 0; break;
    case 0xA: kbps = 12000; nd = 48;  nb = 2; cr = 0; break;  case 0xB: kbps = 6000;  nd = 24;  nb = 1; cr = 0; break;
    case 0xC: kbps = 54000; nd = 216; nb = 6; cr = 2; break;  case 0xD: kbps = 36000; nd = 144; nb = 4; cr = 2; break;
    case 0xE: kbps = 18000; nd = 72;  nb = 2; cr = 2; break;  case 0xF: kbps = 9000;  nd = 36;  nb = 1; cr = 2; break;
    default: ok = false; break;
    }
    const uint32_t len = (sig >> 5) & 0xFFF;
    if (len > 2500) ok = false;
    SigOut O;
    O.ok = ok ? 1u : 0u; O.kbps = kbps; O.len = len; O.cr = (uint32_t)cr; O.nb = (uint32_t)nb;
    O.nsym = ok ? (len * 8 + 16 + 6 + (uint32_t)nd - 1) / (uint32_t)nd : 0u;                  // B11aGetSymbolCount
    O.cfo_comp = cfo_comp; O.sfo_comp = sfo_comp; O.cfo_tr = cfo_tracker; O.sfo_tr = sfo_tracker; O.cfo = cfo;
    __threadfence_block();
    sync();
    return O;
}

// Stream continuation: the record of a resume point (see k_scan).  Out of line, everything by value: its selects and shuffles stay out of the
// carrier-sense loop's register allocation (inlined at its two call sites it cost the loop 56 more SGPR spills and 48 bytes of scratch).
constexpr uint32_t kContMagic = 0x534F5241u;
__device__ __noinline__ void cont_store(uint32_t* crec, uint32_t* consumed, uint32_t Hv, uint32_t Zr, uint32_t Zi, uint32_t Ze, int rr, int ri, int re,
                                        uint32_t sense_count, uint32_t high_count, int peak_corr, int peak_index, uint32_t dc_cnt, int sum_dc_re, int sum_dc_im,
                                        int dc_re, int dc_im, uint32_t at)
{
    const int l = threadIdx.x;
    // window element (l & 3) sits in lanes 4 (l & 3) ..
    const uint32_t zr = (uint32_t)__shfl((int)Zr, 4 * (l & 3)), zi = (uint32_t)__shfl((int)Zi, 4 * (l & 3)), ze = (uint32_t)__shfl((int)Ze, 4 * (l & 3));
    const uint32_t hv = (uint32_t)__shfl((int)Hv, l & 15);
    uint32_t v = l < 16 ? hv : l < 20 ? zr : l < 24 ? zi : l < 28 ? ze : 0u;
    const uint32_t sc[16] = { (uint32_t)rr, (uint32_t)ri, (uint32_t)re, sense_count, high_count, (uint32_t)peak_corr, (uint32_t)peak_index, dc_cnt,
                              (uint32_t)sum_dc_re, (uint32_t)sum_dc_im, (uint32_t)dc_re, (uint32_t)dc_im, at, kContMagic, 0u, 0u };
#pragma unroll
    for (int k = 0; k < 16; k++) if (l == 28 + k) v = sc[k];
    crec[l] = v;
    if (l == 0) *consumed = at;
}

// x / 14 and ceil(x / 14) for x < 2^31 (a source call = 14 samples at the 20 MHz rate) as a multiply: 0x92492493 = ceil(2^35 / 14)
__device__ __forceinline__ uint32_t div14(uint32_t x) { return (uint32_t)(((uint64_t)x * 0x92492493ull) >> 35); }
__device__ __forceinline__ uint32_t call_end(uint32_t burst_end) { return div14(burst_end + 13u) * 14u; }   // end of the source call that delivers the burst ending at burst_end

__global__ void __launch_bounds__(64, 4) k_scan(ScanArgs A)
{
    const uint32_t cap_i = blockIdx.x;
    if (cap_i >= A.ncaps) return;
    const int lane = threadIdx.x;
    const CapDesc cd = A.caps[cap_i];
    const uint32_t* iq = A.iq + cd.offset;
    const uint32_t sh = A.str - 1u;                                     // 40 MHz input: 20 MHz-rate sample s is raw sample 2 s (TDownSample2 keeps the even ones, samples.hpp:27-45)
    const uint32_t NS = (cd.nsamples / (14u << sh)) * 14u;              // 20 MHz-rate samples in whole source calls
    const Tables& T = A.T;
    const ScanTabs tabs = { T.uatan2, T.rot, T.demap, T.deint, T.tw64, T.tw16 };

    // ---- the sample ring: s_ring holds [r_lo, r_hi), the 256 samples behind r_hi are in flight in four registers
    uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0, r_lo = 0, r_hi = 0;
    auto issue = [&](uint32_t base) {
        const uint32_t i = base + (uint32_t)lane;
        n0 = i < NS ? iq[(size_t)i << sh] : 0u;
        n1 = i + 64u < NS ? iq[(size_t)(i + 64u) << sh] : 0u;
        n2 = i + 128u < NS ? iq[(size_t)(i + 128u) << sh] : 0u;
        n3 = i + 192u < NS ? iq[(size_t)(i + 192u) << sh] : 0u;
    };
    auto aim = [&](uint32_t at) { if (at < r_lo || at >= r_hi + 256u) { r_lo = r_hi = at; issue(at); } };      // the walk goes on at `at`: start fetching there unless it is at hand
    auto fill = [&](uint32_t at, uint32_t len) {                        // [at, at + len) readable from s_ring (len <= 256)
        aim(at);
        while (at + len > r_hi) {
            const uint32_t w = r_hi + (uint32_t)lane;
            s_ring[w & (kRing - 1u)] = n0; s_ring[(w + 64u) & (kRing - 1u)] = n1; s_ring[(w + 128u) & (kRing - 1u)] = n2; s_ring[(w + 192u) & (kRing - 1u)] = n3;
            r_hi += 256u;
            if (r_hi - r_lo > kRing) r_lo = r_hi - kRing;
            issue(r_hi);
        }
    };
    issue(0);

    // the fills a call used to get from a kernel of their own in front of this one (kernels.h ScanArgs): nobody reads any of these words before this kernel has ended
    if (A.own_slots && A.slot_row) for (uint32_t i = (uint32_t)lane; i < cd.nslots; i += 64u) A.slot_row[cd.slot_base + i] = 0xFFFFFFFFu;
    if (cap_i == 0) {
        if (A.zero_a) for (uint32_t i = (uint32_t)lane; i < A.nzero_a; i += 64u) A.zero_a[i] = 0u;
        if (A.zero_b) for (uint32_t i = (uint32_t)lane; i < A.nzero_b; i += 64u) A.zero_b[i] = 0u;
    }
    // STS correlation patterns (cca.hpp:268-277), held for the whole walk: check_sync only ever uses patterns 0..3, one tap per lane of a row;
    // establish_sync all sixteen: pattern lane >> 2, taps 4 (lane & 3) .. + 3
    const uint32_t chk0 = T.sts[lane & 15], chk1 = T.sts[16 + (lane & 15)], chk2 = T.sts[32 + (lane & 15)], chk3 = T.sts[48 + (lane & 15)];
    const uint32_t est_base = (uint32_t)(lane >> 2) * 16u + 4u * (uint32_t)(lane & 3);
    const uint32_t est0 = T.sts[est_base], est1 = T.sts[est_base + 1], est2 = T.sts[est_base + 2], est3 = T.sts[est_base + 3];

    // ---- carrier-sense state (cca.hpp:126-158), wave-uniform
    uint32_t Hv = 0;                         // sample_his in TIME ORDER, one packed sample per lane (lane & 15, oldest = 0): 4 bursts of 4, already >>2
    Acc4 ac_re, ac_im, energy;
    uint32_t auto_count = 0, sense_count = 0, high_count = 0; int sync_high = 0, peak_corr = 0, peak_index = 0;
    uint32_t dc_cnt = 8; int sum_dc_re = 0, sum_dc_im = 0;            // TDCEstimator (dc.hpp:92-166); all 4 lanes of the vcs are equal
    int dc_re = 0, dc_im = 0;                                          // CF_VecDC (survives frame resets)
    int cca_detected = 0;
    uint32_t frame_start = 0, nfr = 0;
    bool to_pending = false; uint32_t pend_end = 0;                    // E_CS_TIMEOUT raised: RxThread resets the bricks at the end of that source call (fb11a_demod.cpp:41-45)

    auto cs_reset = [&]() {
        Hv = 0;
        acc_clear(ac_re); acc_clear(ac_im); acc_clear(energy);
        auto_count = sense_count = high_count = 0; sync_high = 0; peak_corr = 0; peak_index = 0;
        dc_cnt = 8; sum_dc_re = sum_dc_im = 0;
    };
    cs_reset();

    // ---- stream continuation (sora_rx_set_stream_mode; TRxStream hands the graph an endless stream, rxstream.hpp:34-66: the DC estimate of
    // dc.hpp:92-166 integrates for ever, the carrier-sense windows and counters carry over from read to read).  A RESUME POINT is a position
    // where a burst boundary falls on a source-call boundary while the graph is in plain carrier sense (no detection under way, no event
    // pending): everything the graph knows there is the record below.  The capture's last resume point is published (A.consumed) and its
    // record kept; the next call's capture k starts AT that point of the stream and the record is its initial state -- so what the graph
    // reports from there on is what it reports on the uncut stream, and a frame cut by the end of a capture is simply found again.
    const bool streaming = A.cont != nullptr;
    auto cont_save = [&](uint32_t at) {
        cont_store(A.cont + (size_t)cap_i * kContWords, A.consumed + cap_i, Hv, ac_re.Z, ac_im.Z, energy.Z, ac_re.reg, ac_im.reg, energy.reg, sense_count,
                high_count, peak_corr, peak_index,
                   dc_cnt, sum_dc_re, sum_dc_im, dc_re, dc_im, at << sh);
    };
    if (streaming) {
        if (lane == 0) A.consumed[cap_i] = 0;
        const uint32_t v = A.cont[(size_t)cap_i * kContWords + lane];
        if ((uint32_t)__builtin_amdgcn_readlane((int)v, 41) == kContMagic) {       // a record exists: this capture continues a stream
            auto sc = [&](int k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 28 + k); };
            Hv = (uint32_t)__shfl((int)v, lane & 15);
            ac_re.Z = (uint32_t)__shfl((int)v, 16 + ((lane & 15) >> 2)); ac_im.Z = (uint32_t)__shfl((int)v, 20 + ((lane & 15) >> 2));
                energy.Z = (uint32_t)__shfl((int)v, 24 + ((lane & 15) >> 2));
            ac_re.reg = (int)sc(0); ac_im.reg = (int)sc(1); energy.reg = (int)sc(2); sense_count = sc(3); high_count = sc(4); peak_corr = (int)sc(5); peak_index = (int)sc(6);
            dc_cnt = sc(7); sum_dc_re = (int)sc(8); sum_dc_im = (int)sc(9); dc_re = (int)sc(10); dc_im = (int)sc(11);
        }
    }

    // frames queued for the per-frame kernels and not yet in joblist[]: lane q < nq holds (code rate << 28) | frame-table row
    uint32_t q_job = 0, nq = 0;
    auto flush_jobs = [&]() {
        if (nq == 0) return;
        const bool have = (uint32_t)lane < nq;
        const uint32_t cr = q_job >> 28, row = q_job & 0x0FFFFFFFu;
        const uint64_t m0 = __ballot(have && cr == 0u), m1 = __ballot(have && cr == 1u), m2 = __ballot(have && cr == 2u);
        // lane r < 3 reserves list r's entries (the three atomics are in flight together)
        const uint32_t cnt = lane == 0 ? (uint32_t)__popcll(m0) : lane == 1 ? (uint32_t)__popcll(m1) : (uint32_t)__popcll(m2);
        uint32_t base = 0;
        if (lane < 3 && cnt) base = atomicAdd(A.njobs + lane, cnt);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)base, 0), b1 = (uint32_t)__builtin_amdgcn_readlane((int)base, 1), b2 = (uint32_t)__builtin_amdgcn_readlane((int)base, 2);
        if (have) {
            const uint64_t mine = cr == 0u ? m0 : cr == 1u ? m1 : m2, below = ((uint64_t)1 << lane) - 1u;
            A.joblist[(size_t)cr * A.nrows + (cr == 0u ? b0 : cr == 1u ? b1 : b2) + (uint32_t)__popcll(mine & below)] = row;
        }
        nq = 0;
    };
    uint32_t s = 0;                                 // next burst, 20 MHz-rate sample index
    PROBE_DECL();
    PROBE_TK();
    // Every frame of the capture is found and counted, as RxThread reports every frame; those past the row limit get no row and
    // no decode job (their per-frame context goes to the capture's spare FrameCtx), and the host flags the capture's last row.
    auto ctx_of = [&](uint32_t k) { return A.fctx + (k < A.max_frames ? (size_t)cap_i * A.max_frames + k : (size_t)A.nrows + cap_i); };
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

    for (;;) {
        PROBE_T(_tg);
        if (to_pending && s + 4u > pend_end) {      // the source call that raised E_CS_TIMEOUT is over: ResetCarrierSense(); scs->Reset()
            to_pending = false; cca_detected = 0; cs_reset();
        }
        if (s + 4u > NS) break;
        if (!sync_high) {
            // ================= TDCRemoveEx<4> -> TCCA11a -> TDCEstimator, up to sixteen bursts at once, one sample per lane (lane = 4 b + e).
            // The per-sample products run once, the sliding sums are prefix sums over the burst groups, and the reference's test (cca.hpp:386-437)
            // is evaluated for every burst of the pass; what the tests mean -- auto_count reaching four (establish_sync), sense_count, the time-out -- follows
            // from the ballot.  The DC estimate moves every eighth burst and changes what the bursts behind it see: a pass holds at most one such update
            // inside (burst p: the lanes behind it subtract the new estimate, which only needs the sums up to p) and one at its last burst.  A pass ends with
            // the burst that calls establish_sync.
            const bool plain = !to_pending && auto_count == 0;
            const uint32_t p = dc_cnt;                                           // the burst of this pass that updates the estimate
            uint32_t K = min(min((NS - s) >> 2, p + 9u), 16u);
            uint32_t jraise = 0xFFFFu, ce_raise = 0;                             // the burst of this pass at which E_CS_TIMEOUT would be raised, and the end of its source call
            if (to_pending) K = min(K, (pend_end - s) >> 2);
            else {
                // sense_count reaches 84 at burst jraise unless a test before it is true; the bursts behind it that the same source call delivers still run, and the
                // bricks are reset at that call's end: a pass does not go beyond it
                jraise = sense_count >= 84u ? 0u : (84u - sense_count + 3u) / 4u - 1u;
                if (jraise < K) { const uint32_t be = s + 4u * jraise + 4u; ce_raise = call_end(be); K = min(K, jraise + 1u + ((ce_raise - be) >> 2)); }
            }
            if (streaming) {
                const uint32_t m = s - div14(s) * 14u;                          // position in the source call
                if (m == 0u && plain) cont_save(s);
                // a pass ends at the next resume point (burst boundary = source-call boundary), so that it is seen: j bursts on, m + 4 j = 0 (mod 14): j = 3 (m / 2) (mod 7), 0 -> 7
                const uint32_t j = (3u * (m >> 1)) % 7u;
                K = min(K, j ? j : 7u);
            }
            PROBE_A(_tg, 1);
            PROBE_T0();
            fill(s, 64u);
            const cpx x = unpack(s_ring[(s + (uint32_t)lane) & (kRing - 1u)]);
            int dcr = dc_re, dci = dc_im;                                        // TDCRemoveEx's operand, per lane
            if (p + 1u < K) {
                // the update at burst p: TDCEstimator's sums (dc.hpp:132-163) over bursts 0 .. p with the present estimate
                const unsigned hr = group_scan64(quad_sum((unsigned)(w16(x.re - dc_re) >> 5))), hi_ = group_scan64(quad_sum((unsigned)(w16(x.im - dc_im) >> 5)));
                const int sr = w16(sum_dc_re + __builtin_amdgcn_readlane((int)hr, (int)(4u * p))), si = w16(sum_dc_im + __builtin_amdgcn_readlane((int)hi_, (int)(4u * p)));
                const int d1r = w16(dc_re + (sr >> 2)), d1i = w16(dc_im + (si >> 2));
                const bool behind = (uint32_t)lane > 4u * p + 3u;
                dcr = behind ? d1r : dc_re; dci = behind ? d1i : dc_im;
            }
            const cpx pi = mk(w16(x.re - dcr), w16(x.im - dci));                    // TDCRemoveEx
            const cpx pii = sra(pi, 2);
            const uint32_t ppk = pack(pii);
            const uint32_t prev = shift16(ppk, Hv);                                 // the sample 16 earlier: the history for the first four bursts
            int re, im; conj_mul32(pii, unpack(prev), re, im);                      // GetAutoCorrelation (cca.hpp:165-186)
            unsigned vr = (unsigned)(re >> 4), vi = (unsigned)(im >> 4), ve = (unsigned)(sqnorm(pii) >> 4);   // GetEnergy (cca.hpp:188-193)
            vr = quad_sum(vr); vi = quad_sum(vi); ve = quad_sum(ve);                // a burst's four samples sit in one quad
            const uint32_t Dr = group_scan64(quad_sum((unsigned)(pi.re >> 5))), Di = group_scan64(quad_sum((unsigned)(pi.im >> 5)));   // TDCEstimator terms
            // the K bursts' sliding sums at once.  After burst b the accumulator holds reg + sum_{i<=b} (d_i - z_i), z = the value that leaves the
            // 4-element window: the old elements for b < 4, d_{b-4} after that.  One prefix sum per stream over the burst groups.
            const uint32_t Rr = (uint32_t)ac_re.reg + group_scan64(vr - shift16(vr, ac_re.Z)), Ri = (uint32_t)ac_im.reg + group_scan64(vi - shift16(vi, ac_im.Z)),
                           Re = (uint32_t)energy.reg + group_scan64(ve - shift16(ve, energy.Z));
            const int iAuto_v = abs((int)Rr) + abs((int)Ri), iEnergy_v = (int)Re;
            const bool carrier = iEnergy_v > (int)A.thr && iAuto_v >= iEnergy_v - (iEnergy_v >> 3);
            // bit b: burst b's test
            const uint32_t hits = compress4((uint64_t)__ballot(carrier) & 0x1111111111111111ull) & ((1u << K) - 1u);
            // auto_count (cca.hpp:400-414): consecutive true tests, establish_sync at every true test from the fourth on: the first run of four in
            // {the a tests that were true before the pass, this pass's tests}
            const uint32_t a = min(auto_count, 3u);
            const uint32_t M = (hits << a) | ((1u << a) - 1u);
            const uint32_t run = M & (M >> 1) & (M >> 2) & (M >> 3);
            const bool est = run != 0;
            const uint32_t done = est ? (uint32_t)__builtin_ctz(run) + 4u - a : K;                   // bursts taken
            const uint32_t bits = (1u << done) - 1u;
            const uint32_t th = hits & bits;                                         // the tests of the bursts taken
            if (est) auto_count = 4;
            else {
                const uint32_t inv = ~th & bits;                                     // ... that were false: auto_count = the true tests behind the last of them
                auto_count = inv ? done - 1u - (31u - (uint32_t)__builtin_clz(inv)) : min(auto_count + done, 4u);
            }
            // sense_count (cca.hpp:398, :402, :433-437): + 4 per burst, 0 at a true test; E_CS_TIMEOUT from 84 on
            if (!to_pending && jraise < min(th ? (uint32_t)__builtin_ctz(th) : done, done)) { to_pending = true; pend_end = ce_raise; }
            sense_count = th ? 4u * (done - 1u - (31u - (uint32_t)__builtin_clz(th))) : sense_count + 4u * done;
            {
                const int last = (int)(4u * (done - 1u));
                ac_re.reg = __builtin_amdgcn_readlane((int)Rr, last); ac_im.reg = __builtin_amdgcn_readlane((int)Ri, last); energy.reg = __builtin_amdgcn_readlane((int)Re, last);
                // the windows <- the last four of {old window, d_0 .. d_{done-1}} (element g in lanes 4 g .. 4 g + 3 of every row); history <- its last 16 samples
                const uint32_t src = ((uint32_t)lane & 15u) + 4u * done;            // index in {old 0..15, new 16..79}
                const int from_old = (int)(((uint32_t)lane & 48u) | (src & 15u)), from_new = (int)(src - 16u);
                const uint32_t fr = (uint32_t)__shfl((int)vr, from_new), fi = (uint32_t)__shfl((int)vi, from_new), fe = (uint32_t)__shfl((int)ve, from_new);
                const uint32_t fresh = (uint32_t)__shfl((int)ppk, from_new);
                if (done < 4u) {
                    const bool old = src < 16u;
                    const uint32_t kr = (uint32_t)__shfl((int)ac_re.Z, from_old), ki = (uint32_t)__shfl((int)ac_im.Z, from_old), ke = (uint32_t)__shfl((int)energy.Z, from_old);
                    const uint32_t keep = (uint32_t)__shfl((int)Hv, from_old);
                    ac_re.Z = old ? kr : fr; ac_im.Z = old ? ki : fi; energy.Z = old ? ke : fe; Hv = old ? keep : fresh;
                } else { ac_re.Z = fr; ac_im.Z = fi; energy.Z = fe; Hv = fresh; }
            }
            s += 4u * done;
            PROBE_ADD(3);
            if (est) {
                PROBE_T0();
                // establish_sync (cca.hpp:220-243) on the history as it is after this burst: GetCrossCorrelation (cca.hpp:202-218) with all sixteen patterns,
                // four taps per lane (the partial sums are 32-bit wrapping adds: any order)
                unsigned sr = 0, si = 0;
                {
                    const int q4 = 4 * (lane & 3);
                    const uint32_t h0 = (uint32_t)__shfl((int)Hv, q4), h1 = (uint32_t)__shfl((int)Hv, q4 + 1), h2 = (uint32_t)__shfl((int)Hv, q4 + 2), h3 = (uint32_t)__shfl((int)Hv, q4 + 3);
                    int cr_, ci_;
                    conj_mul32(unpack(est0), unpack(h0), cr_, ci_); sr += (unsigned)cr_; si += (unsigned)ci_;
                    conj_mul32(unpack(est1), unpack(h1), cr_, ci_); sr += (unsigned)cr_; si += (unsigned)ci_;
                    conj_mul32(unpack(est2), unpack(h2), cr_, ci_); sr += (unsigned)cr_; si += (unsigned)ci_;
                    conj_mul32(unpack(est3), unpack(h3), cr_, ci_); sr += (unsigned)cr_; si += (unsigned)ci_;
                }
                sr = quad_sum(sr); si = quad_sum(si);
                const int corr = abs((int)sr) + abs((int)si);
                int sum_corr = 0, best = 0, best_i = 0;
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const int cp = __builtin_amdgcn_readlane(corr, 4 * p);
                    if (cp > best) { best = cp; best_i = p; }
                    sum_corr += cp;
                }
                peak_corr = best; if (best > 0) peak_index = best_i;
                if (peak_corr > (sum_corr >> 3)) {
                    sync_high = 1; high_count = 0;
                    if (peak_index > 3) { high_count = (uint32_t)peak_index / 4; peak_index &= 3; }
                }
                PROBE_ADD(0);
            }
            // TDCEstimator (dc.hpp:92-166) is fed by every burst that leaves carrier sense unlocked: 16-bit running sums, the estimate moves when dc_cnt is used up
            // (burst p, and eight bursts on: a pass's last burst at most)
            const uint32_t ndc = done - (sync_high ? 1u : 0u);
            if (ndc > p) {
                const int at_p = __builtin_amdgcn_readlane((int)Dr, (int)(4u * p)), at_pi = __builtin_amdgcn_readlane((int)Di, (int)(4u * p));
                dc_re = w16(dc_re + (w16(sum_dc_re + at_p) >> 2)); dc_im = w16(dc_im + (w16(sum_dc_im + at_pi) >> 2));
                const uint32_t n1 = ndc - 1u - p;                                    // bursts fed behind the update
                const int lastd = (int)(4u * (ndc - 1u));
                const int seg_re = w16(__builtin_amdgcn_readlane((int)Dr, lastd) - at_p), seg_im = w16(__builtin_amdgcn_readlane((int)Di, lastd) - at_pi);
                if (n1 == 8u) { dc_re = w16(dc_re + (seg_re >> 2)); dc_im = w16(dc_im + (seg_im >> 2)); sum_dc_re = sum_dc_im = 0; dc_cnt = 7; }
                else { sum_dc_re = seg_re; sum_dc_im = seg_im; dc_cnt = 7u - n1; }
            } else if (ndc) {
                const int lastd = (int)(4u * (ndc - 1u));
                sum_dc_re = w16(sum_dc_re + __builtin_amdgcn_readlane((int)Dr, lastd));
                sum_dc_im = w16(sum_dc_im + __builtin_amdgcn_readlane((int)Di, lastd));
                dc_cnt -= ndc;
            }
            continue;
        }

        // ================= locked to the short training symbols: a burst only enters the history, every fourth one the history is correlated with the winning
        // pattern (check_sync, cca.hpp:245-265).  Sixteen bursts = four checks per pass: lane l holds sample l - 4 o of the pass, o = bursts of the current group
        // of four that are already in the history (lanes below 4 o carry those), so that row r of sixteen lanes is exactly the history check r looks at.
        {
            PROBE_T0();
            const uint32_t o = high_count & 3u;
            uint32_t Kn = min(16u - o, (NS - s) >> 2);
            if (to_pending) Kn = min(Kn, (pend_end - s) >> 2);
            fill(s, 64u);
            const cpx x = unpack(s_ring[(s + (uint32_t)lane - 4u * o) & (kRing - 1u)]);
            const cpx pi = mk(w16(x.re - dc_re), w16(x.im - dc_im));
            const uint32_t ppk = pack(sra(pi, 2));
            uint32_t hvr = Hv;                                                       // row_ror:4 o: lane l < 4 o <- history element 16 - 4 o + l
            if (o == 1u) hvr = sdpp<0x124>(Hv); else if (o == 2u) hvr = sdpp<0x128>(Hv); else if (o == 3u) hvr = sdpp<0x12C>(Hv);
            const uint32_t c = (uint32_t)lane < 4u * o ? hvr : ppk;
            const uint32_t pat = peak_index == 0 ? chk0 : peak_index == 1 ? chk1 : peak_index == 2 ? chk2 : chk3;
            int re, im; conj_mul32(unpack(pat), unpack(c), re, im);                  // GetCrossCorrelation, one tap per lane
            const unsigned ur = row_sum((unsigned)re), ui = row_sum((unsigned)im);
            unsigned dr = (unsigned)(pi.re >> 5), di = (unsigned)(pi.im >> 5);      // (the burst that drops the lock also feeds TDCEstimator)
            dr = quad_sum(dr); di = quad_sum(di);
            const uint32_t nchk = (o + Kn) >> 2;
            uint32_t taken = Kn;
#pragma unroll
            for (uint32_t r = 0; r < 4u; r++) {
                if (r < nchk && taken == Kn) {
                    const int corr = abs(__builtin_amdgcn_readlane((int)ur, 16 * (int)r)) + abs(__builtin_amdgcn_readlane((int)ui, 16 * (int)r));
                    if (corr < (peak_corr >> 1)) {
                        const uint32_t b = 4u * (r + 1u) - o - 1u;                  // the burst that carried the failing check
                        taken = b + 1u;
                        if (high_count + taken > 8u) { cca_detected = 1; frame_start = s + 4u * b + 4u; }
                        else {
                            sync_high = 0; sense_count = 0;
                            sum_dc_re = w16(sum_dc_re + w16(__builtin_amdgcn_readlane((int)dr, 16 * (int)r + 12)));
                            sum_dc_im = w16(sum_dc_im + w16(__builtin_amdgcn_readlane((int)di, 16 * (int)r + 12)));
                            if (dc_cnt == 0) {
                                dc_re = w16(dc_re + (sum_dc_re >> 2)); dc_im = w16(dc_im + (sum_dc_im >> 2));
                                dc_cnt = 8; sum_dc_re = sum_dc_im = 0;
                            }
                            dc_cnt--;
                        }
                        // (taken < Kn now ends the checks -- unless this was the pass's last burst, behind which there is no check anyway)
                        if (taken == Kn) break;
                    } else if (corr > peak_corr) peak_corr = corr;
                }
            }
            high_count += taken;
            {   // history <- the last 16 samples of {history, the bursts taken}
                const uint32_t src = ((uint32_t)lane & 15u) + 4u * taken;           // index in {old 0..15, new 16..}
                const uint32_t fresh = (uint32_t)__shfl((int)ppk, (int)((src - 16u + 4u * o) & 63u));
                if (taken < 4u) { const uint32_t keep = (uint32_t)__shfl((int)Hv, (int)(((uint32_t)lane & 48u) | (src & 15u))); Hv = src < 16u ? keep : fresh; }
                else Hv = fresh;
            }
            s += 4u * taken;
            PROBE_ADD(4);
        }
        if (!cca_detected) continue;

        // ================= a frame: T11aLTS takes the 144 samples behind the detection point (channel_11a.hpp:206-229), T11aDataSymbol 80 per symbol
        // (PHY_11a.hpp:389-428).  Nothing observable happens while the bricks collect their samples (no carrier sense, no error source): the SIGNAL symbol and
        // the frame's last burst are positions.
        const uint32_t lts_start = s, sym_start = s + 144u;
        if (sym_start + 80u > NS) break;                                             // the capture ends inside the preamble: nothing more to report
        PROBE_T(_tf);
        fill(lts_start, 224u);
        PROBE_A(_tf, 6);
        FrameCtx* const fx = ctx_of(nfr);
        PROBE_T(_ts);
        const SigOut so = header_section(tabs, lts_start, fx);
        const int r_cfo = uni(so.cfo);
        PROBE_A(_ts, 2);
        PROBE_T(_tc);
        const bool ok = uni((int)so.ok) != 0;
        const uint32_t r_nsym = ok ? (uint32_t)uni((int)so.nsym) : 0u, r_cr = ok ? (uint32_t)uni((int)so.cr) : 0u;
        // the burst that carries the event: the SIGNAL symbol's last one (E_PLCP_HEADER_FAIL), else the last burst of the last data symbol (the Viterbi
        // sub-graph will raise FRAME_OK / CRC32_FAIL); remain_symbols is a ushort (PHY_11a.hpp:405), nsym <= 835
        const uint64_t ev = ok ? (uint64_t)sym_start + 80ull * (r_nsym + 1u) : (uint64_t)sym_start + 80ull;      // end of that burst
        if (ev > NS) break;                                                          // frame runs past the capture: nothing more to report
        // RxThread looks at error_code after the source call: the bursts that call still delivers are consumed, then Flush + Reset drop the queued tail
        // (TMemSamples' queue; not TDownSample44_40's)
        const uint32_t ce = call_end((uint32_t)ev), consumed_to = (uint32_t)ev + (((ce - (uint32_t)ev) >> 2) << 2);
        const uint32_t s_next = A.keep_queue ? consumed_to : ce;
        aim(s_next);                                                                 // (its samples are on their way while the row is written)
        {
            const bool has_row = nfr < A.max_frames;
            const uint32_t slot0 = cd.slot_base + sym_start / 80u;
            if (ok && has_row) {
                // queue the frame for the per-frame kernels: remembered here (lane q holds the q-th frame's code rate and row), handed over when the walk is done or the
                // lanes are used up -- ONE round trip to the counters per capture instead of one per frame in the walk's critical path
                if ((uint32_t)lane == nq) q_job = (r_cr << 28) | (cap_i * A.max_frames + nfr);
                if (++nq == 64u) flush_jobs();
                // its data symbols' slots (k_sym_front / k_sym_back)
                if (A.slot_row) for (uint32_t sy = 1u + (uint32_t)lane; sy <= r_nsym; sy += 64u) A.slot_row[slot0 + sy] = cap_i * A.max_frames + nfr;
            }
            if (lane == 0 && has_row) {
                FrameRow row;
                row.capture = cap_i; row.start_sample = frame_start; row.end_sample = ok ? (uint32_t)ev : consumed_to;
                row.error_code = ok ? 0u : E_PLCP_HEADER_FAIL;                         // 0 = pending: decided by k_finish
                row.rate_kbps = ok ? (uint32_t)uni((int)so.kbps) : 0u; row.length = (uint16_t)(ok ? (uint32_t)uni((int)so.len) : 0u); row.nsym = (uint16_t)r_nsym;
                row.code_rate = (uint16_t)r_cr; row.nbpsc = (uint16_t)(ok ? (uint32_t)uni((int)so.nb) : 0u); row.slot0 = slot0; row.crc32 = 0;
                row.cfo_est = (int16_t)r_cfo; row.cfo_comp = (int16_t)uni(so.cfo_comp); row.sfo_comp = (int16_t)uni(so.sfo_comp);
                row.cfo_tracker = (int16_t)uni(so.cfo_tr); row.sfo_tracker = (int16_t)uni(so.sfo_tr); row.valid = 1;
                row.data_start = sym_start; row.pad[0] = row.pad[1] = row.pad[2] = 0;
                A.frames[(size_t)cap_i * A.max_frames + nfr] = row;
            }
        }
        nfr++;
        s = s_next;
        to_pending = false; cca_detected = 0; cs_reset();                            // BB11aDemodCtx.Reset(), every brick's Reset (fb11a_demod.cpp:64-70)
        PROBE_A(_tc, 7);
    }
    // the capture ends in plain carrier sense: all of it is final
    if (streaming && s == NS && !cca_detected && !sync_high && auto_count == 0 && !to_pending) cont_save(s);
    flush_jobs();
    if (lane == 0) A.nframes[cap_i] = nfr;
    PROBE_ADDK(5);
}

}  // namespace sora

#ifdef SORA_SCAN_PROBE
extern "C" __attribute__((visibility("default"))) int sora_debug_scan_probe(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sora::g_scan_probe), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(sora::g_scan_probe), z, sizeof(z)); }
    return 0;
}
#endif
