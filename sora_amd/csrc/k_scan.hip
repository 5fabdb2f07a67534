// k_scan.hip -- per-capture front end: carrier sense + frame synchronisation + LTS + SIGNAL decode.
//
// One wave64 per capture walks the sample stream exactly the way the reference's source thread does
// (kernel/bb/demod11/fb11a_demod.cpp:29-81 driving CreateDemodGraph11a_40M, fb11ademod_config.hpp:168-233):
// 28 raw samples per TMemSamples::Process() call, 4-sample bursts through
//   TDownSample2 -> TBB11bRxSwitch -> [TDCRemoveEx -> TCCA11a -> TDCEstimator] | [T11aLTS | T11aDataSymbol ...]
// with the error_code test after every source call.  The carrier-sense state machine is inherently serial
// and runs wave-uniform; the per-frame work it triggers (T11aLTS: CFO estimate, frequency shift, FFT<64>,
// channel inverse; the SIGNAL symbol: FFT<64>, equalise, pilot track, BPSK demap, de-interleave, 24-step
// Viterbi, parse) is spread over the 64 lanes.  Output: the frame table, per-frame contexts and the
// symbol-slot map that the batched per-symbol kernels consume.  Data symbols are NOT decoded here.
//
// Capture contract: nsamples is a whole number of source bursts (28 raw samples @40 MHz / 14 @20 MHz) --
// true of every Sora dump (RX_BLOCK = 28 samples, core/inc/_rx_manager.h:96-137); a trailing partial burst
// is ignored (the reference would run it padded with stale pin-queue memory, memsource.hpp:99-107).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace sora {

#ifdef SORA_SCAN_PROBE
__device__ unsigned long long g_scan_probe[16];              // [0..7] ticks per region, [8..15] how often (capture 0 only; tools/probe_scan.py)
#define PROBE_DECL() long long _pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned _pn[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROBE_T0() const long long _t0 = clock64()
#define PROBE_ADD(i) do { _pa[i] += clock64() - _t0; _pn[i]++; } while (0)
#define PROBE_T(n) const long long n = clock64()
#define PROBE_A(n, i) do { _pa[i] += clock64() - n; _pn[i]++; } while (0)
#define PROBE_TK() const long long _tk = clock64()
#define PROBE_ADDK(i) do { _pa[i] += clock64() - _tk; _pn[i]++; if (blockIdx.x == 0 && threadIdx.x == 0) for (int q = 0; q < 8; q++) { g_scan_probe[q] += (unsigned long long)_pa[q]; g_scan_probe[8 + q] += _pn[q]; } } while (0)
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

// ---- T11aLTS on the 144 samples at lts_start (channel_11a.hpp:206-330): CFO estimate, frequency shift, FFT<64>, channel inverse -> *fx.
// Out of line (once per frame): its temporaries and table pointers stay out of the carrier-sense loop's register allocation.  Returns the CFO.
__device__ __noinline__ int lts_section(ScanTabs tabs, const uint32_t* iq_, uint32_t lts_start_, uint32_t STR_, FrameCtx* fx_)
{
    const Tables T = tables_of(tabs);
    const uint32_t* iq = uni_ptr(iq_); FrameCtx* fx = uni_ptr(fx_);
    const uint32_t lts_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)lts_start_), STR = (uint32_t)__builtin_amdgcn_readfirstlane((int)STR_);
    __shared__ uint32_t s_fft[64];
    __shared__ uint32_t s_x[144];
    const int lane = threadIdx.x;
    auto sync = []() { __syncthreads(); };
                        // stage the 144 samples (20 MHz rate) in LDS
                        for (int i = lane; i < 144; i += 64) s_x[i] = iq[lts_start + (uint32_t)i * STR];
                        sync();
                        // x[n] = s_x[8+n]; first 64 are >>1 (rep_shift_right<16>, :216)
                        cpx x1 = sra(unpack(s_x[8 + lane]), 1);
                        cpx x2 = unpack(s_x[8 + 64 + lane]);
                        int re, im; conj_mul32(x2, x1, re, im);                       // FreqOffsetEstimate<16> (dspalg.hpp:226-243)
                        const int sum_re = __builtin_amdgcn_readfirstlane(wave_sum(re >> 5)), sum_im = __builtin_amdgcn_readfirstlane(wave_sum(im >> 5));
                        const int arg = __builtin_amdgcn_readfirstlane(uatan2(T, sum_im, sum_re));
                        const int cfo = arg >> 6;                   // size_t divisor: unsigned division = floor (dspalg.hpp:242)
                        // BuildFrequencyShiftCoeffs<64>(.., 0, CFO_est): ph = lane*cfo (mod 2^16)   (dspalg.hpp:200-208)
                        const cpx fc = rot_coeff(T, w16(lane * cfo));
                        fx->freq[lane] = pack(fc);
                        cpx xs = mul_q15(x1, fc);                                     // FrequencyShift (:120)
                        sync();
                        s_x[lane] = pack(xs);
                        sync();
                        // FFT<64> on lanes 0..15
                        cpx Y[4];
                        {
                            const int e = lane & 15;
                            cpx xin[4];
    #pragma unroll
                            for (int m = 0; m < 4; m++) xin[m] = unpack(s_x[e + 16 * m]);
                            fft64_group(xin, Y, s_fft, e, T, sync);                    // all lanes call (barriers); results identical per 16-lane group
                        }
                        // lane L (0..63) takes bin L: Y of group lane (L&15), register (L>>4)
                        {
                            cpx Yb = (lane >> 4) == 0 ? Y[0] : (lane >> 4) == 1 ? Y[1] : (lane >> 4) == 2 ? Y[2] : Y[3];
                            uint32_t coef = 0;
                            if (!(lane >= 28 && lane < 36)) {                          // _channel_estimation (:125-178)
                                const int e = sqnorm(Yb) >> 8;
                                const cpx L = mk(kLtsSeq[lane] ? 1600 : -1600, 0);
                                int cre, cim; conj_mul32(L, Yb, cre, cim);
                                int rre = 0, rim = 0;
                                if (e != 0) { rre = cre / e; rim = cim / e; }
                                coef = pack(mk(w16(rre), w16(rim)));
                            }
                            fx->chan[lane] = coef;
                        }
                        __threadfence_block();
                        sync();
    return cfo;
}

// ---- the SIGNAL symbol at sym_start: T11aDataSymbol .. T11aViterbiSig .. T11aPLCPParser (PHY_11a.hpp:389-580), lane-parallel; out of line
// like lts_section.  Every field of the result is wave-uniform.
struct SigOut { uint32_t ok, kbps, len, nsym, cr, nb; int cfo_comp, sfo_comp, cfo_tr, sfo_tr; };
__device__ __noinline__ SigOut signal_section(ScanTabs tabs, const uint32_t* iq_, uint32_t sym_start_, uint32_t STR_, const FrameCtx* fx_)
{
    const Tables T = tables_of(tabs);
    const uint32_t* iq = uni_ptr(iq_); const FrameCtx* fx = uni_ptr(fx_);
    const uint32_t sym_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)sym_start_), STR = (uint32_t)__builtin_amdgcn_readfirstlane((int)STR_);
    __shared__ uint32_t s_fft[64];
    __shared__ uint8_t  s_soft[48];
    const int lane = threadIdx.x;
    auto sync = []() { __syncthreads(); };
                            // ---- the SIGNAL symbol: full header chain, lane-parallel
                            const int e = lane & 15;
                            cpx xin[4], Y[4];
    #pragma unroll
                            for (int m = 0; m < 4; m++) {                              // skip CP 8, >>1, x FreqCoeffs (channel_11a.hpp:643-644)
                                const int n = e + 16 * m;
                                cpx x = sra(unpack(iq[sym_start + (uint32_t)(8 + n) * STR]), 1);
                                xin[m] = mul_q15(x, unpack(fx->freq[n]));
                            }
                            fft64_group(xin, Y, s_fft, e, T, sync);
                            cpx Yb = (lane >> 4) == 0 ? Y[0] : (lane >> 4) == 1 ? Y[1] : (lane >> 4) == 2 ? Y[2] : Y[3];
                            cpx eq = mk(0, 0);
                            if (!(lane >= 28 && lane < 36)) {                          // TChannelEqualization (channel_11a.hpp:548-574)
                                int re, im; mul32(Yb, unpack(fx->chan[lane]), re, im);
                                eq = mk(w16(re >> 8), w16(im >> 8));
                            }
                            // TPhaseCompensate with the reset CompCoeffs (0x7fff, 0) (ieee80211facade.hpp:198-206)
                            cpx pc = mul_q15(eq, mk(0x7fff, 0));
                            sync();
                            s_fft[lane] = pack(pc);
                            sync();
                            // _pilot_track (pilot.hpp:166-233), symbol_count = 127 -> PilotSgn[127] = 0
                            cpx p43 = unpack(s_fft[43]), p57 = unpack(s_fft[57]), p7 = unpack(s_fft[7]), p21 = unpack(s_fft[21]);
                            const int th1 = __builtin_amdgcn_readfirstlane(uatan2(T, p43.im, p43.re)), th2 = __builtin_amdgcn_readfirstlane(uatan2(T, p57.im, p57.re));
                            const int th3 = __builtin_amdgcn_readfirstlane(uatan2(T, p7.im, p7.re)),   th4 = __builtin_amdgcn_readfirstlane(uatan2(T, -p21.im, -p21.re));
                            const int avg = w16((th1 + th2 + th3 + th4) / 4);
                            const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
                            const int cfo_tracker = w16(avg >> 2), sfo_tracker = w16(del >> 2);
                            const int cfo_comp = w16(avg + cfo_tracker), sfo_comp = w16(del + sfo_tracker);
                            // T11aDemapBPSK on the rotated carriers -> T11aDeinterleaveBPSK
                            if (lane < 48) {
                                const int bin = carrier_bin(lane);
                                const int cidx = bin < 32 ? bin : bin - 64;            // signed carrier number
                                cpx r = mul_q15(unpack(s_fft[bin]), rot_coeff(T, w16(avg + cidx * del)));
                                int v = r.re >> 4; v = min(max(v, -128), 127);         // demap_limit (demapper.h:141-151)
                                s_soft[lane] = T.demap[(unsigned)v & 0xFF];
                            }
                            sync();
                            uint8_t sa = 0, sb = 0;                                     // de-interleaved soft pair of trellis step t = lane (t < 24)
                            if (lane < 24) { sa = s_soft[T.deint[2 * lane]]; sb = s_soft[T.deint[2 * lane + 1]]; }
                            // ---- Viterbi_sig11 (viterbicore.h:35-261): lane = state
                            const int n = lane;
                            const int r0 = n, r1 = 64 | n;
                            const int cA0 = __popc(r0 & 0155) & 1, cB0 = __popc(r0 & 0117) & 1;
                            const int cA1 = __popc(r1 & 0155) & 1, cB1 = __popc(r1 & 0117) & 1;
                            unsigned m = (n == 0) ? 0u : 0x30u;
                            uint64_t dec[25];                                           // the 64 states' decisions per step: scalars (the loops are unrolled), not an LDS array
                            dec[0] = 0;
    #pragma unroll
                            for (int t = 1; t <= 24; t++) {
                                const int va = __shfl((int)sa, t - 1), vb = __shfl((int)sb, t - 1);
                                const unsigned m0 = (unsigned)__shfl((int)m, n >> 1), m1 = (unsigned)__shfl((int)m, 32 + (n >> 1));
                                const unsigned b0 = (cA0 ? 2 * (7 - va) : 2 * va) + (cB0 ? 2 * (7 - vb) : 2 * vb);
                                const unsigned b1 = (cA1 ? 2 * (7 - va) : 2 * va) + (cB1 ? 2 * (7 - vb) : 2 * vb);
                                const unsigned c0 = (m0 + b0) & 0xFE, c1 = ((m1 + b1) & 0xFF) | 1;
                                m = min(c0, c1);
                                dec[t] = __ballot(m & 1);
                                if ((t & 7) == 0) m = (m - (wave_min(m) & 0xFE)) & 0xFF;
                            }
                            // (the extra normalisation before the trace-back does not change LSBs or the arg-min order)
                            const unsigned key = (m << 8) | ((unsigned)n << 2);
                            const unsigned kmin = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min(key));
                            int pos = (int)((kmin >> 2) & 0x3F) | (int)(((kmin >> 8) & 1) << 6);
                            uint32_t sig = 0;
    #pragma unroll
                            for (int b = 0; b < 24; b++) {
                                // reference emits MSB-first per byte while walking back: bit b of the walk is output bit (23-b)
                                sig |= (uint32_t)((pos >> 6) & 1) << (23 - b);
                                pos = (pos >> 1) & 0x3F;
                                pos |= (int)((dec[23 - b] >> pos) & 1) << 6;
                            }
                            sig = (uint32_t)__builtin_amdgcn_readfirstlane((int)sig) >> 6;     // viterbi.hpp:39 (wave-uniform)
                            // ---- T11aPLCPParser::_parse_plcp (PHY_11a.hpp:548-580)
                            bool ok = true;
                            sig &= 0xFFFFFF;
                            if (sig & 0xFC0010) ok = false;
                            uint32_t par = (sig >> 16) ^ sig; par = (par >> 8) ^ par; par = (par >> 4) ^ par; par = (par >> 2) ^ par; par = (par >> 1) ^ par;
                            if (par & 1) ok = false;
                            uint32_t kbps = 0; int nd = 0, nb = 0, cr = 0;
                            switch (sig & 0xF) {                                        // ieee80211a_cmn.h:97-107, :65-94, :114-149
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
                            O.cfo_comp = cfo_comp; O.sfo_comp = sfo_comp; O.cfo_tr = cfo_tracker; O.sfo_tr = sfo_tracker;
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

__global__ void __launch_bounds__(64, 4) k_scan(ScanArgs A)
{
    const uint32_t cap_i = blockIdx.x;
    if (cap_i >= A.ncaps) return;
    const int lane = threadIdx.x;
    const CapDesc cd = A.caps[cap_i];
    const uint32_t* iq = A.iq + cd.offset;
    const uint32_t STR = A.str, APP = 28 / (2 / STR), BUR = 8 / (2 / STR);
    const uint32_t nunits = (cd.nsamples / APP) * APP;
    const Tables& T = A.T;
    const ScanTabs tabs = { T.uatan2, T.rot, T.demap, T.deint, T.tw64, T.tw16 };
    // the fills a call used to get from a kernel of their own in front of this one (kernels.h ScanArgs): nobody reads any of these words before this kernel has ended
    if (A.own_slots && A.slot_row) for (uint32_t i = (uint32_t)lane; i < cd.nslots; i += 64u) A.slot_row[cd.slot_base + i] = 0xFFFFFFFFu;
    if (cap_i == 0) {
        if (A.zero_a) for (uint32_t i = (uint32_t)lane; i < A.nzero_a; i += 64u) A.zero_a[i] = 0u;
        if (A.zero_b) for (uint32_t i = (uint32_t)lane; i < A.nzero_b; i += 64u) A.zero_b[i] = 0u;
    }

    // ---- carrier-sense state (cca.hpp:126-158), wave-uniform
    uint32_t Hv = 0;                         // sample_his in TIME ORDER, one packed sample per lane (lane & 15, oldest = 0): 4 bursts of 4, already >>2
    Acc4 ac_re, ac_im, energy;
    uint32_t auto_count = 0, sense_count = 0, high_count = 0; int sync_high = 0, peak_corr = 0, peak_index = 0;
    uint32_t dc_cnt = 8; int sum_dc_re = 0, sum_dc_im = 0;            // TDCEstimator (dc.hpp:92-166); all 4 lanes of the vcs are equal
    int dc_re = 0, dc_im = 0;                                          // CF_VecDC (survives frame resets)
    // ---- context
    uint32_t error_code = 0; int cca_detected = 0, symbol_is_data = 0, plcp_is_data = 0;
    uint32_t lts_n = 0, sym_n = 0, lts_start = 0, sym_start = 0, frame_start = 0;
    uint32_t remain_symbols = 0, sym_idx = 0;
    uint32_t nfr = 0;
    // the frame row being assembled (wave-uniform scalars; written out once)
    uint32_t r_start = 0, r_end = 0, r_rate = 0, r_slot0 = 0, r_data_start = 0, r_len = 0, r_nsym = 0, r_cr = 0, r_nb = 0;
    int r_cfo = 0, r_cfo_comp = 0, r_sfo_comp = 0, r_cfo_tr = 0, r_sfo_tr = 0;

    auto cs_reset = [&]() {
        Hv = 0;
        acc_clear(ac_re); acc_clear(ac_im); acc_clear(energy);
        auto_count = sense_count = high_count = 0; sync_high = 0; peak_corr = 0; peak_index = 0;
        dc_cnt = 8; sum_dc_re = sum_dc_im = 0;
    };
    auto frame_reset = [&]() {
        error_code = 0; cca_detected = 0; symbol_is_data = 0; plcp_is_data = 0;
        lts_n = sym_n = 0; remain_symbols = 0; sym_idx = 0;
        cs_reset();
    };
    cs_reset();
    auto sync = []() { __syncthreads(); };

    // ---- stream continuation (sora_rx_set_stream_mode; TRxStream hands the graph an endless stream, rxstream.hpp:34-66: the DC estimate of
    // dc.hpp:92-166 integrates for ever, the carrier-sense windows and counters carry over from read to read).  A RESUME POINT is a position
    // where a burst boundary falls on a source-call boundary while the graph is in plain carrier sense (no detection under way, no event
    // pending): everything the graph knows there is the record below.  The capture's last resume point is published (A.consumed) and its
    // record kept; the next call's capture k starts AT that point of the stream and the record is its initial state -- so what the graph
    // reports from there on is what it reports on the uncut stream, and a frame cut by the end of a capture is simply found again.
    // (one flag lives through the loop; the pointers are re-derived where they are used)
    const bool streaming = A.cont != nullptr;
    auto cont_save = [&](uint32_t at) {
        cont_store(A.cont + (size_t)cap_i * kContWords, A.consumed + cap_i, Hv, ac_re.Z, ac_im.Z, energy.Z, ac_re.reg, ac_im.reg, energy.reg, sense_count,
                high_count, peak_corr, peak_index,
                   dc_cnt, sum_dc_re, sum_dc_im, dc_re, dc_im, at);
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

    // GetCrossCorrelation (cca.hpp:202-218) for pattern p: the reference starts at the oldest burst, i.e. h[0]
    auto cross_corr = [&](int p) -> int {
        int sre[4] = {0, 0, 0, 0}, sim[4] = {0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < 4; v++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int re, im; conj_mul32(unpack(T.sts[p * 16 + 4 * v + e]), unpack((uint32_t)__builtin_amdgcn_readlane((int)Hv, 4 * v + e)), re, im);
                sre[e] = (int)((unsigned)sre[e] + (unsigned)re); sim[e] = (int)((unsigned)sim[e] + (unsigned)im);
            }
        }
        int r = (int)((unsigned)sre[0] + (unsigned)sre[1] + (unsigned)sre[2] + (unsigned)sre[3]);
        int i = (int)((unsigned)sim[0] + (unsigned)sim[1] + (unsigned)sim[2] + (unsigned)sim[3]);
        return abs(r) + abs(i);
    };
    auto his_push = [&](const cpx (&v)[4]) {                                    // drop the oldest burst, append v (wave-uniform values)
        const uint32_t l = (uint32_t)lane & 15u;
        const uint32_t fresh = (l & 3u) == 0 ? pack(v[0]) : (l & 3u) == 1 ? pack(v[1]) : (l & 3u) == 2 ? pack(v[2]) : pack(v[3]);
        const uint32_t keep = (uint32_t)__shfl((int)Hv, (int)(((uint32_t)lane & 48u) | ((l + 4u) & 15u)));
        Hv = l < 12u ? keep : fresh;
    };

    // Carrier sense reads the stream 4 samples at a time, wave-uniformly: stage 64 consecutive units per
    // coalesced 256-byte load (one per lane) and hand them out with v_readlane instead of paying one global
    // round trip per burst.  (Keeping the following 64 units in flight in a second register was measured in round 4: the kernel
    // alone stays at 0.102-0.108 ms -- the loads are L2 hits behind the staging kernel, not what the wave waits for -- so it is not done.)
    uint32_t win_base = 0xFFFFFFFFu, win = 0;
    auto stage = [&](uint32_t u) { win_base = u; win = (u + (uint32_t)lane < nunits) ? iq[u + (uint32_t)lane] : 0u; };
    auto sample = [&](uint32_t u) -> uint32_t {
        if (u - win_base >= 64u) stage(u);
        return (uint32_t)__builtin_amdgcn_readlane((int)win, (int)(u - win_base));
    };

    uint32_t vpos = 0;                              // next burst start, in queue units
    // ---- idle fast path.  While no carrier is in sight (energy/auto-correlation test false, cca.hpp:386-437) a burst only
    // feeds the sliding sums, the history, the DC estimator and the time-out counter.  Up to 8 bursts are then taken at
    // once, one sample per lane (lane = 4 b + e): the per-sample products run once instead of per burst per lane, the
    // per-burst bookkeeping (sliding-sum updates, the test, counters) stays scalar.  The group contains a DC update at
    // most at its last burst (the estimate changes what the next burst sees), stays inside the source call when the
    // carrier-sense time-out will fire in it (error_code is examined, and the brick reset, at the end of that call), and
    // stops in front of the first burst whose test is true -- that burst goes through the full path below.
    auto fast_idle = [&](uint32_t K) -> uint32_t {
        if (vpos - win_base + K * BUR > 64u || win_base == 0xFFFFFFFFu) stage(vpos);
        const uint32_t l = (uint32_t)lane & 31u, hi = (uint32_t)lane & 32u;    // lanes 32..63 mirror lanes 0..31
        const uint32_t raw = (uint32_t)__shfl((int)win, (int)(vpos - win_base + (l >> 2) * BUR + (l & 3u) * STR));
        const cpx x = unpack(raw);
        const cpx pi = mk(w16(x.re - dc_re), w16(x.im - dc_im));                // TDCRemoveEx
        const cpx pii = sra(pi, 2);
        const uint32_t ppk = pack(pii);
        const uint32_t prev = other_row_even(ppk);                              // the sample 16 lanes below (used by the second row of each half only)
        int re, im; conj_mul32(pii, unpack(l < 16u ? Hv : prev), re, im);       // against the sample 16 earlier
        unsigned vr = (unsigned)(re >> 4), vi = (unsigned)(im >> 4), ve = (unsigned)(sqnorm(pii) >> 4);
        unsigned dr = (unsigned)(pi.re >> 5), di = (unsigned)(pi.im >> 5);     // TDCEstimator terms
        vr = quad_sum(vr); vi = quad_sum(vi); ve = quad_sum(ve);                // a burst's four samples sit in one quad
        dr = quad_sum(dr); di = quad_sum(di);
        // ---- the K bursts' sliding sums at once.  After burst b the accumulator holds reg + sum_{i<=b} (d_i - z_i), z = the value that leaves the
        // 4-element window: the old elements for b < 4, d_{b-4} after that.  One prefix sum per stream over the burst groups, the reference's test
        // (cca.hpp:386-437) in every group, and the first burst whose test is true ends the pass.
        const bool second_row = (l & 16u) != 0u;
        // (every lane takes part: a cross-lane read inside a lane-dependent branch would find its source lanes switched off)
        const uint32_t pr = other_row_even(vr), pim = other_row_even(vi), pe = other_row_even(ve);
        const uint32_t zr = second_row ? pr : ac_re.Z, zi = second_row ? pim : ac_im.Z, ze = second_row ? pe : energy.Z;
        const uint32_t Rr = (uint32_t)ac_re.reg + group_scan(vr - zr), Ri = (uint32_t)ac_im.reg + group_scan(vi - zi), Re = (uint32_t)energy.reg + group_scan(ve - ze);
        const int iAuto_v = abs((int)Rr) + abs((int)Ri), iEnergy_v = (int)Re;
        const bool carrier = iEnergy_v > (int)A.thr && iAuto_v >= iEnergy_v - (iEnergy_v >> 3);
        const uint32_t hits = (uint32_t)__ballot(carrier) & 0x11111111u & (K >= 8u ? 0xFFFFFFFFu : ((1u << (4u * K)) - 1u));
        const uint32_t done = hits ? (uint32_t)__builtin_ctz(hits) >> 2 : K;
        if (done) {
            const int last = (int)(4u * (done - 1u));
            ac_re.reg = __builtin_amdgcn_readlane((int)Rr, last); ac_im.reg = __builtin_amdgcn_readlane((int)Ri, last); energy.reg = __builtin_amdgcn_readlane((int)Re, last);
            {   // the windows <- the last four of {old window, d_0 .. d_{done-1}} (element g in lanes 4 g .. 4 g + 3 of every row)
                const uint32_t a16 = (uint32_t)lane & 15u, src = a16 + 4u * done;
                const int from_old = (int)(((uint32_t)lane & 48u) | (src & 15u)), from_new = (int)(hi | ((src - 16u) & 31u));
                const bool old = src < 16u;
                const uint32_t kr = (uint32_t)__shfl((int)ac_re.Z, from_old), fr = (uint32_t)__shfl((int)vr, from_new);
                const uint32_t ki = (uint32_t)__shfl((int)ac_im.Z, from_old), fi = (uint32_t)__shfl((int)vi, from_new);
                const uint32_t ke = (uint32_t)__shfl((int)energy.Z, from_old), fe = (uint32_t)__shfl((int)ve, from_new);
                ac_re.Z = old ? kr : fr; ac_im.Z = old ? ki : fi; energy.Z = old ? ke : fe;
            }
            auto_count = 0; sense_count += 4u * done;
            // TDCEstimator (dc.hpp:92-166): 16-bit running sums of the bursts taken; the estimate moves at most at the pass's last burst (K <= dc_cnt + 1)
            sum_dc_re = w16(sum_dc_re + __builtin_amdgcn_readlane((int)group_scan(dr), last));
            sum_dc_im = w16(sum_dc_im + __builtin_amdgcn_readlane((int)group_scan(di), last));
            if (done == dc_cnt + 1u) {
                dc_re = w16(dc_re + (sum_dc_re >> 2)); dc_im = w16(dc_im + (sum_dc_im >> 2));
                dc_cnt = 7; sum_dc_re = sum_dc_im = 0;
            } else dc_cnt -= done;
            if (sense_count >= 84) error_code = E_CS_TIMEOUT;                   // cca.hpp:433-437
        }
        if (done) {                                                             // history <- its last 16 samples
            const uint32_t a16 = (uint32_t)lane & 15u, src = a16 + 4u * done;   // index in {old history 0..15, new samples 16..47}
            const uint32_t keep = (uint32_t)__shfl((int)Hv, (int)(((uint32_t)lane & 48u) | (src & 15u)));
            const uint32_t fresh = (uint32_t)__shfl((int)ppk, (int)(hi | ((src - 16u) & 31u)));
            Hv = src < 16u ? keep : fresh;
            vpos += done * BUR;
        }
        return done;
    };

    // ---- the same for the phase between establish_sync and the end of the short training symbols (check_sync,
    // cca.hpp:245-265): a burst only enters the history; every fourth one the history is correlated with the winning
    // STS pattern.  Up to 4 bursts per pass, one sample per lane; the correlation runs one tap per lane.
    auto fast_sync = [&](uint32_t K) {                                          // K = bursts taken (1..4), at most up to the next check
        if (vpos - win_base + 4u * BUR > 64u || win_base == 0xFFFFFFFFu) stage(vpos);
        const uint32_t l = (uint32_t)lane & 15u;
        const uint32_t raw = (uint32_t)__shfl((int)win, (int)(vpos - win_base + (l >> 2) * BUR + (l & 3u) * STR));
        const cpx x = unpack(raw);
        const cpx pi = mk(w16(x.re - dc_re), w16(x.im - dc_im));
        const uint32_t src = ((uint32_t)lane & 48u) | ((l + 4u * K) & 15u);
        const uint32_t keep = (uint32_t)__shfl((int)Hv, (int)src), fresh = (uint32_t)__shfl((int)pack(sra(pi, 2)), (int)src);
        Hv = (l + 4u * K < 16u) ? keep : fresh;
        const uint32_t last_v = vpos + (K - 1u) * BUR;                          // the burst that may carry the check
        vpos += K * BUR;
        high_count += K;
        if (high_count % 4 == 0) {
            int re, im; conj_mul32(unpack(T.sts[peak_index * 16 + (int)l]), unpack(Hv), re, im);   // GetCrossCorrelation, one tap per lane
            unsigned ur = (unsigned)re, ui = (unsigned)im;
            ur = row_sum(ur); ui = row_sum(ui);
            const int corr = abs(__builtin_amdgcn_readfirstlane((int)ur)) + abs(__builtin_amdgcn_readfirstlane((int)ui));
            if (corr < (peak_corr >> 1)) {
                if (high_count > 8) { cca_detected = 1; frame_start = last_v / STR + 4; }
                else {
                    sync_high = 0; sense_count = 0;
                    // the burst that dropped the lock also feeds TDCEstimator (dc.hpp:92-166)
                    unsigned dr = (unsigned)(pi.re >> 5), di = (unsigned)(pi.im >> 5);
                    dr = quad_sum(dr); di = quad_sum(di);
                    sum_dc_re = w16(sum_dc_re + w16(__builtin_amdgcn_readlane((int)dr, (int)(4 * (K - 1)))));
                    sum_dc_im = w16(sum_dc_im + w16(__builtin_amdgcn_readlane((int)di, (int)(4 * (K - 1)))));
                    if (dc_cnt == 0) {
                        dc_re = w16(dc_re + (sum_dc_re >> 2)); dc_im = w16(dc_im + (sum_dc_im >> 2));
                        dc_cnt = 8; sum_dc_re = sum_dc_im = 0;
                    }
                    dc_cnt--;
                }
            } else if (corr > peak_corr) peak_corr = corr;
        }
    };

    const uint32_t nchunks = nunits / APP;
    PROBE_DECL();
    PROBE_TK();
    // Every frame of the capture is found and counted, as RxThread reports every frame; those past the row limit get no row and
    // no decode job (their per-frame context goes to the capture's spare FrameCtx), and the host flags the capture's last row.
    auto ctx_of = [&](uint32_t k) { return A.fctx + (k < A.max_frames ? (size_t)cap_i * A.max_frames + k : (size_t)A.nrows + cap_i); };
    for (uint32_t c = 0; c < nchunks; c++) {
        const uint32_t avail_end = (c + 1) * APP;
        while (vpos + BUR <= avail_end) {
            // (inside this loop the only multiple of APP vpos can be is the chunk's start)
            if (streaming && vpos == avail_end - APP && !cca_detected && !sync_high && auto_count == 0 && error_code == 0) cont_save(vpos);
            if (!cca_detected && !sync_high && auto_count == 0) {
                // bursts until the carrier-sense time-out is raised; if that is near, stay inside this source call
                const uint32_t to_timeout = sense_count >= 84 ? 0u : (84u - sense_count + 3u) / 4u;
                uint32_t room = to_timeout <= 8u ? (avail_end - vpos) / BUR : (nunits - vpos) / BUR;
                // a pass ends at the next resume point (burst boundary = source-call boundary), so that it is seen:
                if (streaming) {
                    // in units of half a burst vpos sits m past the chunk's start and a source call is 7; j bursts further it is m + 2 j: j = -m / 2 = 3 m (mod 7), 0 -> 7
                    const uint32_t m7 = (((vpos + APP - (avail_end - APP)) / (BUR / 2u)) % 7u);
                    const uint32_t j = (3u * m7) % 7u;
                    room = min(room, j ? j : 7u);
                }
                PROBE_T0();
                const uint32_t took = fast_idle(min(min(room, dc_cnt + 1u), 8u));
                PROBE_ADD(3);
                if (took) continue;
            }
            if (!cca_detected && sync_high) {
                PROBE_T0();
                fast_sync(min((avail_end - vpos) / BUR, 4u - high_count % 4u));
                PROBE_ADD(4);
                continue;
            }
            const uint32_t pos20 = vpos / STR;
            PROBE_T(_tb);
            if (!cca_detected) {
                PROBE_T0();
                // ================= TDCRemoveEx<4> -> TCCA11a -> TDCEstimator (power_clear path)
                uint32_t raw[4]; cpx pi[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { raw[e] = sample(vpos + e * STR); cpx x = unpack(raw[e]); pi[e] = mk(w16(x.re - dc_re), w16(x.im - dc_im)); }
                if (!sync_high) {
                    cpx pii[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) pii[e] = sra(pi[e], 2);
                    int sr = 0, si = 0, se = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        int re, im; conj_mul32(pii[e], unpack((uint32_t)__builtin_amdgcn_readlane((int)Hv, e)), re, im);   // sample_his.First(): 16 samples ago
                        sr = (int)((unsigned)sr + (unsigned)(re >> 4)); si = (int)((unsigned)si + (unsigned)(im >> 4));
                        se = (int)((unsigned)se + (unsigned)(sqnorm(pii[e]) >> 4));
                    }
                    acc_push(ac_re, sr); acc_push(ac_im, si);
                    const int iAuto = abs(ac_re.reg) + abs(ac_im.reg);
                    acc_push(energy, se);
                    const int iEnergy = energy.reg;
                    his_push(pii);
                    sense_count += 4;
                    if (iEnergy > (int)A.thr && iAuto >= iEnergy - (iEnergy >> 3)) {
                        auto_count++; sense_count = 0;
                        if (auto_count >= 4) {
                            // establish_sync (cca.hpp:220-243): lanes 0..15 take one pattern each
                            int corr = cross_corr(lane & 15);
                            int sum_corr = 0, best = 0, best_i = 0;
#pragma unroll
                            for (int p = 0; p < 16; p++) {
                                const int cp = __builtin_amdgcn_readlane(corr, p);          // wave-uniform from here on
                                if (cp > best) { best = cp; best_i = p; }
                                sum_corr += cp;
                            }
                            peak_corr = best; if (best > 0) peak_index = best_i;
                            if (peak_corr > (sum_corr >> 3)) {
                                sync_high = 1; high_count = 0;
                                if (peak_index > 3) { high_count = (uint32_t)peak_index / 4; peak_index &= 3; }
                            }
                        }
                    } else {
                        auto_count = 0;
                    }
                }
                if (!sync_high) {
                    int hr = 0, hi = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) { hr = w16(hr + (pi[e].re >> 5)); hi = w16(hi + (pi[e].im >> 5)); }
                    sum_dc_re = w16(sum_dc_re + hr); sum_dc_im = w16(sum_dc_im + hi);
                    if (dc_cnt == 0) {
                        dc_re = w16(dc_re + (sum_dc_re >> 2)); dc_im = w16(dc_im + (sum_dc_im >> 2));
                        dc_cnt = 8; sum_dc_re = sum_dc_im = 0;
                    }
                    dc_cnt--;
                }
                if (sense_count >= 84 && !sync_high) error_code = E_CS_TIMEOUT;     // cca.hpp:433-437
                PROBE_ADD(0);
            } else if (!symbol_is_data) {
                // ================= T11aLTS: IPORT COMPLEX16 x 144 (channel_11a.hpp:206-229)
                if (lts_n == 0) {
                    // Nothing observable happens while the brick collects its 144 samples (no carrier sense, no error source): go straight
                    // to the burst that completes them instead of counting 35 bursts.
                    lts_start = vpos;
                    const uint32_t last_v = vpos + 35u * BUR;
                    if (last_v + BUR > nunits) { c = nchunks; vpos = nunits; break; }       // the capture ends inside the LTS: nothing more to report
                    lts_n = 140; vpos = last_v;
                    // the for-loop increment lands on the chunk that delivers that burst
                    c = (vpos + BUR + APP - 1) / APP - 2;
                    break;
                }
                lts_n += 4;
                if (lts_n == 144) {
                    PROBE_T0();
                    lts_n = 0; symbol_is_data = 1;
                    r_cfo = __builtin_amdgcn_readfirstlane(lts_section(tabs, iq, lts_start, STR, ctx_of(nfr)));
                    PROBE_ADD(1);
                }
            } else {
                // ================= T11aDataSymbol: IPORT COMPLEX16 x 80 (PHY_11a.hpp:389-428)
                if (sym_n == 0) sym_start = vpos;
                // likewise: to the symbol's 20th burst (not once an event is pending: the rest of that source call still counts bursts)
                if (sym_n == 0 && error_code == 0) {
                    const uint32_t last_v = vpos + 19u * BUR;
                    if (last_v + BUR > nunits) { c = nchunks; vpos = nunits; break; }
                    sym_n = 76; vpos = last_v;
                    c = (vpos + BUR + APP - 1) / APP - 2;
                    break;
                }
                sym_n += 4;
                if (sym_n == 80) {
                    sym_n = 0;
                    if (sym_idx == 0) {
                        PROBE_T0();
                        // ---- the SIGNAL symbol: full header chain, lane-parallel (signal_section, out of line)
                        const SigOut so = signal_section(tabs, iq, sym_start, STR, ctx_of(nfr));
                        auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
                        r_start = frame_start; r_slot0 = cd.slot_base + (sym_start / STR) / 80; r_data_start = sym_start / STR;
                        r_cfo_comp = uni(so.cfo_comp); r_sfo_comp = uni(so.sfo_comp); r_cfo_tr = uni(so.cfo_tr); r_sfo_tr = uni(so.sfo_tr);
                        if (uni((int)so.ok)) {
                            r_rate = (uint32_t)uni((int)so.kbps); r_len = (uint32_t)uni((int)so.len); r_nsym = (uint32_t)uni((int)so.nsym);
                            r_cr = (uint32_t)uni((int)so.cr); r_nb = (uint32_t)uni((int)so.nb);
                            remain_symbols = r_nsym + 1; plcp_is_data = 1;
                        } else {
                            r_rate = 0; r_len = 0; r_nsym = 0; r_cr = 0; r_nb = 0;
                            error_code = E_PLCP_HEADER_FAIL;
                        }
                        PROBE_ADD(2);
                    }
                    sym_idx++;
                    remain_symbols = (remain_symbols - 1) & 0xFFFF;                // ushort (PHY_11a.hpp:405)
                    if (sym_idx == 1 && plcp_is_data && remain_symbols > 1) {
                        // Nothing observable happens between here and the last burst of the frame (no carrier sense,
                        // no error source): jump straight to it instead of counting ~20 bursts per symbol.
                        const uint64_t last_v = (uint64_t)sym_start + 80ull * STR * (remain_symbols + 1) - BUR;
                        if (last_v + BUR > nunits) { c = nchunks; vpos = nunits; break; }   // frame runs past the capture: nothing more to report
                        sym_idx += remain_symbols - 1; remain_symbols = 1; sym_n = 76; sym_start = (uint32_t)(last_v + BUR) - 80u * STR;
                        vpos = (uint32_t)last_v;
                        // the for-loop increment lands on the chunk that delivers that burst
                        c = (vpos + BUR + APP - 1) / APP - 2;
                        break;
                    }
                    if (remain_symbols == 0 && plcp_is_data) {
                        // all data symbols are in: the Viterbi sub-graph will raise FRAME_OK / CRC32_FAIL
                        r_end = pos20 + 4;
                        error_code = E_FRAME_OK;                                    // provisional: "frame complete"
                    }
                }
            }
            vpos += BUR;
            PROBE_A(_tb, 6);
        }
        // ---- RxThread bookkeeping after each source call (fb11a_demod.cpp:37-71)
        PROBE_T(_tc);
        if (error_code != 0) {
            if (error_code == E_CS_TIMEOUT) {
                error_code = 0; cca_detected = 0; cs_reset();
            } else {
                const bool plcp_fail = error_code == E_PLCP_HEADER_FAIL;
                const bool has_row = nfr < A.max_frames;
                if (plcp_fail) r_end = vpos / STR;
                else if (has_row) {
                    // queue the frame for the per-frame kernels
                    if (lane == 0) A.joblist[(size_t)r_cr * A.nrows + atomicAdd(A.njobs + r_cr, 1u)] = cap_i * A.max_frames + nfr;
                    // its data symbols' slots (k_sym_front / k_sym_back)
                    if (A.slot_row) for (uint32_t sy = 1u + (uint32_t)lane; sy <= r_nsym; sy += 64u) A.slot_row[r_slot0 + sy] = cap_i * A.max_frames + nfr;
                }
                if (lane == 0 && has_row) {
                    FrameRow row;
                    row.capture = cap_i; row.start_sample = r_start; row.end_sample = r_end;
                    row.error_code = plcp_fail ? E_PLCP_HEADER_FAIL : 0u;                 // 0 = pending: decided by k_finish
                    row.rate_kbps = r_rate; row.length = (uint16_t)r_len; row.nsym = (uint16_t)r_nsym;
                    row.code_rate = (uint16_t)r_cr; row.nbpsc = (uint16_t)r_nb; row.slot0 = r_slot0; row.crc32 = 0;
                    row.cfo_est = (int16_t)r_cfo; row.cfo_comp = (int16_t)r_cfo_comp; row.sfo_comp = (int16_t)r_sfo_comp;
                    row.cfo_tracker = (int16_t)r_cfo_tr; row.sfo_tracker = (int16_t)r_sfo_tr; row.valid = 1;
                    row.data_start = r_data_start; row.pad[0] = row.pad[1] = row.pad[2] = 0;
                    A.frames[(size_t)cap_i * A.max_frames + nfr] = row;
                }
                nfr++;
                // Flush + Reset drop the queued tail (TMemSamples' queue; not TDownSample44_40's)
                if (!A.keep_queue) vpos = avail_end;
                frame_reset();
            }
        }
        PROBE_A(_tc, 7);
        // Nothing pending and the next burst lies source calls ahead (an idle pass takes up to eight bursts, a source call holds three and a half): go straight to the call
        // that delivers it -- the calls in between would find no burst to run and no event to report.  (x / APP for APP = 14 or 28 as a multiply: exact below 2^31.)
        if (error_code == 0 && !streaming && vpos + BUR > avail_end + APP && vpos < 0x7FFFFF00u) {
            const uint32_t calls = (uint32_t)(((uint64_t)(vpos + BUR + APP - 1u) * 0x92492493ull) >> (STR == 1u ? 35 : 36));   // ceil((vpos + BUR) / APP)
            c = calls - 2u;                                                      // (the for-loop's increment lands on call `calls - 1`, whose avail_end = calls x APP)
        }
    }
    // the capture ends in plain carrier sense: all of it is final
    if (streaming && vpos == nunits && !cca_detected && !sync_high && auto_count == 0 && error_code == 0) cont_save(vpos);
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
