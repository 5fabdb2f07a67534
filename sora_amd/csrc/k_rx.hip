// k_rx.hip -- batched per-symbol / per-frame kernels of the 802.11a receive path (gfx950).
//
//   k_sym_front   T11aDataSymbol -> TFreqCompensation -> TFFT64 -> TChannelEqualization   (parallel over symbols)
//   k_track       TPhaseCompensate/TPilotTrack loop-carried state on the 4 pilots          (serial per frame)
//   k_demap       TPhaseCompensate + TPilotTrack rotation + T11aDemap<N> + T11aDeinterleave (parallel over symbols)
//   k_viterbi<CR> T11aViterbi<5000*8,48,256,24>: 64-state ACS, wave64 = 64 states          (one wave per frame)
//   k_traceback   TViterbiCore::Traceback for every window of the schedule                 (one thread per window)
//   k_finish      T11aDesc + TBB11aFrameSink (descramble, CRC-32, FRAME_OK / CRC32_FAIL)   (one thread per frame)
// plus the stand-alone stage kernels behind the per-stage C entry points.
#include <hip/hip_runtime.h>
#include <utility>
#include "kernels.h"

namespace sora {

__device__ __constant__ uint8_t kPilotSgn[128] = {        // pilot.hpp:10-28: 1 <=> polarity -1
    0,0,0,1,1,1,0,1, 1,1,1,0,0,1,0,1, 1,0,0,1,0,0,1,0, 0,0,0,0,0,1,0,0,
    0,1,0,0,1,1,0,0, 0,1,0,1,1,1,0,1, 0,1,1,0,1,1,0,0, 0,0,0,1,1,0,0,1,
    1,0,1,0,1,0,0,1, 1,1,0,0,1,1,1,1, 0,1,1,0,1,0,0,0, 0,1,0,1,0,1,0,1,
    1,1,1,1,0,1,0,0, 1,0,1,0,0,0,1,1, 0,1,1,1,0,0,0,1, 1,1,1,1,1,1,0,0 };

__device__ __forceinline__ int carrier_bin48(int k)       // demap order -26..-1,+1..+26 without pilots (demapper11a.hpp:20-37)
{
    if (k < 24) { int b = 38 + k; if (b >= 43) b++; if (b >= 57) b++; return b; }
    int b = 1 + (k - 24); if (b >= 7) b++; if (b >= 21) b++; return b;
}

// ------------------------------------------------------------------------------------------------
// k_sym_front: 16 lanes per OFDM symbol, 16 symbols per 256-thread block.
//   algorithmic bytes per symbol: 256 read (64 samples) + 256 written (64 equalised bins)
__global__ void __launch_bounds__(256) k_sym_front(RxArgs A)
{
    __shared__ uint32_t s_all[16][64];
    const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
    const uint32_t slot = blockIdx.x * 16 + g;
    int fr = -1, sym = 0;
    if (slot < A.total_slots) { fr = A.slot_frame[slot]; sym = A.slot_sym[slot]; }
    const bool active = fr >= 0 && sym > 0;
    cpx x[4], Y[4];
    const FrameCtx* fx = A.fctx + (active ? fr : 0);
    if (active) {
        const FrameRow& r = A.frames[fr];
        const uint32_t* iq = A.iq + A.caps[r.capture].offset;
        const uint32_t p0 = r.data_start + 80u * (uint32_t)sym + 8u;           // skip_cp = 8 (PHY_11a.hpp:365,394)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int n = e + 16 * m;
            cpx s = sra(unpack(iq[(size_t)(p0 + n) * A.str]), 1);             // TFreqCompensation: >>1 (channel_11a.hpp:643)
            x[m] = mul_q15(s, unpack(fx->freq[n]));                            //   x FreqCoeffs (:644)
        }
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = mk(0, 0);
    }
    fft64_group(x, Y, s_all[g], e, A.T, []() { __syncthreads(); });           // TFFT64
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; q++) {                                          // TChannelEqualization (channel_11a.hpp:548-574)
            const int bin = e + 16 * q;
            cpx o = mk(0, 0);
            if (!(bin >= 28 && bin < 36)) {
                int re, im; mul32(Y[q], unpack(fx->chan[bin]), re, im);
                o = mk(w16(re >> 8), w16(im >> 8));
            }
            A.eq[(size_t)slot * 64 + bin] = pack(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_track: the loop-carried part of TPhaseCompensate/TPilotTrack (freqoffset.hpp:28-30, pilot.hpp:166-233):
// only the 4 pilot bins take part.  One thread per frame, sequential over its data symbols.
__global__ void __launch_bounds__(64) k_track(RxArgs A)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *A.njobs) return;
    const uint32_t f = A.joblist[j];
    const FrameRow r = A.frames[f];
    VitJob J; J.pad = 0;
    J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftPerSlot * 2u; J.nsoft = (uint32_t)r.nsym * 48u * r.nbpsc; J.length = r.length;
    J.dec_off = r.slot0 * (uint32_t)kDecPerSlot; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate;
    A.jobs[j] = J;
    const Tables& T = A.T;
    int cfo_comp = r.cfo_comp, sfo_comp = r.sfo_comp, cfo_tr = r.cfo_tracker, sfo_tr = r.sfo_tracker;
    unsigned symbol_count = 0;                                                 // 127 -> 0 after the SIGNAL symbol
    for (uint32_t s = 1; s <= r.nsym; s++) {
        const uint32_t slot = r.slot0 + s;
        const uint32_t* eq = A.eq + (size_t)slot * 64;
        // CompCoeffs at carriers -21, -7, +7, +21: th = CFO_comp + c*SFO_comp (pilot.hpp:138-164)
        cpx p43 = mul_q15(unpack(eq[43]), rot_coeff(T, w16(cfo_comp - 21 * sfo_comp)));
        cpx p57 = mul_q15(unpack(eq[57]), rot_coeff(T, w16(cfo_comp - 7 * sfo_comp)));
        cpx p7  = mul_q15(unpack(eq[7]),  rot_coeff(T, w16(cfo_comp + 7 * sfo_comp)));
        cpx p21 = mul_q15(unpack(eq[21]), rot_coeff(T, w16(cfo_comp + 21 * sfo_comp)));
        int th1 = uatan2(T, p43.im, p43.re), th2 = uatan2(T, p57.im, p57.re);
        int th3 = uatan2(T, p7.im, p7.re),   th4 = uatan2(T, -p21.im, -p21.re);
        if (kPilotSgn[symbol_count]) { th1 = w16(th1 + 0x8000); th2 = w16(th2 + 0x8000); th3 = w16(th3 + 0x8000); th4 = w16(th4 + 0x8000); }
        symbol_count++; if (symbol_count >= 127) symbol_count = 0;
        const int avg = w16((th1 + th2 + th3 + th4) / 4);
        const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
        TrackRec tr; tr.cfo_comp = (int16_t)cfo_comp; tr.sfo_comp = (int16_t)sfo_comp; tr.avg = (int16_t)avg; tr.del = (int16_t)del;
        A.track[slot] = tr;
        cfo_tr = w16(cfo_tr + (avg >> 2)); sfo_tr = w16(sfo_tr + (del >> 2));
        cfo_comp = w16(cfo_comp + avg + cfo_tr); sfo_comp = w16(sfo_comp + del + sfo_tr);
    }
}

// ------------------------------------------------------------------------------------------------
// k_demap: one wave per data symbol (4 symbols per block): 48 data carriers -> phase compensation ->
// pilot rotation -> soft demap -> de-interleave -> contiguous per-frame soft stream.
//   algorithmic bytes per symbol: 256 read + N_CBPS written
__global__ void __launch_bounds__(256) k_demap(RxArgs A)
{
    __shared__ uint8_t s_soft[4][288];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 4 + w;
    int fr = -1, sym = 0;
    if (slot < A.total_slots) { fr = A.slot_frame[slot]; sym = A.slot_sym[slot]; }
    const bool active = fr >= 0 && sym > 0;
    const Tables& T = A.T;
    int nb = 1; uint32_t slot0 = 0;
    if (active) {
        const FrameRow& r = A.frames[fr];
        nb = r.nbpsc; slot0 = r.slot0;
        if (lane < 48) {
            const TrackRec tr = A.track[slot];
            const int bin = carrier_bin48(lane);
            const int c = bin < 32 ? bin : bin - 64;
            cpx v = unpack(A.eq[(size_t)slot * 64 + bin]);
            v = mul_q15(v, rot_coeff(T, w16(tr.cfo_comp + c * tr.sfo_comp)));      // TPhaseCompensate
            v = mul_q15(v, rot_coeff(T, w16(tr.avg + c * tr.del)));                // TPilotTrack::_rotate
            int re = v.re >> 4, im = v.im >> 4;                                     // demap_limit<64> (demapper.h:141-151)
            re = min(max(re, -128), 127); im = min(max(im, -128), 127);
            const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
            uint8_t* o = s_soft[w] + lane * nb;                                     // DemapperCore::Demap<N_BPSC> (demapper.h:16-45)
            if (nb == 1) { o[0] = T.demap[ur]; }
            else if (nb == 2) { o[0] = T.demap[ur]; o[1] = T.demap[ui]; }
            else if (nb == 4) { o[0] = T.demap[ur]; o[1] = T.demap[256 + ur]; o[2] = T.demap[ui]; o[3] = T.demap[256 + ui]; }
            else { o[0] = T.demap[ur]; o[1] = T.demap[512 + ur]; o[2] = T.demap[768 + ur];
                   o[3] = T.demap[ui]; o[4] = T.demap[512 + ui]; o[5] = T.demap[768 + ui]; }
        }
    }
    __syncthreads();
    if (active) {
        const int ncbps = 48 * nb;
        const int di = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3;
        const uint16_t* map = T.deint + di * 288;
        // soft stream of the frame, one 16-bit field per soft value, already in the Viterbi kernel's metric format (v << 9)
        uint32_t* dst = reinterpret_cast<uint32_t*>(A.soft + ((size_t)slot0 * kSoftPerSlot + (size_t)(sym - 1) * ncbps) * 2);
        for (int k2 = lane; k2 < ncbps / 2; k2 += 64)                               // T11aDeinterleave* : out[k] = in[j(k)]
            dst[k2] = ((uint32_t)s_soft[w][map[2 * k2]] << 9) | ((uint32_t)s_soft[w][map[2 * k2 + 1]] << 25);
    }
}

// ------------------------------------------------------------------------------------------------
// k_viterbi: forward add-compare-select of the K=7 (133,171) code exactly as TViterbiCore does it
// (viterbicore.h:293-465): 8-bit WRAPPING path metrics with the decision in the metric LSB (&0xFE / |1),
// unsigned minimum, normalisation whenever (trellis_index & 7) == 0 after a puncture group, and the window
// schedule of T11aViterbi<..,256,24>::Process (viterbi.hpp:189-214).
//
// Arithmetic.  A reference metric byte is m = 2u + d (d = decision mark).  Branch metrics are even, so
//   c0 = (x0 + bm0) & 0xFE = 2((u0 + b0) mod 128),  c1 = ((x1 + bm1) & 0xFE) | 1 = 2((u1 + b1) mod 128) + 1,  b = bm/2
//   min(c0, c1) picks c1 iff (u1 + b1) mod 128 < (u0 + b0) mod 128  (a tie keeps c0), and the new u is that minimum.
// The kernel therefore carries U = u << 25 in a 32-bit VGPR: the 7-bit wrap is the natural 32-bit wrap, the
// unsigned compare is a plain v_cmp_lt_u32, no masking at all.  Normalisation subtracts min(U) (= (min m & 0xFE)/2).
// Branch metric of soft value v (0..7) for expected bit c: b = v ^ (c ? 7 : 0) (VIT_MA/VIT_MB, viterbilut.h:50-185,
// halved); expected bits = parity((branch<<6 | n) & 0155) for A, & 0117 for B, n = new state.  Both generators
// tap the oldest bit, so the decision-1 branch costs K - b0, K = 14 (7 on a punctured step).
//
// CDNA4 mapping.  wave64 = the 64 states, run as an in-place butterfly {p, p+32} -> {2p, 2p+1}: after t steps
// lane L holds state rol6^t(L), and the two predecessors of its next state sit in lanes L and L ^ (32 >> (t mod 6)).
// Every lane needs (x0, x1) = (metric of the pair member holding the decision-0 predecessor, the other one):
//     t mod 6 = 0,1 : v_permlane32_swap / v_permlane16_swap (gfx950) return exactly that pair
//     t mod 6 = 2,3 : two bank-masked DPP moves (row_ror:8 / row_shl:4 + row_shr:4)
//     t mod 6 = 4,5 : quad_perm DPP operands folded into the two adds
// i.e. ~9 VALU instructions per trellis step, no LDS, no ds_bpermute (a ds_bpermute formulation of the same
// recurrence was bounded by the dependent LDS round trip: 3.4 ms for the 4096-frame batch).
// Decisions: the compare result is carried into a per-lane history register with one v_addc (hist = 2 hist + d);
// every 24 steps (4 butterfly cycles = 8 punctured groups at 3/4) the 64 lanes store their 24-bit histories as one
// coalesced 256-byte row.  Soft values are prefetched 64 words at a time (one per lane), handed out with v_readlane and unpacked on the SALU.
__device__ __forceinline__ unsigned rol6(unsigned v, unsigned r) { r %= 6; return ((v << r) | (v >> (6 - r))) & 63u; }

__device__ __forceinline__ unsigned dpp_min_u32_wave(unsigned v)      // wave-wide unsigned minimum, VALU latency only
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // ^1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // ^2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true));   // row_ror:4
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true));   // row_ror:8
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = min(r16[0], r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return min(r32[0], r32[1]);
}

// Two frames per wave.  The metric of frame A lives in the low 16-bit half of the lane's register (u in bits 15:9),
// frame B's in the high half (bits 31:25): v_pk_add_u16 / v_pk_min_u16 run both trellises with one instruction,
// and the cross-lane exchanges move both halves at once.  Integer VALU ops issue at one wave64 instruction per
// ~4.4 cycles per SIMD on this part (measured: 10.3 VALU/step <-> 46 cycles/step at 4 waves/SIMD), so the kernel
// is bound by VALU instructions per trellis step per frame: 10.3 with one frame per wave, 6.5 with two.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) + __builtin_bit_cast(u16x2_t, b))); }
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) - __builtin_bit_cast(u16x2_t, b))); }
__device__ __forceinline__ unsigned pk_min16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b))); }

__device__ __forceinline__ unsigned dpp_pkmin_wave(unsigned v)         // per-half wave-wide unsigned minimum, broadcast to all lanes
{
    v = pk_min16(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // ^1
    v = pk_min16(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // ^2
    v = pk_min16(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true));   // row_ror:4
    v = pk_min16(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true));   // row_ror:8
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = pk_min16(r16[0], r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return pk_min16(r32[0], r32[1]);
}

struct VitLane {
    unsigned U;              // (u_B << 25) | (u_A << 9): metrics of the state this lane currently holds, frames A and B
    unsigned histA, histB;   // decisions of this lane, newest in bit 0
    unsigned MA[6], MB[6];   // (expected bit ? 7 : 0) in both fields, for the decision-0 branch into the lane's next state, per phase
};

constexpr unsigned kFld = (1u << 9) | (1u << 25);        // one unit of u in both halves

template <int PH, int WHICH>   // PH = t mod 6; WHICH 0: (A,B)  1: A only  2: B only.  a, b = packed soft values (wave-uniform)
__device__ __forceinline__ void acs_bfly(VitLane& V, unsigned a, unsigned b)
{
    constexpr unsigned K = (WHICH == 0 ? 14u : 7u) * kFld;
    unsigned b0;
    if (WHICH == 0)      b0 = (a ^ V.MA[PH]) + (b ^ V.MB[PH]);                   // per field <= 14: no carry between the halves
    else if (WHICH == 1) b0 = a ^ V.MA[PH];
    else                 b0 = b ^ V.MB[PH];
    const unsigned b1 = K - b0;
    unsigned x0, x1;
    const int u = (int)V.U;
    if (PH == 0)      { auto r = __builtin_amdgcn_permlane32_swap(V.U, V.U, false, false); x0 = r[0]; x1 = r[1]; }
    else if (PH == 1) { auto r = __builtin_amdgcn_permlane16_swap(V.U, V.U, false, false); x0 = r[0]; x1 = r[1]; }
    else if (PH == 2) { x0 = (unsigned)__builtin_amdgcn_update_dpp(u, u, 0x128, 0xF, 0xC, false);      // lanes 8-15 of a row <- L-8
                        x1 = (unsigned)__builtin_amdgcn_update_dpp(u, u, 0x128, 0xF, 0x3, false); }    // lanes 0-7          <- L+8
    else if (PH == 3) { x0 = (unsigned)__builtin_amdgcn_update_dpp(u, u, 0x114, 0xF, 0xA, false);      // bit2 = 1 lanes <- L-4
                        x1 = (unsigned)__builtin_amdgcn_update_dpp(u, u, 0x104, 0xF, 0x5, false); }    // bit2 = 0 lanes <- L+4
    else if (PH == 4) { x0 = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0x44, 0xF, 0xF, true);        // quad_perm [0,1,0,1]
                        x1 = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0xEE, 0xF, 0xF, true); }      // quad_perm [2,3,2,3]
    else              { x0 = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0xA0, 0xF, 0xF, true);        // quad_perm [0,0,2,2]
                        x1 = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0xF5, 0xF, 0xF, true); }      // quad_perm [1,1,3,3]
    const unsigned c0 = pk_add16(x0, b0), c1 = pk_add16(x1, b1);
    const uint64_t dA = __ballot((unsigned short)c1 < (unsigned short)c0);      // decision of frame A: strict compare, a tie keeps branch 0
    const uint64_t dB = __ballot((c1 >> 16) < (c0 >> 16));                      // decision of frame B
    V.U = pk_min16(c0, c1);
    uint64_t carry_out;
    asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(V.histA), "=s"(carry_out) : "s"(dA));   // hist = 2*hist + decision, one VALU op
    asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(V.histB), "=s"(carry_out) : "s"(dB));
}

constexpr int kColsPerRow = 24;            // trellis columns per stored 256-byte decision row

struct VitSide {            // wave-uniform per-frame bookkeeping
    const uint32_t* soft; uint32_t* decT; uint32_t* tbk; uint32_t nsteps, last_chunk, tr_end, nw; bool on, done;
};

// Soft input: 16 bits per soft value, v << 9 (what k_demap / k_soft_widen write), so a packed branch-metric operand is
// one s_pack_ll/hh_b32_b16 of a word of frame A and a word of frame B.  The words arrive through the scalar cache
// (s_load_dwordx8 per 12-step chunk per frame, prefetched one chunk ahead): no VALU work, no LDS.
template <int CR>
__device__ __forceinline__ void viterbi_forward(const VitJob& JA, const VitJob& JB, bool hasB, const uint8_t* __restrict__ soft_base, uint64_t* __restrict__ dec_base,
                                                uint32_t* __restrict__ tbkA, uint32_t* __restrict__ tbkB, uint32_t* __restrict__ nwinA, uint32_t* __restrict__ nwinB)
{
    constexpr int GB = CR == 0 ? 2 : CR == 2 ? 4 : 3;                           // soft values per puncture group (CR: 0=1/2, 1=2/3, 2=3/4)
    constexpr int GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;                           // trellis steps per group
    constexpr int CW = 12 / GS * GB / 2;                                        // 32-bit soft words per 12-step chunk: 12 / 9 / 8
    const unsigned lane = threadIdx.x & 63;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    VitSide A, B;
    const uint32_t softA = uni(JA.soft_off), softB = uni(hasB ? JB.soft_off : JA.soft_off);
    const uint32_t nsA = uni(JA.nsoft), nsB = uni(hasB ? JB.nsoft : JA.nsoft);
    A.soft = reinterpret_cast<const uint32_t*>(soft_base + softA); A.decT = reinterpret_cast<uint32_t*>(dec_base + uni(JA.dec_off));
    A.tbk = tbkA; A.nsteps = nsA / GB * GS; A.last_chunk = (A.nsteps - 1) / 12; A.tr_end = uni(JA.length) * 8u + 16u + 6u; A.nw = 0; A.on = true; A.done = false;
    B.soft = reinterpret_cast<const uint32_t*>(soft_base + softB); B.decT = reinterpret_cast<uint32_t*>(dec_base + uni(hasB ? JB.dec_off : JA.dec_off));
    B.tbk = tbkB; B.nsteps = hasB ? nsB / GB * GS : 0u; B.last_chunk = (nsB / GB * GS - 1) / 12; B.tr_end = hasB ? uni(JB.length) * 8u + 16u + 6u : 0u; B.nw = 0; B.on = hasB; B.done = !hasB;

    VitLane V;
    V.U = lane == 0 ? 0u : 0x18u * kFld;                                       // ALL_INIT0 / ALL_INIT = 0x00 / 0x30 (viterbilut.h:22-30)
    V.histA = V.histB = 0;
#pragma unroll
    for (int ph = 0; ph < 6; ph++) {
        const unsigned n = rol6(lane, ph + 1);                                  // label held after a phase-ph step
        V.MA[ph] = (__popc(n & 0155) & 1) ? 7u * kFld : 0u;
        V.MB[ph] = (__popc(n & 0117) & 1) ? 7u * kFld : 0u;
    }

    const uint32_t nsteps = max(A.nsteps, B.nsteps);
    uint32_t tr = 0, ob = 0;                                                    // ob: bits handed out by the partial windows (same schedule for both frames)

    auto normalize = [&]() { V.U = pk_sub16(V.U, dpp_pkmin_wave(V.U)); };       // Normalize (viterbicore.h:444-465), both frames
    auto record = [&](VitSide& S, unsigned mbyte, uint32_t cnt, uint32_t look) {
        // arg-min with the reference's tie-break: metric<<8 | state<<2 (viterbicore.h:479-524); metric = 2u + last decision
        const unsigned kmin = dpp_min_u32_wave((mbyte << 8) | (rol6(lane, tr) << 2));
        if (lane == 0) {
            S.tbk[S.nw * 3 + 0] = tr;
            S.tbk[S.nw * 3 + 1] = look | (cnt << 16);
            S.tbk[S.nw * 3 + 2] = ((kmin >> 2) & 0x3F) | (((kmin >> 8) & 1) << 6) | (ob << 8);
        }
        S.nw++;
    };
    auto next_event = [&]() -> uint32_t {
        uint32_t t = ob + 256u + 24u + 6u;
        if (!A.done) t = min(t, A.tr_end);
        if (!B.done) t = min(t, B.tr_end);
        return t;
    };
    uint32_t next_thr = next_event();
    auto check = [&]() {                                                        // trace-back schedule (viterbi.hpp:196-214), per frame
        if (tr >= next_thr) {
            const unsigned mA = ((V.U & 0xFFFFu) >> 8) | (V.histA & 1u), mB = (V.U >> 24) | (V.histB & 1u);
            const bool partial = tr >= ob + 256u + 24u + 6u;
            const uint32_t plook = 24 + (tr - (ob + 256 + 24 + 6)) % 8;
            if (!A.done) {
                if (tr >= A.tr_end) { record(A, mA, A.tr_end - ob - 6, tr - A.tr_end); A.done = true; }
                else if (partial) record(A, mA, 256, plook);
            }
            if (!B.done) {
                if (tr >= B.tr_end) { record(B, mB, B.tr_end - ob - 6, tr - B.tr_end); B.done = true; }
                else if (partial) record(B, mB, 256, plook);
            }
            if (partial) ob += 256;
            next_thr = next_event();
        }
    };
    auto store_row = [&](uint32_t row, unsigned sh) {
        if (row * kColsPerRow < A.nsteps + kColsPerRow) A.decT[row * 64 + lane] = V.histA << sh;
        if (B.on && row * kColsPerRow < B.nsteps + kColsPerRow) B.decT[row * 64 + lane] = V.histB << sh;
    };
    struct Chunk { uint32_t a[CW], b[CW]; };
    auto load_chunk = [&](uint32_t c, uint32_t zero) -> Chunk {                 // chunk c of both frames; past a frame's end: its last chunk again
        Chunk K;                                                                //   (that frame is done by then; the frame's spare slot covers a ragged tail)
        const uint32_t* pa = A.soft + min(c, A.last_chunk) * CW + zero;
        const uint32_t* pb = B.soft + min(c, B.last_chunk) * CW + zero;
#pragma unroll
        for (int i = 0; i < CW; i++) { K.a[i] = pa[i]; K.b[i] = pb[i]; }
        return K;
    };
    auto sv = [](const Chunk& K, int k) -> unsigned {                           // soft value k of the chunk, frames A | B
        return (k & 1) ? ((K.a[k >> 1] >> 16) | (K.b[k >> 1] & 0xFFFF0000u)) : ((K.a[k >> 1] & 0xFFFFu) | (K.b[k >> 1] << 16));
    };
    auto step = [&](int ph, int which, unsigned a, unsigned b) {               // ph, which are constants after unrolling
        switch (ph * 3 + which) {
        case 0:  acs_bfly<0, 0>(V, a, b); break; case 1:  acs_bfly<0, 1>(V, a, b); break; case 2:  acs_bfly<0, 2>(V, a, b); break;
        case 3:  acs_bfly<1, 0>(V, a, b); break; case 4:  acs_bfly<1, 1>(V, a, b); break; case 5:  acs_bfly<1, 2>(V, a, b); break;
        case 6:  acs_bfly<2, 0>(V, a, b); break; case 7:  acs_bfly<2, 1>(V, a, b); break; case 8:  acs_bfly<2, 2>(V, a, b); break;
        case 9:  acs_bfly<3, 0>(V, a, b); break; case 10: acs_bfly<3, 1>(V, a, b); break; case 11: acs_bfly<3, 2>(V, a, b); break;
        case 12: acs_bfly<4, 0>(V, a, b); break; case 13: acs_bfly<4, 1>(V, a, b); break; case 14: acs_bfly<4, 2>(V, a, b); break;
        case 15: acs_bfly<5, 0>(V, a, b); break; case 16: acs_bfly<5, 1>(V, a, b); break; default: acs_bfly<5, 2>(V, a, b); break;
        }
    };
    // one puncture group = GS steps starting at step index i0 of a 12-step chunk (tr % 12 == 0 at the chunk start, so phase = i0 % 6)
    auto group = [&](const Chunk& K, int i0) {
        const int k0 = i0 / GS * GB;
        step(i0 % 6, 0, sv(K, k0), sv(K, k0 + 1));                              // ACS(A,B)
        if (CR != 0) step((i0 + 1) % 6, 1, sv(K, k0 + 2), 0);                   // ACS(A)     2/3, 3/4 (viterbi.hpp:173-187)
        if (CR == 2) step((i0 + 2) % 6, 2, 0, sv(K, k0 + 3));                   // ACS(B)     3/4
    };
    auto chunk = [&](const Chunk& K) {                                          // up to 12 steps
        if (tr + 12 <= nsteps && next_thr > tr + 12) {
            // fast path: no trace-back due inside the chunk -- straight-line code, no per-group tests.
            // Normalize whenever (trellis index & 7) == 0 after a group: tr % 8 is 0 or 4 here.
            const bool lo = (tr & 7) == 0;
#pragma unroll
            for (int g = 0; g < 12 / GS; g++) {
                group(K, g * GS);
                const int s = (g + 1) * GS;
                if (s == 4 && !lo) normalize();
                if (s == 8 && lo) normalize();
            }
            tr += 12;
            if (!lo) normalize();
        } else {
#pragma unroll
            for (int g = 0; g < 12 / GS; g++) {
                if (tr < nsteps && !(A.done && B.done)) {
                    group(K, g * GS);
                    tr += GS;
                    if ((tr & 7) == 0) normalize();
                    check();
                }
            }
        }
    };

    // Scalar loads return out of order, so the only wait the hardware offers is lgkmcnt(0), and the compiler puts it at the
    // first use of the loaded registers.  The prefetch of chunk c+1 must therefore be ISSUED after the first use of chunk c
    // (or that wait would cover the prefetch too) and is then covered by a whole chunk of ACS work.  The order is pinned by
    // data flow: the prefetch address takes a bit of chunk c that is always zero (soft fields are v << 9).
    uint32_t row = 0, c = 0;
    Chunk cur = load_chunk(0, 0);
    while (tr < nsteps && !(A.done && B.done)) {
        const Chunk nxt = load_chunk(c + 1, (cur.a[0] | cur.b[0]) & 1u);
        chunk(cur);
        if ((tr % kColsPerRow) == 0) { store_row(row, 8); row++; }
        cur = nxt; c++;
    }
    // last, partial row: left-align so that column c always sits at bit 31 - ((c - 1) % 24)
    if ((tr % kColsPerRow) != 0) store_row(row, 32 - (tr % kColsPerRow));
    if (lane == 0) { *nwinA = A.nw; if (hasB) *nwinB = B.nw; }
}

// Four waves per 256-thread workgroup, two frames per wave (no cross-wave traffic).  One-wave workgroups were kept to
// 8 per CU by the dispatcher: 2 waves per SIMD and a second round for a 4096-frame batch.
// Frames are paired in job order when their code rates agree; otherwise each runs alone in the low half.
__global__ void __launch_bounds__(256) k_viterbi(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs_ptr, uint32_t njobs_max, const uint8_t* __restrict__ soft, uint64_t* __restrict__ dec, uint32_t* __restrict__ tbk, uint32_t* __restrict__ nwin)
{
    const uint32_t njobs = njobs_ptr ? *njobs_ptr : njobs_max;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };   // everything below is per-wave uniform: keep it in SGPRs
    const uint32_t fa = uni((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2), fb = fa + 1;
    if (fa >= njobs) return;
    auto load_job = [&](uint32_t f) {
        const VitJob& G = jobs[f];
        VitJob J;
        J.soft_off = uni(G.soft_off); J.nsoft = uni(G.nsoft); J.length = uni(G.length); J.dec_off = uni(G.dec_off);
        J.out_off = uni(G.out_off); J.valid = uni(G.valid); J.code_rate = uni(G.code_rate); J.pad = 0;
        return J;
    };
    const VitJob JA = load_job(fa);
    VitJob JB = JA;
    bool hasB = fb < njobs;
    if (hasB) { JB = load_job(fb); hasB = JB.valid != 0; }
    uint32_t* ta = tbk + (size_t)fa * kMaxWindows * 3;
    uint32_t* tb = tbk + (size_t)fb * kMaxWindows * 3;
    const bool pair = JA.valid && hasB && JA.code_rate == JB.code_rate;
    for (int pass = 0; pass < (pair ? 1 : 2); pass++) {                         // unpaired: A alone, then B alone (one call site per code rate)
        const bool second = pass == 1;
        if (second ? !hasB : !JA.valid) continue;
        const VitJob X = second ? JB : JA;
        uint32_t* tx = second ? tb : ta;
        uint32_t* nx = nwin + (second ? fb : fa);
        if (X.code_rate == 0)      viterbi_forward<0>(X, JB, pair, soft, dec, tx, tb, nx, nwin + fb);
        else if (X.code_rate == 1) viterbi_forward<1>(X, JB, pair, soft, dec, tx, tb, nx, nwin + fb);
        else                       viterbi_forward<2>(X, JB, pair, soft, dec, tx, tb, nx, nwin + fb);
    }
}

// ------------------------------------------------------------------------------------------------
// k_traceback: TViterbiCore::Traceback (viterbicore.h:468-555) -- every window of a frame is an independent
// walk through the stored decisions, one thread each (a 54 Mbps 1500-byte frame has 47 windows), one wave per
// frame.  Decisions live in rows of 24 trellis columns x 64 lanes (see k_viterbi): row R, word L, bit 31-i holds
// the decision that lane L took in column 24R+1+i, and lane L held state rol6^c(L) in column c.  Lanes walk in
// lock-step one row (24 columns) at a time: the wave first copies every window's current 256-byte row into LDS
// with coalesced loads, then each lane chases through its own copy.  Bytes come out LSB-first in time.
__global__ void __launch_bounds__(64) k_traceback(const VitJob* jobs, const uint32_t* njobs_ptr, uint32_t njobs_max, const uint64_t* dec_base, const uint32_t* tbk, const uint32_t* nwin, uint8_t* out_base)
{
    __shared__ uint32_t s_row[64][65];                                           // one decision row per window-thread (+1 pad)
    const uint32_t f = blockIdx.x;
    if (f >= (njobs_ptr ? *njobs_ptr : njobs_max)) return;
    const VitJob J = jobs[f];
    if (!J.valid) return;
    const int lane = threadIdx.x;
    const uint32_t nw = nwin[f];
    const uint32_t* decT = reinterpret_cast<const uint32_t*>(dec_base + J.dec_off);
    uint8_t* out = out_base + J.out_off;
    for (uint32_t w0 = 0; w0 < nw; w0 += 64) {
        const uint32_t w = w0 + (uint32_t)lane;
        uint32_t col = 0, look = 0, cnt = 0, ob = 0; unsigned pos = 0; int left = 0;
        if (w < nw) {
            const uint32_t* t = tbk + ((size_t)f * kMaxWindows + w) * 3;
            col = t[0]; look = t[1] & 0xFFFF; cnt = t[1] >> 16; pos = t[2] & 0x7F; ob = t[2] >> 8; left = (int)(look + cnt);
        }
        // The walk reads columns col-1, col-2, ... (the decision of the start state is already in `pos`).
        uint32_t c = col - 1;                                                    // next column to read (>= 6 whenever left > 0)
        uint32_t i_out = 0;                                                      // steps taken
        unsigned oc = 0;
        uint8_t* ob_ptr = out + (ob >> 3);
        int any = __any(left > 0);
        while (any) {
            const uint32_t rowi = left > 0 ? (c - 1) / kColsPerRow : 0xFFFFFFFFu;
            __syncthreads();
            // coalesced: the whole wave copies window tw's row; all loads of a half are in flight together
            const int nwin_here = (int)min(64u, nw - w0);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (h * 32 < nwin_here) {
                    uint32_t v[32];
#pragma unroll
                    for (int k = 0; k < 32; k++) {
                        const uint32_t r_tw = (uint32_t)__builtin_amdgcn_readlane((int)rowi, h * 32 + k);
                        v[k] = (r_tw != 0xFFFFFFFFu) ? decT[(size_t)r_tw * 64 + lane] : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 32; k++) s_row[h * 32 + k][lane] = v[k];
                }
            }
            __syncthreads();
            if (left > 0) {
                const uint32_t row_first = rowi * kColsPerRow + 1;               // first column of this row
                unsigned r = c % 6u;
                while (left > 0 && c >= row_first) {
                    if (i_out >= look) {
                        const uint32_t bi = cnt - 1 - (i_out - look);
                        oc = (oc << 1) | ((pos >> 6) & 1u);
                        if ((bi & 7) == 0) { ob_ptr[bi >> 3] = (uint8_t)oc; oc = 0; }
                    }
                    pos = (pos >> 1) & 0x3Fu;                                    // predecessor state, lives in column c
                    const unsigned L = ((pos >> r) | (pos << (6 - r))) & 63u;    // ror6^c(state)
                    const unsigned bit = 31u - (c - row_first);
                    pos |= ((s_row[lane][L] >> bit) & 1u) << 6;
                    r = r == 0 ? 5u : r - 1;
                    c--; left--; i_out++;
                }
            }
            any = __any(left > 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_finish: T11aDesc (scramble.hpp:267-353) + TBB11aFrameSink (PHY_11a.hpp:607-702).  One wave per frame.
//   * descrambling is data-parallel: the x^7+x^4+1 sequence has period 127, every non-zero seed is a phase of
//     the same cycle, so scrambler byte j = seqbyte[(phase(seed) + 8 j) mod 127]  (two small tables, no chain);
//   * the 64 lanes load / descramble / store the MPDU coalesced and park it in LDS;
//   * CRC-32 is a byte-serial recurrence: lane 0 runs it slicing-by-8 out of LDS (8 x 1 KiB tables in LDS).
__global__ void __launch_bounds__(256) k_finish(RxArgs A)
{
    __shared__ uint32_t s_crc[8][256];
    __shared__ uint32_t s_bufs[4][2504 / 4 + 2];
    for (int i = threadIdx.x; i < 2048; i += 256) s_crc[i >> 8][i & 255] = A.T.crc8[i];
    __syncthreads();
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= *A.njobs) return;
    const uint32_t f = A.joblist[j];
    FrameRow& r = A.frames[f];
    if (!r.valid || r.error_code != 0) return;
    const int lane = threadIdx.x & 63;
    const Tables& T = A.T;
    uint32_t* s_buf = s_bufs[threadIdx.x >> 6];
    const uint8_t* dec = A.vout + (size_t)r.slot0 * kOutPerSlot;
    uint8_t* mp = A.mpdu + (size_t)r.slot0 * kOutPerSlot;
    const uint32_t L = r.length;
    const unsigned seed = dec[1] >> 1;                                           // byte 0 dropped, byte 1 >> 1 seeds the register
    const unsigned phase = T.scr_phase[seed & 0x7F];                             // 255: seed 0 (sequence stays 0)
    uint8_t* bytes = reinterpret_cast<uint8_t*>(s_buf);
    for (uint32_t i = lane; i < L; i += 64) {
        const unsigned sb = phase == 255 ? 0u : T.scr_seq[(phase + 8u * i) % 127u];
        const unsigned o = dec[2 + i] ^ sb;
        bytes[i] = (uint8_t)o;
        mp[i] = (uint8_t)o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                             // the LDS buffer is private to this wave
    if (lane == 0) {
        uint32_t crc = 0xFFFFFFFFu;
        const uint32_t n = L >= 4 ? L - 4 : 0;                                    // PHY_11a.hpp:668-673: the FCS bytes are not fed to the CRC
        uint32_t i = 0;
        for (; i + 8 <= n; i += 8) {
            const uint32_t lo = s_buf[i >> 2] ^ crc, hi = s_buf[(i >> 2) + 1];
            crc = s_crc[7][lo & 0xFF] ^ s_crc[6][(lo >> 8) & 0xFF] ^ s_crc[5][(lo >> 16) & 0xFF] ^ s_crc[4][lo >> 24] ^
                  s_crc[3][hi & 0xFF] ^ s_crc[2][(hi >> 8) & 0xFF] ^ s_crc[1][(hi >> 16) & 0xFF] ^ s_crc[0][hi >> 24];
        }
        for (; i < n; i++) crc = (crc >> 8) ^ s_crc[0][(bytes[i] ^ crc) & 0xFF];
        uint32_t fcs = 0;
        if (L >= 4) fcs = (uint32_t)bytes[L - 4] | ((uint32_t)bytes[L - 3] << 8) | ((uint32_t)bytes[L - 2] << 16) | ((uint32_t)bytes[L - 1] << 24);
        r.crc32 = fcs;
        r.error_code = ((~crc) == fcs) ? E_FRAME_OK : E_CRC32_FAIL;              // PHY_11a.hpp:688-692
    }
}

// ------------------------------------------------------------------------------------------------
// k_pack: compacts the per-capture frame table into dense sora_frame_result rows in (capture, time) order, on the
// device, so the rows can feed an RCCL all-gather without a host round trip.  mpdu_offset = slot0 * 32 indexes the
// device MPDU array directly.  One 1024-thread block: captures are scanned in tiles of 1024.
struct PackedRow { uint32_t capture_id, start_sample, end_sample, error_code, rate_kbps; uint16_t length, nsym; uint32_t crc32; int16_t cfo_est; uint16_t reserved; uint32_t mpdu_offset; };
__global__ void __launch_bounds__(1024) k_pack(const FrameRow* frames, const uint32_t* nframes, const CapDesc* caps, uint32_t ncaps, uint32_t max_frames,
                                               PackedRow* rows, uint32_t* nrows_out)
{
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_base;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < ncaps; c0 += 1024) {
        const uint32_t c = c0 + t;
        const uint32_t n = c < ncaps ? min(nframes[c], max_frames) : 0u;
        s_scan[t] = n;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {                                // Hillis-Steele inclusive scan
            const uint32_t v = t >= o ? s_scan[t - o] : 0u;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const uint32_t first = s_base + s_scan[t] - n;
        for (uint32_t i = 0; i < n; i++) {
            const FrameRow& r = frames[(size_t)c * max_frames + i];
            PackedRow o;
            o.capture_id = caps[c].capture_id; o.start_sample = r.start_sample; o.end_sample = r.end_sample; o.error_code = r.error_code;
            o.rate_kbps = r.rate_kbps; o.length = r.length; o.nsym = r.nsym; o.crc32 = r.crc32; o.cfo_est = r.cfo_est; o.reserved = 0;
            o.mpdu_offset = r.slot0 * (uint32_t)kOutPerSlot;
            rows[first + i] = o;
        }
        __syncthreads();
        if (t == 1023) s_base += s_scan[1023];
        __syncthreads();
    }
    if (t == 0) *nrows_out = s_base;
}

// ================================================================================================
// stand-alone stage kernels (per-stage C entry points)
__global__ void __launch_bounds__(256) k_fft64_batch(const uint32_t* in, uint32_t* out, uint32_t n, Tables T)
{
    __shared__ uint32_t s_all[16][64];
    const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
    const uint32_t i = blockIdx.x * 16 + g;
    cpx x[4], Y[4];
#pragma unroll
    for (int m = 0; m < 4; m++) x[m] = i < n ? unpack(in[(size_t)i * 64 + e + 16 * m]) : mk(0, 0);
    fft64_group(x, Y, s_all[g], e, T, []() { __syncthreads(); });
    if (i < n) {
#pragma unroll
        for (int q = 0; q < 4; q++) out[(size_t)i * 64 + e + 16 * q] = pack(Y[q]);
    }
}

__global__ void __launch_bounds__(64) k_demap_batch(const uint32_t* in, uint8_t* soft, int nb, uint32_t n, Tables T)
{
    const uint32_t i = blockIdx.x; const int lane = threadIdx.x;
    if (i >= n || lane >= 48) return;
    const int bin = carrier_bin48(lane);
    cpx v = unpack(in[(size_t)i * 64 + bin]);
    int re = v.re >> 4, im = v.im >> 4;
    re = min(max(re, -128), 127); im = min(max(im, -128), 127);
    const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
    uint8_t* o = soft + (size_t)i * 48 * nb + lane * nb;
    if (nb == 1) { o[0] = T.demap[ur]; }
    else if (nb == 2) { o[0] = T.demap[ur]; o[1] = T.demap[ui]; }
    else if (nb == 4) { o[0] = T.demap[ur]; o[1] = T.demap[256 + ur]; o[2] = T.demap[ui]; o[3] = T.demap[256 + ui]; }
    else { o[0] = T.demap[ur]; o[1] = T.demap[512 + ur]; o[2] = T.demap[768 + ur]; o[3] = T.demap[ui]; o[4] = T.demap[512 + ui]; o[5] = T.demap[768 + ui]; }
}

__global__ void __launch_bounds__(64) k_deint_batch(const uint8_t* in, uint8_t* out, int nb, uint32_t n, Tables T)
{
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const int ncbps = 48 * nb, di = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3;
    for (int k = threadIdx.x; k < ncbps; k += 64) out[(size_t)i * ncbps + k] = in[(size_t)i * ncbps + T.deint[di * 288 + k]];
}

// sora_hip_viterbi11a takes the reference's soft format (one byte per soft value, 3 significant bits);
// the trellis kernel reads 16-bit fields v << 9.
__global__ void __launch_bounds__(256) k_soft_widen(const uint8_t* soft8, const uint32_t* off8, const uint32_t* nsoft, const uint32_t* off16, uint8_t* soft16)
{
    const uint32_t j = blockIdx.x;
    const uint8_t* in = soft8 + off8[j];
    uint16_t* out = reinterpret_cast<uint16_t*>(soft16 + off16[j]);
    for (uint32_t k = threadIdx.x; k < nsoft[j]; k += blockDim.x) out[k] = (uint16_t)((in[k] & 7u) << 9);
}

__global__ void __launch_bounds__(64) k_make_vitjobs(VitJob* jobs, const uint32_t* soft_off, const uint32_t* nsoft, const uint16_t* flen,
                                                      const uint32_t* out_off, const uint32_t* dec_off, int code_rate, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    VitJob J; J.soft_off = soft_off[i]; J.nsoft = nsoft[i]; J.length = flen[i]; J.dec_off = dec_off[i]; J.out_off = out_off[i];
    J.valid = 1; J.code_rate = (uint32_t)code_rate; J.pad = 0;
    jobs[i] = J;
}

}  // namespace sora
