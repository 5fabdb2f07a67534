// k_rx.hip -- batched per-symbol / per-frame kernels of the 802.11a receive path (gfx950).
//
//   k_frame       T11aDataSymbol -> TFreqCompensation -> TFFT64 -> TChannelEqualization -> TPhaseCompensate ->
//                 TPilotTrack -> T11aDemap<N> -> T11aDeinterleave                           (one wave per frame, 4 symbols per pass)
//   k_viterbi<CR> T11aViterbi<5000*8,48,256,24>: 64-state ACS + windowed trace-back out of LDS (one wave per two frames)
//   k_finish      T11aDesc + TBB11aFrameSink (descramble, CRC-32, FRAME_OK / CRC32_FAIL)   (one wave per frame)
// plus the stand-alone stage kernels behind the per-stage C entry points.
#include <hip/hip_runtime.h>
#include <utility>
#include "kernels.h"
#include "dev_viterbi.h"
#include "dev_winplan.h"
#include "dev_vitwin.h"

namespace sora {

__device__ __constant__ uint8_t kPilotSgn[128] = {        // pilot.hpp:10-28: 1 <=> polarity -1
    0,0,0,1,1,1,0,1, 1,1,1,0,0,1,0,1, 1,0,0,1,0,0,1,0, 0,0,0,0,0,1,0,0,
    0,1,0,0,1,1,0,0, 0,1,0,1,1,1,0,1, 0,1,1,0,1,1,0,0, 0,0,0,1,1,0,0,1,
    1,0,1,0,1,0,0,1, 1,1,0,0,1,1,1,1, 0,1,1,0,1,0,0,0, 0,1,0,1,0,1,0,1,
    1,1,1,1,0,1,0,0, 1,0,1,0,0,0,1,1, 0,1,1,1,0,0,0,1, 1,1,1,1,1,1,0,0 };

// ------------------------------------------------------------------------------------------------
// k_frame: everything between the frame table and the soft stream, one wave per frame:
//   TFreqCompensation -> TFFT64 -> TChannelEqualization -> TPhaseCompensate -> TPilotTrack -> T11aDemap -> T11aDeinterleave
// (fb11ademod_config.hpp:200-222).  Four OFDM symbols per pass, one per 16-lane group.  The front end (FFT, equaliser)
// and the back end (rotation, demap, de-interleave) of a symbol do not depend on the tracking state; the tracking loop
// itself (freqoffset.hpp:28-30, pilot.hpp:166-233: four pilot bins, two dependent LUT reads per symbol) is run for
// the pass's four symbols in order, pilot k in lane k, the loop state in scalar registers.  The equalised symbol never
// leaves LDS; the small tables (demap steps, de-interleaver map) live in LDS, the FFT twiddles in registers; the next
// pass's samples are requested before the tracking loop so that their latency hides behind it.
//   algorithmic bytes per data symbol: 256 read (64 of the 80 samples) + 3 N_CBPS / 8 written (the packed soft stream, rx_types.h)
struct FrameLds {
    uint32_t eq[4][4][64];                                                       // [wave][symbol of the pass]: FFT staging, then the equalised bins
    uint8_t  soft[4][4][288];                                                    // [wave][symbol of the pass]: soft values in carrier order
    uint8_t  demap[1024];                                                        // DemapperCore step tables (filled by the caller, 256 threads, in front of a block barrier)
    // SHARED tracker (k_frame): what wave 0 -- which runs the pilot tracker of all four frames of the workgroup -- hands back per [frame][symbol of the pass]:
    // {CFO_comp, SFO_comp before the symbol, the symbol's mean phase and slope}; and what it needs of the frames at the start
    int4     trk[4][4];
    int      nsym_of[4];
    short    st0[4][4];                                                          // [frame] { CFO_comp, SFO_comp, CFO_tracker, SFO_tracker } behind the SIGNAL symbol
};
// One wave, the frame queued at slot j of jobs[] / joblist[]: its VitJob and its packed soft stream.
// SHARED (k_frame; all four waves of the workgroup call it together, `valid` = the wave has a frame): the loop-carried part -- TPilotTrack's chain (pilot.hpp:166-233:
// four pilot bins, two dependent LUT reads per symbol) -- of the workgroup's four frames runs in wave 0, four lanes per frame, between two block barriers per pass.
// A wave spent 36 % of its vector instructions on that chain for the four lanes that hold ONE frame's pilots (and 40 scalar instructions per symbol on its state);
// one wave doing it for four frames at once is the same latency chain at a quarter of the instructions -- and this path is priced by its instructions.
// !SHARED: a wave on its own (the redo path behind k_pipe runs a frame at a time).
template <bool SHARED>
__device__ __forceinline__ void frame_symbols(const RxArgs& A, uint32_t j, FrameLds& lds, bool valid = true)
{
    uint32_t (&s_eq)[4][4][64] = lds.eq; uint8_t (&s_soft)[4][4][288] = lds.soft; const uint8_t* s_demap = lds.demap;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    const Tables& T = A.T;
    const uint32_t f = valid ? A.joblist[j] : 0u;
    FrameRow r = A.frames[f];
    if (!valid) { r.nsym = 0; r.nbpsc = 1; }
    const uint32_t my_nsoft = (uint32_t)r.nsym * 48u * r.nbpsc;
    if (lane == 0 && valid) {
        VitJob J;
        J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftBytesPerSlot; J.nsoft = my_nsoft; J.length = r.length;
        J.dec_off = 0; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate; J.soft_bits = 3;
        A.jobs[j] = J;
    }
    const FrameCtx* fx = A.fctx + f;
    const uint32_t* iq = A.iq + A.caps[r.capture].offset;
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    // FreqCoeffs / ChannelCoeffs as the operand pairs of the packed complex product
    PkTw fq[4], ch[4];
#pragma unroll
    for (int m = 0; m < 4; m++) { fq[m] = pk_tw_mul(fx->freq[e + 16 * m]); ch[m] = pk_tw_mul(fx->chan[e + 16 * m]); }
    // (wave-uniform: scalar branches on the modulation)
    const int nb = __builtin_amdgcn_readfirstlane((int)r.nbpsc), ncbps = 48 * nb, nsym = __builtin_amdgcn_readfirstlane((int)r.nsym);
    // de-interleaver source indices of output positions 8 lane .. 8 lane + 7 (lane < N_CBPS / 8: one three-byte group of the packed stream), two per register
    uint32_t mp[4];
    const bool packs = 8 * lane < ncbps;
    {
        const uint16_t* map = T.deint + (nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3) * 288;
#pragma unroll
        for (int t = 0; t < 4; t++) mp[t] = packs ? (uint32_t)map[8 * lane + 2 * t] | ((uint32_t)map[8 * lane + 2 * t + 1] << 16) : 0u;
    }
    // the frame's packed soft stream: symbol s (1-based) at 3 N_CBPS / 8 * (s - 1) bytes
    uint8_t* dst = A.soft + (size_t)r.slot0 * kSoftBytesPerSlot;
    // pilot k in lane k: bins 43, 57, 7, 21 = carriers -21, -7, +7, +21 (pilot.hpp:138-164)
    const int pk = lane & 3;
    const int pbin = pk == 0 ? 43 : pk == 1 ? 57 : pk == 2 ? 7 : 21, pc = pk == 0 ? -21 : pk == 1 ? -7 : pk == 2 ? 7 : 21;
    int cfo_comp = r.cfo_comp, sfo_comp = r.sfo_comp, cfo_tr = r.cfo_tracker, sfo_tr = r.sfo_tracker;
    unsigned symbol_count = 0;                                                   // 127 -> 0 after the SIGNAL symbol

    auto load_samples = [&](int s0, uint32_t raw[4]) {                           // the 64 samples after the cyclic prefix, symbol s0 + g
        const uint32_t p0 = r.data_start + 80u * (uint32_t)(s0 + g) + 8u;        // skip_cp = 8 (PHY_11a.hpp:365,394)
#pragma unroll
        for (int m = 0; m < 4; m++) raw[m] = (s0 + g <= nsym) ? iq[(size_t)(p0 + (uint32_t)(e + 16 * m)) * A.str] : 0u;
    };
    // ---- SHARED: the frames' lengths and tracker states to wave 0, lanes 4 f + k (pilot k of the frame of wave f)
    int nsym_loop = nsym, nsym_f = nsym;
    if (SHARED) {
        if (lane == 0) { lds.nsym_of[w] = nsym; lds.st0[w][0] = (short)cfo_comp; lds.st0[w][1] = (short)sfo_comp; lds.st0[w][2] = (short)cfo_tr; lds.st0[w][3] = (short)sfo_tr; }
        __syncthreads();
        nsym_loop = max(max(lds.nsym_of[0], lds.nsym_of[1]), max(lds.nsym_of[2], lds.nsym_of[3]));
        const int fr = (lane >> 2) & 3;
        nsym_f = lds.nsym_of[fr];
        cfo_comp = lds.st0[fr][0]; sfo_comp = lds.st0[fr][1]; cfo_tr = lds.st0[fr][2]; sfo_tr = lds.st0[fr][3];   // (only wave 0 uses them from here on)
    }
    uint32_t raw[4];
    load_samples(1, raw);
    for (int s0 = 1; s0 <= nsym_loop; s0 += 4) {
        const int sym = s0 + g;
        const bool active = sym <= nsym;
        const bool mine = s0 <= nsym;                                            // (SHARED: a wave whose frame is shorter only keeps the barriers company)
        int my_cfo = 0, my_sfo = 0, my_avg = 0, my_del = 0;
        if (mine) {
        // ---- TFreqCompensation + TFFT64 + TChannelEqualization on packed COMPLEX16 (dev_arith.h), symbol `sym` in group g
        pcx x[4], Y[4];
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = pk_cmul<15>(pk_sra(raw[m], 1), fq[m]);   // >>1, x FreqCoeffs (channel_11a.hpp:643-644)
        if (s0 + 4 <= nsym) load_samples(s0 + 4, raw);                           // next pass: in flight during the tracking loop
        fft64_core_pk(x, s_eq[w][g], e, W, wsync);
#pragma unroll
        for (int q = 0; q < 4; q++) Y[q] = s_eq[w][g][__brev((unsigned)(e + 16 * q)) >> 26];
        wsync();
#pragma unroll
        for (int q = 0; q < 4; q++) {                                            // channel_11a.hpp:548-574
            const int bin = e + 16 * q;
            s_eq[w][g][bin] = (bin >= 28 && bin < 36) ? 0u : pk_cmul<8>(Y[q], ch[q]);
        }
        wsync();
        }
        // ---- the loop-carried part, symbols s0 .. s0+3 in order
        if (SHARED) {
            __syncthreads();                                                     // every wave's four equalised symbols are in LDS
            if (w == 0) {
                const int fr = (lane >> 2) & 3;                                  // lanes 16 .. 63 repeat lanes 0 .. 15 (one instruction stream either way)
#pragma unroll
                for (int gg = 0; gg < 4; gg++) {
                    const bool act = s0 + gg <= nsym_f;
                    const int c0 = cfo_comp, s0c = sfo_comp;
                    const cpx p = mul_q15(unpack(s_eq[fr][gg][pbin]), rot_coeff(T, w16(cfo_comp + pc * sfo_comp)));
                    int th = pk == 3 ? uatan2(T, -p.im, -p.re) : uatan2(T, p.im, p.re);
                    // (the symbol index is the same for every frame -- all of them start behind their SIGNAL symbol -- and wave-uniform: a scalar)
                    if (kPilotSgn[(symbol_count + (unsigned)gg) % 127u]) th = w16(th + 0x8000);
                    // the frame's four angles in each of its four lanes: quad_perm broadcasts
                    int th1 = __builtin_amdgcn_update_dpp(0, th, 0x00, 0xF, 0xF, true), th2 = __builtin_amdgcn_update_dpp(0, th, 0x55, 0xF, 0xF, true);
                    int th3 = __builtin_amdgcn_update_dpp(0, th, 0xAA, 0xF, 0xF, true), th4 = __builtin_amdgcn_update_dpp(0, th, 0xFF, 0xF, 0xF, true);
                    // The four moves stay moves.  Left alone, hipcc (ROCm 7.2) folds them into the sums below as v_add_u32_dpp / v_subrev_u32_dpp, and on gfx950 that
                    // form gave th2 - th4 where the source says th4 - th2 (measured round 6: one frame in sixteen lost its CRC through the slope's sign; the same
                    // arithmetic through four separate v_mov_b32_dpp, or through ds_bpermute, is bit-exact: DESIGN.md section 3.11).
                    asm volatile("" : "+v"(th1), "+v"(th2), "+v"(th3), "+v"(th4));
                    const int avg = w16((th1 + th2 + th3 + th4) / 4);
                    const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
                    if (act) {
                        cfo_tr = w16(cfo_tr + (avg >> 2)); sfo_tr = w16(sfo_tr + (del >> 2));
                        cfo_comp = w16(cfo_comp + avg + cfo_tr); sfo_comp = w16(sfo_comp + del + sfo_tr);
                    }
                    if (lane < 16 && pk == 0) lds.trk[fr][gg] = int4{ c0, s0c, act ? avg : 0, act ? del : 0 };
                }
            }
            symbol_count = (symbol_count + 4u) % 127u;
            __syncthreads();
            const int4 t = lds.trk[w][g];
            my_cfo = t.x; my_sfo = t.y; my_avg = t.z; my_del = t.w;
        } else {
        int t_cfo[4], t_sfo[4], t_avg[4], t_del[4];
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            t_cfo[gg] = cfo_comp; t_sfo[gg] = sfo_comp; t_avg[gg] = 0; t_del[gg] = 0;
            if (s0 + gg <= nsym) {
                cpx p = mul_q15(unpack(s_eq[w][gg][pbin]), rot_coeff(T, w16(cfo_comp + pc * sfo_comp)));
                int th = pk == 3 ? uatan2(T, -p.im, -p.re) : uatan2(T, p.im, p.re);
                if (kPilotSgn[symbol_count]) th = w16(th + 0x8000);
                symbol_count++; if (symbol_count >= 127) symbol_count = 0;
                const int th1 = __builtin_amdgcn_readlane(th, 0), th2 = __builtin_amdgcn_readlane(th, 1);
                const int th3 = __builtin_amdgcn_readlane(th, 2), th4 = __builtin_amdgcn_readlane(th, 3);
                const int avg = w16((th1 + th2 + th3 + th4) / 4);
                const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
                t_avg[gg] = avg; t_del[gg] = del;
                cfo_tr = w16(cfo_tr + (avg >> 2)); sfo_tr = w16(sfo_tr + (del >> 2));
                cfo_comp = w16(cfo_comp + avg + cfo_tr); sfo_comp = w16(sfo_comp + del + sfo_tr);
            }
        }
        my_cfo = g == 0 ? t_cfo[0] : g == 1 ? t_cfo[1] : g == 2 ? t_cfo[2] : t_cfo[3];
        my_sfo = g == 0 ? t_sfo[0] : g == 1 ? t_sfo[1] : g == 2 ? t_sfo[2] : t_sfo[3];
        my_avg = g == 0 ? t_avg[0] : g == 1 ? t_avg[1] : g == 2 ? t_avg[2] : t_avg[3];
        my_del = g == 0 ? t_del[0] : g == 1 ? t_del[1] : g == 2 ? t_del[2] : t_del[3];
        }
        if (!mine) continue;
        // ---- TPhaseCompensate + TPilotTrack::_rotate + T11aDemap, 3 data carriers per lane
        if (active) {
            cpx c1[3], c2[3];
#pragma unroll
            for (int m = 0; m < 3; m++) {                                        // all six coefficient reads in flight together
                const int bin = carrier_bin48(e + 16 * m);
                const int c = bin < 32 ? bin : bin - 64;
                c1[m] = rot_coeff(T, w16(my_cfo + c * my_sfo));
                c2[m] = rot_coeff(T, w16(my_avg + c * my_del));
            }
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const int k = e + 16 * m;
                cpx v = unpack(s_eq[w][g][carrier_bin48(k)]);
                v = mul_q15(v, c1[m]);
                v = mul_q15(v, c2[m]);
                int re = v.re >> 4, im = v.im >> 4;                               // demap_limit<64> (demapper.h:141-151)
                re = min(max(re, -128), 127); im = min(max(im, -128), 127);
                const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
                uint8_t* o = s_soft[w][g] + k * nb;                               // DemapperCore::Demap<N_BPSC> (demapper.h:16-45)
                if (nb == 1) { o[0] = s_demap[ur]; }
                else if (nb == 2) { o[0] = s_demap[ur]; o[1] = s_demap[ui]; }
                else if (nb == 4) { o[0] = s_demap[ur]; o[1] = s_demap[256 + ur]; o[2] = s_demap[ui]; o[3] = s_demap[256 + ui]; }
                else { o[0] = s_demap[ur]; o[1] = s_demap[512 + ur]; o[2] = s_demap[768 + ur];
                       o[3] = s_demap[ui]; o[4] = s_demap[512 + ui]; o[5] = s_demap[768 + ui]; }
            }
        }
        wsync();
        // ---- T11aDeinterleave*: out[k] = in[j(k)], eight values -> three bytes of the frame's stream (soft3_store8), symbol by symbol
        {
            const int nact = min(4, nsym - s0 + 1);
            const uint32_t sym_bytes = 3u * (uint32_t)ncbps / 8u;
            uint8_t* d = dst + (size_t)(s0 - 1) * sym_bytes;
            for (int gs = 0; gs < nact; gs++, d += sym_bytes) {
                if (packs) {
                    const uint8_t* src = s_soft[w][gs];
                    uint32_t v[8];
#pragma unroll
                    for (int t = 0; t < 4; t++) { v[2 * t] = src[mp[t] & 0xFFFFu]; v[2 * t + 1] = src[mp[t] >> 16]; }
                    soft3_store8(d, (uint32_t)lane, soft3_pack8(v));
                }
            }
        }
        wsync();
    }
}
__global__ void __launch_bounds__(256) k_frame(RxArgs A)
{
    __shared__ FrameLds lds;
    reinterpret_cast<uint32_t*>(lds.demap)[threadIdx.x] = reinterpret_cast<const uint32_t*>(A.T.demap)[threadIdx.x];
    __syncthreads();                                                             // the only block barrier: the waves are independent from here on
    if (!locate_job(blockIdx.x * 4, A.njobs).ok) return;                         // (the whole workgroup)
    const JobRef jr = locate_job(blockIdx.x * 4 + (threadIdx.x >> 6), A.njobs);
#ifdef SORA_DBG_KFRAME_PRIVATE                                                   // (tools variant: every wave tracks its own frame, as before round 6)
    if (jr.ok) frame_symbols<false>(A, jr.list * A.nrows + jr.idx, lds);
#else
    frame_symbols<true>(A, jr.ok ? jr.list * A.nrows + jr.idx : 0u, lds, jr.ok);   // slot of the job in jobs[] / joblist[]; a wave without one still meets the others at the barriers
#endif
}

// ------------------------------------------------------------------------------------------------
// Round 4: the data field's symbol chain as THREE kernels (VERDICT r3 #3c) instead of one wave per frame.
//
// What a symbol needs from its predecessor is four numbers (CFO_comp, SFO_comp and the two trackers of TPilotTrack,
// pilot.hpp:213-232 -> freqoffset.hpp:28); everything else -- TFreqCompensation, TFFT64, TChannelEqualization in front of
// the tracker, the rotation by CompCoeffs and by the pilots' mean phase / slope, T11aDemap and T11aDeinterleave behind
// it -- is per-symbol work.  k_frame ran the tracker's chain (two dependent LUT gathers, ~50 vector + ~40 scalar
// instructions per symbol) on a whole wave for the four lanes that hold a symbol's pilots: 36 % of its vector
// instructions, and a frame's symbols one pass after the other (fsample-6's 465 symbols: 117 passes, 0.30 ms for one wave).
//   k_sym_front  per SYMBOL SLOT (rx_types.h): samples -> equalised bins eq[slot][64] in HBM      (16 lanes per symbol)
//   k_track      per FRAME, four lanes (= the four pilots) each, sixteen frames per wave: the chain over the frame's
//                symbols in order, reading 16 bytes per symbol, writing the rotation parameters track[slot] (TrackRec)
//   k_sym_back   per SYMBOL SLOT: eq[slot] x CompCoeffs x rotation -> demap -> de-interleave -> packed soft stream
// The symbol kernels find a slot's frame through slot_row[] (written by k_scan for the data symbols of every frame it
// queues).  HBM is the idle resource of this path (4.6 % of the roofline in round 3): the equalised symbols cost
// 256 B written + ~210 B read per symbol (59 + 48 MB per 4096-frame call) and buy the tracker's instructions back.
struct SlotOwner { uint32_t row; bool ok; };

// pilot.hpp:10-28 as a bit string (bit i of word i >> 5 = polarity -1 of symbol count i)
__device__ __forceinline__ unsigned pilot_sgn(unsigned count)
{
    const unsigned w = count < 32 ? 0x2049a7b8u : count < 64 ? 0x9836ba32u : count < 96 ? 0xaa16f395u : 0x3f8ec52fu;
    return (w >> (count & 31u)) & 1u;
}

constexpr int kSlotIters = 4;                                                    // quads of slots per wave: 16 consecutive slots

// One quad of slots through TFreqCompensation, TFFT64 and TChannelEqualization: group g's symbol from raw[] (its 64 samples behind the cyclic
// prefix, sample e + 16 m) with the frame's FreqCoeffs / ChannelCoeffs already in their packed-product form; bins 4e .. 4e+3 out.
template <typename SYNC>
__device__ __forceinline__ void sym_front_quad(const uint32_t raw[4], const PkTw fq[4], const PkTw ch[4], uint32_t* sl, int e, const Fft64TwPk& W, SYNC wsync, uint32_t o[4])
{
    pcx x[4];
#pragma unroll
    for (int m = 0; m < 4; m++) x[m] = pk_cmul<15>(pk_sra(raw[m], 1), fq[m]);   // >>1, x FreqCoeffs (channel_11a.hpp:643-644)
    fft64_core_pk(x, sl, e, W, wsync);
    const unsigned rv = __brev((unsigned)e) >> 28;                               // bin 4e+q sits at slot bitrev6(4e+q) = bitrev4(e) + 16 bitrev2(q)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int bin = 4 * e + q;
        o[q] = (bin >= 28 && bin < 36) ? 0u : pk_cmul<8>(sl[rv + 16u * ((q & 1) * 2 + (q >> 1))], ch[q]);   // channel_11a.hpp:548-574
    }
    wsync();
}

// Stores another workgroup of the SAME launch reads (k_pipe): written through to memory (sc1), so that the publishing flag needs no write-back of the XCD's L2 behind it
// (cdna_hip_programming.md, guideline 16, form R1: sc1 payload, every storing wave drains, one relaxed agent-scope flag).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_through(uint4* p, uint4 v)
{
    const u32x4_t x = { v.x, v.y, v.z, v.w };
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void store4_through(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The slots of workgroup `bid` (64 of them, 16 per wave) through TFreqCompensation, TFFT64 and TChannelEqualization.  THROUGH: the results are for workgroups of this launch.
template <bool THROUGH>
__device__ __forceinline__ void sym_front_block(const RxArgs& A, uint32_t bid, uint32_t (*s_eq)[4][64])
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    const Tables& T = A.T;
    const uint32_t slot_first = (bid * 4u + (uint32_t)w) * (4u * kSlotIters);
    if (slot_first >= A.total_slots) return;
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // the owners of the wave's 16 slots: lane l < 16 asks for slot slot_first + l, the groups pick theirs up by cross-lane reads
    const uint32_t my_slot = slot_first + (uint32_t)(lane & 15);
    const uint32_t my_own = (lane < 16 && my_slot < A.total_slots) ? A.slot_row[my_slot] : 0xFFFFFFFFu;
    const unsigned long long owned = __ballot(my_own != 0xFFFFFFFFu);
    if (owned == 0) return;                                                      // preamble / silence only
    const uint32_t row0 = (uint32_t)__builtin_amdgcn_readlane((int)my_own, __builtin_ctzll(owned));
    // the usual case: every owned slot of the wave belongs to ONE frame
    const bool one_frame = __ballot(lane < 16 && my_own != 0xFFFFFFFFu && my_own != row0) == 0;
    uint32_t own[kSlotIters];
#pragma unroll
    for (int it = 0; it < kSlotIters; it++) own[it] = (uint32_t)__shfl((int)my_own, 4 * it + g);
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    uint32_t* sl = s_eq[w][g];
    uint4* eq4 = reinterpret_cast<uint4*>(A.eq);
    auto store = [&](uint32_t slot, const uint32_t o[4]) {
        if (THROUGH) store16_through(eq4 + ((size_t)slot * 16u + (uint32_t)e), uint4{ o[0], o[1], o[2], o[3] });
        else eq4[(size_t)slot * 16u + (uint32_t)e] = uint4{ o[0], o[1], o[2], o[3] };
        // the four pilot bins once more, densely (16 bytes per slot: k_track reads nothing else): bins 43, 57, 7, 21 = (e, q) (10,3), (14,1), (1,3), (5,1)
        auto pilot = [&](uint32_t k, uint32_t v) { if (THROUGH) store4_through(A.pil + ((size_t)slot * 4u + k), v); else A.pil[(size_t)slot * 4u + k] = v; };
        if (e == 10) pilot(0u, o[3]);
        if (e == 14) pilot(1u, o[1]);
        if (e == 1)  pilot(2u, o[3]);
        if (e == 5)  pilot(3u, o[1]);
    };
    if (one_frame) {
        // ---- one frame: its row and its coefficients once per wave, every sample load in flight before the first butterfly
        const FrameRow& r = A.frames[row0];
        const uint32_t* iq = A.iq + A.caps[r.capture].offset;
        const uint32_t ds = r.data_start, s0 = r.slot0, str = A.str;
        const FrameCtx* fx = A.fctx + row0;
        uint32_t raw[kSlotIters][4];
#pragma unroll
        for (int it = 0; it < kSlotIters; it++) {
            const uint32_t p0 = ds + 80u * (slot_first + 4u * it + (uint32_t)g - s0) + 8u;   // skip_cp = 8 (PHY_11a.hpp:365,394)
#pragma unroll
            for (int m = 0; m < 4; m++) raw[it][m] = own[it] != 0xFFFFFFFFu ? iq[(size_t)(p0 + (uint32_t)(e + 16 * m)) * str] : 0u;
        }
        PkTw fq[4], ch[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { fq[m] = pk_tw_mul(fx->freq[e + 16 * m]); ch[m] = pk_tw_mul(fx->chan[4 * e + m]); }
#pragma unroll
        for (int it = 0; it < kSlotIters; it++) {
            if (((owned >> (4 * it)) & 0xFull) == 0) continue;                   // (wave-uniform)
            uint32_t o[4];
            sym_front_quad(raw[it], fq, ch, sl, e, W, wsync, o);
            if (own[it] != 0xFFFFFFFFu) store(slot_first + 4u * it + (uint32_t)g, o);
        }
        return;
    }
    // ---- several frames meet in these 16 slots (the end of one and the start of the next, captures of a few symbols): per group and quad
#pragma unroll 1
    for (int it = 0; it < kSlotIters; it++) {
        if (((owned >> (4 * it)) & 0xFull) == 0) continue;
        const uint32_t ow = (uint32_t)__shfl((int)my_own, 4 * it + g);
        const bool mine = ow != 0xFFFFFFFFu;
        const uint32_t slot = slot_first + 4u * it + (uint32_t)g;
        uint32_t raw[4] = { 0u, 0u, 0u, 0u };
        PkTw fq[4], ch[4];
        const FrameCtx* fx = A.fctx + (mine ? ow : 0u);
        if (mine) {
            const FrameRow& r = A.frames[ow];
            const uint32_t* iq = A.iq + A.caps[r.capture].offset;
            const uint32_t p0 = r.data_start + 80u * (slot - r.slot0) + 8u;
#pragma unroll
            for (int m = 0; m < 4; m++) raw[m] = iq[(size_t)(p0 + (uint32_t)(e + 16 * m)) * A.str];
        }
#pragma unroll
        for (int m = 0; m < 4; m++) { fq[m] = pk_tw_mul(mine ? fx->freq[e + 16 * m] : 0u); ch[m] = pk_tw_mul(mine ? fx->chan[4 * e + m] : 0u); }
        uint32_t o[4];
        sym_front_quad(raw, fq, ch, sl, e, W, wsync, o);
        if (mine) store(slot, o);
    }
}

__global__ void __launch_bounds__(256) k_sym_front(RxArgs A)
{
    __shared__ uint32_t s_eq[4][4][64];                                          // [wave][group]: FFT staging
    sym_front_block<false>(A, blockIdx.x, s_eq);
}

// The loop-carried part (freqoffset.hpp:28-30, pilot.hpp:166-233): pilot k of a frame in lane 4 f + k, sixteen frames per wave, every
// frame stepping through its own symbols; the four angles of a frame meet through quad broadcasts.
__global__ void __launch_bounds__(256) k_track(RxArgs A)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, pk = lane & 3;
    const Tables& T = A.T;
    const JobRef jr = locate_job((blockIdx.x * 4u + (uint32_t)w) * 16u + (uint32_t)(lane >> 2), A.njobs);
    const uint32_t j = jr.ok ? jr.list * A.nrows + jr.idx : 0u;
    const uint32_t f = jr.ok ? A.joblist[j] : 0u;
    FrameRow r = A.frames[f];
    const int nsym = jr.ok ? (int)r.nsym : 0;
    if (jr.ok && pk == 0) {
        VitJob J;
        J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftBytesPerSlot; J.nsoft = (uint32_t)r.nsym * 48u * r.nbpsc; J.length = r.length;
        J.dec_off = 0; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate; J.soft_bits = 3;
        A.jobs[j] = J;
    }
    // pilot k in lane k: bins 43, 57, 7, 21 = carriers -21, -7, +7, +21 (pilot.hpp:138-164)
    const int pc = pk == 0 ? -21 : pk == 1 ? -7 : pk == 2 ? 7 : 21;
    int cfo_comp = r.cfo_comp, sfo_comp = r.sfo_comp, cfo_tr = r.cfo_tracker, sfo_tr = r.sfo_tracker;
    unsigned symbol_count = 0;                                                   // 127 -> 0 after the SIGNAL symbol
    int nmax = nsym;
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    // pilot k of data symbol s at pp[4 (s - 1)]: 16 bytes per symbol and frame (k_sym_front)
    const uint32_t* pp = A.pil + (size_t)(r.slot0 + 1u) * 4u + (uint32_t)pk;
    TrackRec* trk = A.track + r.slot0 + 1u;
    // symbols requested ahead of the one in the chain (each step is two dependent table reads long)
    constexpr int kAhead = 4;
    uint32_t q[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; i++) q[i] = i < nsym ? pp[4 * i] : 0u;
    for (int s = 1; s <= nmax; s++) {
        const uint32_t cur = q[0];
#pragma unroll
        for (int i = 0; i + 1 < kAhead; i++) q[i] = q[i + 1];
        q[kAhead - 1] = s + kAhead <= nsym ? pp[4 * (s + kAhead - 1)] : 0u;
        if (s <= nsym) {                                                         // (uniform inside a quad: the cross-lane reads below see their whole quad)
            const cpx p = mul_q15(unpack(cur), rot_coeff(T, w16(cfo_comp + pc * sfo_comp)));
            int th = pk == 3 ? uatan2(T, -p.im, -p.re) : uatan2(T, p.im, p.re);
            if (pilot_sgn(symbol_count)) th = w16(th + 0x8000);
            symbol_count++; if (symbol_count >= 127) symbol_count = 0;
            // The four angles of the quad, in every lane.  Written as assembler on purpose: with __builtin_amdgcn_update_dpp the compiler folds two of
            // the broadcasts into the arithmetic that follows (v_add_u32_dpp / v_subrev_u32_dpp writing the register it reads through the DPP
            // selector) and `del` comes out a few LSB off -- reproduced in round 4 (tools/dbg_arrays.py), the same fault round 3 noted in k_frame.
            // (s_nop 1: a VALU write of th followed by a DPP read needs two wait states, and the hazard pass does not look into assembler.)
            int th1, th2, th3, th4;
            asm volatile("s_nop 1\n\t"
                         "v_mov_b32_dpp %0, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %2, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %3, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(th1), "=&v"(th2), "=&v"(th3), "=&v"(th4) : "v"(th));
            const int avg = w16((th1 + th2 + th3 + th4) / 4);
            const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
#ifdef SORA_DBG_TRACK_TH
            if (pk == 0) { TrackRec t; t.cfo_comp = (int16_t)th1; t.sfo_comp = (int16_t)th2; t.avg = (int16_t)th3; t.del = (int16_t)th4; trk[s - 1] = t; }
#else
            if (pk == 0) { TrackRec t; t.cfo_comp = (int16_t)cfo_comp; t.sfo_comp = (int16_t)sfo_comp; t.avg = (int16_t)avg; t.del = (int16_t)del; trk[s - 1] = t; }
#endif
            cfo_tr = w16(cfo_tr + (avg >> 2)); sfo_tr = w16(sfo_tr + (del >> 2));
            cfo_comp = w16(cfo_comp + avg + cfo_tr); sfo_comp = w16(sfo_comp + del + sfo_tr);
        }
    }
}

// One symbol of the chain out of the folded tables in LDS (freqoffset.hpp:28-30, pilot.hpp:166-233): pilot k of the frame in lane k of a quad, `cur` its equalised bin, `flip`
// 0x8000 for a symbol of pilot polarity -1.  Advances the state and returns the symbol's TrackRec { cfo_comp, sfo_comp, avg, del } as two words.
__device__ __forceinline__ uint2 track_step(const TrkTables& s_t, uint32_t cur, int flip, int pc, int m3, int& cfo, int& sfo, int& ctr, int& str)
{
    // 0x3D124925: trunc((float)d * kInv28) == d / 28 (C division) for every |d| <= 65535
    constexpr float kInv28 = 0.0357142873108387f;
    // rot_coeff(cfo + pc sfo) = (ucos, -usin) out of the quarter wave
    const unsigned a = (unsigned)(cfo + pc * sfo) & 0xFFFFu;
    const int qi = trk_quarter_index(a);
    const int sraw = s_t.q[qi], craw = s_t.q[16384 - qi];
    const uint32_t ws = s_t.e2s[a >> 4], wc = s_t.e2c[a >> 4];
    const int ms = __builtin_amdgcn_sbfe((int)a, 15, 1), mc = __builtin_amdgcn_sbfe((int)(a ^ (a << 1)), 15, 1);
    const int sn = ((sraw ^ ms) - ms) + trk_sext2(ws, a), cs = ((craw ^ mc) - mc) + trk_sext2(wc, a);
    // p = pilot x (cs, -sn), Q15 with a wrapping pack (vector128.h:1201-1211)
    const int pr = (int)(short)cur, pi = (int)cur >> 16;
    const int re = __builtin_amdgcn_sbfe(pr * cs + pi * sn, 15, 16), im = __builtin_amdgcn_sbfe(pi * cs - pr * sn, 15, 16);
    const int x = (re ^ m3) - m3, y = (im ^ m3) - m3;
    // uatan2 (intalg.h:100-113): the larger magnitude's top bit to bit 6, then the table
    const int sh = max(25 - __builtin_clz((unsigned)(max(x, -x) | max(y, -y)) | 1u), 0);
    int th = trk_uatan2_entry(s_t, y >> sh, x >> sh);
    th = __builtin_amdgcn_sbfe(th ^ flip, 0, 16);                                // + 0x8000 mod 2^16 for a pilot of polarity -1
    int th1, th2, th3, th4;                                                      // the four angles of the quad, in every lane (assembler: see k_track)
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %2, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %3, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "=&v"(th1), "=&v"(th2), "=&v"(th3), "=&v"(th4) : "v"(th));
    const int sum = th1 + th2 + th3 + th4;
    const int avg = (sum + ((sum >> 31) & 3)) >> 2;                              // (th1 + th2 + th3 + th4) / 4, towards zero: within 16 bits
    const int del = ((int)((float)(th3 - th1) * kInv28) + (int)((float)(th4 - th2) * kInv28)) >> 1;
    // the record's halves by one byte permute each (low half of the second operand | low half of the first << 16)
    const uint2 rec = uint2{ __builtin_amdgcn_perm((uint32_t)sfo, (uint32_t)cfo, 0x05040100u), __builtin_amdgcn_perm((uint32_t)del, (uint32_t)avg, 0x05040100u) };
    ctr += avg >> 2; str += del >> 2;
    // cfo runs free like the trackers (it is only ever added, and read modulo 2^16: unsigned, so that the wrap of 32 bits is defined); sfo is a factor of the 24-bit product above
    cfo = (int)((unsigned)cfo + (unsigned)avg + (unsigned)ctr); sfo = __builtin_amdgcn_sbfe(sfo + del + str, 0, 16);
    return rec;
}

// The tracker's tables into a workgroup's LDS (256 threads; the caller's barrier follows).
__device__ __forceinline__ void track_tables_to_lds(const uint32_t* __restrict__ trk, TrkTables& s_t, uint32_t* s_pol)
{
    if (threadIdx.x < 127) {
        unsigned b = 0;
#pragma unroll
        for (unsigned u = 0; u < 8; u++) b |= pilot_sgn((threadIdx.x + u) % 127u) << u;
        s_pol[threadIdx.x] = b;
    }
    // 129 KB from L2 / HBM: eleven 16-byte loads per thread in flight at a time (a loop that waits for every load costs a memory latency per 4 KB: 33 of them)
    const uint4* src = reinterpret_cast<const uint4*>(trk);
    uint4* dst = reinterpret_cast<uint4*>(&s_t);
    constexpr uint32_t kWords = sizeof(TrkTables) / 16, kBatch = 11;
    for (uint32_t i0 = threadIdx.x; i0 < kWords; i0 += 256u * kBatch) {
        uint4 v[kBatch];
#pragma unroll
        for (uint32_t b = 0; b < kBatch; b++) v[b] = src[min(i0 + 256u * b, kWords - 1u)];
#pragma unroll
        for (uint32_t b = 0; b < kBatch; b++) if (i0 + 256u * b < kWords) dst[i0 + 256u * b] = v[b];
    }
}

// Round 5: the same chain with its three tables in LDS -- the tracker for FEW frames in flight (a single capture: fsample-6's 465 symbols).
// k_track's step is two dependent L2 gathers long (rot[] is 256 KB, uatan2[] 128 KB: 0.7 us per symbol); here a workgroup first copies the folded tables
// (dev_arith.h TrkTables, 129 KB: quarter-wave sine + two-bit corrections, uatan2 for y >= 0) into its LDS, and the step is three LDS reads deep.  One wave is
// alone on its SIMD, so the step costs what its instructions cost to ISSUE (~5 cycles each): the loop is written for few instructions, not for few dependent ones --
// no exec-mask juggling (frames shorter than the wave's longest keep stepping on clamped reads and write nothing), the pilot polarity a scalar, the state wrapped
// only where a product needs it, the two divisions by 28 as float products (exact for every difference of two 16-bit angles: tests/test_track_division.py).
// One workgroup serves up to 64 frames (four lanes each); the tables are exact by the host's exhaustive check (sora_hip.cpp: trk_tables_exact).
__global__ void __launch_bounds__(256) k_track_lds(RxArgs A)
{
    __shared__ TrkTables s_t;
    // the pilot polarities (pilot.hpp:10-28, period 127) of the eight symbols from count c on, one bit each
    __shared__ uint32_t s_pol[128];
    track_tables_to_lds(A.T.trk, s_t, s_pol);
    __syncthreads();
    const int lane = threadIdx.x & 63, pk = lane & 3;
    const JobRef jr = locate_job(blockIdx.x * 64u + (threadIdx.x >> 2), A.njobs);
    const uint32_t j = jr.ok ? jr.list * A.nrows + jr.idx : 0u;
    const uint32_t f = jr.ok ? A.joblist[j] : 0u;
    const FrameRow r = A.frames[f];
    const int nsym = jr.ok ? (int)r.nsym : 0;
    if (jr.ok && pk == 0) {
        VitJob J;
        J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftBytesPerSlot; J.nsoft = (uint32_t)r.nsym * 48u * r.nbpsc; J.length = r.length;
        J.dec_off = 0; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate; J.soft_bits = 3;
        A.jobs[j] = J;
    }
    const int pc = pk == 0 ? -21 : pk == 1 ? -7 : pk == 2 ? 7 : 21;             // pilot k in lane k: carriers -21, -7, +7, +21 (pilot.hpp:138-164)
    const int m3 = pk == 3 ? -1 : 0;                                             // the fourth pilot's angle is taken of -p (pilot.hpp:166-233)
    // cfo / sfo wrapped to 16 bits after every step; the trackers run free (they are only ever added)
    int cfo = r.cfo_comp, sfo = r.sfo_comp, ctr = r.cfo_tracker, str = r.sfo_tracker;
    int nmax = nsym;
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    // pilot k of data symbol s is word 4 (slot0 + s) + k of pil[] (k_sym_front); a record goes to track[slot0 + s].  32-bit word offsets from the arrays' bases, so that a
    // request is a minimum, a shift-add and the load.  Symbols past the frame's last re-read its last pilots and their records land on the slot behind the frame --
    // preamble or silence of whatever follows, a slot no frame owns and k_sym_back never reads.
    // BYTE offsets: a uniform base plus a 32-bit lane offset is one address operand pair
    const uint32_t w0 = (jr.ok ? (r.slot0 + 1u) * 4u + (uint32_t)pk : (uint32_t)pk) * 4u;
    const uint32_t t0 = (jr.ok ? r.slot0 + 1u : A.total_slots + 1u) * 8u;        // (no job: a slot in the arrays' slack)
    const char* __restrict__ pil = reinterpret_cast<const char*>(A.pil);
    char* __restrict__ trk2 = reinterpret_cast<char*>(A.track);
    const unsigned last = (unsigned)max(nsym, 1) - 1u, nrec = (unsigned)nsym;
    // symbols requested ahead of the one in the chain (a step is ~0.1 us, an L2 miss ten times that)
    constexpr int kAhead = 8;
    uint32_t q[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; i++) q[i] = *reinterpret_cast<const uint32_t*>(pil + (w0 + 16u * min((unsigned)i, last)));
    // symbol_count: 127 -> 0 after the SIGNAL symbol; the same in every frame of the wave
    unsigned cnt = 0;
    for (int s0 = 1; s0 <= nmax; s0 += kAhead) {
        const unsigned pol = (unsigned)__builtin_amdgcn_readfirstlane((int)s_pol[cnt]);   // the polarities of the block's eight symbols, one scalar byte
        cnt = cnt + (unsigned)kAhead >= 127u ? cnt + (unsigned)kAhead - 127u : cnt + (unsigned)kAhead;
#pragma unroll
        for (int u = 0; u < kAhead; u++) {                                       // (unrolled: the request ring's slots are registers)
            const int s = s0 + u;
            const uint32_t cur = q[u];
            q[u] = *reinterpret_cast<const uint32_t*>(pil + (w0 + 16u * min((unsigned)(s + kAhead - 1), last)));
            const int flip = (int)((pol << (15 - u)) & 0x8000u);
            // TrackRec { cfo_comp, sfo_comp, avg, del } of THIS symbol: the quad's four lanes store the same eight bytes (no exec juggling)
            *reinterpret_cast<uint2*>(trk2 + (t0 + 8u * min((unsigned)(s - 1), nrec))) = track_step(s_t, cur, flip, pc, m3, cfo, sfo, ctr, str);
        }
    }
}

// TPhaseCompensate + TPilotTrack::_rotate + T11aDemap for one group's symbol (3 data carriers per lane, 16 lanes per symbol): the three bins v3
// (carriers e, e + 16, e + 32 in demap order) x CompCoeffs(cfo, sfo) x rotation(avg, del) -> soft values in carrier order at `dst`.
__device__ __forceinline__ void sym_back_demap(const Tables& T, const uint8_t* s_demap, const uint32_t v3[3], TrackRec tr, int nb, int e, uint8_t* dst)
{
    cpx c1[3], c2[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {                                                // all six coefficient reads in flight together
        const int bin = carrier_bin48(e + 16 * m);
        const int c = bin < 32 ? bin : bin - 64;
        c1[m] = rot_coeff(T, w16((int)tr.cfo_comp + c * (int)tr.sfo_comp));
        c2[m] = rot_coeff(T, w16((int)tr.avg + c * (int)tr.del));
    }
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int k = e + 16 * m;
        cpx v = mul_q15(unpack(v3[m]), c1[m]);
        v = mul_q15(v, c2[m]);
        int re = v.re >> 4, im = v.im >> 4;                                       // demap_limit<64> (demapper.h:141-151)
        re = min(max(re, -128), 127); im = min(max(im, -128), 127);
        const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
        uint8_t* o = dst + k * nb;                                                // DemapperCore::Demap<N_BPSC> (demapper.h:16-45)
        if (nb == 1) { o[0] = s_demap[ur]; }
        else if (nb == 2) { o[0] = s_demap[ur]; o[1] = s_demap[ui]; }
        else if (nb == 4) { o[0] = s_demap[ur]; o[1] = s_demap[256 + ur]; o[2] = s_demap[ui]; o[3] = s_demap[256 + ui]; }
        else { o[0] = s_demap[ur]; o[1] = s_demap[512 + ur]; o[2] = s_demap[768 + ur];
               o[3] = s_demap[ui]; o[4] = s_demap[512 + ui]; o[5] = s_demap[768 + ui]; }
    }
}

// ... then T11aDeinterleave + the packed three-bit stream, a symbol at a time across the wave (eight values -> three bytes per lane).
__global__ void __launch_bounds__(256) k_sym_back(RxArgs A)
{
    __shared__ uint8_t s_soft[4][4][288];                                        // [wave][group]: soft values in carrier order
    __shared__ uint8_t s_demap[1024];                                            // DemapperCore step tables
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    const Tables& T = A.T;
    reinterpret_cast<uint32_t*>(s_demap)[threadIdx.x] = reinterpret_cast<const uint32_t*>(T.demap)[threadIdx.x];
    __syncthreads();                                                             // the only block barrier: the waves are independent from here on
    const uint32_t slot_first = (blockIdx.x * 4u + (uint32_t)w) * (4u * kSlotIters);
    if (slot_first >= A.total_slots) return;
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    const uint32_t my_slot = slot_first + (uint32_t)(lane & 15);
    const uint32_t my_own = (lane < 16 && my_slot < A.total_slots) ? A.slot_row[my_slot] : 0xFFFFFFFFu;
    const unsigned long long owned = __ballot(my_own != 0xFFFFFFFFu);
    if (owned == 0) return;
    const uint32_t row0 = (uint32_t)__builtin_amdgcn_readlane((int)my_own, __builtin_ctzll(owned));
    const bool one_frame = __ballot(lane < 16 && my_own != 0xFFFFFFFFu && my_own != row0) == 0;
    int bins[3];
#pragma unroll
    for (int m = 0; m < 3; m++) bins[m] = carrier_bin48(e + 16 * m);
    if (one_frame) {
        // ---- one frame: modulation, stream position and de-interleaver entries once per wave; the loads of all four quads up front
        const FrameRow& r = A.frames[row0];
        const int nb = __builtin_amdgcn_readfirstlane((int)r.nbpsc), ncbps = 48 * nb;
        const uint32_t slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.slot0), sym_bytes = 3u * (uint32_t)ncbps / 8u;
        uint8_t* stream = A.soft + (size_t)slot0 * kSoftBytesPerSlot;
        const bool packs = 8 * lane < ncbps;
        uint32_t mp[4];
        {
            const uint16_t* map = T.deint + (nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3) * 288;
#pragma unroll
            for (int t = 0; t < 4; t++) mp[t] = packs ? (uint32_t)map[8 * lane + 2 * t] | ((uint32_t)map[8 * lane + 2 * t + 1] << 16) : 0u;
        }
        TrackRec tr[kSlotIters]; uint32_t v3[kSlotIters][3];
#pragma unroll
        for (int it = 0; it < kSlotIters; it++) {
            const uint32_t slot = slot_first + 4u * it + (uint32_t)g;
            const bool mine = (owned >> (4 * it + g)) & 1ull;
            tr[it] = mine ? A.track[slot] : TrackRec{ 0, 0, 0, 0 };
#pragma unroll
            for (int m = 0; m < 3; m++) v3[it][m] = mine ? A.eq[(size_t)slot * 64u + (uint32_t)bins[m]] : 0u;
        }
#pragma unroll
        for (int it = 0; it < kSlotIters; it++) {
            const unsigned quad = (unsigned)(owned >> (4 * it)) & 0xFu;
            if (quad == 0) continue;                                             // (wave-uniform)
            if ((quad >> g) & 1u) sym_back_demap(T, s_demap, v3[it], tr[it], nb, e, s_soft[w][g]);
            wsync();
#pragma unroll
            for (int gs = 0; gs < 4; gs++) {
                if (!((quad >> gs) & 1u) || !packs) continue;
                const uint32_t sg = slot_first + 4u * it + (uint32_t)gs;         // data symbol sg - slot0 of the frame
                const uint8_t* src = s_soft[w][gs];
                uint32_t v[8];
#pragma unroll
                for (int t = 0; t < 4; t++) { v[2 * t] = src[mp[t] & 0xFFFFu]; v[2 * t + 1] = src[mp[t] >> 16]; }
                soft3_store8(stream + (size_t)(sg - slot0 - 1u) * sym_bytes, (uint32_t)lane, soft3_pack8(v));
            }
            wsync();
        }
        return;
    }
    // ---- several frames meet in these 16 slots: per group and quad
    // de-interleaver entries of output positions 8 lane .. 8 lane + 7 for modulation cur_nb
    int cur_nb = 0; uint32_t mp[4] = { 0, 0, 0, 0 };
#pragma unroll 1
    for (int it = 0; it < kSlotIters; it++) {
        const unsigned quad = (unsigned)(owned >> (4 * it)) & 0xFu;
        if (quad == 0) continue;
        const uint32_t ow = (uint32_t)__shfl((int)my_own, 4 * it + g);
        const bool mine = ow != 0xFFFFFFFFu;
        const uint32_t slot = slot_first + 4u * it + (uint32_t)g;
        int nb = 0; uint32_t slot0 = 0;
        if (mine) {
            const FrameRow& r = A.frames[ow];
            nb = r.nbpsc; slot0 = r.slot0;
            uint32_t v3[3];
#pragma unroll
            for (int m = 0; m < 3; m++) v3[m] = A.eq[(size_t)slot * 64u + (uint32_t)bins[m]];
            sym_back_demap(T, s_demap, v3, A.track[slot], nb, e, s_soft[w][g]);
        }
        wsync();
#pragma unroll 1
        for (int gs = 0; gs < 4; gs++) {
            if (!((quad >> gs) & 1u)) continue;
            const int nbg = __shfl(nb, 16 * gs);
            const uint32_t slot0g = (uint32_t)__shfl((int)slot0, 16 * gs);
            const int ncbps = 48 * nbg;
            if (nbg != cur_nb) {
                cur_nb = nbg;
                const uint16_t* map = T.deint + (nbg == 1 ? 0 : nbg == 2 ? 1 : nbg == 4 ? 2 : 3) * 288;
                const bool packs = 8 * lane < ncbps;
#pragma unroll
                for (int t = 0; t < 4; t++) mp[t] = packs ? (uint32_t)map[8 * lane + 2 * t] | ((uint32_t)map[8 * lane + 2 * t + 1] << 16) : 0u;
            }
            if (8 * lane < ncbps) {
                const uint32_t sg = slot_first + 4u * it + (uint32_t)gs;
                uint8_t* d = A.soft + (size_t)slot0g * kSoftBytesPerSlot + (size_t)(sg - slot0g - 1u) * (3u * (uint32_t)ncbps / 8u);
                const uint8_t* src = s_soft[w][gs];
                uint32_t v[8];
#pragma unroll
                for (int t = 0; t < 4; t++) { v[2 * t] = src[mp[t] & 0xFFFFu]; v[2 * t + 1] = src[mp[t] >> 16]; }
                soft3_store8(d, (uint32_t)lane, soft3_pack8(v));
            }
        }
        wsync();
    }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the data field of a HANDFUL of frames (a single capture: BASELINE configs[1]) as ONE launch -- k_pipe.
//
// Behind one another, k_sym_front -> k_track_lds -> k_sym_back -> k_viterbi16w cost fsample-6 9 + 112 + 10 + 34 us plus three kernel boundaries, and all but the
// tracker's 112 us (a serial chain: DESIGN.md section 3.8) is work that could run BESIDE it: a symbol's soft values can be made as soon as the chain has passed
// it, a trace-back window's unit can be decoded as soon as its soft values exist.  k_pipe is those four kernels' code in one grid whose workgroups take ROLES
// by their index and hand their results on INSIDE the launch:
//   [0, nfront)                 sym_front_block: 64 symbol slots each; eq[] / pil[] written through (sc1), then ONE flag per workgroup
//   [nfront, nfront + ntrack)   one frame each.  Wave 0 runs the chain (the frame's pilots and the records it produces are in LDS: no address arithmetic, no
//                               memory latency in the loop); waves 1-3 follow it through an LDS counter, four symbols at a time in turn: rotate, demap,
//                               de-interleave, pack, write the quad's soft bytes through (sc1 dwords) and publish a count per wave
//   the rest                    four waves each, a wave = eight units of the window-parallel trellis (dev_vitwin.h) -- it works out its units, waits until the
//                               three counters of its frames say that every soft value it will read has been written, acquires, and decodes
// Waiting only ever looks at a workgroup of a LOWER role, the launch is small enough for every workgroup to be resident at once (sora_hip.cpp: pipe_fits -- 160 KB
// of LDS each, one per CU, at most ~190 of the 256), and every wait is bounded (PipeArgs::wait_ticks): a wait that expires sets flags[0] and gives up, and the finishing
// kernel behind this launch (k_win_redo_finish_pipe) then makes the whole data field again with the plain chain's code
// -- a call never delivers anything but the reference's rows.  Hand-offs follow cdna_hip_programming.md guideline 16.
// The proof of the units and the serial decode of what fails it stay a kernel of their own behind this one (k_win_redo), then k_finish.
// pilots kept in LDS: 1366 data symbols (4095 bytes at 6 Mbps) + the chain's overshoot to a multiple of eight
constexpr uint32_t kPipeMaxSym = 1376;
constexpr uint32_t kPipeRing = 512;                                              // records kept in LDS (a ring: the helpers are a few symbols behind the chain)
struct PipeTrackLds {
    TrkTables t;
    uint32_t pol[128];
    uint32_t pil[kPipeMaxSym][4];
    uint2 rec[kPipeRing];
    uint8_t demap[1024];
    // [helper][symbol of the quad]: soft values in carrier order, then the quad's packed bytes on their way out
    uint8_t soft[3][4][288];
    uint32_t done;                                                               // symbols the chain has passed
    uint32_t next_quad[3];                                                       // the quad each helper works on (all below the smallest are finished)
    uint32_t give_up;
    uint32_t pad[11];
};
constexpr uint32_t kPipeLdsBytes = sizeof(PipeTrackLds);
static_assert(kPipeLdsBytes <= 160 * 1024 && sizeof(TrkTables) % 16 == 0, "k_pipe: a tracker workgroup's LDS");
static_assert(4 * sizeof(Lds16<256, 24>) <= kPipeLdsBytes, "k_pipe: four trellis waves' LDS");

#ifdef SORA_DBG_PIPE_TIMELINE                                                    // tools/pipe_timeline.py: 10 ns stamps of the launch's hand-offs, behind the hand-off words
#define PIPE_STAMP(P, i) do { if ((threadIdx.x & 63) == 0) (P).flags[(P).stamp_base + (i)] = (uint32_t)wall_clock64(); } while (0)
#else
#define PIPE_STAMP(P, i) do {} while (0)
#endif
__device__ __forceinline__ uint32_t flag_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool pipe_expired(long long t0, uint32_t wait_ticks) { return wall_clock64() - t0 > (long long)wait_ticks; }   // ticks of the 100 MHz counter

// ---- role 2: one frame's chain and everything behind it up to the soft stream
__device__ __forceinline__ void pipe_track_block(const RxArgs& A, const PipeArgs& P, uint32_t t, PipeTrackLds& L)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const JobRef jr = locate_job(t, A.njobs);
    if (!jr.ok) return;                                                          // (the whole workgroup)
    track_tables_to_lds(A.T.trk, L.t, L.pol);                                    // (its loads are in flight while wave 0 waits for the front workgroups)
    reinterpret_cast<uint32_t*>(L.demap)[threadIdx.x] = reinterpret_cast<const uint32_t*>(A.T.demap)[threadIdx.x];
    const uint32_t j = jr.list * A.nrows + jr.idx, f = A.joblist[j];
    const FrameRow r = A.frames[f];
    const uint32_t nsym = min((uint32_t)r.nsym, kPipeMaxSym - 8u), slot0 = r.slot0;
    if (threadIdx.x == 0) {
        VitJob J;
        J.valid = 1; J.soft_off = slot0 * (uint32_t)kSoftBytesPerSlot; J.nsoft = (uint32_t)r.nsym * 48u * r.nbpsc; J.length = r.length;
        J.dec_off = 0; J.out_off = slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate; J.soft_bits = 3;
        // (for the kernels behind this launch; the trellis waves of this one work it out themselves)
        A.jobs[j] = J;
        L.done = 0; L.give_up = 0; L.next_quad[0] = 0; L.next_quad[1] = 1; L.next_quad[2] = 2;
    }
    if (w == 0) {
        // the front workgroups that hold this frame's slots: one flag per lane (a frame is at most 22 of them)
        const uint32_t b0 = (slot0 + 1u) >> 6, b1 = (slot0 + max(nsym, 1u)) >> 6;
        const uint32_t* fl = P.flags + 4u + 4u * A.nrows + b0 + (uint32_t)lane;
        const bool mine = b0 + (uint32_t)lane <= b1;
        const long long t0 = wall_clock64();
        bool gave_up = false;
        while (!__all(!mine || flag_load(fl) != 0u)) {
            if (pipe_expired(t0, P.wait_ticks)) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (gave_up && lane == 0) { L.give_up = 1; atomicOr(P.flags, 1u); }
        PIPE_STAMP(P, 1);
    }
    __syncthreads();
    if (L.give_up) return;
    // the frame's pilots (16 bytes per symbol: k_sym_front's dense copy) -> LDS, by loads
    // that pass this CU's L1 (sc1: the acquire that plain loads would need is ~1.7 us
    {
        // in front of the chain; the helpers, who read eq[] with plain loads, make theirs beside it); the overshoot reads zeros
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(A.pil) + 2u * (size_t)(slot0 + 1u);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(&L.pil[0][0]);
        for (uint32_t i = threadIdx.x; i < 2u * (nsym + 8u); i += 256u) dst[i] = i < 2u * nsym ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    __syncthreads();
    if (w == 0) {
        // ---- the chain (k_track_lds's step), lanes 0-3
        __builtin_amdgcn_s_setprio(3);
        PIPE_STAMP(P, 2);
        if (lane < 4) {
            const int pk = lane & 3;
            const int pc = pk == 0 ? -21 : pk == 1 ? -7 : pk == 2 ? 7 : 21, m3 = pk == 3 ? -1 : 0;
            int cfo = r.cfo_comp, sfo = r.sfo_comp, ctr = r.cfo_tracker, str = r.sfo_tracker;
            unsigned cnt = 0;
            volatile uint32_t* done = &L.done; volatile uint32_t* nq = L.next_quad;
            for (uint32_t s0 = 1; s0 <= nsym; s0 += 8u) {
                // the records about to be overwritten must have been used (frames of more than 512 symbols only)
                if (s0 + 7u > kPipeRing) {
                    while (4u * min(min(nq[0], nq[1]), nq[2]) + kPipeRing < s0 + 7u) __builtin_amdgcn_s_sleep(1);
                }
                const unsigned pol = (unsigned)__builtin_amdgcn_readfirstlane((int)L.pol[cnt]);
                cnt = cnt + 8u >= 127u ? cnt + 8u - 127u : cnt + 8u;
                uint32_t q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) q[u] = L.pil[s0 - 1u + (uint32_t)u][pk];
                const uint32_t ring = (s0 - 1u) & (kPipeRing - 1u);
#pragma unroll
                for (int u = 0; u < 8; u++) L.rec[ring + (uint32_t)u] = track_step(L.t, q[u], (int)((pol << (15 - u)) & 0x8000u), pc, m3, cfo, sfo, ctr, str);
                asm volatile("" ::: "memory");                                   // (the records before the count: a wave's LDS accesses are served in order)
                if (lane == 0) *done = min(s0 + 7u, nsym);
            }
        }
        PIPE_STAMP(P, 3);
        return;
    }
    // ---- the helpers: TPhaseCompensate + the pilots' rotation + T11aDemap + T11aDeinterleave + the packed stream, quad by quad behind the chain
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                           // (eq[] was written by the front workgroups of this launch)
    const int h = w - 1, g = lane >> 4, e = lane & 15;
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    const int nb = __builtin_amdgcn_readfirstlane((int)r.nbpsc), ncbps = 48 * nb;
    const uint32_t sym_bytes = 3u * (uint32_t)ncbps / 8u;
    uint8_t* stream = A.soft + (size_t)slot0 * kSoftBytesPerSlot;
    const bool packs = 8 * lane < ncbps;
    uint32_t mp[4];
    {
        const uint16_t* map = A.T.deint + (nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3) * 288;
#pragma unroll
        for (int tt = 0; tt < 4; tt++) mp[tt] = packs ? (uint32_t)map[8 * lane + 2 * tt] | ((uint32_t)map[8 * lane + 2 * tt + 1] << 16) : 0u;
    }
    int bins[3];
#pragma unroll
    for (int m = 0; m < 3; m++) bins[m] = carrier_bin48(e + 16 * m);
    volatile uint32_t* done = &L.done; volatile uint32_t* nq = L.next_quad;
    uint32_t* my_count = P.flags + 4u + 4u * f + (uint32_t)h;
    uint32_t quads = 0;
    // (the quad's packed bytes reuse the helper's soft values' place once those are gathered)
    uint8_t* bytes = &L.soft[h][0][0];
    for (uint32_t k = (uint32_t)h; 4u * k < nsym; k += 3u) {
        const uint32_t nact = min(4u, nsym - 4u * k), s = 4u * k + 1u + (uint32_t)g;   // group g's symbol (1-based)
        const bool mine = (uint32_t)g < nact;
        uint32_t v3[3];
#pragma unroll
        for (int m = 0; m < 3; m++) v3[m] = mine ? A.eq[(size_t)(slot0 + s) * 64u + (uint32_t)bins[m]] : 0u;   // (in flight while the chain gets there)
        while (*done < 4u * k + nact) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const uint2 rw = L.rec[(s - 1u) & (kPipeRing - 1u)];
        TrackRec tr;
        tr.cfo_comp = (int16_t)rw.x; tr.sfo_comp = (int16_t)(rw.x >> 16); tr.avg = (int16_t)rw.y; tr.del = (int16_t)(rw.y >> 16);
        if (mine) sym_back_demap(A.T, L.demap, v3, tr, nb, e, L.soft[h][g]);
        wsync();
        uint32_t b24[4];
#pragma unroll
        for (int gs = 0; gs < 4; gs++) {
            const uint8_t* src = L.soft[h][gs];
            uint32_t v[8];
#pragma unroll
            for (int tt = 0; tt < 4; tt++) { v[2 * tt] = src[mp[tt] & 0xFFFFu]; v[2 * tt + 1] = src[mp[tt] >> 16]; }
            b24[gs] = soft3_pack8(v);
        }
        wsync();
        if (packs) {
#pragma unroll
            for (int gs = 0; gs < 4; gs++) soft3_store8(bytes + (uint32_t)gs * sym_bytes, (uint32_t)lane, b24[gs]);
        }
        wsync();
        // the quad's bytes (a multiple of eight from a multiple-of-four address: 72 N_BPSC per quad; a last, shorter quad is rounded up into the frame's own spare slot)
        const uint32_t ndw = (nact * sym_bytes + 3u) / 4u;
        uint32_t* out32 = reinterpret_cast<uint32_t*>(stream + (size_t)(4u * k) * sym_bytes);
        for (uint32_t i = (uint32_t)lane; i < ndw; i += 64u) store4_through(out32 + i, reinterpret_cast<const uint32_t*>(bytes)[i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wsync();
        quads++;
        if (lane == 0) { store4_through(my_count, quads); nq[h] = k + 3u; }
    }
    if (lane == 0) nq[h] = 0x3FFFFFFFu;
    PIPE_STAMP(P, 4 + h);
}

// A trellis wave in front of its first soft value: wait until the three counts of its frames cover everything its units will read (M: this lane's unit; polls: whether
// this lane is the one that looks for it), then acquire.  false: the wait ran into its bound.
__device__ __forceinline__ bool pipe_units_ready(const RxArgs& A, const PipeArgs& P, uint32_t list, const UnitGeom& M, bool polls_, uint32_t wave_index)
{
    const unsigned lane = threadIdx.x & 63;
    const bool polls = polls_ && M.valid;
    const uint32_t f = A.joblist[(size_t)list * A.nrows + M.idx];
    const uint32_t ncbps = 48u * A.frames[f].nbpsc;
    const uint32_t quads = polls ? ((M.need + ncbps - 1u) / ncbps + 3u) / 4u : 0u;   // quads of symbols that hold the unit's soft values
    const uint32_t* c = P.flags + 4u + 4u * f;
    const long long t0 = wall_clock64();
    for (;;) {
        bool ok = true;
#pragma unroll
        for (uint32_t h = 0; h < 3; h++) ok = ok && flag_load(c + h) >= (quads + 2u - h) / 3u;   // helper h has the quads = h (mod 3)
        if (__all(ok)) break;
        if (pipe_expired(t0, P.wait_ticks)) { if (lane == 0) atomicOr(P.flags, 1u); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    PIPE_STAMP(P, 16 + 2 * wave_index);
    return true;
}

// ---- role 3: a wave of the window-parallel trellis
__device__ __forceinline__ void pipe_trellis_wave(const RxArgs& A, const PipeArgs& P, uint32_t wave_index, Lds16<256, 24>& S)
{
    auto jobs_of = [&](uint32_t list) {
        const uint32_t* jl = A.joblist + (size_t)list * A.nrows; const FrameRow* fr = A.frames;
        return [jl, fr](uint32_t idx) {
            const FrameRow& r = fr[jl[idx]];
            VitJob J;
            J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftBytesPerSlot; J.nsoft = (uint32_t)r.nsym * 48u * r.nbpsc; J.length = r.length;
            J.dec_off = 0; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate; J.soft_bits = 3;
            return J;
        };
    };
    auto ready = [&](const UnitGeom& GA, const UnitGeom& GB_, uint32_t list) -> bool {
        const unsigned lane = threadIdx.x & 63;
        return pipe_units_ready(A, P, list, (lane & 1u) ? GB_ : GA, (lane & 15u) < 2u, wave_index);      // one lane per unit polls
    };
    viterbi16w_wave<256, 24, 3>(S, wave_index, jobs_of, ready, A.njobs, P.target, P.vstride, (const uint8_t*)A.soft, A.vout, P.vecs);
    PIPE_STAMP(P, 17 + 2 * wave_index);
}

// (k_pipe's 64-lane trellis role and the kernel itself follow viterbi_forward_unit below)

// (the trellis machinery -- metric representation, ACS step, trace-back -- lives in dev_viterbi.h)
struct VitSide {            // wave-uniform per-frame bookkeeping
    uint8_t* out; uint32_t nsteps, tr_end; bool done;
};

// Soft input: the two frames a wave decodes (consecutive jobs of one code-rate list) each have their own packed stream (rx_types.h: three
// bits per value, or a byte).  Per 12-step chunk, lane k < 32 fetches value k of frame A, lane 32 + k value k of frame B (one 16-bit load
// each, two chunks ahead), shifts it into a 16-bit metric field and writes it to the wave's operand table in LDS; five or six broadcast
// ds_read_b128 then give every lane the chunk's operands, dword i = field A | field B << 16 -- exactly what acs_step xors with the lane's
// mask.  (Round 3 first kept such operand dwords in HBM -- 264 MB per call written and read, fetched through the scalar cache; round 2 a
// 16-bit stream per frame that cost one s_pack per value.)  Past a frame's end its last value is repeated: its trellis half keeps stepping
// on well-formed operands (a half fed garbage could carry twice into the guard bit within one block).
//
// Trace-back (TViterbiCore::Traceback, viterbicore.h:468-555) runs in the same wave, out of LDS, whenever the window
// schedule of T11aViterbi<..,256,24>::Process (viterbi.hpp:196-214) fires: the ring holds, per 8-column block j
// (columns 8j+1..8j+8) and per state at column 8j+8, the 8 decisions of the survivor path into that state (bit i =
// column 8j+1+i).  One lookup walks 8 columns: the decisions are the decoded bits, and the state at column 8j is the 6
// oldest decisions, newest in bit 0 (s' = d << 5 | s >> 1 applied 8 times).  Decoded bit i of the frame is the
// decision at column i + 7 on the traced path (6-bit decoder delay), so output byte m is (block m >> 6) | (block m+1
// & 0x3F) << 2.
// WIN / LOOK: the window schedule of T11aViterbi<.., N_INPUT, TRELLIS_DEPTH = WIN, TRELLIS_LOOKAHEAD = LOOK> -- 256 / 24 in the 802.11a graph
// (fb11ademod_config.hpp:199), 192 / 36 in the 802.11n graph (fb11ndemod_config.hpp:199); a walk touches at most (WIN + LOOK + 7) / 8 + 2 <= 38 blocks.
template <int CR, int WIN, int LOOK, int BITS>
__device__ __forceinline__ void viterbi_forward(const VitJob& JA, const VitJob& JB, bool hasB, const uint8_t* __restrict__ soft_base,
        uint8_t* __restrict__ out_base, uint16_t* ring, uint16_t* ops)
{
    using RG = RingGeom<WIN, LOOK>;
    constexpr int P = RG::P;
    constexpr int GB = CR == 0 ? 2 : CR == 2 ? 4 : 3;                           // soft values per puncture group (CR: 0=1/2, 1=2/3, 2=3/4)
    constexpr int GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;                           // trellis steps per group
    constexpr int CW = 12 / GS * GB;                                            // operands (dwords) per 12-step chunk: 24 / 18 / 16
    const unsigned lane = threadIdx.x & 63;
    VitSide A, B;
    A.out = out_base + JA.out_off; A.nsteps = JA.nsoft / GB * GS; A.tr_end = JA.length * 8u + 16u + 6u; A.done = false;
    B.out = out_base + JB.out_off; B.nsteps = hasB ? JB.nsoft / GB * GS : 0u; B.tr_end = hasB ? JB.length * 8u + 16u + 6u : 0u; B.done = !hasB;
    const uint32_t nsteps = max(A.nsteps, B.nsteps);
    // this lane's part in fetching a chunk: value (lane & 31) of frame lane >> 5
    const bool mineB = lane >= 32u && hasB;
    const uint32_t my_soft_off = mineB ? JB.soft_off : JA.soft_off;
    const uint32_t my_last = max(mineB ? JB.nsoft : JA.nsoft, 1u) - 1u, my_k = lane & 31u;

    auto which_of = [](int ph) { return CR == 0 ? 0 : CR == 1 ? (ph & 1) : ph % 3; };   // step kinds of a puncture group (viterbi.hpp:167-187)
    VitLane V;
    const unsigned vl = lane_map(lane);                                         // label lane: holds state rol6^t(vl) after t steps
    V.U = vl == 0 ? 0u : 0x18u * kFld;                                         // ALL_INIT0 / ALL_INIT = 0x00 / 0x30 (viterbilut.h:22-30)
    V.ring = ring; V.rowpos = 0;
    V.sidx[0] = __brev(rol6(vl, 2)) >> 26; V.sidx[1] = __brev(rol6(vl, 4)) >> 26; V.sidx[2] = __brev(vl) >> 26;   // rev6 of the state: (8j + 8) mod 6 = 2, 4, 0
#pragma unroll
    for (int t = 0; t < 24; t++) {
        const int ph = t % 6, k = t % 8;
        const unsigned n = rol6(vl, ph + 1);                                    // state held after a phase-ph step
        const bool own1 = ph >= 2 && ((vl >> (5 - ph)) & 1);                    // DPP phases: the lane's own metric is the decision-1 candidate
        const unsigned ma = (__popc(n & 0155) & 1) ? 7u * kFld : 0u, mb = (__popc(n & 0117) & 1) ? 7u * kFld : 0u;
        const unsigned mx = which_of(ph) == 2 ? mb : ma;
        V.MX[t] = own1 ? ((mx ^ (7u * kFld)) | (kOne << k)) : mx;
        if (t < 6) V.MY[t] = own1 ? (mb ^ (7u * kFld)) : mb;
    }

    uint32_t tr = 0, ob = 0;                                                    // ob: bits handed out by the partial windows (same schedule for both frames)

    // Normalize (viterbicore.h:444-465), both frames; marks and guard are clear here and no half borrows (its minimum is subtracted): one 32-bit VOP2
    auto normalize = [&]() { V.U = V.U - dpp_pkmin_wave(V.U); };
#ifdef SORA_DBG_NO_TRACE                                                        // experiment (tools/ab_decode.sh): the forward pass alone -- results are wrong, only the duration means something
    auto trace = [&](unsigned, unsigned, uint32_t, uint32_t, uint32_t) {};
#else
    auto trace = [&](unsigned mA, unsigned mB, uint32_t cntA, uint32_t cntB, uint32_t top) { viterbi_trace<RG::kMaxWalk>(V.U, ring, tr, ob, mA, mB, cntA, cntB,
            A.out, B.out, top); };
#endif
    auto next_event = [&]() -> uint32_t {
        uint32_t t = ob + (uint32_t)(WIN + LOOK + 6);
        if (!A.done) t = min(t, A.tr_end);
        if (!B.done) t = min(t, B.tr_end);
        return t;
    };
    uint32_t next_thr = next_event();
    auto check = [&](int t24_last) {                                            // trace-back schedule (viterbi.hpp:196-214), per frame
        if (tr >= next_thr) {
            const int k = t24_last % 8;                                         // the last decision: mark k of the field, or bit 7 of the block just banked
            const uint32_t pos = V.rowpos + (uint32_t)(t24_last / 8) * 64u;     // ring position (x 64) of block (tr - 1) >> 3
            unsigned lastA, lastB;
            if (k == 7) { const unsigned w = ring[pos + V.sidx[t24_last / 8]]; lastA = (w >> 7) & 1u; lastB = (w >> 15) & 1u; }
            else { lastA = (V.U >> k) & 1u; lastB = (V.U >> (17 + k)) & 1u; }
            const unsigned mA = ((V.U & 0xFFFFu) >> 9 << 1) | lastA, mB = (V.U >> 25 << 1) | lastB;
            const bool partial = tr >= ob + (uint32_t)(WIN + LOOK + 6);
            uint32_t cntA = 0, cntB = 0;
            if (!A.done) {
                if (tr >= A.tr_end) { cntA = A.tr_end - ob - 6; A.done = true; }
                else if (partial) cntA = WIN;
            }
            if (!B.done) {
                if (tr >= B.tr_end) { cntB = B.tr_end - ob - 6; B.done = true; }
                else if (partial) cntB = WIN;
            }
            if (cntA | cntB) trace(mA, mB, cntA, cntB, (pos >> 6) + (uint32_t)P);
            if (partial) ob += WIN;
            next_thr = next_event();
        }
    };
    struct Chunk { uint32_t v[(CW + 3) / 4 * 4]; };
    SoftCursor<BITS, CW> cur;
    cur.init(my_soft_off, my_k, my_last);
    auto fetch = [&](uint32_t c) -> SoftRaw { return cur.fetch(soft_base, c); };
    // operand k, frame's half (k up to 31: the table has 32 operands, those past CW are never read)
    uint16_t* my_op = ops + 2u * my_k + (lane >> 5);
    auto lds_order = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    auto unpack = [&](const SoftRaw& R) -> Chunk {
        *my_op = (uint16_t)cur.field(R);
        lds_order();
        Chunk K;
#pragma unroll
        for (int i = 0; i < (CW + 3) / 4; i++) {
            const uint4 x = reinterpret_cast<const uint4*>(ops)[i];
            K.v[4 * i] = x.x; K.v[4 * i + 1] = x.y; K.v[4 * i + 2] = x.z; K.v[4 * i + 3] = x.w;
        }
        lds_order();
        return K;
    };
    auto op = [](const Chunk& K, int i) -> uint32_t { return K.v[i]; };
    // one puncture group = GS steps; i0 = step index inside the 12-step chunk, h = which half of the 24-step row
    auto group = [&](const Chunk& K, int h, int i0) {
        const int k0 = i0 / GS * GB, t24 = 12 * h + i0;
        acs_step<0, P>(V, t24, op(K, k0), op(K, k0 + 1));                       // ACS(A,B)
        if (CR != 0) acs_step<1, P>(V, t24 + 1, op(K, k0 + 2), 0);              // ACS(A)     2/3, 3/4 (viterbi.hpp:173-187)
        if (CR == 2) acs_step<2, P>(V, t24 + 2, 0, op(K, k0 + 3));              // ACS(B)     3/4
        if ((t24 + GS) % 8 == 0) normalize();                                   // (trellis index & 7) == 0 after a group
    };
    auto end_row = [&]() { V.rowpos = V.rowpos + 3 * 64 == (unsigned)P * 64 ? 0u : V.rowpos + 3 * 64; };   // P is a multiple of 3: the wrap falls between rows
    auto fast_chunk = [&](const Chunk& K, int h) {                              // 12 steps, no trace-back due inside: straight-line code
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) group(K, h, g * GS);
        tr += 12;
    };
    auto slow_chunk = [&](const Chunk& K, int h) {                              // up to 12 steps with the schedule examined after every group
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) {
            if (tr < nsteps && !(A.done && B.done)) {
                group(K, h, g * GS);
                tr += GS;
                check(12 * h + g * GS + GS - 1);
            }
        }
    };
    auto chunk = [&](const Chunk& K, int h) {                                   // tr % 24 == 12 h on entry
        if (tr + 12 <= nsteps && next_thr > tr + 12) fast_chunk(K, h); else slow_chunk(K, h);
    };

    // Vector loads return in order: chunk c + 2 is requested before chunk c is stepped through (the compiler's vmcnt waits follow from that).
    uint32_t c = 0;
    SoftRaw r0 = fetch(0), r1 = fetch(1);
    while (tr < nsteps && !(A.done && B.done)) {
        // rows (2 chunks) that certainly need no look at the schedule: run them back to back, 9 rows out of 10
        const uint32_t lim = min(nsteps, next_thr - 1);
        for (uint32_t rows = lim > tr ? (lim - tr) / 24 : 0; rows > 0; rows--) {
            const Chunk K0 = unpack(r0); r0 = fetch(c + 2);
            fast_chunk(K0, 0);
            const Chunk K1 = unpack(r1); r1 = fetch(c + 3);
            fast_chunk(K1, 1);
            c += 2;
            end_row();
        }
        if (!(tr < nsteps)) break;
        const Chunk K0 = unpack(r0); r0 = fetch(c + 2);
        chunk(K0, 0);
        if (!(tr < nsteps && !(A.done && B.done))) break;
        const Chunk K1 = unpack(r1); r1 = fetch(c + 3);
        chunk(K1, 1);
        c += 2;
        end_row();
    }
}

// The same wave decoding one UNIT of the window-parallel trellis per half instead of a whole frame (dev_vitwin.h; k_pipe's 64-lane form for a lone capture: a unit is
// 33 ns per step here against 50 in the sixteen-lane layout, and the launch's last unit is what the capture waits for).  GA / GB_ are unit `u` of two frames of one
// code-rate list (or GB_ invalid): the same unit index, hence the same distance ob between the unit's first step and its first window and ONE trace-back schedule for
// the wave, as in the whole-frame form.  What a unit adds: all-equal metrics at its start (the frame's first unit starts like the frame), its stream read from its
// first step on, its metric vector stored at its verify point and at the next unit's, and an end after its windows.  ready(): see forward16w.
template <int CR, int WIN, int LOOK, int BITS, typename READY>
__device__ __forceinline__ void viterbi_forward_unit(const UnitGeom& GA, const UnitGeom& GB_, const uint8_t* __restrict__ soft_base, uint16_t* ring, uint16_t* ops,
        uint16_t* __restrict__ vecs, READY ready)
{
    const bool hasB = GB_.valid;
    using RG = RingGeom<WIN, LOOK>;
    constexpr int P = RG::P;
    constexpr int GB = CR == 0 ? 2 : CR == 2 ? 4 : 3;                           // soft values per puncture group (CR: 0=1/2, 1=2/3, 2=3/4)
    constexpr int GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;                           // trellis steps per group
    constexpr int CW = 12 / GS * GB;                                            // operands (dwords) per 12-step chunk: 24 / 18 / 16
    const unsigned lane = threadIdx.x & 63;
    VitSide A, B;
    A.out = GA.out; A.nsteps = GA.nsteps; A.tr_end = GA.tr_end; A.done = false;
    B.out = GB_.out; B.nsteps = hasB ? GB_.nsteps : 0u; B.tr_end = hasB ? GB_.tr_end : 0u; B.done = !hasB;
    const uint32_t nsteps = max(A.nsteps, B.nsteps);
    // this lane's part in fetching a chunk: value (lane & 31) of frame lane >> 5
    const bool mineB = lane >= 32u && hasB;
    const uint32_t my_soft_off = mineB ? GB_.soft_off : GA.soft_off;
    const uint32_t my_last = mineB ? GB_.last : GA.last, my_k = lane & 31u;
    const uint32_t my_first = my_k + (mineB ? GB_.i0 : GA.i0);                 // (the unit's first value: a whole number of puncture groups into the stream)

    auto which_of = [](int ph) { return CR == 0 ? 0 : CR == 1 ? (ph & 1) : ph % 3; };   // step kinds of a puncture group (viterbi.hpp:167-187)
    VitLane V;
    const unsigned vl = lane_map(lane);                                         // label lane: holds state rol6^t(vl) after t steps
    V.U = vl == 0 ? 0u : ((GA.first ? 0x18u << 9 : 0u) | ((hasB && GB_.first) ? 0x18u << 25 : 0u));   // a frame's first unit: ALL_INIT0 / ALL_INIT (viterbilut.h:22-30); any other: all equal
    V.ring = ring; V.rowpos = 0;
    V.sidx[0] = __brev(rol6(vl, 2)) >> 26; V.sidx[1] = __brev(rol6(vl, 4)) >> 26; V.sidx[2] = __brev(vl) >> 26;   // rev6 of the state: (8j + 8) mod 6 = 2, 4, 0
#pragma unroll
    for (int t = 0; t < 24; t++) {
        const int ph = t % 6, k = t % 8;
        const unsigned n = rol6(vl, ph + 1);                                    // state held after a phase-ph step
        const bool own1 = ph >= 2 && ((vl >> (5 - ph)) & 1);                    // DPP phases: the lane's own metric is the decision-1 candidate
        const unsigned ma = (__popc(n & 0155) & 1) ? 7u * kFld : 0u, mb = (__popc(n & 0117) & 1) ? 7u * kFld : 0u;
        const unsigned mx = which_of(ph) == 2 ? mb : ma;
        V.MX[t] = own1 ? ((mx ^ (7u * kFld)) | (kOne << k)) : mx;
        if (t < 6) V.MY[t] = own1 ? (mb ^ (7u * kFld)) : mb;
    }

    uint32_t tr = 0, ob = GA.ob;                                                // steps taken / where the next window's bits begin, both in the units' own step count (the same for both)
    uint32_t wA = GA.wleft, wB = GB_.wleft;
    uint32_t vstepA = GA.vstep, vstepB = hasB ? GB_.vstep : kNever, estepA = GA.estep, estepB = hasB ? GB_.estep : kNever;
    // a vector: the lane's 16-bit field of the unit's half -- taken at a multiple of 24 of the unit's own steps: straight after a normalisation, marks and guard clear,
    // the state <-> lane map the identity (the same at both ends of a comparison)
    auto save = [&](uint32_t vec, int which, bool hi) { vecs[((size_t)vec * 2u + (uint32_t)which) * 64u + lane] = (uint16_t)(hi ? V.U >> 16 : V.U); };

    // Normalize (viterbicore.h:444-465), both frames; marks and guard are clear here and no half borrows (its minimum is subtracted): one 32-bit VOP2
    auto normalize = [&]() { V.U = V.U - dpp_pkmin_wave(V.U); };
#ifdef SORA_DBG_NO_TRACE                                                        // experiment (tools/ab_decode.sh): the forward pass alone -- results are wrong, only the duration means something
    auto trace = [&](unsigned, unsigned, uint32_t, uint32_t, uint32_t) {};
#else
    auto trace = [&](unsigned mA, unsigned mB, uint32_t cntA, uint32_t cntB, uint32_t top) { viterbi_trace<RG::kMaxWalk>(V.U, ring, tr, ob, mA, mB, cntA, cntB,
            A.out, B.out, top); };
#endif
    auto next_event = [&]() -> uint32_t {
        uint32_t t = ob + (uint32_t)(WIN + LOOK + 6);
        if (!A.done) t = min(t, A.tr_end);
        if (!B.done) t = min(t, B.tr_end);
        return min(min(t, min(vstepA, vstepB)), min(estepA, estepB));
    };
    uint32_t next_thr = next_event();
    auto check = [&](int t24_last) {                                            // trace-back schedule (viterbi.hpp:196-214), per frame
        if (tr >= next_thr) {
            if (tr == vstepA) { save(GA.vec, 0, false); vstepA = kNever; }
            if (tr == vstepB) { save(GB_.vec, 0, true); vstepB = kNever; }
            if (tr == estepA) { save(GA.vec, 1, false); estepA = kNever; }
            if (tr == estepB) { save(GB_.vec, 1, true); estepB = kNever; }
            const int k = t24_last % 8;                                         // the last decision: mark k of the field, or bit 7 of the block just banked
            const uint32_t pos = V.rowpos + (uint32_t)(t24_last / 8) * 64u;     // ring position (x 64) of block (tr - 1) >> 3
            unsigned lastA, lastB;
            if (k == 7) { const unsigned w = ring[pos + V.sidx[t24_last / 8]]; lastA = (w >> 7) & 1u; lastB = (w >> 15) & 1u; }
            else { lastA = (V.U >> k) & 1u; lastB = (V.U >> (17 + k)) & 1u; }
            const unsigned mA = ((V.U & 0xFFFFu) >> 9 << 1) | lastA, mB = (V.U >> 25 << 1) | lastB;
            const bool partial = tr >= ob + (uint32_t)(WIN + LOOK + 6);
            uint32_t cntA = 0, cntB = 0;
            if (!A.done) {
                if (tr >= A.tr_end) { cntA = A.tr_end - ob - 6; A.done = true; }
                else if (partial) cntA = WIN;
            }
            if (!B.done) {
                if (tr >= B.tr_end) { cntB = B.tr_end - ob - 6; B.done = true; }
                else if (partial) cntB = WIN;
            }
            if (cntA | cntB) trace(mA, mB, cntA, cntB, (pos >> 6) + (uint32_t)P);
            if (partial) {                                                      // a unit ends behind its last window (the frame's last unit: at the frame's end)
                ob += WIN;
                if (!A.done && --wA == 0) A.done = true;
                if (!B.done && --wB == 0) B.done = true;
            }
            next_thr = next_event();
        }
    };
    struct Chunk { uint32_t v[(CW + 3) / 4 * 4]; };
    SoftCursor<BITS, CW> cur;
    cur.init(my_soft_off, my_first, my_last);
    auto fetch = [&](uint32_t c) -> SoftRaw { return cur.fetch(soft_base, c); };
    // operand k, frame's half (k up to 31: the table has 32 operands, those past CW are never read)
    uint16_t* my_op = ops + 2u * my_k + (lane >> 5);
    auto lds_order = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    auto unpack = [&](const SoftRaw& R) -> Chunk {
        *my_op = (uint16_t)cur.field(R);
        lds_order();
        Chunk K;
#pragma unroll
        for (int i = 0; i < (CW + 3) / 4; i++) {
            const uint4 x = reinterpret_cast<const uint4*>(ops)[i];
            K.v[4 * i] = x.x; K.v[4 * i + 1] = x.y; K.v[4 * i + 2] = x.z; K.v[4 * i + 3] = x.w;
        }
        lds_order();
        return K;
    };
    auto op = [](const Chunk& K, int i) -> uint32_t { return K.v[i]; };
    // one puncture group = GS steps; i0 = step index inside the 12-step chunk, h = which half of the 24-step row
    auto group = [&](const Chunk& K, int h, int i0) {
        const int k0 = i0 / GS * GB, t24 = 12 * h + i0;
        acs_step<0, P>(V, t24, op(K, k0), op(K, k0 + 1));                       // ACS(A,B)
        if (CR != 0) acs_step<1, P>(V, t24 + 1, op(K, k0 + 2), 0);              // ACS(A)     2/3, 3/4 (viterbi.hpp:173-187)
        if (CR == 2) acs_step<2, P>(V, t24 + 2, 0, op(K, k0 + 3));              // ACS(B)     3/4
        if ((t24 + GS) % 8 == 0) normalize();                                   // (trellis index & 7) == 0 after a group
    };
    auto end_row = [&]() { V.rowpos = V.rowpos + 3 * 64 == (unsigned)P * 64 ? 0u : V.rowpos + 3 * 64; };   // P is a multiple of 3: the wrap falls between rows
    auto fast_chunk = [&](const Chunk& K, int h) {                              // 12 steps, no trace-back due inside: straight-line code
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) group(K, h, g * GS);
        tr += 12;
    };
    auto slow_chunk = [&](const Chunk& K, int h) {                              // up to 12 steps with the schedule examined after every group
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) {
            if (tr < nsteps && !(A.done && B.done)) {
                group(K, h, g * GS);
                tr += GS;
                check(12 * h + g * GS + GS - 1);
            }
        }
    };
    auto chunk = [&](const Chunk& K, int h) {                                   // tr % 24 == 12 h on entry
        if (tr + 12 <= nsteps && next_thr > tr + 12) fast_chunk(K, h); else slow_chunk(K, h);
    };

    // Vector loads return in order: chunk c + 2 is requested before chunk c is stepped through (the compiler's vmcnt waits follow from that).
    uint32_t c = 0;
    if (!ready()) return;
    SoftRaw r0 = fetch(0), r1 = fetch(1);
    while (tr < nsteps && !(A.done && B.done)) {
        // rows (2 chunks) that certainly need no look at the schedule: run them back to back, 9 rows out of 10
        const uint32_t lim = min(nsteps, next_thr - 1);
        for (uint32_t rows = lim > tr ? (lim - tr) / 24 : 0; rows > 0; rows--) {
            const Chunk K0 = unpack(r0); r0 = fetch(c + 2);
            fast_chunk(K0, 0);
            const Chunk K1 = unpack(r1); r1 = fetch(c + 3);
            fast_chunk(K1, 1);
            c += 2;
            end_row();
        }
        if (!(tr < nsteps)) break;
        const Chunk K0 = unpack(r0); r0 = fetch(c + 2);
        chunk(K0, 0);
        if (!(tr < nsteps && !(A.done && B.done))) break;
        const Chunk K1 = unpack(r1); r1 = fetch(c + 3);
        chunk(K1, 1);
        c += 2;
        end_row();
    }
}

// ---- role 3, 64-lane form (P.lanes64: the handle's calls in flight are so few that two units per wave still fit the chip): wave w of a code-rate list of n frames holds
// unit w / ceil(n / 2) of frames 2 j and 2 j + 1, j = w mod ceil(n / 2)
struct PipeUnitLds { uint16_t ring[RingGeom<256, 24>::kEntries]; uint16_t ops[64]; };
static_assert(4 * sizeof(PipeUnitLds) <= kPipeLdsBytes, "k_pipe: four 64-lane trellis waves' LDS");
__device__ __forceinline__ void pipe_trellis_wave64(const RxArgs& A, const PipeArgs& P, uint32_t wave_index, PipeUnitLds& L)
{
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t n[3] = { A.njobs[0], A.njobs[1], A.njobs[2] };
    const uint32_t q = uni(win_units_per_frame(n[0] + n[1] + n[2], P.target));
    uint32_t w = uni(wave_index), list = 0;
    while (list < 3 && w >= q * ((n[list] + 1u) / 2u)) { w -= q * ((n[list] + 1u) / 2u); list++; }
    if (list >= 3) return;
    const uint32_t nl = uni(n[list]), pairs = (nl + 1u) / 2u, u = w / pairs, ia = 2u * (w - u * pairs), ib = ia + 1u;
    const uint32_t* jl = A.joblist + (size_t)list * A.nrows; const FrameRow* fr = A.frames;
    auto job_at = [jl, fr](uint32_t idx) {
        const FrameRow& r = fr[jl[idx]];
        VitJob J;
        J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftBytesPerSlot; J.nsoft = (uint32_t)r.nsym * 48u * r.nbpsc; J.length = r.length;
        J.dec_off = 0; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate; J.soft_bits = 3;
        return J;
    };
    const uint32_t code_rate = uni(job_at(0).code_rate), vbase = list * P.vstride;
    auto run = [&](auto cr) {
        constexpr int CR = decltype(cr)::value;
        UnitGeom GA = unit_geom_direct<CR, 256, 24>(job_at, ia, u, true, q, vbase, A.vout);
        UnitGeom GB_ = unit_geom_direct<CR, 256, 24>(job_at, ib < nl ? ib : ia, u, ib < nl, q, vbase, A.vout);
        // the second frame is shorter / the first is: one unit in the wave.  Unequal cuts of a pair do not occur for as few frames as this form is used for (every frame comes
        // out cut into single windows); should they ever, the second frame's unit would go undecoded: say so, and the finishing kernel makes the call again (PipeFallbackGate)
        if (GB_.valid && GA.valid && GB_.ob != GA.ob && (threadIdx.x & 63) == 0) atomicOr(P.flags, 1u);
        if (GB_.valid && (!GA.valid || GB_.ob != GA.ob)) { if (!GA.valid) { GA = GB_; } const bool no = false; GB_.valid = no; }
        if (!GA.valid) return;
        const unsigned lane = threadIdx.x & 63;
        viterbi_forward_unit<CR, 256, 24, 3>(GA, GB_, (const uint8_t*)A.soft, L.ring, L.ops, P.vecs,
                                             [&]() { return pipe_units_ready(A, P, list, lane >= 32u ? GB_ : GA, (lane & 31u) == 0u, wave_index); });
    };
    if (code_rate == 0) run(std::integral_constant<int, 0>{});
    else if (code_rate == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    PIPE_STAMP(P, 17 + 2 * wave_index);
}

__global__ void __launch_bounds__(256) k_pipe(RxArgs A, PipeArgs P)
{
    __shared__ __attribute__((aligned(16))) char lds[kPipeLdsBytes];
    uint32_t b = blockIdx.x;
    if (b == 0) PIPE_STAMP(P, 0);
    if (b < P.nfront) {
        sym_front_block<true>(A, b, reinterpret_cast<uint32_t (*)[4][64]>(lds));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // every storing wave drains, then ONE flag
        __syncthreads();
#ifndef SORA_DBG_PIPE_LOSE_FLAGS                                                 // (tools/pipe_timeout_check.py: what happens when a hand-off never arrives)
        if (threadIdx.x == 0) store4_through(P.flags + 4u + 4u * A.nrows + b, 1u);
#endif
        if (b == 0) PIPE_STAMP(P, 8);
        return;
    }
    b -= P.nfront;
    if (b < P.ntrack) { pipe_track_block(A, P, b, *reinterpret_cast<PipeTrackLds*>(lds)); return; }
    b -= P.ntrack;
    if (P.lanes64) pipe_trellis_wave64(A, P, b * 4u + (threadIdx.x >> 6), reinterpret_cast<PipeUnitLds*>(lds)[threadIdx.x >> 6]);
    else pipe_trellis_wave(A, P, b * 4u + (threadIdx.x >> 6), reinterpret_cast<Lds16<256, 24>*>(lds)[threadIdx.x >> 6]);
}


// Four waves per 256-thread workgroup, two frames per wave (no cross-wave traffic).  One-wave workgroups were kept to
// 8 per CU by the dispatcher: 2 waves per SIMD and a second round for a 4096-frame batch.
// Frames are queued per code rate (k_scan), so the two frames of a wave always share the puncture pattern; the last
// frame of an odd list runs alone in the low half.  (Jobs given through sora_hip_viterbi11a are one list of one rate.)
struct DecodeEveryPair { __device__ __forceinline__ bool operator()(uint32_t, uint32_t, uint32_t, bool) const { return true; } };
struct NothingAfter { __device__ __forceinline__ void operator()(uint32_t, uint32_t, uint32_t, bool) const {} };
// (gate(list, fa, fb, hasB): decode this pair at all?  after(...): what the wave does with its pair once the pair's bytes are final -- k_win_redo_finish's T11aDesc / frame sink)
template <int WIN, int LOOK, int BITS, typename GATE = DecodeEveryPair, typename AFTER = NothingAfter>
__device__ __forceinline__ void viterbi_kernel_body(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single,
        uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out,
                                                    GATE gate = GATE(), AFTER after = AFTER())
{
    // 39 KB / 33 KB: survivor history, two copies of every block (RingGeom), per wave
    __shared__ uint16_t s_ring[4][RingGeom<WIN, LOOK>::kEntries];
    // [wave][operand of the chunk][frame]: the soft values as metric fields (viterbi_forward); with it under 40 KB: four workgroups per CU
    __shared__ uint16_t s_ops[4][64];
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };   // everything below is per-wave uniform: keep it in SGPRs
    // wave -> (code-rate list, pair): list r has ceil(n_r / 2) pairs (njobs3 == nullptr: one list of njobs_single jobs)
    uint32_t n[3] = { njobs_single, 0, 0 };
    if (njobs3) { n[0] = njobs3[0]; n[1] = njobs3[1]; n[2] = njobs3[2]; }
    uint32_t pw = uni(blockIdx.x * 4 + (threadIdx.x >> 6)), list = 0;
    while (list < 3 && pw >= (n[list] + 1) / 2) { pw -= (n[list] + 1) / 2; list++; }
    if (list >= 3) return;
    const uint32_t njobs = uni(n[list]);
    jobs += (size_t)list * stride;
    const uint32_t fa = pw * 2, fb = fa + 1;
    if (gate(list, fa, fb, fb < njobs)) {                                        // (k_win_redo: false = the pair's proofs hold, nothing to decode again)
        uint16_t* ring = s_ring[threadIdx.x >> 6];
        auto load_job = [&](uint32_t f) {
            const VitJob& G = jobs[f];
            VitJob J;
            J.soft_off = uni(G.soft_off); J.nsoft = uni(G.nsoft); J.length = uni(G.length); J.dec_off = uni(G.dec_off);
            J.out_off = uni(G.out_off); J.valid = uni(G.valid); J.code_rate = uni(G.code_rate); J.soft_bits = uni(G.soft_bits);
            return J;
        };
        const VitJob JA = load_job(fa);
        const bool hasB = fb < njobs;
        const VitJob JB = hasB ? load_job(fb) : JA;
        if (JA.code_rate == 0)      viterbi_forward<0, WIN, LOOK, BITS>(JA, JB, hasB, soft, out, ring, s_ops[threadIdx.x >> 6]);
        else if (JA.code_rate == 1) viterbi_forward<1, WIN, LOOK, BITS>(JA, JB, hasB, soft, out, ring, s_ops[threadIdx.x >> 6]);
        else                        viterbi_forward<2, WIN, LOOK, BITS>(JA, JB, hasB, soft, out, ring, s_ops[threadIdx.x >> 6]);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");                   // (after() reads the bytes this wave has just written)
    }
    after(list, fa, fb, fb < njobs);
}

__global__ void __launch_bounds__(256) k_viterbi(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride,
        const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{ viterbi_kernel_body<256, 24, 3>(jobs, njobs3, njobs_single, stride, soft, out); }
// the 802.11n graph's decoder: T11aViterbi<5000*8, 312, 192, 36> (fb11ndemod_config.hpp:199)
__global__ void __launch_bounds__(256) k_viterbi11n(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single,
        uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{ viterbi_kernel_body<192, 36, 8>(jobs, njobs3, njobs_single, stride, soft, out); }

// The window-parallel trellis's proof AND the serial decode of what failed it, in one launch (k_vitwin.hip describes the proof; a verify kernel + k_viterbi as two
// launches cost a lone capture a kernel and a queue gap).  A wave owns a PAIR of frames of one code-rate list, as in k_viterbi: its lower half compares the unit
// boundaries of frame A (unit u's vector at its verify point against unit u - 1's vector at the same step), its upper half those of frame B; if every boundary of both
// holds the wave is done -- what k_viterbi16w wrote is the reference's decode -- else it decodes the pair serially, overwriting it.
template <uint32_t WIN, uint32_t LOOK>
struct WinProofGateT {
    const VitJob* jobs; const uint32_t* hdr; uint32_t jstride, target, vstride; const uint16_t* vecs; unsigned long long* stats;
    __device__ __forceinline__ bool operator()(uint32_t list, uint32_t fa, uint32_t fb, bool hasB) const
    {
        const unsigned lane = threadIdx.x & 63, side = lane >> 5, l32 = lane & 31u;
        const uint32_t q = win_units_per_frame(hdr[0] + hdr[1] + hdr[2], target);
        const uint32_t idx = side ? fb : fa;
        const bool have = side ? hasB : true;
        const VitJob& J = jobs[(size_t)list * jstride + (have ? idx : fa)];
        const uint32_t nev = win_events(J.length, J.code_rate, WIN, LOOK), m = win_per_unit(nev, q), nun = have ? (nev + m - 1u) / m : 1u;
        const size_t vec0 = (size_t)list * vstride + (size_t)idx * q;
        uint32_t bad = 0;
        for (uint32_t u = 1u + l32; u < nun; u += 32u) {
            const uint4* a = reinterpret_cast<const uint4*>(vecs + ((vec0 + u) * 2u) * 64u);
            const uint4* b = reinterpret_cast<const uint4*>(vecs + ((vec0 + u - 1u) * 2u + 1u) * 64u);
            uint32_t d = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { const uint4 x = a[i], y = b[i]; d |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w); }
            bad += d != 0u;
        }
        const unsigned long long ba = __ballot(bad != 0u);
        uint32_t nbad = bad;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) nbad += __shfl_xor(nbad, o);             // per side
        if (stats && l32 == 0 && have) {                                          // the record, banked (rx_types.h kWinStatBanks)
            unsigned long long* bk = stats + 4u * ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (kWinStatBanks - 1u));
            atomicAdd(&bk[0], (unsigned long long)(nun - 1u)); atomicAdd(&bk[3], (unsigned long long)nun);
            if (nbad) { atomicAdd(&bk[1], (unsigned long long)nbad); atomicAdd(&bk[2], 1ull); }
        }
        return ba != 0ull;
    }
};
using WinProofGate = WinProofGateT<256u, 24u>;
__global__ void __launch_bounds__(256) k_win_redo(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ hdr, uint32_t jstride, uint32_t target,
        uint32_t vstride, const uint16_t* __restrict__ vecs,
                                                  const uint8_t* __restrict__ soft, uint8_t* __restrict__ out, unsigned long long* __restrict__ stats)
{ viterbi_kernel_body<256, 24, 3>(jobs, hdr, 0u, jstride, soft, out, WinProofGate{ jobs, hdr, jstride, target, vstride, vecs, stats }); }
// ... for the 802.11n graph's decoder (windows of 192 bits, 36 of look-ahead, one byte per soft value)
__global__ void __launch_bounds__(256) k_win_redo_11n(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ hdr, uint32_t jstride, uint32_t target,
        uint32_t vstride, const uint16_t* __restrict__ vecs,
                                                      const uint8_t* __restrict__ soft, uint8_t* __restrict__ out, unsigned long long* __restrict__ stats)
{ viterbi_kernel_body<192, 36, 8>(jobs, hdr, 0u, jstride, soft, out, WinProofGateT<192u, 36u>{ jobs, hdr, jstride, target, vstride, vecs, stats }); }

// ------------------------------------------------------------------------------------------------
// k_finish: T11aDesc (scramble.hpp:267-353) + TBB11aFrameSink (PHY_11a.hpp:607-702).  One wave per frame.
//   * descrambling is data-parallel: the x^7+x^4+1 sequence has period 127, every non-zero seed is a phase of
//     the same cycle, so scrambler byte j = seqbyte[(phase(seed) + 8 j) mod 127]  (two small tables, no chain);
//   * the 64 lanes load / descramble / store the MPDU coalesced and park it in LDS;
//   * CRC-32 in parallel: the register update is linear over GF(2), so CRC(init, M) = CRC(0, M') with the first four
//     bytes complemented, and CRC(0, M1 | M2) = Z_|M2|(CRC(0, M1)) ^ CRC(0, M2) with Z_m = "m zero bytes".  Lane l takes
//     the 40 bytes that END 40 l bytes before the end of the message (table-driven, bytes out of LDS), then six tree
//     levels fold lane l + 2^k into lane l through Z_(40 * 2^k) (8 nibble look-ups each).  ~450 instructions per frame
//     instead of a 4500-instruction byte-serial chain on one lane.
struct FinishLds {
    uint32_t crc[256]; uint32_t z[6 * 8 * 16]; uint32_t bufs[4][2504 / 4 + 2];
    // the scrambler sequence's bytes at phases p, p + 8, p + 16, p + 24 (mod 127) as one word: what four consecutive MPDU bytes are xored with
    uint32_t seq4[128];
    uint8_t  phase[128];         // T.scr_phase
};
__device__ __forceinline__ void finish_tables_to_lds(const RxArgs& A, FinishLds& L)      // (256 threads; the caller's barrier follows)
{
    L.crc[threadIdx.x] = A.T.crc[threadIdx.x];
    for (int i = threadIdx.x; i < 6 * 8 * 16; i += 256) L.z[i] = A.T.crcz[i];
    if (threadIdx.x < 127) {
        const uint8_t* q = A.T.scr_seq; const uint32_t p = threadIdx.x;
        L.seq4[p] = (uint32_t)q[p] | ((uint32_t)q[(p + 8u) % 127u] << 8) | ((uint32_t)q[(p + 16u) % 127u] << 16) | ((uint32_t)q[(p + 24u) % 127u] << 24);
    }
    if (threadIdx.x >= 128) L.phase[threadIdx.x - 128] = A.T.scr_phase[threadIdx.x - 128];
}
// one wave, frame row f
__device__ __forceinline__ void finish_frame(const RxArgs& A, uint32_t f, FinishLds& lds)
{
    const uint32_t* s_crc = lds.crc; const uint32_t* s_z = lds.z; uint32_t (*s_bufs)[2504 / 4 + 2] = lds.bufs;
    FrameRow& r = A.frames[f];
    if (!r.valid || r.error_code != 0) return;
    const int lane = threadIdx.x & 63;
    uint32_t* s_buf = s_bufs[threadIdx.x >> 6];
    const uint32_t* dec32 = reinterpret_cast<const uint32_t*>(A.vout + (size_t)r.slot0 * kOutPerSlot);   // (32-byte aligned; the MPDU starts at its byte 2)
    uint32_t* mp32 = reinterpret_cast<uint32_t*>(A.mpdu + (size_t)r.slot0 * kOutPerSlot);
    // (sora_rx_bind_mpdu: the same words straight into the host's array, 256 contiguous bytes per store instruction -- a lone call's MPDUs travel while the frames finish)
    uint32_t* hp32 = A.mpdu_host ? reinterpret_cast<uint32_t*>(A.mpdu_host + (size_t)r.slot0 * kOutPerSlot) : nullptr;
    const uint32_t L = min((uint32_t)r.length, 2500u), nwords = (L + 3u) / 4u;   // (k_scan queues nothing longer)
    // Four bytes per lane and pass, every load of the frame in flight before the first is used (a byte per lane and pass with the store behind it was a memory round
    // trip per 64 bytes: 22 of them in a row for fsample-6, most of what this kernel cost a lone capture).  Word w of the MPDU = bytes 2 + 4w .. 5 + 4w of the decoder's
    // output: two aligned words and a byte shift.  The words past the frame's last byte carry garbage into the LDS copy and the MPDU array's slack; nothing reads them.
    constexpr int kPasses = (2500 / 4 + 63) / 64;                                // 10
    uint32_t lo[kPasses], hi[kPasses];
#pragma unroll
    for (int k = 0; k < kPasses; k++) {
        const uint32_t w = (uint32_t)lane + 64u * (uint32_t)k, wc = min(w, nwords);
        lo[k] = dec32[wc]; hi[k] = dec32[wc + 1u];
    }
    const unsigned seed = (unsigned)__builtin_amdgcn_readfirstlane((int)lo[0]) >> 9;   // byte 0 dropped, byte 1 >> 1 seeds the register (lane 0 holds word 0)
    const unsigned phase = lds.phase[seed & 0x7F];                               // 255: seed 0 (sequence stays 0)
    uint8_t* bytes = reinterpret_cast<uint8_t*>(s_buf);
#pragma unroll
    for (int k = 0; k < kPasses; k++) {
        const uint32_t w = (uint32_t)lane + 64u * (uint32_t)k;
        if (w < nwords) {
            const uint32_t sb = phase == 255 ? 0u : lds.seq4[(phase + 32u * w) % 127u];   // scrambler byte j = seqbyte[(phase + 8 j) mod 127]
            const uint32_t o = __builtin_amdgcn_alignbyte(hi[k], lo[k], 2u) ^ sb;
            s_buf[w] = o;
            mp32[w] = o;
            if (hp32) hp32[w] = o;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                             // the LDS buffer is private to this wave
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = L >= 4 ? (int)L - 4 : 0;                                       // PHY_11a.hpp:668-673: the FCS bytes are not fed to the CRC
    uint32_t crc;
    if (n >= 4) {
        crc = crc32_wave(bytes, n, s_crc, s_z, lane);                             // lane 0 holds the register after the whole message
    } else {
        crc = 0xFFFFFFFFu;
        for (int i = 0; i < n; i++) crc = (crc >> 8) ^ s_crc[(bytes[i] ^ crc) & 0xFF];
    }
    if (lane == 0) {
        uint32_t fcs = 0;
        if (L >= 4) fcs = (uint32_t)bytes[L - 4] | ((uint32_t)bytes[L - 3] << 8) | ((uint32_t)bytes[L - 2] << 16) | ((uint32_t)bytes[L - 1] << 24);
        r.crc32 = fcs;
        r.error_code = ((~crc) == fcs) ? E_FRAME_OK : E_CRC32_FAIL;              // PHY_11a.hpp:688-692
    }
}

__global__ void __launch_bounds__(256) k_finish(RxArgs A)
{
    __shared__ FinishLds L;
    finish_tables_to_lds(A, L);
    __syncthreads();
    const JobRef jr = locate_job(blockIdx.x * 4 + (threadIdx.x >> 6), A.njobs);
    if (!jr.ok) return;
    finish_frame(A, A.joblist[jr.list * A.nrows + jr.idx], L);
}

// Behind the window-parallel trellis: k_win_redo AND k_finish as one launch -- the wave that holds a pair's proof (and decodes the pair again if it fails) descrambles and
// checks its two frames itself.  A kernel boundary less for a lone capture (1.5 us of packet + the next kernel's ramp); the same work for a batch.
//
// PIPE = the launch behind k_pipe.  k_pipe's workgroups wait for one another inside one launch, and every such wait is bounded (PipeArgs::wait_ticks): when one of them
// gave up (flags[0] != 0 -- the chip was not the launch's alone) the call's data field is simply made again HERE by the plain chain's code: the wave that owns a pair
// of frames runs k_frame's body for them (frame_symbols: VitJob + packed soft stream), decodes them with the serial trellis and finishes them.  What a call delivers
// is the reference's result whatever happened inside k_pipe; the only trace is the count in the handle's record (stats[4 kWinStatBanks]) and the note the host reads to
// leave k_pipe alone for a while (sora_hip.cpp: pipe_backoff).
struct PipeFallbackGate {
    WinProofGate proof; const RxArgs* A; FrameLds* lds; bool gave_up;
    __device__ __forceinline__ bool operator()(uint32_t list, uint32_t fa, uint32_t fb, bool hasB) const
    {
        if (!gave_up) return proof(list, fa, fb, hasB);
        frame_symbols<false>(*A, list * A->nrows + fa, *lds);
        if (hasB) frame_symbols<false>(*A, list * A->nrows + fb, *lds);
        // (the same wave reads the jobs and the soft stream back through the L2)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return true;
    }
};
template <bool PIPE>
__device__ __forceinline__ void win_redo_finish_body(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride,
                                                     const uint16_t* vecs, const uint8_t* soft, uint8_t* out, unsigned long long* stats, const RxArgs& A, uint32_t* host_note)
{
    __shared__ FinishLds L;
    finish_tables_to_lds(A, L);
    auto after = [&](uint32_t list, uint32_t fa, uint32_t fb, bool hasB) {
        finish_frame(A, A.joblist[list * A.nrows + fa], L);
        if (hasB) finish_frame(A, A.joblist[list * A.nrows + fb], L);
    };
    const WinProofGate proof{ jobs, hdr, jstride, target, vstride, vecs, stats };
    if constexpr (PIPE) {
        __shared__ FrameLds F;
        reinterpret_cast<uint32_t*>(F.demap)[threadIdx.x] = reinterpret_cast<const uint32_t*>(A.T.demap)[threadIdx.x];
        __syncthreads();
        const bool gave_up = __builtin_amdgcn_readfirstlane((int)A.pipe_flags[0]) != 0;
        if (gave_up && blockIdx.x == 0 && threadIdx.x == 0) {
            if (stats) atomicAdd(&stats[4u * kWinStatBanks], 1ull);
            if (host_note) __hip_atomic_store(host_note, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        viterbi_kernel_body<256, 24, 3>(jobs, hdr, 0u, jstride, soft, out, PipeFallbackGate{ proof, &A, &F, gave_up }, after);
    } else {
        __syncthreads();
        viterbi_kernel_body<256, 24, 3>(jobs, hdr, 0u, jstride, soft, out, proof, after);
    }
}
__global__ void __launch_bounds__(256) k_win_redo_finish(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride,
                                                         const uint16_t* vecs, const uint8_t* soft, uint8_t* out, unsigned long long* stats, RxArgs A)
{ win_redo_finish_body<false>(jobs, hdr, jstride, target, vstride, vecs, soft, out, stats, A, nullptr); }
__global__ void __launch_bounds__(256) k_win_redo_finish_pipe(const VitJob* jobs, const uint32_t* hdr, uint32_t jstride, uint32_t target, uint32_t vstride,
                                                              const uint16_t* vecs, const uint8_t* soft, uint8_t* out, unsigned long long* stats, RxArgs A, uint32_t* host_note)
{ win_redo_finish_body<true>(jobs, hdr, jstride, target, vstride, vecs, soft, out, stats, A, host_note); }

#ifdef SORA_EXP_FIN
// Round 6, MEASURED AND NOT ADOPTED (tools variant only: -DSORA_EXP_FIN=3; profiles/r06_c_unit_finish_experiment.json): the window-parallel trellis whose LAST UNIT OF A
// FRAME TO ARRIVE finishes the frame (a frame is complete when its own bytes are, not when the call's last kernel has run).
// k_viterbi16w's wave, then a tail: the wave publishes its units' bytes and vectors (release), counts each of its up to eight
// units in at its frame (wdone[list][idx]); the wave whose count completes a frame -- every other unit of the frame has published before it counted -- checks the
// frame's boundaries exactly as k_win_redo's gate does and, if they all hold, descrambles the frame, checks its CRC, stores the MPDU (into the host's page-locked
// array too when one is bound: sora_rx_bind_mpdu -- the bytes cross PCIe WHILE the other frames are still being decoded, which is what a lone call's 0.13 ms of
// finishing kernel was) and writes the row's verdict.  Nobody waits for anybody: a frame whose proof fails is simply left to k_win_redo_finish behind this kernel,
// which decodes it again and finishes it as before; for a frame finished here its finish_frame returns at once (error_code is set).  The counter is reset by the
// wave that completes it: every call finds zeros.  The finishing tables live in the trellis's own LDS block, which is free by then.
// Result, a lone 4096-capture call: identical rows, and SLOWER -- k_viterbi16w 0.262 -> 0.297 ms with the release and the counting alone (SORA_EXP_FIN=1), 0.317 with the
// proofs' reads behind the acquire (=2), 0.372 with the frames finished (0.442 when the MPDUs also cross PCIe from here), against 0.027 (0.134) ms of k_win_redo_finish
// saved: all units of a call run side by side (2048 waves on 1792 slots), so every frame's last unit arrives at the END of the launch, and the 512 waves that hold the
// frames' last pieces then finish eight frames each, one after the other, where the finishing kernel has 2048 waves doing it at once.
struct FinTail {
    const RxArgs* A; uint32_t* wdone; uint32_t jstride, vstride; const uint16_t* vecs; Lds16<256, 24>* S; const VitJob* jobs;
    // (out of line: the tail gets registers of its own instead of living beside the forward pass's 143)
    template <typename CRT, typename JOBS>
    __device__ __forceinline__ void operator()(CRT, uint32_t list, uint32_t nl, uint32_t w, uint32_t q, JOBS) const { tail<CRT::value>(list, nl, w, q); }
    template <int CR>
    __device__ __attribute__((noinline)) void tail(uint32_t list, uint32_t nl, uint32_t w, uint32_t q) const
    {
        const VitJob* __restrict__ jl = jobs + (size_t)list * jstride;
        auto job_at = [jl](uint32_t idx) { return jl[idx]; };
        static_assert(sizeof(FinishLds) <= sizeof(Lds16<256, 24>), "the finishing tables take the place of the trellis's LDS block");
        const unsigned lane = threadIdx.x & 63;
        FinishLds& L = *reinterpret_cast<FinishLds*>(S);
        bool tables = false;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                       // this wave's bytes and vectors, before its units are counted
        for (uint32_t sidx = 0; sidx < 8u; sidx++) {
            const UnitRef R = unit_ref_at<CR, 256, 24>(job_at, nl, 8u * w + sidx, q);      // (uniform: every lane works out the same position)
            if (R.uu == kWinNone) continue;
            uint32_t* cnt = wdone + (size_t)list * jstride + R.idx;
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
            if (old + 1u != R.nun) continue;
            if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(SORA_EXP_FIN) && SORA_EXP_FIN == 1
            continue;                                                            // experiment: what do the fence and the counting cost by themselves?
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                   // ... what the frame's other units published before they counted
            const size_t vec0 = (size_t)list * vstride + (size_t)R.idx * q;
            uint32_t bad = 0;
            for (uint32_t u = 1u + lane; u < R.nun; u += 64u) {
                const uint4* a = reinterpret_cast<const uint4*>(vecs + ((vec0 + u) * 2u) * 64u);
                const uint4* b = reinterpret_cast<const uint4*>(vecs + ((vec0 + u - 1u) * 2u + 1u) * 64u);
                uint32_t d = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) { const uint4 x = a[i], y = b[i]; d |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w); }
                bad |= d;
            }
            if (__ballot(bad != 0u) != 0ull) continue;                           // (k_win_redo_finish decodes the pair again and finishes it)
#if defined(SORA_EXP_FIN) && SORA_EXP_FIN == 2
            continue;                                                            // experiment: ... and the proof's reads behind the acquire?
#endif
            if (!tables) {
                tables = true;
                for (uint32_t i = lane; i < 256u; i += 64u) L.crc[i] = A->T.crc[i];
                for (uint32_t i = lane; i < 6u * 8u * 16u; i += 64u) L.z[i] = A->T.crcz[i];
                for (uint32_t pp = lane; pp < 127u; pp += 64u) {
                    const uint8_t* qq = A->T.scr_seq;
                    L.seq4[pp] = (uint32_t)qq[pp] | ((uint32_t)qq[(pp + 8u) % 127u] << 8) | ((uint32_t)qq[(pp + 16u) % 127u] << 16) | ((uint32_t)qq[(pp + 24u) % 127u] << 24);
                }
                for (uint32_t i = lane; i < 128u; i += 64u) L.phase[i] = A->T.scr_phase[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            finish_frame(*A, A->joblist[list * A->nrows + R.idx], L);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
};
__global__ void __launch_bounds__(64) k_viterbi16w_fin(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ hdr, uint32_t jstride, uint32_t target, uint32_t vstride,
                                                       const uint8_t* __restrict__ soft, uint8_t* out, uint16_t* vecs, uint32_t* wdone, RxArgs A)
{
    __shared__ Lds16<256, 24> S;
    auto jobs_of = [&](uint32_t list) { const VitJob* __restrict__ jl = jobs + (size_t)list * jstride; return [jl](uint32_t idx) { return jl[idx]; }; };
    auto ready = [](const UnitGeom&, const UnitGeom&, uint32_t) { return true; };
    viterbi16w_wave<256, 24, 3>(S, blockIdx.x, jobs_of, ready, hdr, target, vstride, soft, out, vecs, FinTail{ &A, wdone, jstride, vstride, vecs, &S, jobs });
}

#endif  // SORA_EXP_FIN

// ------------------------------------------------------------------------------------------------
// k_pack: compacts the per-capture frame table into dense sora_frame_result rows in (capture, time) order, on the
// device, so the rows can feed an RCCL all-gather without a host round trip.  mpdu_offset = slot0 * 32 indexes the
// device MPDU array directly.  One 1024-thread block: captures are scanned in tiles of 1024.
struct PackedRow { uint32_t capture_id, start_sample, end_sample, error_code, rate_kbps; uint16_t length, nsym; uint32_t crc32; int16_t cfo_est;
    uint16_t flags; uint32_t mpdu_offset; };
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
        const uint32_t found = c < ncaps ? nframes[c] : 0u;                      // k_scan counts every frame, also those past the row limit
        const uint32_t n = min(found, max_frames);
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
            o.rate_kbps = r.rate_kbps; o.length = r.length; o.nsym = r.nsym; o.crc32 = r.crc32; o.cfo_est = r.cfo_est;
            o.flags = (i + 1 == n && found > max_frames) ? 1u : 0u;               // SORA_ROW_TRUNCATED: later frames of this capture have no row
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
// (k_fft64_batch, k_demap_batch, k_deint_batch: k_stage.hip)

// sora_hip_viterbi11a takes the reference's soft format (one byte per soft value, 3 significant bits); the trellis kernels read packed
// streams out of buffers that have slack behind them (their 16-bit fetches reach one byte past a stream's end).  Block j packs job j's
// values into the workspace at byte ceil(off8[j] / 2) -- FOUR bits of room per value of the caller's range for three bits of stream, so
// that the streams of jobs whose ranges are disjoint stay disjoint for ANY offsets and ANY nsoft >= 11, in any job order: the stream
// of n values takes ceil(3 n / 8) bytes (the last, partial group of eight is written byte by byte, not padded to three bytes), and
// ceil(off / 2) + ceil(3 n / 8) <= floor((off + n) / 2) from n = 11 on.  (Round 3 placed a job at byte 3 ceil(off / 8) and padded its last
// group: disjoint only for nsoft % 8 == 0.)
__global__ void __launch_bounds__(256) k_soft_pack3(const uint8_t* soft8, const uint32_t* off8, const uint32_t* nsoft, const uint16_t* flen, const uint32_t* out_off,
                                                    int code_rate, uint8_t* packed, VitJob* jobs)
{
    const uint32_t j = blockIdx.x, mine = nsoft[j], at = (off8[j] + 1u) / 2u;
    const uint8_t* in = soft8 + off8[j];
    for (uint32_t g = threadIdx.x; g < (mine + 7u) / 8u; g += blockDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = 8u * g + k < mine ? in[8u * g + k] & 7u : 0u;
        const uint32_t bits24 = soft3_pack8(v), left = mine - 8u * g;
        if (left >= 8u) soft3_store8(packed + at, g, bits24);
        else for (uint32_t b = 0; b < (3u * left + 7u) / 8u; b++) packed[at + 3u * g + b] = (uint8_t)(bits24 >> (8u * b));
    }
    if (threadIdx.x == 0) {
        VitJob J; J.soft_off = at; J.nsoft = mine; J.length = flen[j]; J.dec_off = 0; J.out_off = out_off[j];
        J.valid = 1; J.code_rate = (uint32_t)code_rate; J.soft_bits = 3;
        jobs[j] = J;
    }
}

}  // namespace sora
