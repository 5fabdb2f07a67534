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
__global__ void __launch_bounds__(256) k_frame(RxArgs A)
{
    __shared__ uint32_t s_eq[4][4][64];                                          // [wave][symbol of the pass]: FFT staging, then the equalised bins
    __shared__ uint8_t  s_soft[4][4][288];                                       // [wave][symbol of the pass]: soft values in carrier order
    __shared__ uint8_t  s_demap[1024];                                           // DemapperCore step tables
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    const Tables& T = A.T;
    reinterpret_cast<uint32_t*>(s_demap)[threadIdx.x] = reinterpret_cast<const uint32_t*>(T.demap)[threadIdx.x];
    __syncthreads();                                                             // the only block barrier: the waves are independent from here on
    const JobRef jr = locate_job(blockIdx.x * 4 + w, A.njobs);
    if (!jr.ok) return;
    const uint32_t j = jr.list * A.nrows + jr.idx;                               // slot of the job in jobs[] / joblist[]
    const uint32_t f = A.joblist[j];
    const FrameRow r = A.frames[f];
    const uint32_t my_nsoft = (uint32_t)r.nsym * 48u * r.nbpsc;
    if (lane == 0) {
        VitJob J;
        J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftBytesPerSlot; J.nsoft = my_nsoft; J.length = r.length;
        J.dec_off = 0; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate; J.soft_bits = 3;
        A.jobs[j] = J;
    }
    const FrameCtx* fx = A.fctx + f;
    const uint32_t* iq = A.iq + A.caps[r.capture].offset;
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    PkTw fq[4], ch[4];                                                           // FreqCoeffs / ChannelCoeffs as the operand pairs of the packed complex product
#pragma unroll
    for (int m = 0; m < 4; m++) { fq[m] = pk_tw_mul(fx->freq[e + 16 * m]); ch[m] = pk_tw_mul(fx->chan[e + 16 * m]); }
    const int nb = __builtin_amdgcn_readfirstlane((int)r.nbpsc), ncbps = 48 * nb, nsym = __builtin_amdgcn_readfirstlane((int)r.nsym);   // (wave-uniform: scalar branches on the modulation)
    // de-interleaver source indices of output positions 8 lane .. 8 lane + 7 (lane < N_CBPS / 8: one three-byte group of the packed stream), two per register
    uint32_t mp[4];
    const bool packs = 8 * lane < ncbps;
    {
        const uint16_t* map = T.deint + (nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3) * 288;
#pragma unroll
        for (int t = 0; t < 4; t++) mp[t] = packs ? (uint32_t)map[8 * lane + 2 * t] | ((uint32_t)map[8 * lane + 2 * t + 1] << 16) : 0u;
    }
    uint8_t* dst = A.soft + (size_t)r.slot0 * kSoftBytesPerSlot;                  // the frame's packed soft stream: symbol s (1-based) at 3 N_CBPS / 8 * (s - 1) bytes
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
    uint32_t raw[4];
    load_samples(1, raw);
    for (int s0 = 1; s0 <= nsym; s0 += 4) {
        const int sym = s0 + g;
        const bool active = sym <= nsym;
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
        // ---- the loop-carried part, symbols s0 .. s0+3 in order
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
        // ---- TPhaseCompensate + TPilotTrack::_rotate + T11aDemap, 3 data carriers per lane
        if (active) {
            const int my_cfo = g == 0 ? t_cfo[0] : g == 1 ? t_cfo[1] : g == 2 ? t_cfo[2] : t_cfo[3];
            const int my_sfo = g == 0 ? t_sfo[0] : g == 1 ? t_sfo[1] : g == 2 ? t_sfo[2] : t_sfo[3];
            const int my_avg = g == 0 ? t_avg[0] : g == 1 ? t_avg[1] : g == 2 ? t_avg[2] : t_avg[3];
            const int my_del = g == 0 ? t_del[0] : g == 1 ? t_del[1] : g == 2 ? t_del[2] : t_del[3];
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

__global__ void __launch_bounds__(256) k_sym_front(RxArgs A)
{
    __shared__ uint32_t s_eq[4][4][64];                                          // [wave][group]: FFT staging
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    const Tables& T = A.T;
    const uint32_t slot_first = (blockIdx.x * 4u + (uint32_t)w) * (4u * kSlotIters);
    if (slot_first >= A.total_slots) return;
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // the owners of the wave's 16 slots: lane l < 16 asks for slot slot_first + l, the groups pick theirs up by cross-lane reads
    const uint32_t my_slot = slot_first + (uint32_t)(lane & 15);
    const uint32_t my_own = (lane < 16 && my_slot < A.total_slots) ? A.slot_row[my_slot] : 0xFFFFFFFFu;
    const unsigned long long owned = __ballot(my_own != 0xFFFFFFFFu);
    if (owned == 0) return;                                                      // preamble / silence only
    const uint32_t row0 = (uint32_t)__builtin_amdgcn_readlane((int)my_own, __builtin_ctzll(owned));
    const bool one_frame = __ballot(lane < 16 && my_own != 0xFFFFFFFFu && my_own != row0) == 0;   // the usual case: every owned slot of the wave belongs to ONE frame
    uint32_t own[kSlotIters];
#pragma unroll
    for (int it = 0; it < kSlotIters; it++) own[it] = (uint32_t)__shfl((int)my_own, 4 * it + g);
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    uint32_t* sl = s_eq[w][g];
    uint4* eq4 = reinterpret_cast<uint4*>(A.eq);
    auto store = [&](uint32_t slot, const uint32_t o[4]) {
        eq4[(size_t)slot * 16u + (uint32_t)e] = uint4{ o[0], o[1], o[2], o[3] };
        // the four pilot bins once more, densely (16 bytes per slot: k_track reads nothing else): bins 43, 57, 7, 21 = (e, q) (10,3), (14,1), (1,3), (5,1)
        if (e == 10) A.pil[(size_t)slot * 4u + 0u] = o[3];
        if (e == 14) A.pil[(size_t)slot * 4u + 1u] = o[1];
        if (e == 1)  A.pil[(size_t)slot * 4u + 2u] = o[3];
        if (e == 5)  A.pil[(size_t)slot * 4u + 3u] = o[1];
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
    const uint32_t* pp = A.pil + (size_t)(r.slot0 + 1u) * 4u + (uint32_t)pk;    // pilot k of data symbol s at pp[4 (s - 1)]: 16 bytes per symbol and frame (k_sym_front)
    TrackRec* trk = A.track + r.slot0 + 1u;
    constexpr int kAhead = 4;                                                    // symbols requested ahead of the one in the chain (each step is two dependent table reads long)
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
    __shared__ uint32_t s_pol[128];                                              // the pilot polarities (pilot.hpp:10-28, period 127) of the eight symbols from count c on, one bit each
    if (threadIdx.x < 127) {
        unsigned b = 0;
#pragma unroll
        for (unsigned u = 0; u < 8; u++) b |= pilot_sgn((threadIdx.x + u) % 127u) << u;
        s_pol[threadIdx.x] = b;
    }
    {
        // 129 KB from L2 / HBM: eleven 16-byte loads per thread in flight at a time (a loop that waits for every load costs a memory latency per 4 KB: 33 of them)
        const uint4* src = reinterpret_cast<const uint4*>(A.T.trk);
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
    int cfo = r.cfo_comp, sfo = r.sfo_comp, ctr = r.cfo_tracker, str = r.sfo_tracker;   // cfo / sfo wrapped to 16 bits after every step; the trackers run free (they are only ever added)
    int nmax = nsym;
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    // pilot k of data symbol s is word 4 (slot0 + s) + k of pil[] (k_sym_front); a record goes to track[slot0 + s].  32-bit word offsets from the arrays' bases, so that a
    // request is a minimum, a shift-add and the load.  Symbols past the frame's last re-read its last pilots and their records land on the slot behind the frame --
    // preamble or silence of whatever follows, a slot no frame owns and k_sym_back never reads.
    const uint32_t w0 = (jr.ok ? (r.slot0 + 1u) * 4u + (uint32_t)pk : (uint32_t)pk) * 4u;   // BYTE offsets: a uniform base plus a 32-bit lane offset is one address operand pair
    const uint32_t t0 = (jr.ok ? r.slot0 + 1u : A.total_slots + 1u) * 8u;        // (no job: a slot in the arrays' slack)
    const char* __restrict__ pil = reinterpret_cast<const char*>(A.pil);
    char* __restrict__ trk2 = reinterpret_cast<char*>(A.track);
    const unsigned last = (unsigned)max(nsym, 1) - 1u, nrec = (unsigned)nsym;
    constexpr int kAhead = 8;                                                    // symbols requested ahead of the one in the chain (a step is ~0.1 us, an L2 miss ten times that)
    uint32_t q[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; i++) q[i] = *reinterpret_cast<const uint32_t*>(pil + (w0 + 16u * min((unsigned)i, last)));
    unsigned cnt = 0;                                                            // symbol_count: 127 -> 0 after the SIGNAL symbol; the same in every frame of the wave
    constexpr float kInv28 = 0.0357142873108387f;                                // 0x3D124925: trunc((float)d * kInv28) == d / 28 (C division) for every |d| <= 65535
    for (int s0 = 1; s0 <= nmax; s0 += kAhead) {
        const unsigned pol = (unsigned)__builtin_amdgcn_readfirstlane((int)s_pol[cnt]);   // the polarities of the block's eight symbols, one scalar byte
        cnt = cnt + (unsigned)kAhead >= 127u ? cnt + (unsigned)kAhead - 127u : cnt + (unsigned)kAhead;
#pragma unroll
        for (int u = 0; u < kAhead; u++) {                                       // (unrolled: the request ring's slots are registers)
            const int s = s0 + u;
            const uint32_t cur = q[u];
            q[u] = *reinterpret_cast<const uint32_t*>(pil + (w0 + 16u * min((unsigned)(s + kAhead - 1), last)));
            const int flip = (int)((pol << (15 - u)) & 0x8000u);
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
            th = __builtin_amdgcn_sbfe(th ^ flip, 0, 16);                        // + 0x8000 mod 2^16 for a pilot of polarity -1
            int th1, th2, th3, th4;                                              // the four angles of the quad, in every lane (assembler: see k_track)
            asm volatile("s_nop 1\n\t"
                         "v_mov_b32_dpp %0, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %2, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %3, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(th1), "=&v"(th2), "=&v"(th3), "=&v"(th4) : "v"(th));
            const int sum = th1 + th2 + th3 + th4;
            const int avg = (sum + ((sum >> 31) & 3)) >> 2;                      // (th1 + th2 + th3 + th4) / 4, towards zero: within 16 bits
            const int del = ((int)((float)(th3 - th1) * kInv28) + (int)((float)(th4 - th2) * kInv28)) >> 1;
            // TrackRec { cfo_comp, sfo_comp, avg, del } of THIS symbol: the quad's four lanes store the same eight bytes (no exec juggling)
            *reinterpret_cast<uint2*>(trk2 + (t0 + 8u * min((unsigned)(s - 1), nrec))) = uint2{ ((uint32_t)cfo & 0xFFFFu) | ((uint32_t)sfo << 16), ((uint32_t)avg & 0xFFFFu) | ((uint32_t)del << 16) };
            ctr += avg >> 2; str += del >> 2;
            cfo = __builtin_amdgcn_sbfe(cfo + avg + ctr, 0, 16); sfo = __builtin_amdgcn_sbfe(sfo + del + str, 0, 16);
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
    int cur_nb = 0; uint32_t mp[4] = { 0, 0, 0, 0 };                             // de-interleaver entries of output positions 8 lane .. 8 lane + 7 for modulation cur_nb
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
__device__ __forceinline__ void viterbi_forward(const VitJob& JA, const VitJob& JB, bool hasB, const uint8_t* __restrict__ soft_base, uint8_t* __restrict__ out_base, uint16_t* ring, uint16_t* ops)
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

    auto normalize = [&]() { V.U = V.U - dpp_pkmin_wave(V.U); };                // Normalize (viterbicore.h:444-465), both frames; marks and guard are clear here and no half borrows (its minimum is subtracted): one 32-bit VOP2
#ifdef SORA_DBG_NO_TRACE                                                        // experiment (tools/ab_decode.sh): the forward pass alone -- results are wrong, only the duration means something
    auto trace = [&](unsigned, unsigned, uint32_t, uint32_t, uint32_t) {};
#else
    auto trace = [&](unsigned mA, unsigned mB, uint32_t cntA, uint32_t cntB, uint32_t top) { viterbi_trace<RG::kMaxWalk>(V.U, ring, tr, ob, mA, mB, cntA, cntB, A.out, B.out, top); };
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
    uint16_t* my_op = ops + 2u * my_k + (lane >> 5);                           // operand k, frame's half (k up to 31: the table has 32 operands, those past CW are never read)
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

// Four waves per 256-thread workgroup, two frames per wave (no cross-wave traffic).  One-wave workgroups were kept to
// 8 per CU by the dispatcher: 2 waves per SIMD and a second round for a 4096-frame batch.
// Frames are queued per code rate (k_scan), so the two frames of a wave always share the puncture pattern; the last
// frame of an odd list runs alone in the low half.  (Jobs given through sora_hip_viterbi11a are one list of one rate.)
struct DecodeEveryPair { __device__ __forceinline__ bool operator()(uint32_t, uint32_t, uint32_t, bool) const { return true; } };
template <int WIN, int LOOK, int BITS, typename GATE = DecodeEveryPair>
__device__ __forceinline__ void viterbi_kernel_body(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out,
                                                    GATE gate = GATE())
{
    __shared__ uint16_t s_ring[4][RingGeom<WIN, LOOK>::kEntries];                // 39 KB / 33 KB: survivor history, two copies of every block (RingGeom), per wave
    __shared__ uint16_t s_ops[4][64];                                            // [wave][operand of the chunk][frame]: the soft values as metric fields (viterbi_forward); with it under 40 KB: four workgroups per CU
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
    if (!gate(list, fa, fb, fb < njobs)) return;                                 // (k_win_redo: the pair's proofs hold, nothing to decode again)
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
}

__global__ void __launch_bounds__(256) k_viterbi(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{ viterbi_kernel_body<256, 24, 3>(jobs, njobs3, njobs_single, stride, soft, out); }
// the 802.11n graph's decoder: T11aViterbi<5000*8, 312, 192, 36> (fb11ndemod_config.hpp:199)
__global__ void __launch_bounds__(256) k_viterbi11n(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{ viterbi_kernel_body<192, 36, 8>(jobs, njobs3, njobs_single, stride, soft, out); }

// The window-parallel trellis's proof AND the serial decode of what failed it, in one launch (k_vitwin.hip describes the proof; a verify kernel + k_viterbi as two
// launches cost a lone capture a kernel and a queue gap).  A wave owns a PAIR of frames of one code-rate list, as in k_viterbi: its lower half compares the unit
// boundaries of frame A (unit u's vector at its verify point against unit u - 1's vector at the same step), its upper half those of frame B; if every boundary of both
// holds the wave is done -- what k_viterbi16w wrote is the reference's decode -- else it decodes the pair serially, overwriting it.
struct WinProofGate {
    const VitJob* jobs; const uint32_t* hdr; uint32_t jstride, target, vstride; const uint16_t* vecs; unsigned long long* stats;
    __device__ __forceinline__ bool operator()(uint32_t list, uint32_t fa, uint32_t fb, bool hasB) const
    {
        const unsigned lane = threadIdx.x & 63, side = lane >> 5, l32 = lane & 31u;
        const uint32_t q = win_units_per_frame(hdr[0] + hdr[1] + hdr[2], target);
        const uint32_t idx = side ? fb : fa;
        const bool have = side ? hasB : true;
        const VitJob& J = jobs[(size_t)list * jstride + (have ? idx : fa)];
        const uint32_t nev = win_events(J.length, J.code_rate, 256u, 24u), m = win_per_unit(nev, q), nun = have ? (nev + m - 1u) / m : 1u;
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
__global__ void __launch_bounds__(256) k_win_redo(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ hdr, uint32_t jstride, uint32_t target, uint32_t vstride, const uint16_t* __restrict__ vecs,
                                                  const uint8_t* __restrict__ soft, uint8_t* __restrict__ out, unsigned long long* __restrict__ stats)
{ viterbi_kernel_body<256, 24, 3>(jobs, hdr, 0u, jstride, soft, out, WinProofGate{ jobs, hdr, jstride, target, vstride, vecs, stats }); }

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
__global__ void __launch_bounds__(256) k_finish(RxArgs A)
{
    __shared__ uint32_t s_crc[256];
    __shared__ uint32_t s_z[6 * 8 * 16];
    __shared__ uint32_t s_bufs[4][2504 / 4 + 2];
    s_crc[threadIdx.x] = A.T.crc[threadIdx.x];
    for (int i = threadIdx.x; i < 6 * 8 * 16; i += 256) s_z[i] = A.T.crcz[i];
    __syncthreads();
    const JobRef jr = locate_job(blockIdx.x * 4 + (threadIdx.x >> 6), A.njobs);
    if (!jr.ok) return;
    const uint32_t f = A.joblist[jr.list * A.nrows + jr.idx];
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

// ------------------------------------------------------------------------------------------------
// k_pack: compacts the per-capture frame table into dense sora_frame_result rows in (capture, time) order, on the
// device, so the rows can feed an RCCL all-gather without a host round trip.  mpdu_offset = slot0 * 32 indexes the
// device MPDU array directly.  One 1024-thread block: captures are scanned in tiles of 1024.
struct PackedRow { uint32_t capture_id, start_sample, end_sample, error_code, rate_kbps; uint16_t length, nsym; uint32_t crc32; int16_t cfo_est; uint16_t flags; uint32_t mpdu_offset; };
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
