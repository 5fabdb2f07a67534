// k_decode.hip -- the data field of every frame, from IQ samples to decoded bytes, in ONE kernel (gfx950):
//
//   T11aDataSymbol -> TFreqCompensation -> TFFT64 -> TChannelEqualization -> TPhaseCompensate -> TPilotTrack -> T11aDemap<N> ->
//   T11aDeinterleave -> [TThreadSeparator] -> T11aViterbi<5000*8,48,256,24>           (fb11ademod_config.hpp:200-228)
//
// The reference runs the two halves of this chain on two threads joined by a ring buffer (TThreadSeparator,
// stdbrick.hpp:89-248: RxThread feeds soft values, ViterbiThread decodes them).  This kernel is the same structure inside a
// workgroup: of its four waves, two are SYMBOL waves (the RxThread half: one OFDM symbol per 16-lane group, four per pass)
// and two are TRELLIS waves (the ViterbiThread half: 64 states = 64 lanes, two frames per wave in packed 16-bit metrics,
// dev_viterbi.h); symbol wave i feeds trellis wave i through a ring in LDS.  No soft value ever reaches HBM (the first
// version of this path wrote 132 MB of them per call and read them back), the latency-bound symbol work (LUT reads for the
// pilot loop) hides behind the issue-bound trellis of the same CU, and the trellis gets its branch-metric operands as
// ready-made VGPRs (one broadcast ds_read_b128 per four soft values) instead of packing them in the scalar unit.
//
// The ring.  A trellis wave decodes frames A and B (same code rate) in lockstep, so the ring holds OPERANDS: dword i =
// (soft value i of A) << 9 | (soft value i of B) << 25, exactly what acs_step adds.  2304 dwords (a common multiple of every
// pass size 4 x N_CBPS and of every 12-step chunk 24 / 18 / 16), produced/consumed counters in LDS with workgroup-scope
// release/acquire; the symbol wave always serves the frame that is further behind.
//   algorithmic bytes per data symbol: 256 read (64 of the 80 samples) + N_DBPS / 8 written
#include <hip/hip_runtime.h>
#include "kernels.h"
#include <type_traits>
#include "dev_viterbi.h"

namespace sora {

__device__ __constant__ uint8_t kPilotSgnD[128] = {       // pilot.hpp:10-28: 1 <=> polarity -1
    0,0,0,1,1,1,0,1, 1,1,1,0,0,1,0,1, 1,0,0,1,0,0,1,0, 0,0,0,0,0,1,0,0,
    0,1,0,0,1,1,0,0, 0,1,0,1,1,1,0,1, 0,1,1,0,1,1,0,0, 0,0,0,1,1,0,0,1,
    1,0,1,0,1,0,0,1, 1,1,0,0,1,1,1,1, 0,1,1,0,1,0,0,0, 0,1,0,1,0,1,0,1,
    1,1,1,1,0,1,0,0, 1,0,1,0,0,0,1,1, 0,1,1,1,0,0,0,1, 1,1,1,1,1,1,0,0 };

constexpr int kSoftRing = 2304;                            // operands per trellis wave (9216 bytes)

struct DecodeLds {
    uint32_t ring[2][kSoftRing];                           // [pair] soft operands
    uint16_t surv[2][RingGeom<256, 24>::kEntries];         // [pair] survivor history of the trellis wave (dev_viterbi.h: two copies of every block)
    uint32_t eq[2][4][64];                                 // [pair][symbol of the pass] FFT staging, then the equalised bins
    uint8_t  soft[2][4][288];                              // [pair][symbol of the pass] soft values in carrier order
    uint16_t map[2][2][288];                               // [pair][frame] de-interleaver source index of the frame's modulation
    uint8_t  demap[1024];                                  // DemapperCore step tables
    uint32_t produced[2][2];                               // [pair][frame] soft values written to the ring so far
    uint32_t consumed[2];                                  // [pair] soft values the trellis wave no longer needs
};

__device__ __forceinline__ uint32_t lds_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct FrameGeom {                                         // wave-uniform description of one frame's data field
    const uint32_t* iq; const FrameCtx* fx;
    uint32_t data_start, nsym, nb, ncbps, length, out_off, nsoft;
    int cfo_comp, sfo_comp, cfo_tr, sfo_tr;
};
__device__ __forceinline__ FrameGeom frame_geom(const RxArgs& A, uint32_t f)
{
    const FrameRow& r = A.frames[f];
    FrameGeom G;
    G.iq = A.iq + A.caps[uni(r.capture)].offset; G.fx = A.fctx + f;
    G.data_start = uni(r.data_start); G.nsym = uni(r.nsym); G.nb = uni(r.nbpsc); G.ncbps = 48u * G.nb; G.length = uni(r.length);
    G.out_off = uni(r.slot0) * (uint32_t)kOutPerSlot; G.nsoft = G.nsym * G.ncbps;
    G.cfo_comp = (int)uni((uint32_t)(int)r.cfo_comp); G.sfo_comp = (int)uni((uint32_t)(int)r.sfo_comp);
    G.cfo_tr = (int)uni((uint32_t)(int)r.cfo_tracker); G.sfo_tr = (int)uni((uint32_t)(int)r.sfo_tracker);
    return G;
}

// ------------------------------------------------------------------------------------------------ the symbol wave
// One pass = up to four consecutive data symbols of ONE frame (one per 16-lane group): TFreqCompensation, TFFT64 and
// TChannelEqualization on packed COMPLEX16 (dev_arith.h), the loop-carried pilot tracking (freqoffset.hpp:28-30,
// pilot.hpp:166-233: four pilot bins, two dependent LUT reads per symbol) for the pass's symbols in order with pilot k in
// lane k, then rotation, demap (demapper.h:16-45) and de-interleave straight into the frame's half of the ring operands.
__device__ __forceinline__ void symbol_wave(const RxArgs& A, DecodeLds& S, int pi, const FrameGeom& GA, const FrameGeom& GB, bool hasB)
{
    const Tables& T = A.T;
    const int lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    uint32_t* s = S.eq[pi][g];
    auto fill_map = [&](int x, const FrameGeom& G) {
        const uint16_t* map = T.deint + (G.nb == 1 ? 0 : G.nb == 2 ? 1 : G.nb == 4 ? 2 : 3) * 288;
        for (uint32_t i = lane; i < G.ncbps; i += 64) S.map[pi][x][i] = map[i];
    };
    fill_map(0, GA);
    if (hasB) fill_map(1, GB);
    wave_lds_sync();
    // pilot k in lane k: bins 43, 57, 7, 21 = carriers -21, -7, +7, +21 (pilot.hpp:138-164)
    const int pk = lane & 3;
    const int pbin = pk == 0 ? 43 : pk == 1 ? 57 : pk == 2 ? 7 : 21, pc = pk == 0 ? -21 : pk == 1 ? -7 : pk == 2 ? 7 : 21;
    // per-frame loop state (wave-uniform): symbols done, values published, tracking loop (symbol_count: 127 -> 0 after SIGNAL)
    uint32_t s0[2] = { 1u, 1u }, wv[2] = { 0u, 0u }, symbol_count[2] = { 0u, 0u };
    int cfo_comp[2] = { GA.cfo_comp, GB.cfo_comp }, sfo_comp[2] = { GA.sfo_comp, GB.sfo_comp }, cfo_tr[2] = { GA.cfo_tr, GB.cfo_tr }, sfo_tr[2] = { GA.sfo_tr, GB.sfo_tr };
    uint32_t seen_consumed = 0;
    // one pass over up to four symbols of frame X (instantiated for X = 0 and X = 1: all per-frame state is addressed statically)
    auto pass = [&](const FrameGeom& G, auto XC) {
        constexpr int X = decltype(XC)::value;
        const uint32_t sym0 = s0[X], wbase = wv[X];
        const uint32_t nact = min(4u, G.nsym - sym0 + 1u), P = nact * G.ncbps;
        while ((int32_t)(wbase + P - seen_consumed) > kSoftRing) {               // room in the ring for this pass?
            seen_consumed = uni(lds_acquire(&S.consumed[pi]));
            if ((int32_t)(wbase + P - seen_consumed) > kSoftRing) __builtin_amdgcn_s_sleep(8);
        }
#ifdef SORA_DBG_NO_SYMBOLS                                                      // experiment: the trellis side alone (operands = whatever the ring holds)
        s0[X] += 4; wv[X] += P; lds_release(&S.produced[pi][X], wbase + P); return;
#endif
        // ---- TFreqCompensation + TFFT64 + TChannelEqualization, symbol sym0 + g in group g
        const uint32_t sym = sym0 + (uint32_t)g;
        const bool active = sym <= G.nsym;
        {
            const uint32_t p0 = G.data_start + 80u * sym + 8u;                   // skip_cp = 8 (PHY_11a.hpp:365,394)
            pcx x[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t raw = active ? G.iq[(size_t)(p0 + (uint32_t)(e + 16 * m)) * A.str] : 0u;
                x[m] = pk_cmul<15>(pk_sra(raw, 1), pk_tw_mul(G.fx->freq[e + 16 * m]));   // >>1, x FreqCoeffs (channel_11a.hpp:643-644)
            }
            fft64_core_pk(x, s, e, W, wave_lds_sync);
            pcx Y[4];
#pragma unroll
            for (int q = 0; q < 4; q++) Y[q] = s[__brev((unsigned)(e + 16 * q)) >> 26];
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < 4; q++) {                                        // channel_11a.hpp:548-574
                const int bin = e + 16 * q;
                s[bin] = (bin >= 28 && bin < 36) ? 0u : pk_cmul<8>(Y[q], pk_tw_mul(G.fx->chan[bin]));
            }
            wave_lds_sync();
        }
        // ---- the loop-carried part, symbols sym0 .. sym0+3 in order
        int cc = cfo_comp[X], sc = sfo_comp[X], ct = cfo_tr[X], st = sfo_tr[X];
        unsigned cnt = symbol_count[X];
        int t_cfo[4], t_sfo[4], t_avg[4], t_del[4];
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            t_cfo[gg] = cc; t_sfo[gg] = sc; t_avg[gg] = 0; t_del[gg] = 0;
            if ((uint32_t)gg < nact) {
                cpx p = mul_q15(unpack(S.eq[pi][gg][pbin]), rot_coeff(T, w16(cc + pc * sc)));
                int th = pk == 3 ? uatan2(T, -p.im, -p.re) : uatan2(T, p.im, p.re);
                if (kPilotSgnD[cnt]) th = w16(th + 0x8000);
                cnt++; if (cnt >= 127) cnt = 0;
                const int th1 = __builtin_amdgcn_readlane(th, 0), th2 = __builtin_amdgcn_readlane(th, 1);
                const int th3 = __builtin_amdgcn_readlane(th, 2), th4 = __builtin_amdgcn_readlane(th, 3);
                const int avg = w16((th1 + th2 + th3 + th4) / 4);
                const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
                t_avg[gg] = avg; t_del[gg] = del;
                ct = w16(ct + (avg >> 2)); st = w16(st + (del >> 2));
                cc = w16(cc + avg + ct); sc = w16(sc + del + st);
            }
        }
        cfo_comp[X] = cc; sfo_comp[X] = sc; cfo_tr[X] = ct; sfo_tr[X] = st; symbol_count[X] = cnt;
        // ---- TPhaseCompensate + TPilotTrack::_rotate + T11aDemap, 3 data carriers per lane
        const int nb = (int)G.nb;
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
                cpx v = unpack(s[carrier_bin48(k)]);
                v = mul_q15(v, c1[m]);
                v = mul_q15(v, c2[m]);
                int re = v.re >> 4, im = v.im >> 4;                               // demap_limit<64> (demapper.h:141-151)
                re = min(max(re, -128), 127); im = min(max(im, -128), 127);
                const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
                uint8_t* o = S.soft[pi][g] + k * nb;                             // DemapperCore::Demap<N_BPSC> (demapper.h:16-45)
                if (nb == 1) { o[0] = S.demap[ur]; }
                else if (nb == 2) { o[0] = S.demap[ur]; o[1] = S.demap[ui]; }
                else if (nb == 4) { o[0] = S.demap[ur]; o[1] = S.demap[256 + ur]; o[2] = S.demap[ui]; o[3] = S.demap[256 + ui]; }
                else { o[0] = S.demap[ur]; o[1] = S.demap[512 + ur]; o[2] = S.demap[768 + ur];
                       o[3] = S.demap[ui]; o[4] = S.demap[512 + ui]; o[5] = S.demap[768 + ui]; }
            }
        }
        wave_lds_sync();
        // ---- T11aDeinterleave*: out[k] = in[j(k)], written as this frame's 16-bit half (v << 9) of the ring operands.  A pass
        // never straddles the end of the ring: its start is a multiple of the pass size, which divides the ring size.
        {
            uint16_t* half = reinterpret_cast<uint16_t*>(S.ring[pi]) + X;
            const uint16_t* mp = S.map[pi][X];
            uint32_t at = wbase % (uint32_t)kSoftRing;
            for (uint32_t gs = 0; gs < nact; gs++, at += G.ncbps)
                for (uint32_t k = lane; k < G.ncbps; k += 64) half[2u * (at + k)] = (uint16_t)((uint32_t)S.soft[pi][gs][mp[k]] << 9);
        }
        s0[X] += 4; wv[X] += P;
        lds_release(&S.produced[pi][X], wbase + P);                             // the operands above are visible before the counter
    };
    for (;;) {
        const bool doneA = s0[0] > GA.nsym, doneB = !hasB || s0[1] > GB.nsym;
        if (doneA && doneB) break;
        const bool pickB = doneA || (!doneB && wv[1] < wv[0]);                   // the frame that is further behind
        if (pickB) pass(GB, std::integral_constant<int, 1>{}); else pass(GA, std::integral_constant<int, 0>{});
    }
}

// ------------------------------------------------------------------------------------------------ the trellis wave
// T11aViterbi<5000*8,48,256,24>::Process over two frames in lockstep (dev_viterbi.h has the arithmetic): the forward loop of
// k_viterbi (k_rx.hip) with its soft operands read out of the ring -- 12-step chunks of 24 / 18 / 16 operands, every lane
// reading the same address (a broadcast), the next chunk requested before the current one is stepped through.
struct VitSideR { uint8_t* out; uint32_t nsteps, tr_end, nsoft; bool done; };

template <int CR>
__device__ __forceinline__ void trellis_wave(DecodeLds& S, int pi, const FrameGeom& GA, const FrameGeom& GB, bool hasB, uint8_t* __restrict__ out_base)
{
    constexpr int GBv = CR == 0 ? 2 : CR == 2 ? 4 : 3;                          // soft values per puncture group (CR: 0=1/2, 1=2/3, 2=3/4)
    constexpr int GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;                           // trellis steps per group
    constexpr int VC = 12 / GS * GBv;                                           // operands per 12-step chunk: 24 / 18 / 16
    static_assert(kSoftRing % VC == 0, "a chunk never straddles the end of the ring");
    const unsigned lane = threadIdx.x & 63;
    uint16_t* ring = S.surv[pi];
    const uint32_t* soft = S.ring[pi];
    VitSideR A, B;
    A.out = out_base + GA.out_off; A.nsoft = GA.nsoft; A.nsteps = GA.nsoft / GBv * GS; A.tr_end = GA.length * 8u + 16u + 6u; A.done = false;
    B.out = out_base + GB.out_off; B.nsoft = hasB ? GB.nsoft : 0u; B.nsteps = hasB ? GB.nsoft / GBv * GS : 0u; B.tr_end = hasB ? GB.length * 8u + 16u + 6u : 0u; B.done = !hasB;

    auto which_of = [](int ph) { return CR == 0 ? 0 : CR == 1 ? (ph & 1) : ph % 3; };   // step kinds of a puncture group (viterbi.hpp:167-187)
    VitLane V;
    const unsigned vl = lane_map(lane);                                         // label lane: holds state rol6^t(vl) after t steps
    V.U = vl == 0 ? 0u : 0x18u * kFld;                                         // ALL_INIT0 / ALL_INIT = 0x00 / 0x30 (viterbilut.h:22-30)
    V.ring = ring; V.rowpos = 0;
    constexpr int P = RingGeom<256, 24>::P;
    V.sidx[0] = __brev(rol6(vl, 2)) >> 26; V.sidx[1] = __brev(rol6(vl, 4)) >> 26; V.sidx[2] = __brev(vl) >> 26;
#pragma unroll
    for (int t = 0; t < 24; t++) {
        const int ph = t % 6, k = t % 8;
        const unsigned n = rol6(vl, ph + 1);
        const bool own1 = ph >= 2 && ((vl >> (5 - ph)) & 1);
        const unsigned ma = (__popc(n & 0155) & 1) ? 7u * kFld : 0u, mb = (__popc(n & 0117) & 1) ? 7u * kFld : 0u;
        const unsigned mx = which_of(ph) == 2 ? mb : ma;
        V.MX[t] = own1 ? ((mx ^ (7u * kFld)) | (kOne << k)) : mx;
        if (t < 6) V.MY[t] = own1 ? (mb ^ (7u * kFld)) : mb;
    }
    const uint32_t nsteps = max(A.nsteps, B.nsteps);
    uint32_t tr = 0, ob = 0;

    auto normalize = [&]() { V.U = V.U - dpp_pkmin_wave(V.U); };                // (no half borrows: a plain 32-bit subtraction)
    auto trace = [&](unsigned mA, unsigned mB, uint32_t cntA, uint32_t cntB, uint32_t top) { viterbi_trace<RingGeom<256, 24>::kMaxWalk>(V.U, ring, tr, ob, mA,
            mB, cntA, cntB, A.out, B.out, top); };
    auto next_event = [&]() -> uint32_t {
        uint32_t t = ob + 256u + 24u + 6u;
        if (!A.done) t = min(t, A.tr_end);
        if (!B.done) t = min(t, B.tr_end);
        return t;
    };
    uint32_t next_thr = next_event();
    auto check = [&](int t24_last) {                                            // trace-back schedule (viterbi.hpp:196-214), per frame
        if (tr >= next_thr) {
            const int k = t24_last % 8;
            const uint32_t pos = V.rowpos + (uint32_t)(t24_last / 8) * 64u;     // ring position (x 64) of block (tr - 1) >> 3
            unsigned lastA, lastB;
            if (k == 7) { const unsigned w = ring[pos + V.sidx[t24_last / 8]]; lastA = (w >> 7) & 1u; lastB = (w >> 15) & 1u; }
            else { lastA = (V.U >> k) & 1u; lastB = (V.U >> (17 + k)) & 1u; }
            const unsigned mA = ((V.U & 0xFFFFu) >> 9 << 1) | lastA, mB = (V.U >> 25 << 1) | lastB;
            const bool partial = tr >= ob + 256u + 24u + 6u;
            uint32_t cntA = 0, cntB = 0;
            if (!A.done) {
                if (tr >= A.tr_end) { cntA = A.tr_end - ob - 6; A.done = true; }
                else if (partial) cntA = 256;
            }
            if (!B.done) {
                if (tr >= B.tr_end) { cntB = B.tr_end - ob - 6; B.done = true; }
                else if (partial) cntB = 256;
            }
            if (cntA | cntB) trace(mA, mB, cntA, cntB, (pos >> 6) + (uint32_t)P);
            if (partial) ob += 256;
            next_thr = next_event();
        }
    };
    struct Chunk { uint32_t v[VC]; };
    uint32_t seenA = 0, seenB = 0;
    auto load_chunk = [&](uint32_t c) -> Chunk {                                // waits until both frames' operands of chunk c are in the ring
        const uint32_t end = (c + 1u) * (uint32_t)VC;
        const uint32_t needA = min(end, A.nsoft), needB = min(end, B.nsoft);    // past a frame's end: whatever the ring holds (that frame is done by then)
        while (seenA < needA) { seenA = uni(lds_acquire(&S.produced[pi][0])); if (seenA < needA) __builtin_amdgcn_s_sleep(4); }
        while (seenB < needB) { seenB = uni(lds_acquire(&S.produced[pi][1])); if (seenB < needB) __builtin_amdgcn_s_sleep(4); }
        Chunk K;
        const uint32_t* p = soft + (c * (uint32_t)VC) % (uint32_t)kSoftRing;
        if (VC % 4 == 0) {
#pragma unroll
            for (int i = 0; i < VC / 4; i++) { const uint4 q = reinterpret_cast<const uint4*>(p)[i]; K.v[4 * i] = q.x; K.v[4 * i + 1] = q.y;
                K.v[4 * i + 2] = q.z; K.v[4 * i + 3] = q.w; }
        } else {
#pragma unroll
            for (int i = 0; i < VC / 2; i++) { const uint2 q = reinterpret_cast<const uint2*>(p)[i]; K.v[2 * i] = q.x; K.v[2 * i + 1] = q.y; }
        }
        return K;
    };
    auto release = [&](uint32_t c) { lds_release(&S.consumed[pi], c * (uint32_t)VC); };   // everything below chunk c has been read
    // one puncture group = GS steps; i0 = step inside the chunk, h = half of the 24-step row
    auto group = [&](const Chunk& K, int h, int i0) {
        const int k0 = i0 / GS * GBv, t24 = 12 * h + i0;
        acs_step<0, P>(V, t24, K.v[k0], K.v[k0 + 1]);                              // ACS(A,B)
        if (CR != 0) acs_step<1, P>(V, t24 + 1, K.v[k0 + 2], 0);                   // ACS(A)     2/3, 3/4 (viterbi.hpp:173-187)
        if (CR == 2) acs_step<2, P>(V, t24 + 2, 0, K.v[k0 + 3]);                   // ACS(B)     3/4
        if ((t24 + GS) % 8 == 0) normalize();
    };
    auto end_row = [&]() { V.rowpos = V.rowpos + 3 * 64 == (unsigned)P * 64 ? 0u : V.rowpos + 3 * 64; };
    auto fast_chunk = [&](const Chunk& K, int h) {
#ifndef SORA_DBG_NO_TRELLIS                                                     // experiment: the symbol side alone (operands consumed, no ACS)
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) group(K, h, g * GS);
#else
        V.U ^= K.v[0];
#endif
        tr += 12;
    };
    auto slow_chunk = [&](const Chunk& K, int h) {
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) {
            if (tr < nsteps && !(A.done && B.done)) {
                group(K, h, g * GS);
                tr += GS;
                check(12 * h + g * GS + GS - 1);
            }
        }
    };
    auto chunk = [&](const Chunk& K, int h) { if (tr + 12 <= nsteps && next_thr > tr + 12) fast_chunk(K, h); else slow_chunk(K, h); };

    uint32_t c = 0;
    Chunk cur = load_chunk(0);
    while (tr < nsteps && !(A.done && B.done)) {
        const uint32_t lim = min(nsteps, next_thr - 1);
        for (uint32_t rows = lim > tr ? (lim - tr) / 24 : 0; rows > 0; rows--) {  // rows that certainly need no look at the schedule
            Chunk nxt = load_chunk(c + 1);
            fast_chunk(cur, 0);
            cur = load_chunk(c + 2);
            fast_chunk(nxt, 1);
            c += 2;
            end_row();
            release(c);
        }
        if (!(tr < nsteps)) break;
        Chunk nxt = load_chunk(c + 1);
        chunk(cur, 0);
        if (!(tr < nsteps && !(A.done && B.done))) break;
        cur = load_chunk(c + 2);
        chunk(nxt, 1);
        c += 2;
        end_row();
        release(c);
    }
    lds_release(&S.consumed[pi], 0x40000000u);                                  // nothing more will be read: the symbol wave never waits again
}

// grid: ceil(pairs / 2) workgroups of 256 threads; waves 0, 1 = trellis waves of pairs 2b, 2b+1, waves 2, 3 = their symbol waves.
// Frames are queued per code rate (k_scan): list r holds njobs[r] frame rows, consecutive rows form a pair, the last frame of an
// odd list runs alone in the low half.
__global__ void __launch_bounds__(256, 4) k_decode(RxArgs A)
{
    __shared__ DecodeLds S;
    const int tid = threadIdx.x, w = tid >> 6, pi = w & 1;
    reinterpret_cast<uint32_t*>(S.demap)[tid] = reinterpret_cast<const uint32_t*>(A.T.demap)[tid];
    // a half nobody writes (no frame B, a frame that has ended) must read as well-formed operands
    for (int i = tid; i < 2 * kSoftRing; i += 256) (&S.ring[0][0])[i] = 0;
    if (tid < 4) S.produced[tid >> 1][tid & 1] = 0;
    if (tid < 2) S.consumed[tid] = 0;
    __syncthreads();                                                            // the only block barrier: the two pairs are independent from here on
    const uint32_t n0 = A.njobs[0], n1 = A.njobs[1], n2 = A.njobs[2];
    const uint32_t p0 = (n0 + 1) / 2, p1 = (n1 + 1) / 2, p2 = (n2 + 1) / 2;
    uint32_t pw = uni(blockIdx.x * 2u + (uint32_t)pi), list, nl;
    if (pw < p0) { list = 0; nl = n0; } else if (pw < p0 + p1) { list = 1; nl = n1; pw -= p0; } else if (pw < p0 + p1 + p2) { list = 2; nl = n2; pw -= p0 + p1; } else return;
    list = uni(list); nl = uni(nl);
    const uint32_t* jl = A.joblist + (size_t)list * A.nrows;
    const uint32_t fa = uni(jl[2 * pw]);
    const bool hasB = 2 * pw + 1 < nl;
    const uint32_t fb = hasB ? uni(jl[2 * pw + 1]) : fa;
    const FrameGeom GA = frame_geom(A, fa), GB = frame_geom(A, fb);
    if (w >= 2) { symbol_wave(A, S, pi, GA, GB, hasB); return; }
    if (list == 0)      trellis_wave<0>(S, pi, GA, GB, hasB, A.vout);
    else if (list == 1) trellis_wave<1>(S, pi, GA, GB, hasB, A.vout);
    else                trellis_wave<2>(S, pi, GA, GB, hasB, A.vout);
}

}  // namespace sora
