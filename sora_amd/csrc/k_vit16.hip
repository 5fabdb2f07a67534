// k_vit16.hip -- the K=7 trellis with SIXTEEN LANES PER FRAME PAIR: eight frames per wave (gfx950).
//
//   k_viterbi16 / k_viterbi16_11n   T11aViterbi<5000*8,48,256,24> / <..,312,192,36>: the same arithmetic, window schedule and
//                                   trace-back as k_viterbi (k_rx.hip, dev_viterbi.h), bit for bit, in a layout that needs
//                                   half the vector instructions per frame.  tools/emu_trellis16.py is the lane-level model
//                                   this file was written against (tests/test_trellis16_model.py runs it against the oracle).
//
// Why.  k_viterbi (64 lanes = the 64 states of one frame pair) spends, per frame pair and step, two operand instructions, an
// add, a cross-lane add and a packed minimum -- and every third step a v_permlane swap with its copy and wait states, because
// two of the six butterfly distances (32, 16) leave the 16-lane row a DPP move can reach.  It is bound by vector issue
// (profiles/r03_b_sq_counters_k_viterbi_pairstream.json: 200 M vector instructions, a wave executing one 62 % of its life).
//
// Layout.  The 64 states of a frame pair live in 16 lanes x 4 registers; a wave holds four pairs (one per row).  W = {0, 21,
// 42, 63} is a subgroup of (Z_2)^6 that the in-place butterfly (state -> rol6(state)) maps onto itself and whose members are
// orthogonal to every rotation of both generator polynomials (0155, 0117): the states s ^ w, w in W, have IDENTICAL branch
// metrics at every step.  A lane holds such a coset, register i <-> w_i.  The butterfly partner of s at step t is s ^ e_j,
// j = 5 - t mod 6, and with lane = c0 ^ c2 << 1 ^ c1 << 3 ^ (c3 ? 7 : 0) for the coset of c0 e0 + c1 e1 + c2 e2 + c3 e3:
//     e0, e2, e4 = e0 ^ e2 ^ 21   ->  lane ^ 1, lane ^ 2, lane ^ 3 (and register ^ 1)     quad_perm
//     e1, e3, e5 = e1 ^ e3 ^ 42   ->  lane ^ 8, lane ^ 7, lane ^ 15 (and register ^ 2)    row_ror:8, row_half_mirror, row_mirror
// Every exchange is ONE row-local DPP move folded into its add (v_add_u32_dpp) plus a register renaming: no v_permlane swaps,
// no wait states, and the step's two operands (P and K - P + mark, dev_viterbi.h) serve all four registers -- which of the two
// a register adds to its own metric is a compile-time property of (step, register) XOR a per-lane bit that is folded into the
// lane's masks exactly as in k_viterbi.  Per step and wave: 2 operand instructions + 4 x (add, add_dpp, pk_min) for EIGHT
// frames (k_viterbi: 5 to 8 for two).
//
// Operands.  The four rows decode different pairs, so the operands are per-lane values.  Lane (row, 2 j + f) fetches soft
// values j, j + 8 (, j + 16) of the chunk from frame f's packed stream (rx_types.h: three bits per value; 16-bit loads, two
// chunks ahead -- vector loads return in order, so the look-ahead is just a deeper vmcnt), shifts them into 16-bit metric
// fields and writes them to the row's operand table in LDS; four to six broadcast ds_read_b128 then hand every lane the
// chunk's operands (field A | field B << 16).  Round 3 first read ready-made operand dwords from HBM (264 MB per call
// written and read; now 50).
//
// Trace-back.  The survivor ring is indexed by rev6(state) as in k_viterbi (the next index is the low six bits of the byte
// read), one 128-byte table per row and block.  The walk is lane-parallel instead of readlane-serial: every lane walks the
// path of frame (row, lane & 1) with one LDS read per block, the eight paths of the wave side by side, and the rows' lanes
// then assemble and store the decoded bytes.  Per window ~450 instructions for eight frames (k_viterbi: ~400 for two).
//
// Cost.  Eight frames per wave means 512 waves for the 4096-frame batch of BASELINE configs[2]: alone on the chip this kernel
// is slower than k_viterbi (half the SIMDs idle), with several calls in flight it is faster; sora_rx_set_trellis selects.
#include <hip/hip_runtime.h>

#include "dev_vit16.h"

namespace sora {

namespace {

template <int CR, int WIN, int LOOK, int BITS>
__device__ __forceinline__ void forward16(Lds16<WIN, LOOK>& S, const uint8_t* __restrict__ soft, uint32_t my_soft_off, uint32_t my_nsoft,
                                          uint32_t nstepsA, uint32_t nstepsB, uint32_t my_tr_end, bool my_valid, uint8_t* my_out)
{
    using G = Geom16<WIN, LOOK>;
    constexpr int P = G::P;
    constexpr int GB = CR == 0 ? 2 : CR == 2 ? 4 : 3;                           // soft values per puncture group
    constexpr int GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;                           // trellis steps per group
    constexpr int CW = 12 / GS * GB;                                            // operands per 12-step chunk: 24 / 18 / 16
    const unsigned lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15, half = lane & 1u;
    const unsigned v0 = v_of_lane(l16);
    const uint32_t row_steps = max(nstepsA, nstepsB);
    const uint32_t nsteps = wave_max_u32(row_steps);
    // the last soft value of this lane's frame (fetches past it repeat it: well-formed operands nobody uses)
    const uint32_t my_last = max(my_nsoft, 1u) - 1u;

    auto which_of = [](int ph) { return CR == 0 ? 0 : CR == 1 ? (ph & 1) : ph % 3; };
    Vit16 V;
#pragma unroll
    for (int i = 0; i < 4; i++) V.U[i] = (v0 ^ kW[i]) == 0 ? 0u : 0x18u * kFld;   // ALL_INIT0 / ALL_INIT (viterbilut.h:22-30)
    const unsigned ring_base = (unsigned)(uintptr_t)&S.ring[0][0][0];           // (the low half of a flat LDS address is the LDS offset)
#pragma unroll
    for (int jb = 0; jb < 3; jb++)
#pragma unroll
        // (8 jb + 8) mod 6
        for (int i = 0; i < 4; i++) V.sadr[jb][i] = ring_base + ((row * 64u + rev6u(rol6(v0 ^ kW[i], jb == 0 ? 2 : jb == 1 ? 4 : 0))) << 1);
#pragma unroll
    for (int t = 0; t < 24; t++) {
        const int ph = t % 6, k = t % 8;
        const unsigned n = rol6(v0, ph + 1);                                    // register 0's state after the step (all four registers agree on the masks)
        const bool vb = (v0 >> (5 - ph)) & 1u;                                  // the lane's half of the role bit
        const unsigned ma = (__popc(n & 0155) & 1) ? 7u * kFld : 0u, mb = (__popc(n & 0117) & 1) ? 7u * kFld : 0u;
        const unsigned mx = which_of(ph) == 2 ? mb : ma;
        V.MX[t] = vb ? ((mx ^ (7u * kFld)) | (kOne << k)) : mx;
        if (t < 6) V.MY[t] = vb ? (mb ^ (7u * kFld)) : mb;
    }

    uint32_t tr = 0, ob = 0;
    uint32_t pos = 0;                                                           // ring position (block index % P) of the current row's first block
    bool my_done = !my_valid;

    auto normalize = [&]() {                                                    // Normalize (viterbicore.h:444-465): the row's minimum, both frames
        const unsigned m = row_pkmin(pk_min16(pk_min16(V.U[0], V.U[1]), pk_min16(V.U[2], V.U[3])));
#pragma unroll
        for (int i = 0; i < 4; i++) V.U[i] -= m;
    };
    auto pos_of = [&](uint32_t p, int jb) -> uint32_t { const uint32_t q = p + (uint32_t)jb; return q >= (uint32_t)P ? q - (uint32_t)P : q; };

    auto trace = [&](uint32_t my_cnt, int t24_last) {
#ifndef SORA_EXP_NORING
        trace16<WIN, LOOK>((unsigned)(uintptr_t)&S, V.U[0], V.U[1], V.U[2], V.U[3], tr, ob, pos_of(pos, t24_last / 8), (uint32_t)(t24_last % 8), my_cnt, my_out);
#endif
    };
    auto next_event = [&]() -> uint32_t {
        const uint32_t mine = my_done ? 0xFFFFFFFFu : my_tr_end;
        return min(ob + (uint32_t)(WIN + LOOK + 6), wave_min_u32(mine));
    };
    uint32_t next_thr = next_event();
    bool all_done = wave_min_u32(my_done ? 1u : 0u) != 0u;
    auto check = [&](int t24_last) {                                            // trace-back schedule (viterbi.hpp:196-214), per frame
        if (tr >= next_thr) {
            const bool partial = tr >= ob + (uint32_t)(WIN + LOOK + 6);
            uint32_t cnt = 0;
            if (!my_done) {
                if (tr >= my_tr_end) { cnt = my_tr_end - ob - 6; my_done = true; }
                else if (partial) cnt = WIN;
            }
            if (wave_max_u32(cnt) != 0u) trace(cnt, t24_last);
            if (partial) ob += WIN;
            next_thr = next_event();
            all_done = wave_min_u32(my_done ? 1u : 0u) != 0u;
        }
    };

    struct Chunk { uint32_t v[CW]; };
    // soft values a lane fetches per chunk: operands j, j + 8 (, j + 16) of its frame
    constexpr int NV = (CW + 7) / 8;
    struct Raw { SoftRaw r[NV]; };
    const uint32_t my_j = l16 >> 1;
    SoftCursor<BITS, CW> cur[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) cur[v].init(my_soft_off, my_j + 8u * v, my_last);
    auto fetch = [&](uint32_t c) -> Raw {                                       // chunk c: the loads only
        Raw R;
#pragma unroll
        for (int v = 0; v < NV; v++) R.r[v] = cur[v].fetch(soft, c);
        return R;
    };
    uint16_t* my_ops = &S.ops[row][my_j][half];
    const uint4* row_ops = reinterpret_cast<const uint4*>(&S.ops[row][0][0]);
    auto unpack = [&](const Raw& R) -> Chunk {                                  // ... their values -> the row's operand table -> every lane's registers
#pragma unroll
        // (operand j + 8 v; slots up to 23 exist, those past CW are never read)
        for (int v = 0; v < NV; v++) my_ops[16 * v] = (uint16_t)cur[v].field(R.r[v]);
        lds_fence();
        Chunk K;
#pragma unroll
        for (int i = 0; i < (CW + 3) / 4; i++) {
            const uint4 x = row_ops[i];
            K.v[4 * i] = x.x; K.v[4 * i + 1] = x.y;
            if (4 * i + 2 < CW) { K.v[4 * i + 2] = x.z; K.v[4 * i + 3] = x.w; }
        }
        lds_fence();
        return K;
    };
    unsigned pos512[3];
    auto set_row_pos = [&]() {
#pragma unroll
#ifdef SORA_EXP_NORING
        for (int jb = 0; jb < 3; jb++) pos512[jb] = 0u;
#else
        for (int jb = 0; jb < 3; jb++) pos512[jb] = pos_of(pos, jb) * 512u;
#endif
    };
    auto end_row = [&]() { pos = pos_of(pos, 3); set_row_pos(); };
    set_row_pos();
    // one puncture group = GS steps; i0 = step inside the chunk, h = half of the 24-step row
    auto group = [&](const Chunk& K, int h, int i0) {
        const int k0 = i0 / GS * GB, t24 = 12 * h + i0;
        acs16<0, P>(V, t24, K.v[k0], K.v[k0 + 1], pos512);                      // ACS(A,B)
        if (CR != 0) acs16<1, P>(V, t24 + 1, K.v[k0 + 2], 0, pos512);           // ACS(A)     2/3, 3/4 (viterbi.hpp:173-187)
        if (CR == 2) acs16<2, P>(V, t24 + 2, 0, K.v[k0 + 3], pos512);           // ACS(B)     3/4
        if ((t24 + GS) % 8 == 0) normalize();                                   // (trellis index & 7) == 0 after a group
    };
    auto fast_chunk = [&](const Chunk& K, int h) {
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) group(K, h, g * GS);
        tr += 12;
    };
    // The fast loop's operand hand-over, off the critical path (round 4).  unpack() writes a chunk's fields, reads the row's table back and uses the
    // operands at once: one LDS round trip per 12 steps sits in a lone wave's dependence chain (SQ_WAIT_ANY 21 % of its cycles,
    // profiles/r04_a_sq_counters_alone.json).  Here chunk c + 1's fields are written at the START of chunk c into the other of two tables and read
    // back in the MIDDLE of chunk c: by the time chunk c + 1 begins its operands have long been in registers.
    uint16_t* my_ops2[2] = { &S.ops2[0][row][my_j][half], &S.ops2[1][row][my_j][half] };
    const uint4* row_ops2[2] = { reinterpret_cast<const uint4*>(&S.ops2[0][row][0][0]), reinterpret_cast<const uint4*>(&S.ops2[1][row][0][0]) };
    auto stage = [&](const Raw& R, int t) {
#pragma unroll
        for (int v = 0; v < NV; v++) my_ops2[t][16 * v] = (uint16_t)cur[v].field(R.r[v]);
    };
    auto collect = [&](int t) -> Chunk {
        lds_fence();
        Chunk K;
#pragma unroll
        for (int i = 0; i < (CW + 3) / 4; i++) {
            const uint4 x = row_ops2[t][i];
            K.v[4 * i] = x.x; K.v[4 * i + 1] = x.y;
            if (4 * i + 2 < CW) { K.v[4 * i + 2] = x.z; K.v[4 * i + 3] = x.w; }
        }
        return K;
    };
    auto fast_chunk_mid = [&](const Chunk& K, int h, Chunk& Knext, int tnext) {
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) {
            if (g == (12 / GS) / 2) Knext = collect(tnext);
            group(K, h, g * GS);
        }
        tr += 12;
    };
    auto slow_chunk = [&](const Chunk& K, int h) {
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) {
            if (tr < nsteps && !all_done) {
                group(K, h, g * GS);
                tr += GS;
                check(12 * h + g * GS + GS - 1);
            }
        }
    };
    auto chunk = [&](const Chunk& K, int h) { if (tr + 12 <= nsteps && next_thr > tr + 12) fast_chunk(K, h); else slow_chunk(K, h); };

    // Vector loads return in order: chunk c + 2 is requested before chunk c is stepped through.  Four fetch buffers in fixed roles, two rows
    // per turn of the fast loop, so that no buffer is ever copied.
    uint32_t c = 0;
    Raw b0 = fetch(0), b1 = fetch(1), b2, b3;
    while (tr < nsteps && !all_done) {
        const uint32_t lim = min(nsteps, next_thr - 1);
        uint32_t rows = lim > tr ? (lim - tr) / 24 : 0;                         // rows that certainly need no look at the schedule
        if (rows >= 2) {
            // invariant at the top of a turn: Ka = chunk c's operands (in registers), b1 / b2 = the raw values of chunks c + 1 / c + 2 (requested)
            Chunk Ka, Kb;
            b2 = fetch(c + 2);
            stage(b0, 0); Ka = collect(0); lds_fence();
            for (; rows >= 2; rows -= 2) {
                b3 = fetch(c + 3); stage(b1, 1); fast_chunk_mid(Ka, 0, Kb, 1);
                b0 = fetch(c + 4); stage(b2, 0); fast_chunk_mid(Kb, 1, Ka, 0);
                end_row();
                b1 = fetch(c + 5); stage(b3, 1); fast_chunk_mid(Ka, 0, Kb, 1);
                b2 = fetch(c + 6); stage(b0, 0); fast_chunk_mid(Kb, 1, Ka, 0);
                end_row();
                c += 4;
            }
            // (the tables share their bytes with unpack()'s and the trace-back's: nothing of them is pending past here)
            lds_fence();
            // b0 holds chunk c's raw values, b1 / b2 those of c + 1 / c + 2: what the code below expects of b0, b1
        }
        if (!(tr < nsteps)) break;
        b2 = fetch(c + 2);
        if (rows) fast_chunk(unpack(b0), 0); else chunk(unpack(b0), 0);
        if (!(tr < nsteps && !all_done)) break;
        b3 = fetch(c + 3);
        if (rows) fast_chunk(unpack(b1), 1); else chunk(unpack(b1), 1);
        end_row();
        b0 = b2; b1 = b3;
        c += 2;
    }
}

// One wave per workgroup: wave w of code-rate list r decodes pairs 4w .. 4w+3 of the list (jobs 8w .. 8w+7), one pair per 16-lane row.
template <int WIN, int LOOK, int BITS>
__device__ __forceinline__ void viterbi16_body(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride,
                                               const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{
    __shared__ Lds16<WIN, LOOK> S;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    uint32_t n[3] = { njobs_single, 0, 0 };
    if (njobs3) { n[0] = njobs3[0]; n[1] = njobs3[1]; n[2] = njobs3[2]; }
    uint32_t w = uni(blockIdx.x), list = 0;
    while (list < 3 && w >= (n[list] + 7) / 8) { w -= (n[list] + 7) / 8; list++; }
    if (list >= 3) return;
    const uint32_t njobs = uni(n[list]);
    jobs += (size_t)list * stride;
    const unsigned lane = threadIdx.x & 63, row = lane >> 4, half = lane & 1u;
    const uint32_t fa = 8u * w + 2u * row, fb = fa + 1u;
    const bool hasA = fa < njobs, hasB = fb < njobs;
    const VitJob& GA = jobs[hasA ? fa : 8u * w];                                // an empty row reads the wave's first pair (its operands are never used)
    const VitJob& GBj = jobs[hasB ? fb : (hasA ? fa : 8u * w)];
    const uint32_t code_rate = uni(jobs[8u * w].code_rate);
    const uint32_t gsd = code_rate == 0 ? 2u : code_rate == 2 ? 4u : 3u, gss = code_rate == 0 ? 1u : code_rate == 2 ? 3u : 2u;
    const uint32_t nstepsA = hasA ? GA.nsoft / gsd * gss : 0u, nstepsB = hasB ? GBj.nsoft / gsd * gss : 0u;
    const VitJob& Mine = half ? GBj : GA;
    const bool my_valid = half ? hasB : hasA;
    const uint32_t my_tr_end = Mine.length * 8u + 16u + 6u;
    uint8_t* my_out = out + Mine.out_off;
    if (code_rate == 0)      forward16<0, WIN, LOOK, BITS>(S, soft, Mine.soft_off, Mine.nsoft, nstepsA, nstepsB, my_tr_end, my_valid, my_out);
    else if (code_rate == 1) forward16<1, WIN, LOOK, BITS>(S, soft, Mine.soft_off, Mine.nsoft, nstepsA, nstepsB, my_tr_end, my_valid, my_out);
    else                     forward16<2, WIN, LOOK, BITS>(S, soft, Mine.soft_off, Mine.nsoft, nstepsA, nstepsB, my_tr_end, my_valid, my_out);
}

}  // namespace

#ifdef SORA_EXP_LB
#define SORA_VIT16_BOUNDS __launch_bounds__(64, SORA_EXP_LB)
#else
#define SORA_VIT16_BOUNDS __launch_bounds__(64)
#endif
__global__ void SORA_VIT16_BOUNDS k_viterbi16(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride,
        const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{
#ifdef SORA_EXP_STAGGER                                         // experiment (round 4): waves of one launch start up to SORA_EXP_STAGGER x 16 x 1.75 us apart (a hash of the workgroup), so that the
    // two waves of a SIMD are not in the add-compare-select stretch and in the trace-back at the same time
    {
        const uint32_t d = ((blockIdx.x * 2654435761u) >> 28) * SORA_EXP_STAGGER;
        for (uint32_t i = 0; i < d; i++) __builtin_amdgcn_s_sleep(63);
    }
#endif
#ifdef SORA_EXP_VIT_PRIO
    // experiment (round 4): the issue-bound trellis waves ahead of the latency-bound front-end waves on their SIMD
    __builtin_amdgcn_s_setprio(SORA_EXP_VIT_PRIO);
#endif
    viterbi16_body<256, 24, 3>(jobs, njobs3, njobs_single, stride, soft, out);
}
__global__ void __launch_bounds__(64) k_viterbi16_11n(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single,
        uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{ viterbi16_body<192, 36, 8>(jobs, njobs3, njobs_single, stride, soft, out); }

}  // namespace sora
