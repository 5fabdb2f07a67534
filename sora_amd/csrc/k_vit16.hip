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
#include "kernels.h"
#include "dev_viterbi.h"

namespace sora {

namespace {

template <int WIN, int LOOK> struct Geom16 {
    static constexpr int kMaxWalk = (WIN + LOOK + 7) / 8 + 2;                   // 37 / 31 blocks a window's walk can touch
    static constexpr int P = kMaxWalk;                                          // ring period: the walk runs while nothing is being banked
    static constexpr int kPathBytes = 40;                                       // per frame: walk positions 0 .. kMaxWalk - 1
};

template <int WIN, int LOOK> struct Lds16 {
#ifdef SORA_EXP_NORING                                                           // experiment (tools/r04_exp_noring.sh): no survivor ring, no trace-back -- results are wrong, only the duration means something
    uint16_t ring[1][4][64];
#else
    uint16_t ring[Geom16<WIN, LOOK>::P][4][64];                                 // [block % P][row][rev6(state)] {frame A's byte, frame B's byte}: 18944 / 15872 B
#endif
    union {
        uint32_t udump[4][64];                                                  // the metrics registers at a trace-back (the start state's unfinished block)
        uint16_t ops[4][24][2];                                                 // [row][operand of the chunk][frame]: the soft values as metric fields -- live only inside
        uint16_t ops2[2][4][24][2];                                             //   forward16's unpack() / the fast loop's two alternating tables, never across a trace-back:
    };                                                                          //   they share their bytes with the trace-back's register dump
    uint8_t  path[8][Geom16<WIN, LOOK>::kPathBytes];                             // [row * 2 + frame][walk position]: the bytes along the traced path
};                                                                              // 20288 / 17216 bytes: eight one-wave workgroups per CU (20480 each)

constexpr unsigned kW[4] = { 0u, 21u, 42u, 63u };

__device__ __forceinline__ unsigned v_of_lane(unsigned l)                       // coset representative (bits e0..e3) held by lane l of a row
{
    const unsigned b0 = l & 1u, b1 = (l >> 1) & 1u, b2 = (l >> 2) & 1u, b3 = (l >> 3) & 1u;
    const unsigned c3 = b2, c1 = b3, c0 = b0 ^ c3, c2 = b1 ^ c3;
    return c0 | (c1 << 1) | (c2 << 2) | (c3 << 3);
}
__device__ __forceinline__ unsigned lane_of_v(unsigned v)                       // inverse: lane of the coset with representative v (4 bits)
{
    const unsigned c0 = v & 1u, c1 = (v >> 1) & 1u, c2 = (v >> 2) & 1u, c3 = (v >> 3) & 1u;
    return c0 ^ (c2 << 1) ^ (c1 << 3) ^ (c3 ? 7u : 0u);
}
__device__ __forceinline__ unsigned rev6u(unsigned x) { return __brev(x) >> 26; }

template <int CTRL> __device__ __forceinline__ unsigned dppx(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

// the partner's metric for phase ph = t mod 6: lane ^ {15, 3, 7, 2, 8, 1}
__device__ __forceinline__ unsigned partner(unsigned v, int ph)
{
    switch (ph) {
    case 0: return dppx<0x140>(v);                                              // row_mirror:      lane ^ 15   (e5; register ^ 2)
    case 1: return dppx<0x1B>(v);                                               // quad_perm [3,2,1,0]: ^ 3     (e4; register ^ 1)
    case 2: return dppx<0x141>(v);                                              // row_half_mirror: lane ^ 7    (e3)
    case 3: return dppx<0x4E>(v);                                               // quad_perm [2,3,0,1]: ^ 2     (e2)
    case 4: return dppx<0x128>(v);                                              // row_ror:8:       lane ^ 8    (e1)
    default: return dppx<0xB1>(v);                                              // quad_perm [1,0,3,2]: ^ 1     (e0)
    }
}

struct Vit16 {
    unsigned U[4];           // register i: the metrics of state (coset of the lane) ^ kW[i]; (field B << 16) | field A as in dev_viterbi.h
    unsigned MX[24];         // mask of the mark-carrying operand per t mod 24 (for lanes whose coset bit j is set: complemented, with the mark)
    unsigned MY[6];          // mask of the second operand of a two-input step per t mod 6
    unsigned sadr[3][4];     // LDS byte address (without the block's position) of the ring entry of register i at the end of block jb of a row
};

// WHICH 0: (A,B) two soft values, 1: A only, 2: B only.  t24 = step index mod 24 (a constant after unrolling).  pos512 = the ring
// position of the row's first block, in bytes (wave-uniform).
template <int WHICH, int P>
__device__ __forceinline__ void acs16(Vit16& V, int t24, unsigned a, unsigned b, unsigned pos512[3])
{
    const int ph = t24 % 6, k = t24 % 8;
    const unsigned Kp = (WHICH == 0 ? 14u : 7u) * kFld + (kOne << k);           // K + mark
    unsigned bm;
    if (WHICH == 0)      bm = (a ^ V.MX[t24]) + (b ^ V.MY[ph]);
    else if (WHICH == 1) bm = a ^ V.MX[t24];
    else                 bm = b ^ V.MX[t24];
    const unsigned bo = Kp - bm;
    const int rx = ph == 0 ? 2 : ph == 1 ? 1 : 0;
    unsigned N[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned X = V.U[i], Y = partner(V.U[i ^ rx], ph);
        const bool wb = (kW[i] >> (5 - ph)) & 1u;                               // the register's half of the role bit (the lane's half is in the masks)
        N[i] = wb ? pk_min16(X + bo, Y + bm) : pk_min16(X + bm, Y + bo);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) V.U[i] = N[i];
    if (k == 7) {                                                               // end of an 8-step block: bank the path histories, clear the marks
        const int jb = t24 / 8;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned w = bank_word(V.U[i]);
            const unsigned addr = V.sadr[jb][i] + pos512[jb];
            asm volatile("ds_write_b16 %0, %1" : : "v"(addr), "v"(w) : "memory");
            V.U[i] &= 0xFE00FE00u;
        }
    }
}

__device__ __forceinline__ unsigned row_min_u32(unsigned v)                     // minimum over the 16 lanes of the row, in every lane
{
    v = min(v, dppx<0xB1>(v)); v = min(v, dppx<0x4E>(v)); v = min(v, dppx<0x141>(v)); v = min(v, dppx<0x128>(v));
    return v;
}
__device__ __forceinline__ unsigned row_pkmin(unsigned v)
{
    v = pk_min16(v, dppx<0xB1>(v)); v = pk_min16(v, dppx<0x4E>(v)); v = pk_min16(v, dppx<0x141>(v)); v = pk_min16(v, dppx<0x128>(v));
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)dpp_min_u32_wave(v)); }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return ~wave_min_u32(~v); }
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// Trace-back of one window for every frame of the wave whose count is non-zero (my_cnt: this lane's frame = (row, lane & 1)); kept out of
// line (it is reached from every puncture group of the slow path), so everything arrives by value and the LDS block by its offset.
// pj = ring position of block j = (tr - 1) >> 3; k = index in its 8-step block of the last step taken.
template <int WIN, int LOOK>
__device__ __noinline__ void trace16(unsigned lds_off, unsigned U0, unsigned U1, unsigned U2, unsigned U3, uint32_t tr_, uint32_t ob_, uint32_t pj_, uint32_t k_,
                                     uint32_t my_cnt, uint8_t* my_out)
{
    using G = Geom16<WIN, LOOK>;
    constexpr int P = G::P;
    typedef __attribute__((address_space(3))) Lds16<WIN, LOOK> lds_t;
    lds_t& S = *(lds_t*)(uintptr_t)lds_off;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t tr = uni(tr_), ob = uni(ob_), pj = uni(pj_), k = uni(k_);
    const unsigned lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15, half = lane & 1u;
    const unsigned v0 = v_of_lane(l16);
    const unsigned U[4] = { U0, U1, U2, U3 };
    const uint32_t j = (tr - 1) >> 3, nn = tr - 8u * j, m_lo = ob >> 3;
    // the metrics registers -> LDS (the start state's unfinished block is read from there); the ring writes of this block are visible after the fence
#pragma unroll
    for (int i = 0; i < 4; i++) S.udump[i][lane] = U[i];
    lds_fence();
    // arg-min with the reference's tie-break metric << 8 | state << 2 (viterbicore.h:479-524), metric = 2u + last decision
    unsigned key[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned st = rol6(v0 ^ kW[i], tr);
        unsigned lastA, lastB;
        if (k == 7) { const unsigned w = S.ring[pj][row][rev6u(st)]; lastA = (w >> 7) & 1u; lastB = (w >> 15) & 1u; }
        else { lastA = (U[i] >> k) & 1u; lastB = (U[i] >> (17 + k)) & 1u; }
        const unsigned mA = ((U[i] & 0xFFFFu) >> 9 << 1) | lastA, mB = (U[i] >> 25 << 1) | lastB;
        key[0] = min(key[0], (mA << 8) | (st << 2)); key[1] = min(key[1], (mB << 8) | (st << 2));
    }
    const unsigned kA = row_min_u32(key[0]), kB = row_min_u32(key[1]);
    const unsigned st = ((half ? kB : kA) >> 2) & 0x3Fu;                        // this lane's frame's start state
    // the slot holding st now: state0 = ror6(st, tr), register from bits 4 / 5, lane from the coset representative
    const unsigned s0 = rol6(st, 6u - tr % 6u);
    const unsigned b4 = (s0 >> 4) & 1u, b5 = (s0 >> 5) & 1u;
    const unsigned ri = b4 | (b5 << 1);
    const unsigned sl = lane_of_v((s0 ^ (b4 ? 21u : 0u) ^ (b5 ? 42u : 0u)) & 15u);
    const unsigned Ust = S.udump[ri][row * 16u + sl];
    unsigned H;
    if (nn == 8) H = ((unsigned)S.ring[pj][row][rev6u(st)] >> (8u * half)) & 0xFFu;
    else H = ((Ust >> (17u * half)) & 0xFFu) & ((1u << nn) - 1u);
    unsigned q = rev6u(((st >> nn) | rev6u(H & 0x3Fu)) & 0x3Fu);                 // ring index at column 8j
    __attribute__((address_space(3))) uint8_t* pth = S.path[row * 2u + half];
    pth[0] = (uint8_t)H;
    // The walk, unrolled in full: the ring position of step i is pj - i (mod P), a scalar -- its row's byte offset is one v_lshl_add off the dependence chain --
    // so that a block costs four vector instructions (shift, field, two address adds) and the chain ds_read -> bfe -> lshl_add -> ds_read (round 4: it was seven,
    // with the position counted down in a vector register).
    typedef __attribute__((address_space(3))) const uint16_t lds_u16;
    const unsigned rowbase = (unsigned)(uintptr_t)&S.ring[0][row][0], sh = 8u * half;
#pragma unroll
    for (int i = 1; i < G::kMaxWalk; i++) {                                     // always the full length: blocks below the window are read and never used
        const int d = (int)pj - i;
        const uint32_t p = (uint32_t)(d < 0 ? d + P : d);
        unsigned base = rowbase + p * 512u;
        asm volatile("" : "+v"(base));                                          // (one register: the chain's add is then v_lshl_add, not a three-input add behind a shift)
        const unsigned raw = *(lds_u16*)(uintptr_t)(base + (q << 1));
        pth[i] = (uint8_t)(raw >> sh);                                          // (this frame's byte of the pair)
        q = __builtin_amdgcn_ubfe(raw, sh, 6u);
    }
    lds_fence();
    // decoded byte m = (block m >> 6) | (block m + 1 & 0x3F) << 2; block m sits at walk position j - m.  Lane (l16 >> 1) of the row's
    // eight lanes with this `half` takes bytes m_lo + (l16 >> 1) + 8 z.
    const uint32_t nbytes = my_cnt >> 3;
    for (uint32_t z = l16 >> 1; z < nbytes; z += 8) {
        const uint32_t m = m_lo + z, i1 = j - m;                                // >= 1: the window ends at least one block below the start column
        my_out[m] = (uint8_t)(((unsigned)pth[i1] >> 6) | (((unsigned)pth[i1 - 1] & 0x3Fu) << 2));
    }
    lds_fence();
}

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
    const uint32_t my_last = max(my_nsoft, 1u) - 1u;                            // the last soft value of this lane's frame (fetches past it repeat it: well-formed operands nobody uses)

    auto which_of = [](int ph) { return CR == 0 ? 0 : CR == 1 ? (ph & 1) : ph % 3; };
    Vit16 V;
#pragma unroll
    for (int i = 0; i < 4; i++) V.U[i] = (v0 ^ kW[i]) == 0 ? 0u : 0x18u * kFld;   // ALL_INIT0 / ALL_INIT (viterbilut.h:22-30)
    const unsigned ring_base = (unsigned)(uintptr_t)&S.ring[0][0][0];           // (the low half of a flat LDS address is the LDS offset)
#pragma unroll
    for (int jb = 0; jb < 3; jb++)
#pragma unroll
        for (int i = 0; i < 4; i++) V.sadr[jb][i] = ring_base + ((row * 64u + rev6u(rol6(v0 ^ kW[i], jb == 0 ? 2 : jb == 1 ? 4 : 0))) << 1);   // (8 jb + 8) mod 6
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
    constexpr int NV = (CW + 7) / 8;                                            // soft values a lane fetches per chunk: operands j, j + 8 (, j + 16) of its frame
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
        for (int v = 0; v < NV; v++) my_ops[16 * v] = (uint16_t)cur[v].field(R.r[v]);      // (operand j + 8 v; slots up to 23 exist, those past CW are never read)
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
    auto group = [&](const Chunk& K, int h, int i0) {                           // one puncture group = GS steps; i0 = step inside the chunk, h = half of the 24-step row
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
            lds_fence();                                                        // (the tables share their bytes with unpack()'s and the trace-back's: nothing of them is pending past here)
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
__global__ void SORA_VIT16_BOUNDS k_viterbi16(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{
#ifdef SORA_EXP_STAGGER                                         // experiment (round 4): waves of one launch start up to SORA_EXP_STAGGER x 16 x 1.75 us apart (a hash of the workgroup), so that the
    {                                                           // two waves of a SIMD are not in the add-compare-select stretch and in the trace-back at the same time
        const uint32_t d = ((blockIdx.x * 2654435761u) >> 28) * SORA_EXP_STAGGER;
        for (uint32_t i = 0; i < d; i++) __builtin_amdgcn_s_sleep(63);
    }
#endif
#ifdef SORA_EXP_VIT_PRIO
    __builtin_amdgcn_s_setprio(SORA_EXP_VIT_PRIO);              // experiment (round 4): the issue-bound trellis waves ahead of the latency-bound front-end waves on their SIMD
#endif
    viterbi16_body<256, 24, 3>(jobs, njobs3, njobs_single, stride, soft, out);
}
__global__ void __launch_bounds__(64) k_viterbi16_11n(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ njobs3, uint32_t njobs_single, uint32_t stride, const uint8_t* __restrict__ soft, uint8_t* __restrict__ out)
{ viterbi16_body<192, 36, 8>(jobs, njobs3, njobs_single, stride, soft, out); }

}  // namespace sora
