// dev_vit16.h -- the pieces of the 16-lanes-per-frame-pair trellis layout shared by k_viterbi16 (k_vit16.hip: one serial chain per frame) and
// k_viterbi16w (k_vitwin.hip: the window-parallel form, round 5): the LDS layout, the coset <-> lane maps, the add-compare-select step and the
// lane-parallel trace-back of one window.  The layout itself is described at the top of k_vit16.hip.
#pragma once
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
// SORA_EXP_NORING: experiment (tools/r04_exp_noring.sh): no survivor ring, no trace-back -- results are wrong, only the duration means something
#ifdef SORA_EXP_NORING
    uint16_t ring[1][4][64];
#else
    // [block % P][row][rev6(state)] {frame A's byte, frame B's byte}: 18944 / 15872 B
    uint16_t ring[Geom16<WIN, LOOK>::P][4][64];
#endif
    union {
        uint32_t udump[4][64];                                                  // the metrics registers at a trace-back (the start state's unfinished block)
        // [row][operand of the chunk][frame]: the soft values as metric fields -- live only inside
        uint16_t ops[4][24][2];
        //   forward16's unpack() / the fast loop's two alternating tables, never across a trace-back:
        uint16_t ops2[2][4][24][2];
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
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// Trace-back of one window for every frame of the wave whose count is non-zero (my_cnt: this lane's frame = (row, lane & 1)); kept out of
// line (it is reached from every puncture group of the slow path), so everything arrives by value and the LDS block by its offset.
// pj = ring position of block j = (tr - 1) >> 3; k = index in its 8-step block of the last step taken.
// LANE_OB (k_vitwin.hip): ob_ is a per-lane value -- the units a wave decodes hand out their bits at different positions of their frames.
template <int WIN, int LOOK, bool LANE_OB = false>
__device__ __noinline__ void trace16(unsigned lds_off, unsigned U0, unsigned U1, unsigned U2, unsigned U3, uint32_t tr_, uint32_t ob_, uint32_t pj_, uint32_t k_,
                                     uint32_t my_cnt, uint8_t* my_out)
{
    using G = Geom16<WIN, LOOK>;
    constexpr int P = G::P;
    typedef __attribute__((address_space(3))) Lds16<WIN, LOOK> lds_t;
    lds_t& S = *(lds_t*)(uintptr_t)lds_off;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t tr = uni(tr_), ob = LANE_OB ? ob_ : uni(ob_), pj = uni(pj_), k = uni(k_);
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
        // (one register: the chain's add is then v_lshl_add, not a three-input add behind a shift)
        asm volatile("" : "+v"(base));
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

}  // namespace
}  // namespace sora
