// dev_viterbi.h -- the 64-state trellis machinery shared by k_viterbi (k_rx.hip: soft values from the frames' packed streams in HBM, handed
// round through an operand table in LDS), k_viterbi16 (k_vit16.hip: the same in the 16-lanes-per-pair layout) and k_decode (k_decode.hip:
// operands out of an LDS ring filled by the symbol waves of the same workgroup).
#pragma once
#include <hip/hip_runtime.h>
#include "rx_types.h"

namespace sora {

// ------------------------------------------------------------------------------------------------
// k_viterbi: forward add-compare-select of the K=7 (133,171) code exactly as TViterbiCore does it
// (viterbicore.h:293-465): 8-bit WRAPPING path metrics with the decision in the metric LSB (&0xFE / |1),
// unsigned minimum, normalisation whenever (trellis_index & 7) == 0 after a puncture group, and the window
// schedule of T11aViterbi<..,256,24>::Process (viterbi.hpp:189-214).
//
// Arithmetic.  A reference metric byte is m = 2u + d (d = decision mark).  Branch metrics are even, so
//   c0 = (x0 + bm0) & 0xFE = 2((u0 + b0) mod 128),  c1 = ((x1 + bm1) & 0xFE) | 1 = 2((u1 + b1) mod 128) + 1,  b = bm/2
//   min(c0, c1) picks c1 iff (u1 + b1) mod 128 < (u0 + b0) mod 128  (a tie keeps c0), and the new u is that minimum.
// The kernel carries u in the top 7 bits of a 16-bit field (u << 9): the 7-bit wrap is the natural 16-bit wrap and the
// unsigned minimum needs no masking.  Normalisation subtracts min(u) (= (min m & 0xFE)/2).
// Branch metric of soft value v (0..7) for expected bit c: b = v ^ (c ? 7 : 0) (VIT_MA/VIT_MB, viterbilut.h:50-185,
// halved); expected bits = parity((branch<<6 | n) & 0155) for A, & 0117 for B, n = new state.  Both generators
// tap the oldest bit, so the decision-1 branch costs K - b0, K = 14 (7 on a punctured step).
//
// Decisions without a compare.  The reference marks the decision in the metric LSB; here the nine spare low bits of
// the field do the same job: step k of an 8-step block adds 1 << k (frame A; 1 << (k + 1) in
// frame B's half, whose bit 0 is a carry guard) to the decision-1 candidate.  That bit breaks
// ties exactly like the reference's LSB (a tie keeps branch 0), marks of earlier steps sit BELOW it and can never
// decide a comparison, and after the minimum it IS the decision.  Because the marks travel with the metric through
// the butterfly, after 8 steps the low byte of a lane is the decision history of the SURVIVOR PATH into the state
// the lane holds (register exchange, 8 deep, for free).  Every 8 steps the byte is moved to a history register
// (one v_perm_b32 per frame) and cleared; every 24 steps the 64 lanes store 3 such bytes as one coalesced 256-byte
// row.  The trace-back then walks 8 columns per lookup: the 6 oldest decisions of a block are the state 8 columns
// earlier, the 8 decisions are the decoded bits.
//
// CDNA4 mapping.  wave64 = the 64 states, run as an in-place butterfly {p, p+32} -> {2p, 2p+1}: after t steps
// lane L holds state rol6^t(L), and the two predecessors of its next state sit in lanes L and P = L ^ (32 >> (t mod 6)).
//     t mod 6 = 0,1 : v_permlane32_swap / v_permlane16_swap (gfx950) return (metric of the decision-0 predecessor,
//                     metric of the decision-1 predecessor) directly
//     t mod 6 = 2..5: the lane adds its OWN metric and the partner's (one DPP move: row_ror:8, quad_perm; two for ^4);
//                     which of the two is the decision-1 candidate depends on the lane, so the lane's soft masks are
//                     complemented (K - b = b ^ 7) and carry the mark for the lanes whose own metric is candidate 1.
// Two frames per wave: frame A in the low 16-bit half of every register, frame B in the high half (v_pk_add_u16,
// v_pk_min_u16; the cross-lane moves carry both).  Measured issue cost on gfx950 at 2 waves/SIMD (tools/gen_probe_issue.py):
// VOP2 add/sub/xor/and/mov 5.0, VOP3/VOP3P/DPP 9.4, compare->SGPR / v_addc 9.9, permlane swap 16.9 (units of 0.67 ns).
// Per packed step: 1 move + 2 adds + 1 min + 2..3 for the branch metrics; no compare, no carry chain, no LDS.
__device__ __forceinline__ unsigned rol6(unsigned v, unsigned r) { r %= 6; return ((v << r) | (v >> (6 - r))) & 63u; }
// Physical lane <-> label lane.  The butterfly partner of label lane v at phase ph is v ^ (32 >> ph); every XOR distance
// except 4 is one cross-lane move (permlane swaps for 32/16, row_ror:8, quad_perm for 2/1).  Placing label lane v in
// physical lane v ^ (v & 4 ? 3 : 0) turns the label distance 4 into the physical distance 7 = row_half_mirror, one DPP
// move too; the other distances are unchanged.  The map is its own inverse.
__device__ __forceinline__ unsigned lane_map(unsigned x) { return x ^ ((x & 4u) ? 3u : 0u); }

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

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (u16x2_t)(__builtin_bit_cast(u16x2_t,
        a) + __builtin_bit_cast(u16x2_t, b))); }
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (u16x2_t)(__builtin_bit_cast(u16x2_t,
        a) - __builtin_bit_cast(u16x2_t, b))); }
__device__ __forceinline__ unsigned pk_min16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned,
        __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b))); }

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

// ------------------------------------------------------------------------------------------------
// The soft stream (rx_types.h).  Producers pack eight values (three bits each, value j in bits 3 j ..) into three bytes; a trellis lane
// fetches the 16 bits that contain value i of its frame -- at any byte address: gfx950 serves unaligned 16-bit loads
// (tools/calib/unaligned_probe.hip) -- and shifts.
__device__ __forceinline__ uint32_t soft3_pack8(const uint32_t v[8])
{
    return v[0] | (v[1] << 3) | (v[2] << 6) | (v[3] << 9) | (v[4] << 12) | (v[5] << 15) | (v[6] << 18) | (v[7] << 21);
}
__device__ __forceinline__ void soft3_store8(uint8_t* stream, uint32_t group, uint32_t bits24)     // values 8 group .. 8 group + 7
{
    uint8_t* p = stream + 3u * group;
    p[0] = (uint8_t)bits24; p[1] = (uint8_t)(bits24 >> 8); p[2] = (uint8_t)(bits24 >> 16);
}
struct SoftRaw { uint32_t w, sh; };                        // a fetched value before the shift
// One lane's view of one frame's stream: value k of every CW-value chunk.  BITS is a property of the kernel (3 for the 802.11a graph's
// producers, 8 for the 802.11n / 40 MHz ones).  When a chunk is a whole number of bytes (every case but three-bit values at rate 2/3) the
// value's bit offset inside its byte never changes: a fetch is an add, a clamp and a load, the field a shift and a mask.  Past the frame's
// end the fetch address stays on the frame's last value (some well-formed value: nobody uses it).
template <int BITS, int CW> struct SoftCursor {
    static constexpr int ADV = BITS * CW;
    static constexpr bool kConst = ADV % 8 == 0;
    uint32_t off;            // kConst: byte offset (from the soft base) of the value in chunk 0; else: of the stream
    uint32_t pos;            // else: bit offset of the value in chunk 0 inside the stream
    uint32_t lim;            // kConst: byte offset of the frame's last value; else: its bit offset inside the stream
    uint32_t lsh;            // kConst: 9 - (bit offset & 7): the shift that puts the value at bits 9..11
    __device__ __forceinline__ void init(uint32_t stream_off, uint32_t k, uint32_t last)
    {
        const uint32_t t0 = (uint32_t)BITS * min(k, last), tl = (uint32_t)BITS * last;
        if (kConst) { off = stream_off + (t0 >> 3); pos = 0; lim = stream_off + (tl >> 3); lsh = 9u - (t0 & 7u); }
        else { off = stream_off; pos = t0; lim = tl; lsh = 0; }
    }
    __device__ __forceinline__ SoftRaw fetch(const uint8_t* __restrict__ base, uint32_t c) const
    {
        SoftRaw r;
        if (kConst) { r.sh = 0; r.w = *reinterpret_cast<const uint16_t*>(base + min(off + c * (uint32_t)(ADV / 8), lim)); }
        else { const uint32_t t = min(pos + c * (uint32_t)ADV, lim); r.sh = t & 7u; r.w = *reinterpret_cast<const uint16_t*>(base + (off + (t >> 3))); }
        return r;
    }
    __device__ __forceinline__ uint32_t field(const SoftRaw& r) const                                   // the value as a 16-bit metric field (u << 9)
    {
        return kConst ? (r.w << lsh) & 0x0E00u : ((r.w >> r.sh) & 7u) << 9;
    }
};

constexpr unsigned kFld = (1u << 9) | (1u << 25);        // one unit of u in both halves
// mark bit 0 of both frames: bit 0 (frame A), bit 17 (frame B; bit 16 is the carry guard, see acs_step)
constexpr unsigned kOne = 0x00020001u;
constexpr unsigned kGuard = 1u << 16;

// Survivor history in LDS, per wave: 8-step blocks of 64 16-bit entries {frame A's byte, frame B's byte}.  A window's walk touches
// kMaxWalk = (WIN + LOOK + 7) / 8 + 2 consecutive blocks ending at the newest one.  The ring has a PERIOD of P blocks and 2 P positions:
// block b is written twice, at b % P and at b % P + P, so the walk's blocks top, top - 1, ... (top = b % P + P) are always contiguous
// and the trace-back reads them with ONE address register and compile-time offsets (the single-copy ring needed a borrow-select and an
// address computation per block).  P is a multiple of 3: a 24-step row is three blocks, rows start at multiples of three blocks, so a
// row never straddles the wrap and the position is advanced once per row.
template <int WIN, int LOOK> struct RingGeom {
    static constexpr int kMaxWalk = (WIN + LOOK + 7) / 8 + 2;                   // 37 for 256 / 24 (802.11a graph), 31 for 192 / 36 (802.11n graph)
    static constexpr int P = (kMaxWalk + 2) / 3 * 3;                            // 39 / 33
    static constexpr int kEntries = 2 * P * 64;                                 // 16-bit entries per wave: 9984 B / 8448 B
};

// The two frames' decision bytes of a finished block as one 16-bit word: frame A's marks are byte 0 of the register, frame B's (bits 17..24,
// above the guard bit) byte 2 of the register shifted right by one: a shift and a byte permute.
__device__ __forceinline__ unsigned bank_word(unsigned U) { return __builtin_amdgcn_perm(U >> 1, U, 0x0C0C0600u); }

struct VitLane {
    unsigned U;              // (field B << 16) | field A; field = u << 9 | marks of the current 8-step block
    unsigned MX[24];         // soft mask (+ mark, + complement for own-is-candidate-1 lanes) of the mark-carrying operand, per t mod 24
    unsigned MY[6];          // soft mask of the second operand of a two-input step, per t mod 6
    uint16_t* ring;          // LDS: [2 P][64] {frame A's block, frame B's block} (one byte each) at the block's end, indexed by rev6(state)
    unsigned rowpos;         // ring position (block index % P) of the first block of the current 24-step row, times 64   (wave-uniform)
    unsigned sidx[3];        // ring index of the state this lane holds at the end of block j of a row: rev6(rol6^(8j+8)(lane))
};

// WHICH 0: (A,B) two soft values, 1: A only, 2: B only.  t24 = trellis step index mod 24 (a constant after unrolling).
template <int WHICH, int P>
__device__ __forceinline__ void acs_step(VitLane& V, int t24, unsigned a, unsigned b)
{
    const int ph = t24 % 6, k = t24 % 8;
    const unsigned Kp = (WHICH == 0 ? 14u : 7u) * kFld + (kOne << k);           // K + mark
    unsigned bm;                                                                // cost added to X (per field: b, or b + mark)
    if (WHICH == 0)      bm = (a ^ V.MX[t24]) + (b ^ V.MY[ph]);                  // per field <= 14 << 9 | mark: no carry between the halves
    else if (WHICH == 1) bm = a ^ V.MX[t24];
    else                 bm = b ^ V.MX[t24];
    const unsigned bo = Kp - bm;                                                // cost added to Y
    unsigned X, Y;
    const int u = (int)V.U;
    switch (ph) {
    case 0: { auto r = __builtin_amdgcn_permlane32_swap(V.U, V.U, false, false); X = r[0]; Y = r[1]; break; }
    case 1: { auto r = __builtin_amdgcn_permlane16_swap(V.U, V.U, false, false); X = r[0]; Y = r[1]; break; }
    case 2: X = V.U; Y = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0x128, 0xF, 0xF, true); break;                  // L ^ 8: row_ror:8
    case 3: X = V.U; Y = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0x141, 0xF, 0xF, true); break;                  // label ^ 4 = lane ^ 7: row_half_mirror
    case 4: X = V.U; Y = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0x4E, 0xF, 0xF, true); break;                   // L ^ 2: quad_perm [2,3,0,1]
    default: X = V.U; Y = (unsigned)__builtin_amdgcn_update_dpp(0, u, 0xB1, 0xF, 0xF, true); break;                  // L ^ 1: quad_perm [1,0,3,2]
    }
    // The two sums are plain 32-bit adds (VOP2, and the cross-lane move of a DPP phase folds into its add: v_add_u32_dpp): the wrap of frame
    // A's 7-bit metric carries into bit 16, which belongs to nobody -- frame B's marks start at bit 17.  That guard bit is the lowest bit of the
    // high half, below the mark of the current step, which always differs between the two candidates: it can never decide the minimum.  It is
    // cleared with the marks at the end of every 8-step block, and that is often enough: a path's metric grows by at most 14 per step, 112 per
    // block, so it passes a multiple of 128 at most once per block and the bit never has to absorb a second carry.  (v_pk_add_u16 is
    // VOP3P-encoded and issues at half the rate of a VOP2, profiles/r01_issue_probe_table.txt; the constants stay 32-bit literals: the same
    // instructions with the constants pinned into SGPRs measured 1.5-10 % slower.)
    V.U = pk_min16(X + bm, Y + bo);
    if (k == 7) {                                                               // end of an 8-step block: bank the path histories, clear the marks
        uint16_t* e = V.ring + V.rowpos + (t24 / 8) * 64 + V.sidx[t24 / 8];
        const uint16_t w = (uint16_t)bank_word(V.U);                            // frame A's block, frame B's block
        e[0] = w; e[P * 64] = w;                                                // both copies (same address register, two immediates)
        V.U &= 0xFE00FE00u;
    }
}

// Trace-back of one window per frame (cnt = 0: none), called from the forward loop whenever the schedule fires (once per
// 256 columns): start at the arg-min state with the reference's tie-break metric<<8 | state<<2 (viterbicore.h:479-524),
// metric = 2u + last decision; walk back look + cnt columns, write cnt / 8 decoded bytes at bit `ob` of the frame.
// The ring is indexed by q = rev6(state), so the index of the next (earlier) block is simply the low 6 bits of the
// block just read.  All blocks the walk can touch (<= 38) are first fetched into registers, lane = ring index, with
// independent LDS reads; the walk itself is then v_readlane + two scalar ops per block and frame, no memory latency.
// Kept out of line: it is reached from every puncture group of the slow path.
template <int kMaxWalk>                                                         // blocks a window's walk can touch: (WIN + LOOK + 7) / 8 + 2
__device__ __noinline__ void viterbi_trace(unsigned U, const uint16_t* ring, uint32_t tr_, uint32_t ob_, unsigned mA, unsigned mB,
                                           uint32_t cntA_, uint32_t cntB_, uint8_t* outA, uint8_t* outB, uint32_t top_)
{                                                                               // top: upper-copy ring position (P .. 2 P - 1) of block j = (tr - 1) >> 3
    const unsigned lane = threadIdx.x & 63;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };   // arguments arrive in VGPRs; these are wave-uniform
    const uint32_t tr = uni(tr_), ob = uni(ob_), cntA = uni(cntA_), cntB = uni(cntB_), top = uni(top_);
    auto rev6 = [](unsigned x) { return __brev(x) >> 26; };
    // vec[lane ln] = val (one SGPR operand per VALU op: the lane select goes through M0)
    auto writelane = [](unsigned& vec, unsigned val, unsigned ln) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(val), "s"(ln) : "m0");
    };
    const unsigned lbl = rol6(lane_map(lane), tr) << 2;
    const unsigned kA = (unsigned)__builtin_amdgcn_readfirstlane((int)dpp_min_u32_wave((mA << 8) | lbl));
    const unsigned kB = (unsigned)__builtin_amdgcn_readfirstlane((int)dpp_min_u32_wave((mB << 8) | lbl));
    const unsigned stA = (kA >> 2) & 0x3F, stB = (kB >> 2) & 0x3F;
    const unsigned back = 6u - tr % 6u;                                         // label lane holding state s now: rol6(s, 6 - tr mod 6)
    const unsigned pA = (unsigned)__builtin_amdgcn_readlane((int)U, (int)lane_map(rol6(stA, back))) & 0xFFu;  // decisions of the unfinished block
    const unsigned pB = ((unsigned)__builtin_amdgcn_readlane((int)U, (int)lane_map(rol6(stB, back))) >> 17) & 0xFFu;  // along the start state's path
    const int m_lo = (int)(ob >> 3);                                            // first output byte of the window
    const int j = (int)((tr - 1) >> 3);                                         // block holding the start column
    const unsigned n = tr - 8u * (unsigned)j;                                   // its decisions known now: 1..8
    const int nblk = j - m_lo;                                                  // blocks below j on the walk
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // W[i] = block j - i, all 64 ring entries (one per lane): ring positions top - i, contiguous thanks to the second copy of every block
    // (RingGeom), so ONE address register and compile-time offsets.  (Read as LDS: the low half of a flat LDS address is the LDS offset.)
    const unsigned low = (unsigned)(uintptr_t)ring + ((((top - (unsigned)(kMaxWalk - 1)) << 6) | lane) << 1);
    uint32_t W[kMaxWalk];
#pragma unroll
    for (int i = 0; i < kMaxWalk; i++)
        // (zero-extends: no masking afterwards)
        asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(W[i]) : "v"(low), "n"((kMaxWalk - 1 - i) * 128) : "memory");
    // (W[1..] are only used by the assembler blocks below, which stay behind this one)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W[0]) : : "memory");
    unsigned HA, HB;
    if (n == 8) {
        HA = (unsigned)__builtin_amdgcn_readlane((int)W[0], (int)rev6(stA)) & 0xFFu;
        HB = ((unsigned)__builtin_amdgcn_readlane((int)W[0], (int)rev6(stB)) >> 8) & 0xFFu;
    } else { HA = pA & ((1u << n) - 1u); HB = pB & ((1u << n) - 1u); }
    unsigned qA = rev6(((stA >> n) | rev6(HA & 0x3Fu)) & 0x3Fu), qB = rev6(((stB >> n) | rev6(HB & 0x3Fu)) & 0x3Fu);   // ring index at column 8j
    // The walk, top down: lane i <- the word of block j - i along frame A's path (tA) and along frame B's (tB).  It always runs its full
    // length (blocks below the window are read and never used; a window shorter than the longest happens once per frame).  Per block and
    // frame one v_readlane -- its lane select only looks at the low six bits, so the word just read IS the next ring index (frame A's as
    // it stands, frame B's after a shift) -- and one v_writelane with a constant lane.  (Assembler blocks are not seen by the compiler's
    // hazard pass, so the spacing is built in.)
    unsigned tA = 0, tB = 0;
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, 0\n\tv_writelane_b32 %1, %3, 0" : "+v"(tA), "+v"(tB) : "s"(HA), "s"(HB << 8));
    // One block = five instructions in a fixed order, which is what keeps them clear of the two hazards involved without a single s_nop:
    // an SGPR written by v_readlane is read as DATA by a VALU two instructions later at the earliest, and as a LANE SELECT four later.
#pragma unroll
    for (int i = 1; i < kMaxWalk; i++) {
        unsigned t;
        asm volatile("v_readlane_b32 %[a], %[w], %[a]\n\t"
                     "v_readlane_b32 %[t], %[w], %[b]\n\t"
                     "s_lshr_b32 %[b], %[t], 8\n\t"
                     "v_writelane_b32 %[ta], %[a], %[i]\n\t"
                     "v_writelane_b32 %[tb], %[t], %[i]"
                     : [a] "+s"(qA), [b] "+s"(qB), [t] "=&s"(t), [ta] "+v"(tA), [tb] "+v"(tB) : [w] "v"(W[i]), [i] "n"(i));
    }
    // decoded byte m = (block m >> 6) | (block m+1 & 0x3F) << 2; lane L <-> byte m_lo + L <-> blocks at walk positions nblk - L and nblk - L - 1
    const int at = (nblk - (int)lane) << 2;
    const unsigned loA = (unsigned)__builtin_amdgcn_ds_bpermute(at, (int)tA) & 0xFFu, upA = (unsigned)__builtin_amdgcn_ds_bpermute(at - 4, (int)tA) & 0xFFu;
    const unsigned loB = ((unsigned)__builtin_amdgcn_ds_bpermute(at, (int)tB) >> 8) & 0xFFu, upB = ((unsigned)__builtin_amdgcn_ds_bpermute(at - 4, (int)tB) >> 8) & 0xFFu;
    if (lane < (cntA >> 3)) outA[m_lo + (int)lane] = (uint8_t)((loA >> 6) | ((upA & 0x3Fu) << 2));
    if (lane < (cntB >> 3)) outB[m_lo + (int)lane] = (uint8_t)((loB >> 6) | ((upB & 0x3Fu) << 2));
}

}  // namespace sora
