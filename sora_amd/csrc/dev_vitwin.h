// dev_vitwin.h -- the window-parallel trellis's device code (the description is at the top of k_vitwin.hip): a unit's geometry, the forward pass of a wave's eight
// units, and the wave's body.  Shared by k_viterbi16w (k_vitwin.hip: one wave per workgroup, launched behind the symbol kernels) and k_pipe (k_rx.hip: the same wave
// inside the single-launch chain for a handful of frames, where it first waits until the soft values it needs have been published).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "dev_vit16.h"
#include "dev_winplan.h"

namespace sora {

namespace {

constexpr uint32_t kNever = 0xFFFFFFFFu;

// what a lane knows about one of the two units of its row
struct UnitGeom {
    uint32_t soft_off, last;     // the frame's stream, its last value
    uint32_t i0;                 // soft value of the unit's first step
    uint32_t nsteps;             // steps from the unit's first to the frame's last
    uint32_t ob;                 // WIN k0 - s0: where, in the unit's own step count, its first window's bits begin
    uint32_t tr_end;             // the frame's last trace-back, in the unit's step count
    uint32_t vstep, estep;       // steps at which the unit's vector goes to vecs[..][0] / [..][1] (kNever: not)
    uint32_t wleft;              // windows the unit decodes before it is done (the frame's end ends it anyway)
    uint32_t vec;
    uint8_t* out;                // the frame's output shifted by the unit's first step (bytes)
    uint32_t need;               // soft values of the frame the unit reads at most (k_pipe: what must have been published before the wave starts)
    uint32_t idx;                // the frame's place in its code-rate list
    bool valid, first;
};

// position p of a code-rate list of n frames -> the unit it holds (if any)
// (job_at(idx): the VitJob of place idx of the list -- out of jobs[] behind the symbol kernels, worked out from the frame table inside k_pipe)
// unit uu (kWinNone: none) of the frame at place idx of its list, the frame cut into units of m windows (nun of them)
template <int CR, int WIN, int LOOK>
__device__ __forceinline__ UnitGeom unit_of(const VitJob& J, uint32_t idx, uint32_t uu, uint32_t m, uint32_t nun, uint32_t q, uint32_t vbase, uint8_t* __restrict__ out)
{
    constexpr uint32_t GB = CR == 0 ? 2 : CR == 2 ? 4 : 3, GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;
    UnitGeom g;
    const bool has = uu != kWinNone;
    const uint32_t u = has ? uu : 0u;
    const uint32_t k0 = u * m, k1 = (u + 1u) * m;
    const bool last = u + 1u >= nun;
    const uint32_t b = (uint32_t)WIN * k0 / 24u * 24u;
    const uint32_t s0 = u == 0 ? 0u : b - (uint32_t)kWinWarm;
    g.valid = has; g.first = u == 0;
    g.soft_off = J.soft_off; g.last = max(J.nsoft, 1u) - 1u;
    g.i0 = s0 / GS * GB;
    g.nsteps = has ? J.nsoft / GB * GS - s0 : 0u;
    g.ob = (uint32_t)WIN * k0 - s0;
    g.tr_end = J.length * 8u + 16u + 6u - s0;
    g.vstep = (has && u != 0) ? (uint32_t)kWinWarm : kNever;
    g.estep = (has && !last) ? (uint32_t)WIN * k1 / 24u * 24u - s0 : kNever;
    g.wleft = last ? 0x10000u : m;
    g.vec = vbase + idx * q + u;
    g.out = out + J.out_off + (s0 >> 3);
    // the unit's last step: its last window's trace-back fires at the first schedule check at or
    // past WIN k1 + LOOK + 6 (checks come every GS steps); the frame's last unit runs to the end
    g.need = last ? J.nsoft : min(J.nsoft, (((uint32_t)WIN * k1 + (uint32_t)LOOK + 6u + GS) / GS + 1u) * GB);
    g.idx = idx;
    return g;
}

// position p of a list of n frames: the frame (place idx of the list), which of its nun units (kWinNone: none), windows per unit
struct UnitRef { uint32_t idx, uu, m, nun; };
template <int CR, int WIN, int LOOK>
__device__ __forceinline__ UnitRef unit_ref(const VitJob& J, uint32_t n, uint32_t p, uint32_t q, bool inside, uint32_t upos, uint32_t idx)
{
    const uint32_t nev = win_events(J.length, CR, WIN, LOOK), m = win_per_unit(nev, q), nun = (nev + m - 1u) / m;
    const uint32_t uu = !inside ? kWinNone : m != 1u ? (upos < nun ? upos : kWinNone) : n == 1u ? win_unit_lone(upos, nun) : (upos < nun ? win_unit_at(upos, nun) : kWinNone);
    return UnitRef{ idx, uu, m, nun };
}
template <int CR, int WIN, int LOOK, typename JOBS>
__device__ __forceinline__ UnitRef unit_ref_at(JOBS job_at, uint32_t n, uint32_t p, uint32_t q)
{
    const bool inside = p < win_slots(n, q);
    const uint32_t upos = inside ? p / n : 0u, idx = inside ? p - upos * n : 0u;
    return unit_ref<CR, WIN, LOOK>(job_at(idx), n, p, q, inside, upos, idx);
}
template <int CR, int WIN, int LOOK, typename JOBS>
__device__ __forceinline__ UnitGeom unit_geom(JOBS job_at, uint32_t n, uint32_t p, uint32_t q, uint32_t vbase, uint8_t* __restrict__ out)
{
    const bool inside = p < win_slots(n, q);
    const uint32_t upos = inside ? p / n : 0u, idx = inside ? p - upos * n : 0u;
    const VitJob J = job_at(idx);
    const UnitRef R = unit_ref<CR, WIN, LOOK>(J, n, p, q, inside, upos, idx);
    return unit_of<CR, WIN, LOOK>(J, idx, R.uu, R.m, R.nun, q, vbase, out);
}
// ... and unit `u` itself of the frame at place idx (the 64-lane form inside k_pipe: a wave holds unit u of two frames, or of one)
template <int CR, int WIN, int LOOK, typename JOBS>
__device__ __forceinline__ UnitGeom unit_geom_direct(JOBS job_at, uint32_t idx, uint32_t u, bool exists, uint32_t q, uint32_t vbase, uint8_t* __restrict__ out)
{
    const VitJob J = job_at(idx);
    const uint32_t nev = win_events(J.length, CR, WIN, LOOK), m = win_per_unit(nev, q), nun = (nev + m - 1u) / m;
    return unit_of<CR, WIN, LOOK>(J, idx, exists && u < nun ? u : kWinNone, m, nun, q, vbase, out);
}

// (ready(): called once, behind the wave's set-up and in front of its first soft value: false = give up)
template <int CR, int WIN, int LOOK, int BITS, typename READY>
__device__ __forceinline__ void forward16w(Lds16<WIN, LOOK>& S, const uint8_t* __restrict__ soft, const UnitGeom& GA, const UnitGeom& GB_, uint16_t* __restrict__ vecs, READY ready)
{
    using G = Geom16<WIN, LOOK>;
    constexpr int P = G::P;
    constexpr int GB = CR == 0 ? 2 : CR == 2 ? 4 : 3;                           // soft values per puncture group
    constexpr int GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;                           // trellis steps per group
    constexpr int CW = 12 / GS * GB;                                            // operands per 12-step chunk: 24 / 18 / 16
    constexpr uint32_t THR = WIN + LOOK + 6;
    const unsigned lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15, half = lane & 1u;
    const unsigned v0 = v_of_lane(l16);
    const UnitGeom& Mine = half ? GB_ : GA;
    const uint32_t nsteps = wave_max_u32(max(GA.nsteps, GB_.nsteps));
    const uint32_t my_last = Mine.last;

    auto which_of = [](int ph) { return CR == 0 ? 0 : CR == 1 ? (ph & 1) : ph % 3; };
    Vit16 V;
    // the frame's first unit starts from ALL_INIT0 / ALL_INIT (viterbilut.h:22-30), every other one from all-equal metrics; each half of the registers is a unit of its own
    {
        const unsigned ia = GA.first ? 0x18u << 9 : 0u, ib = GB_.first ? 0x18u << 25 : 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) V.U[i] = (v0 ^ kW[i]) == 0 ? 0u : (ia | ib);
    }
    const unsigned ring_base = (unsigned)(uintptr_t)&S.ring[0][0][0];
#pragma unroll
    for (int jb = 0; jb < 3; jb++)
#pragma unroll
        for (int i = 0; i < 4; i++) V.sadr[jb][i] = ring_base + ((row * 64u + rev6u(rol6(v0 ^ kW[i], jb == 0 ? 2 : jb == 1 ? 4 : 0))) << 1);
#pragma unroll
    for (int t = 0; t < 24; t++) {
        const int ph = t % 6, k = t % 8;
        const unsigned n = rol6(v0, ph + 1);
        const bool vb = (v0 >> (5 - ph)) & 1u;
        const unsigned ma = (__popc(n & 0155) & 1) ? 7u * kFld : 0u, mb = (__popc(n & 0117) & 1) ? 7u * kFld : 0u;
        const unsigned mx = which_of(ph) == 2 ? mb : ma;
        V.MX[t] = vb ? ((mx ^ (7u * kFld)) | (kOne << k)) : mx;
        if (t < 6) V.MY[t] = vb ? (mb ^ (7u * kFld)) : mb;
    }

    uint32_t tr = 0;                                                            // steps taken, in every unit's own count (wave-uniform)
    uint32_t pos = 0;
    uint32_t my_ob = Mine.ob, my_wleft = Mine.wleft;
    const uint32_t my_tr_end = Mine.tr_end;
    bool my_done = !Mine.valid;
    uint32_t vstepA = GA.vstep, vstepB = GB_.vstep, estepA = GA.estep, estepB = GB_.estep;   // (row-uniform: every lane holds a coset of BOTH units' metrics)

    auto normalize = [&]() {
        const unsigned m = row_pkmin(pk_min16(pk_min16(V.U[0], V.U[1]), pk_min16(V.U[2], V.U[3])));
#pragma unroll
        for (int i = 0; i < 4; i++) V.U[i] -= m;
    };
    auto pos_of = [&](uint32_t p, int jb) -> uint32_t { const uint32_t q = p + (uint32_t)jb; return q >= (uint32_t)P ? q - (uint32_t)P : q; };
    auto trace = [&](uint32_t my_cnt, int t24_last) {
        trace16<WIN, LOOK, true>((unsigned)(uintptr_t)&S, V.U[0], V.U[1], V.U[2], V.U[3], tr, my_ob, pos_of(pos, t24_last / 8), (uint32_t)(t24_last % 8), my_cnt, Mine.out);
    };
    auto next_event = [&]() -> uint32_t {
        uint32_t mine = my_done ? kNever : min(my_ob + THR, my_tr_end);
        mine = min(min(mine, min(vstepA, vstepB)), min(estepA, estepB));
        return wave_min_u32(mine);
    };
    uint32_t next_thr = next_event();
    bool all_done = wave_min_u32(my_done ? 1u : 0u) != 0u;
    // a vector: register i of row-lane l16 at [16 i + l16], the unit's half of the register -- the same order at both ends of a comparison (both are taken at a
    // multiple of 24 of the unit's own steps, where the state <-> lane map is the identity)
    auto save = [&](uint32_t vec, int which, bool hi) {
        uint16_t* d = vecs + ((size_t)vec * 2u + (uint32_t)which) * 64u + l16;
#pragma unroll
        for (int i = 0; i < 4; i++) d[16 * i] = (uint16_t)(hi ? V.U[i] >> 16 : V.U[i]);
    };
    auto check = [&](int t24_last) {
        if (tr >= next_thr) {
            // verification vectors: due only at multiples of 24 steps, i.e. straight after a normalisation, marks and carry guard cleared
            if (tr == vstepA) { save(GA.vec, 0, false); vstepA = kNever; }
            if (tr == vstepB) { save(GB_.vec, 0, true); vstepB = kNever; }
            if (tr == estepA) { save(GA.vec, 1, false); estepA = kNever; }
            if (tr == estepB) { save(GB_.vec, 1, true); estepB = kNever; }
            // trace-back schedule (viterbi.hpp:196-214), per unit
            uint32_t cnt = 0; bool partial = false;
            if (!my_done) {
                if (tr >= my_tr_end) { cnt = my_tr_end - my_ob - 6; my_done = true; }
                else if (tr >= my_ob + THR) { cnt = WIN; partial = true; }
            }
            if (wave_max_u32(cnt) != 0u) trace(cnt, t24_last);
            if (partial) { my_ob += WIN; if (--my_wleft == 0) my_done = true; }
            next_thr = next_event();
            all_done = wave_min_u32(my_done ? 1u : 0u) != 0u;
        }
    };

    struct Chunk { uint32_t v[CW]; };
    constexpr int NV = (CW + 7) / 8;
    struct Raw { SoftRaw r[NV]; };
    const uint32_t my_j = l16 >> 1;
    SoftCursor<BITS, CW> cur[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) cur[v].init(Mine.soft_off, Mine.i0 + my_j + 8u * v, my_last);
    auto fetch = [&](uint32_t c) -> Raw {
        Raw R;
#pragma unroll
        for (int v = 0; v < NV; v++) R.r[v] = cur[v].fetch(soft, c);
        return R;
    };
    uint16_t* my_ops = &S.ops[row][my_j][half];
    const uint4* row_ops = reinterpret_cast<const uint4*>(&S.ops[row][0][0]);
    auto unpack = [&](const Raw& R) -> Chunk {
#pragma unroll
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
        for (int jb = 0; jb < 3; jb++) pos512[jb] = pos_of(pos, jb) * 512u;
    };
    auto end_row = [&]() { pos = pos_of(pos, 3); set_row_pos(); };
    set_row_pos();
    auto group = [&](const Chunk& K, int h, int i0) {
        const int k0 = i0 / GS * GB, t24 = 12 * h + i0;
        acs16<0, P>(V, t24, K.v[k0], K.v[k0 + 1], pos512);
        if (CR != 0) acs16<1, P>(V, t24 + 1, K.v[k0 + 2], 0, pos512);
        if (CR == 2) acs16<2, P>(V, t24 + 2, 0, K.v[k0 + 3], pos512);
        if ((t24 + GS) % 8 == 0) normalize();
    };
    auto fast_chunk = [&](const Chunk& K, int h) {
#pragma unroll
        for (int g = 0; g < 12 / GS; g++) group(K, h, g * GS);
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

    // A unit is a few hundred to a few thousand steps: the plain loop of k_viterbi16 (operands unpacked at the head of every chunk), without its
    // two-table hand-over -- half the code, and what that hand-over buys (2 % for a wave alone on its SIMD) a unit gives back many times over.
    uint32_t c = 0;
    if (!ready()) return;
    Raw b0 = fetch(0), b1 = fetch(1), b2, b3;
    while (tr < nsteps && !all_done) {
        const uint32_t lim = min(nsteps, next_thr - 1);
        uint32_t rows = lim > tr ? (lim - tr) / 24 : 0;                         // rows that certainly need no look at the schedule
        for (; rows > 0; rows--) {
            b2 = fetch(c + 2);
            fast_chunk(unpack(b0), 0);
            b3 = fetch(c + 3);
            fast_chunk(unpack(b1), 1);
            end_row();
            b0 = b2; b1 = b3;
            c += 2;
        }
        if (!(tr < nsteps)) break;
        b2 = fetch(c + 2);
        chunk(unpack(b0), 0);
        if (!(tr < nsteps && !all_done)) break;
        b3 = fetch(c + 3);
        chunk(unpack(b1), 1);
        end_row();
        b0 = b2; b1 = b3;
        c += 2;
    }
}

// (S: the wave's own LDS block.  wave_index: which eighth-of-units of the call.  jobs_of(list): that list's job_at.  ready(A, B, list): called once the wave's units are
// known, before the first soft value is read -- k_pipe waits there for the symbol chain; false = give up.)
// done(cr, list, nl, w, q, job_at): what the wave does once its eight units (positions 8 w .. 8 w + 7 of the list) have written their bytes and vectors -- nothing, or
// k_viterbi16w_fin's tail (k_rx.hip): the LAST unit of a frame to arrive proves and finishes the frame
struct NothingDone { template <typename CRT, typename JOBS> __device__ __forceinline__ void operator()(CRT, uint32_t, uint32_t, uint32_t, uint32_t, JOBS) const {} };
template <int WIN, int LOOK, int BITS, typename JOBSOF, typename READY, typename DONE = NothingDone>
__device__ __forceinline__ void viterbi16w_wave(Lds16<WIN, LOOK>& S, uint32_t wave_index, JOBSOF jobs_of, READY ready, const uint32_t* __restrict__ hdr,
        uint32_t target, uint32_t vstride,
                                                const uint8_t* __restrict__ soft, uint8_t* __restrict__ out, uint16_t* __restrict__ vecs, DONE done = DONE())
{
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t n[3] = { hdr[0], hdr[1], hdr[2] };
    const uint32_t q = uni(win_units_per_frame(n[0] + n[1] + n[2], target));
    uint32_t w = uni(wave_index), list = 0;
    while (list < 3 && w >= (win_slots(n[list], q) + 7) / 8) { w -= (win_slots(n[list], q) + 7) / 8; list++; }
    if (list >= 3) return;
    const uint32_t nl = uni(n[list]);
    const auto job_at = jobs_of(list);
    const unsigned lane = threadIdx.x & 63, row = lane >> 4;
    const uint32_t pa = 8u * w + 2u * row, pb = pa + 1u, vbase = list * vstride;
    const uint32_t code_rate = uni(job_at(0).code_rate);                         // (a list holds one code rate)
    auto run = [&](auto cr) {
        constexpr int CR = decltype(cr)::value;
        UnitGeom A = unit_geom<CR, WIN, LOOK>(job_at, nl, pa, q, vbase, out), B = unit_geom<CR, WIN, LOOK>(job_at, nl, pb, q, vbase, out);
        if (__ballot(A.valid || B.valid) == 0) return;                           // (frames shorter than the longest leave whole waves empty)
        // an empty slot steps through a unit that exists: well-formed operands, nothing written
        if (!B.valid) { const bool v = false; B = A; B.valid = v; B.vstep = B.estep = kNever; B.nsteps = 0; }
        if (!A.valid) { const bool v = false; const UnitGeom T = B; A = T; A.valid = v; A.vstep = A.estep = kNever; A.nsteps = 0; }
        forward16w<CR, WIN, LOOK, BITS>(S, soft, A, B, vecs, [&]() { return ready(A, B, list); });
        done(cr, list, nl, w, q, job_at);
    };
    if (code_rate == 0) run(std::integral_constant<int, 0>{});
    else if (code_rate == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
}

// One wave per workgroup behind the symbol kernels: wave w of code-rate list r holds positions 8w .. 8w+7 of the list, one pair of units per 16-lane row.
template <int WIN, int LOOK, int BITS>
__device__ __forceinline__ void viterbi16w_body(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ hdr, uint32_t jstride, uint32_t target, uint32_t vstride,
                                                const uint8_t* __restrict__ soft, uint8_t* __restrict__ out, uint16_t* __restrict__ vecs)
{
    __shared__ Lds16<WIN, LOOK> S;
    auto jobs_of = [&](uint32_t list) { const VitJob* __restrict__ jl = jobs + (size_t)list * jstride; return [jl](uint32_t idx) { return jl[idx]; }; };
    auto ready = [](const UnitGeom&, const UnitGeom&, uint32_t) { return true; };
    viterbi16w_wave<WIN, LOOK, BITS>(S, blockIdx.x, jobs_of, ready, hdr, target, vstride, soft, out, vecs);
}

}  // namespace

}  // namespace sora
