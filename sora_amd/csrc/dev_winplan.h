// dev_winplan.h -- cutting a frame into the units of the window-parallel trellis (k_vitwin.hip), done by whichever kernel publishes the frame's VitJob
// (k_frame; k_track in the three-kernel symbol chain).
#pragma once
#include "kernels.h"

namespace sora {

// Trace-backs of a frame (T11aViterbi<.., WIN, LOOK>::Process, viterbi.hpp:196-214): window k fires as a partial one iff the first schedule check at or past
// step WIN k + WIN + LOOK + 6 comes before the frame's last step; the frame's end is one more.  Checks happen after every puncture group (GS steps).
__host__ __device__ inline uint32_t win_events(uint32_t length, uint32_t code_rate, uint32_t WIN, uint32_t LOOK)
{
    const uint32_t GS = code_rate == 0 ? 1u : code_rate == 2 ? 3u : 2u, tr_end = length * 8u + 16u + 6u, thr = WIN + LOOK + 6u;
    const uint32_t top = (tr_end - 1u) / GS * GS;
    return (top >= thr ? (top - thr) / WIN + 1u : 0u) + 1u;
}
// Windows per unit: at most q = target / frames-of-the-call units per frame; from two windows on a multiple of three, so that WIN k0 mod 24 -- the distance of a
// unit's verify point from its first window, and with it the step at which the unit's trace-backs fall -- is the same for every unit of the frame.
__host__ __device__ inline uint32_t win_per_unit(uint32_t nev, uint32_t njobs_total, uint32_t target)
{
    const uint32_t q = target / (njobs_total ? njobs_total : 1u);
    uint32_t m = q ? (nev + q - 1u) / q : nev;
    if (m == 0) m = 1;
    if (m >= 2u) m = (m + 2u) / 3u * 3u;
    return m;
}
// unit u of a frame cut into single windows sits at position p of the frame's range: sorted by u mod 3, so that the eight units of a wave mostly share their
// trace-back steps
__host__ __device__ inline uint32_t win_unit_at(uint32_t p, uint32_t nun)
{
    const uint32_t c0 = (nun + 2u) / 3u, c1 = (nun + 1u) / 3u;
    return p < c0 ? 3u * p : p < c0 + c1 ? 3u * (p - c0) + 1u : 3u * (p - c0 - c1) + 2u;
}

// Called by `gsize` lanes (gl = 0 .. gsize - 1) that share the frame; bcast(v) returns lane 0's v in all of them.
template <typename BCAST>
__device__ __forceinline__ void win_plan_frame(const RxArgs& A, uint32_t list, uint32_t idx, uint32_t length, uint32_t code_rate, unsigned gl, unsigned gsize, BCAST bcast)
{
    if (!A.wunits) return;
    const uint32_t njobs_total = A.njobs[0] + A.njobs[1] + A.njobs[2];
    const uint32_t nev = win_events(length, code_rate, 256u, 24u);
    const uint32_t m = win_per_unit(nev, njobs_total, A.wtarget), nun = (nev + m - 1u) / m;
    uint32_t vec0 = 0, pos0 = 0;
    if (gl == 0) { vec0 = atomicAdd(&A.hdr[kHdrVecs], nun); pos0 = atomicAdd(&A.hdr[kHdrUnits + list], nun); }
    vec0 = bcast(vec0); pos0 = bcast(pos0);
    const uint32_t jslot = list * A.nrows + idx;
    if (gl == 0) { WinFrame F; F.vec0 = vec0; F.nunits = nun; A.wframes[jslot] = F; }
    for (uint32_t p = gl; p < nun; p += gsize) {
        if (pos0 + p >= A.wstride) break;                                        // (cannot happen: the lists are sized for target + rows units)
        const uint32_t u = m == 1u ? win_unit_at(p, nun) : p;
        WinUnit W; W.job = jslot; W.k0 = (uint16_t)(u * m); W.k1 = u + 1u == nun ? (uint16_t)0xFFFFu : (uint16_t)((u + 1u) * m); W.vec = vec0 + u; W.u = u;
        A.wunits[(size_t)list * A.wstride + pos0 + p] = W;
    }
}

}  // namespace sora
