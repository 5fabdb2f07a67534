// dev_winplan.h -- how a call's frames are cut into the units of the window-parallel trellis (k_vitwin.hip).  Nothing is planned ahead and no table is kept:
// a frame's cut follows from its length and code rate, the number of frames of the call and the call's unit target, and both kernels (k_viterbi16w,
// k_win_redo) work it out where they need it.
#pragma once
#include "rx_types.h"

namespace sora {

// Trace-backs of a frame (T11aViterbi<.., WIN, LOOK>::Process, viterbi.hpp:196-214): window k fires as a partial one iff the first schedule check at or past
// step WIN k + WIN + LOOK + 6 comes before the frame's last step; the frame's end is one more.  Checks happen after every puncture group (GS steps).
__host__ __device__ inline uint32_t win_events(uint32_t length, uint32_t code_rate, uint32_t WIN, uint32_t LOOK)
{
    const uint32_t GS = code_rate == 0 ? 1u : code_rate == 2 ? 3u : 2u, tr_end = length * 8u + 16u + 6u, thr = WIN + LOOK + 6u;
    const uint32_t top = (tr_end - 1u) / GS * GS;
    return (top >= thr ? (top - thr) / WIN + 1u : 0u) + 1u;
}
constexpr uint32_t kWinMaxUnits = 80;             // a frame has at most 80 trace-backs (2500 bytes)
// units a frame may be cut into at most: the call's target shared among its frames
__host__ __device__ inline uint32_t win_units_per_frame(uint32_t njobs_total, uint32_t target)
{
    const uint32_t q = target / (njobs_total ? njobs_total : 1u);
    return q < 1u ? 1u : q > kWinMaxUnits ? kWinMaxUnits : q;
}
// Windows per unit.  From two on a multiple of three, so that WIN k0 mod 24 -- the distance of a unit's verify point from its first window, and with it the steps
// at which the unit's trace-backs fall -- is the same for every unit of the frame (256 = 16 mod 24).
__host__ __device__ inline uint32_t win_per_unit(uint32_t nev, uint32_t q)
{
    uint32_t m = (nev + q - 1u) / q;
    if (m == 0) m = 1;
    if (m >= 2u) m = (m + 2u) / 3u * 3u;
    return m;
}
// A frame cut into single windows: slot p of the frame holds unit win_unit_at(p): sorted by u mod 3, so that the units of a wave mostly share their trace-back steps
__host__ __device__ inline uint32_t win_unit_at(uint32_t p, uint32_t nun)
{
    const uint32_t c0 = (nun + 2u) / 3u, c1 = (nun + 1u) / 3u;
    return p < c0 ? 3u * p : p < c0 + c1 ? 3u * (p - c0) + 1u : 3u * (p - c0 - c1) + 2u;
}

}  // namespace sora
