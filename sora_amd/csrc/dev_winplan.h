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
// A list of ONE frame cut into single windows (a lone capture) lays them out so that no wave mixes trace-back steps: the three classes u mod 3 each start a wave of
// their own, and the frame's last unit -- whose only trace-back is the frame's end, earlier than its class's -- sits alone behind them.  (A wave steps until its slowest
// unit is done and pays every distinct trace-back step with a walk of its own: the wave that holds the frame's end is the launch's last one, and it should be a short one.)
// The layout needs up to 21 + 8 slots more than the frame has units: a list of one frame gets win_slots() of them.
constexpr uint32_t kWinLonePad = 32;
constexpr uint32_t kWinNone = 0xFFFFFFFFu;
__host__ __device__ inline uint32_t win_slots(uint32_t nframes, uint32_t q) { return nframes == 1u ? q + kWinLonePad : nframes * q; }
__host__ __device__ inline uint32_t win_unit_lone(uint32_t p, uint32_t nun)
{
    const uint32_t N = nun - 1u, c0 = (N + 2u) / 3u, c1 = (N + 1u) / 3u, c2 = N / 3u;
    const uint32_t a1 = (c0 + 7u) & ~7u, a2 = a1 + ((c1 + 7u) & ~7u), a3 = a2 + ((c2 + 7u) & ~7u);
    if (p < a1) return p < c0 ? 3u * p : kWinNone;
    if (p < a2) return p - a1 < c1 ? 3u * (p - a1) + 1u : kWinNone;
    if (p < a3) return p - a2 < c2 ? 3u * (p - a2) + 2u : kWinNone;
    return p == a3 ? N : kWinNone;
}

}  // namespace sora
