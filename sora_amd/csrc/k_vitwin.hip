// k_vitwin.hip -- the K=7 trellis decoded WINDOW-PARALLEL with exact verification (round 5; gfx950).
//
//   k_viterbi16w   eight UNITS per wave in the 16-lanes-per-pair layout of k_vit16.hip (dev_vit16.h): a unit = windows k0 .. k1 - 1 of one frame
//   k_win_redo     (k_rx.hip) per pair of frames: every unit's metric vector at its verify point against the vector its predecessor had there; a pair with a
//                  mismatch is decoded again, serially, by the same wave (k_viterbi's body), which overwrites what the units wrote
//
// Why.  T11aViterbi (viterbi.hpp:148-235) is one serial chain per frame: 8-bit wrapping metrics, the decision in the LSB, unsigned minimum --
// no block decomposition of that arithmetic is exact by itself, so one frame was one wave-slot, a lone 4096-frame call
// filled half the SIMDs (0.62 ms) and a single capture took 33 ns per trellis step (fsample-6: 0.37 ms).  But the chain FORGETS: the decoder's
// whole state at a normalisation point (viterbicore.h:444-465) is the 64 seven-bit metrics relative to their minimum -- the LSBs are
// overwritten by the next step -- and survivor paths merge within a few constraint lengths, after which that vector no longer depends on
// where the decoder started.  The reference itself already cuts the frame into 256-bit trace-back windows (viterbi.hpp:196-214).
//
// How.  A unit starts kWinWarm steps before its VERIFY POINT b = floor24(256 k0) (a step at which every code rate normalises) from all-zero
// metrics, steps through the same add-compare-select code as k_viterbi16 (the state <-> lane map is free at an all-equal start, so the unit
// simply calls its first step "step 0": b - kWinWarm is a multiple of 24, which keeps puncture phase, normalisation points and 8-step decision
// blocks aligned with the frame's), stores its vector when it reaches b, decodes its windows with the unchanged trace-back, and stores its
// vector again at the next unit's verify point.  The frame's first unit starts at step 0 from the reference's initial metrics.  If
// vector(unit u at b_u) == vector(unit u - 1 at b_u) for every u >= 1, then by induction every unit was on the reference decoder's own
// trajectory from its verify point on -- the forward pass is a deterministic function of (vector at a normalisation point, soft values after
// it) -- and every decision a window's walk reads (columns > 256 k0 >= b) is the reference's: bit-exact by construction, not by probability.
// tools/winmodel measured how often the proof fails: never for a frame that passes its CRC (kWinWarm = 96, every rate, noise up to the
// decoding threshold); frames that are noise fail and are decoded again serially, so the worst case costs the serial kernel on top.
//
// Cost: (kWinWarm + 30) extra steps per unit.  A call is cut into about one chip's worth of units (dev_winplan.h; 16384): a lone capture
// becomes units of one window, a 4096-frame call units of twelve.  Unit u of frame idx sits at position u x frames + idx of its code rate's
// list -- the eight units of a wave are the same piece of eight frames, so their trace-backs fall on the same steps.
#include <hip/hip_runtime.h>
#include "dev_vitwin.h"

namespace sora {

__global__ void __launch_bounds__(64) k_viterbi16w(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ hdr, uint32_t jstride, uint32_t target, uint32_t vstride,
                                                   const uint8_t* __restrict__ soft, uint8_t* __restrict__ out, uint16_t* __restrict__ vecs)
{ viterbi16w_body<256, 24, 3>(jobs, hdr, jstride, target, vstride, soft, out, vecs); }
// the 802.11n graph's decoder, T11aViterbi<5000*8, 312, 192, 36> (fb11ndemod_config.hpp:199), one byte per soft value: the same body -- a unit's verify point
// floor24(192 k0) is 192 k0 itself
__global__ void __launch_bounds__(64) k_viterbi16w_11n(const VitJob* __restrict__ jobs, const uint32_t* __restrict__ hdr, uint32_t jstride, uint32_t target, uint32_t vstride,
                                                       const uint8_t* __restrict__ soft, uint8_t* __restrict__ out, uint16_t* __restrict__ vecs)
{ viterbi16w_body<192, 36, 8>(jobs, hdr, jstride, target, vstride, soft, out, vecs); }

// (the proof itself -- every unit's vector at its verify point against its predecessor's at the same step -- and the serial decode of what fails it are ONE kernel, k_win_redo in
// k_rx.hip, beside the serial trellis it shares its body with)

}  // namespace sora
