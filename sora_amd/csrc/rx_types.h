// rx_types.h -- device-resident data layout of one sora_rx_process call (shared by host and kernels).
//
// HBM layout (all arrays allocated once per handle, sized by sora_rx_cfg):
//   iq            input captures, COMPLEX16, read once per stage that needs them
//   caps[]        per capture {offset, nsamples, id, slot_base}
//   frames[]      frame table, max_frames_per_capture rows per capture, filled by k_scan in time order
//   fctx[]        per frame: FreqCoeffs[64] + ChannelCoeffs[64] (CF_FreqCompensate / CF_Channel_11a)
//   soft[]        per frame: its de-interleaved soft values (0..7) as a packed bit stream, THREE BITS per value, value i in bits
//                 3 i .. 3 i + 2 (little-endian), base = slot0 * 108 bytes (a 64-QAM symbol's 288 values are exactly 108 bytes, so a
//                 frame's stream never leaves its own symbol slots).  VitJob::soft_bits = 3: what k_viterbi / k_viterbi16 read.  (The 802.11n and
//                 40 MHz producers write one BYTE per value, soft_bits = 8, read by k_viterbi11n / k_viterbi16_11n: the format is a
//                 property of the kernel.)  Readers fetch 16 bits at byte (bits * i) >> 3 and take three of them.
//   vout[]        per frame: Viterbi output bytes (length+2), base = slot0*32
//   mpdu[]        per frame: descrambled MPDU, base = slot0*32 (same geometry as vout)
//   rows[]        compacted sora_frame_result rows + counter
#pragma once
#include <stdint.h>

namespace sora {

struct CapDesc {
    uint64_t offset;        // samples from iq base
    uint32_t nsamples;      // input-rate samples
    uint32_t capture_id;
    uint32_t slot_base;     // first symbol slot of this capture
    uint32_t nslots;        // floor(n20/80)+1
};

struct FrameRow {           // 64 bytes
    uint32_t capture;       // index into caps[]
    uint32_t start_sample;  // 20 MHz-rate index of the first sample given to T11aLTS
    uint32_t end_sample;
    uint32_t error_code;    // 0: pending data decode; PLCP_HEADER_FAIL; later FRAME_OK / CRC32_FAIL
    uint32_t rate_kbps;
    uint16_t length, nsym;
    uint16_t code_rate, nbpsc;
    uint32_t slot0;         // symbol slot of the SIGNAL symbol; data symbol s (1-based) is slot0+s
    uint32_t crc32;
    int16_t  cfo_est;
    int16_t  cfo_comp, sfo_comp, cfo_tracker, sfo_tracker;   // pilot-tracking state after SIGNAL
    uint16_t valid;
    uint32_t data_start;    // 20 MHz-rate index of the first sample of the SIGNAL symbol (80-sample window)
    uint32_t pad[3];
};

struct FrameCtx {           // 512 bytes
    uint32_t freq[64];      // packed COMPLEX16 FreqCoeffs
    uint32_t chan[64];      // packed COMPLEX16 ChannelCoeffs
};

struct VitJob {             // one Viterbi decode: a frame of the RX path or one job of sora_hip_viterbi11a
    uint32_t soft_off;      // BYTES from the soft base: this frame's soft stream
    uint32_t nsoft;
    uint32_t length;        // frame_length (decoded bytes = length+2)
    uint32_t dec_off;       // unused (decisions live in LDS)
    uint32_t out_off;       // bytes from the output base
    uint32_t valid;
    uint32_t code_rate;
    // 3: packed three bits per soft value; 8: one byte per soft value (its low three bits) -- informative: the trellis kernel's template parameter decides
    uint32_t soft_bits;
};

// ---- the window-parallel trellis (k_vitwin.hip, round 5).  A frame's trace-back windows (256 decoded bits each, viterbi.hpp:196-214) are cut into UNITS of m
// consecutive windows; a unit is decoded on its own -- from all-zero metrics kWinWarm steps before its verify point b = floor24(WIN k0) -- and proven afterwards:
// its metric vector at b must equal the vector the unit before it had there; a frame with any mismatch is decoded again serially (k_win_redo does both).
// warm-up steps in front of a verify point (a multiple of 24).  tools/winmodel: with 96
// no frame that passes its CRC failed a verification at any rate; 144 leaves a margin
constexpr int kWinWarm = 144;
constexpr int kWinVecBytes = 256;             // per unit: two vectors of 64 16-bit metric fields in the kernel's own lane order
constexpr unsigned kWinStatBanks = 64;        // k_win_redo spreads its record over this many banks of four counters (a power of two)

struct TrackRec {           // per data-symbol slot
    int16_t cfo_comp, sfo_comp;   // CompCoeffs of THIS symbol = build_coeff(cfo_comp, sfo_comp)
    int16_t avg, del;             // rotation applied by TPilotTrack to THIS symbol
};

// Frames are queued per code rate (1/2, 2/3, 3/4): k_viterbi runs two frames per wave and both must step through the
// same puncture pattern.  List r holds njobs[r] frame rows at joblist[r * stride ...]; job g of a call = element
// g - (jobs of the lists before) of its list.
struct JobRef { uint32_t list, idx; bool ok; };
__host__ __device__ inline JobRef locate_job(uint32_t g, const uint32_t* njobs)
{
    JobRef r; r.ok = true;
    const uint32_t n0 = njobs[0], n1 = njobs[1], n2 = njobs[2];
    if (g < n0) { r.list = 0; r.idx = g; }
    else if (g < n0 + n1) { r.list = 1; r.idx = g - n0; }
    else if (g < n0 + n1 + n2) { r.list = 2; r.idx = g - n0 - n1; }
    else { r.list = 0; r.idx = 0; r.ok = false; }
    return r;
}

constexpr int kSoftPerSlot = 288;      // N_CBPS max
constexpr int kSoftBytesPerSlot = 108; // ... at three bits each
constexpr int kSoftSlack = 64;         // bytes behind the last stream that a reader's 16-bit fetch / a producer's padded last group may touch
constexpr int kOutPerSlot  = 32;       // decoded bytes per symbol max 27 -> 32

constexpr uint32_t E_FRAME_OK = 0x00000001u, E_PLCP_HEADER_FAIL = 0x80000005u, E_CRC32_FAIL = 0x80000006u,
                   E_CS_TIMEOUT = 0x80000007u,
                   E_INTERNAL_TIMEOUT = 0x8000F001u;   // (SORA_E_INTERNAL_TIMEOUT, sora_hip.h: not one of the reference's codes)

}  // namespace sora
