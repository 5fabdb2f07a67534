// k_rx11b.hip -- the 802.11b receive graph for gfx950: 44 MHz samples in, long preamble, 1 Mbps DBPSK
// and 2 Mbps DQPSK payloads.  Reference: CreateDemodGraph (kernel/bb/demod11/fb11bdemod_config.hpp:122-172) driven by
// MAC11b_Receive (kernel/bb/demod11/fb11b_demod.cpp:27-76):
//   src -> TDCRemove -> TBB11bRxSwitch -+-> TEnergyDetect -> TDCEstimator                                  (carrier sense)
//                                        +-> TSymTiming -> TBarkerSync -> TBB11bRxRateSel -> TBB11bDespread ->
//       TSFDSync | TDBPSKDemap | TDQPSKDemap -> TDesc741 -> TBB11bPlcpSwitch -> TBB11bPlcpParser | TBB11bFrameSink
//
// Mapping: one wavefront per capture (captures are independent; a capture is one serial state machine because the
// early-late symbol timing, the Barker alignment and the differential demapper all carry state from sample to sample).
// A source call is 28 samples = 28 lanes: one coalesced 112-byte load, DC removal, the four sampling-phase energies of
// TSymTiming and the per-burst energies of TEnergyDetect are lane-parallel (cross-lane sums); the <= 8 chips a block
// yields are then pulled out with v_readlane and walk the chip/symbol/byte state machines as wave-uniform scalars.
// The 11-chip Barker correlation needs no chip queue: every operation of QuickBarkerDespread is a wrapping int16 add,
// so the symbol is accumulated chip by chip.  The harness's output buffer (stale bytes survive from frame to frame and
// end up in the reported FCS word and last MPDU byte) lives in LDS, 4 KiB per wave.
// HBM: 4 B per input sample read once; results are a few bytes per frame.  Bound: latency of the serial chain per
// capture, hidden by running 16 captures per CU.
// Not implemented (as in oracle/so_rx11b.c): the CCK decoders; a header announcing 5.5/11 Mbps ends the frame with
// SORA_E_NOT_SUPPORTED.
#include "kernels.h"

namespace sora {

namespace {
constexpr uint32_t E_NOT_SUPPORTED = 0x80000003u, E_SFD_FAIL = 0x80000004u, E_SFD_TIMEOUT = 0x80000008u, E_SYNC_TIMEOUT = 0x80000009u;   // the others: rx_types.h
enum { RATE_SYNC = 0, RATE_1M, RATE_2M, RATE_5P5M, RATE_11M };
enum { NO_PEAK_FOUND = 0, PEAK_FOUND, PEAK_VALID, PEAK_VALIDED, BARKER_SYNCED };
constexpr uint32_t kOutBuf = 4096;                                    // OUTPUTBUF_SIZE (fb11b_demod.cpp:21)

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int lane_of(int v, int l) { return __builtin_amdgcn_readlane(v, uni(l)); }
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ uint32_t dot_sign(int rre, int rim, int xre, int xim)          // (ulong)(ref.re*s.re + ref.im*s.im) >> 31
{ return ((uint32_t)(rre * xre) + (uint32_t)(rim * xim)) >> 31; }
}  // namespace

__global__ void __launch_bounds__(256) k_rx11b(Rx11bArgs A)
{
    __shared__ uint8_t s_out_all[4][kOutBuf];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cap_i = blockIdx.x * 4 + wave;
    if (cap_i >= A.ncaps) return;
    uint8_t* s_out = s_out_all[wave];
    for (int i = lane; i < (int)kOutBuf / 4; i += 64) reinterpret_cast<uint32_t*>(s_out)[i] = 0;
    lds_order();
    const CapDesc cap = A.caps[cap_i];
    const uint32_t* x = A.iq + cap.offset;
    const uint32_t thr = A.thr;

    // ---- context facades: survive every reset except as noted (ieee80211facade.hpp:21-135, stdfacade.h:51-57)
    uint32_t error_code = 0; int power = 0, rxrate = RATE_SYNC, plcp_data = 0;
    int dc_re = 0, dc_im = 0;                                          // CF_VecDC
    int last_re = 0, last_im = 0; uint32_t byte_reg = 0;              // CF_DifferentialDemap::last_symbol, CF_Descramber::byte_reg: never reset
    uint32_t frame_length = 0, rate_kbps = 0, frame_crc32 = 0;
    // ---- brick state
    uint32_t avg_energy = 0, win[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ecount = 0;               // TEnergyDetect
    uint32_t update_cnt = 8; int sdc_re = 0, sdc_im = 0;                                  // TDCEstimator
    int m_index = 2, m_frag = 0;                                                          // TSymTiming
    int sync_flag = NO_PEAK_FOUND, last_peak_cnt = -1, m_max = 0, search_count = 0;      // TBarkerSync
    int p_re[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_im[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int chip_n = 0, acc_re = 0, acc_im = 0;                                               // the current port's despreader
    int bit_one_found = 0; uint32_t word = 0; int bit_err_cnt = 0; uint32_t sync_cnt = 0; // TSFDSync
    int sym_n = 0; uint32_t sym_byte = 0; int ref_re = 0, ref_im = 0;                     // TDBPSKDemap / TDQPSKDemap burst in progress
    int hdr_n = 0; uint32_t hdr_lo = 0, hdr_hi = 0;                                       // TBB11bPlcpParser's 6-byte burst
    uint32_t byte_count = 0, crc32 = 0xFFFFFFFFu;                                         // TBB11bFrameSink
    uint32_t nfr = 0;

    auto graph_reset = [&]() {                                          // BB11bDemodCtx.reset() + pRxSource->Reset()
        error_code = 0; power = 0; rxrate = RATE_SYNC; plcp_data = 0;
        avg_energy = 0; ecount = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) win[i] = 0;
        update_cnt = 8; sdc_re = sdc_im = 0;
        m_index = 2; m_frag = 0;
        sync_flag = NO_PEAK_FOUND; last_peak_cnt = -1; m_max = 0; search_count = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) { p_re[i] = 0; p_im[i] = 0; }
        chip_n = 0; acc_re = acc_im = 0;
        bit_one_found = 0; word = 0; bit_err_cnt = 0; sync_cnt = 0;
        sym_n = 0; sym_byte = 0; hdr_n = 0; hdr_lo = hdr_hi = 0;
        byte_count = 0; crc32 = 0xFFFFFFFFu;
    };

    // ---- TBB11bFrameSink (PHY_11b.hpp:700-740)
    auto frame_sink = [&](uint32_t b) {
        if (byte_count < (uint32_t)((int)frame_length - 4)) {
            if (lane == 0 && byte_count < kOutBuf) s_out[byte_count] = (uint8_t)b;
            byte_count++;
            crc32 = A.crc[(crc32 ^ b) & 0xFF] ^ (crc32 >> 8);
        } else if (byte_count < frame_length) {
            if (lane == 0 && byte_count < kOutBuf) s_out[byte_count] = (uint8_t)b;
            byte_count++;
            if (byte_count == frame_length - 1) {                       // "speculating ACK": three FCS bytes are compared
                lds_order();
                uint32_t p = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { const uint32_t at = byte_count - 3 + k; p |= (at < kOutBuf ? (uint32_t)s_out[at] : 0u) << (8 * k); }
                p = (uint32_t)uni((int)p);
                frame_crc32 = p;
                error_code = ((~crc32 & 0x00FFFFFFu) == (p & 0x00FFFFFFu)) ? E_FRAME_OK : E_CRC32_FAIL;
            }
        }
    };
    // ---- TBB11bPlcpParser (PHY_11b.hpp:560-640)
    auto plcp_parser = [&]() {
        uint32_t c = 0xFFFF;                                            // CalcCRC16 over the first four bytes (CRC16.h: reflected 0x8408, init 0xFFFF, ~)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c ^= (hdr_lo >> (8 * i)) & 0xFF;
#pragma unroll
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x8408u : c >> 1;
        }
        if (((~c) & 0xFFFFu) != (hdr_hi & 0xFFFFu)) { error_code = E_PLCP_HEADER_FAIL; return; }
        const uint32_t signal = hdr_lo & 0xFF, service = (hdr_lo >> 8) & 0xFF; uint32_t len = hdr_lo >> 16;
        switch (signal) {
        case 0x0A: rate_kbps = 1000;  len = len >> 3; rxrate = RATE_1M; break;
        case 0x14: rate_kbps = 2000;  len = len >> 2; rxrate = RATE_2M; break;
        case 0x37: rate_kbps = 5500;  len = (((len * 11) >> 4) - (service >> 7) - ((service >> 3) & 1)) & 0xFFFF; rxrate = RATE_5P5M; break;
        case 0x6E: rate_kbps = 11000; len = (((len * 11) >> 3) - (service >> 7) - ((service >> 3) & 1)) & 0xFFFF; rxrate = RATE_11M; break;
        default:   rate_kbps = 0; len = 0;
        }
        frame_length = len; plcp_data = 1;
        if (rxrate > RATE_2M) error_code = E_NOT_SUPPORTED;             // CCK branches (cck.hpp) are not implemented
    };
    // ---- TDesc741 (scramble.hpp:93-170) -> TBB11bPlcpSwitch (PHY_11b.hpp:459-519)
    auto byte_out = [&](uint32_t b) {
        uint32_t st = byte_reg & 0x7F, xx = b, o = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t o1 = (xx ^ st ^ (st >> 3)) & 1;
            st = (st >> 1) | ((xx & 1) << 6);
            o = (o >> 1) | (o1 << 7);
            xx >>= 1;
        }
        byte_reg = b >> 1;
        if (!plcp_data) {
            if (hdr_n < 4) hdr_lo |= o << (8 * hdr_n); else hdr_hi |= o << (8 * (hdr_n - 4));
            if (++hdr_n == 6) { plcp_parser(); hdr_n = 0; hdr_lo = hdr_hi = 0; }
        } else frame_sink(o);
    };
    // ---- one despread symbol into the brick the rate selector's port leads to
    auto symbol_out = [&](int port, int sre, int sim) {
        if (port == RATE_SYNC) {                                        // TSFDSync (sfd_sync.hpp:76-126)
            const uint32_t bit = dot_sign(last_re, last_im, sre, sim);
            last_re = sre; last_im = sim;
            byte_reg &= 0x7F;
            const uint32_t sbit = (bit ^ byte_reg ^ (byte_reg >> 3)) & 1;
            byte_reg = (byte_reg >> 1) | (bit << 6);
            word = ((word >> 1) | (sbit << 15)) & 0xFFFF;
            sync_cnt++;
            if (!bit_one_found) { if (word == 0xFFFF) bit_one_found = 1; }
            else if (word == 0xF3A0) rxrate = RATE_1M;                  // DOT11B_PLCP_LONG_PREAMBLE_SFD
            else if (word != 0xFFFF) { if (bit_err_cnt++ > 32) { error_code = E_SFD_FAIL; return; } }
            if (sync_cnt > 128 + 16) error_code = E_SFD_TIMEOUT;
        } else if (port == RATE_1M) {                                   // TDBPSKDemap (barkerspread.hpp:312-390): 8 symbols -> 1 byte
            if (sym_n == 0) { ref_re = last_re; ref_im = last_im; }
            sym_byte |= dot_sign(ref_re, ref_im, sre, sim) << sym_n;
            ref_re = sre; ref_im = sim;
            if (++sym_n == 8) { last_re = sre; last_im = sim; const uint32_t b = sym_byte; sym_n = 0; sym_byte = 0; byte_out(b); }
        } else {                                                        // TDQPSKDemap (barkerspread.hpp:396-454): 4 symbols -> 1 byte
            if (sym_n == 0) { ref_re = last_re; ref_im = last_im; }
            const int re = (int)((uint32_t)(ref_re * sre) + (uint32_t)(ref_im * sim));
            const int im = (int)((uint32_t)(ref_re * sim) - (uint32_t)(ref_im * sre));
            sym_byte |= (((uint32_t)re + (uint32_t)im) >> 31) << (2 * sym_n);
            sym_byte |= (((uint32_t)re - (uint32_t)im) >> 31) << (2 * sym_n + 1);
            ref_re = sre; ref_im = sim;
            if (++sym_n == 4) { last_re = sre; last_im = sim; const uint32_t b = sym_byte; sym_n = 0; sym_byte = 0; byte_out(b); }
        }
    };
    // ---- TBarkerSync (symtiming.hpp:229-313) -> TBB11bRxRateSel -> TBB11bDespread::QuickBarkerDespread (barkerspread.hpp:277-303)
    auto chip_in = [&](int cre, int cim) {
        if (sync_flag == BARKER_SYNCED) {
            if (rxrate > RATE_2M) return;
            const int port = rxrate;
            int tre, tim;
            if (chip_n == 1 || chip_n == 4) { tre = neg16(cre) >> 4; tim = neg16(cim) >> 4; }          // negated before the shift
            else { tre = cre >> 4; tim = cim >> 4; if (chip_n >= 8) { tre = -tre; tim = -tim; } }       // chips 8..10 subtracted after it
            acc_re = w16(acc_re + tre); acc_im = w16(acc_im + tim);
            if (++chip_n == 11) { const int sre = acc_re, sim = acc_im; chip_n = 0; acc_re = acc_im = 0; symbol_out(port, sre, sim); }
            return;
        }
        search_count++;
        if (search_count >= 11 * 4) { error_code = E_SYNC_TIMEOUT; return; }
        const int sr = cre >> 4, si = cim >> 4;
        const int o_re = w16(p_re[0] - sr), o_im = w16(p_im[0] - si);
#define P_SUB(d, s) p_re[d] = w16(p_re[s] - sr); p_im[d] = w16(p_im[s] - si);
#define P_ADD(d, s) p_re[d] = w16(p_re[s] + sr); p_im[d] = w16(p_im[s] + si);
        P_SUB(0, 1) P_SUB(1, 2) P_ADD(2, 3) P_ADD(3, 4) P_ADD(4, 5) P_SUB(5, 6) P_ADD(6, 7) P_ADD(7, 8) P_SUB(8, 9)
#undef P_SUB
#undef P_ADD
        p_re[9] = sr; p_im[9] = si;
        const int corr = (int)((uint32_t)(o_re * o_re) + (uint32_t)(o_im * o_im));
        switch (sync_flag) {
        case NO_PEAK_FOUND:
            if (corr > m_max) { m_max = corr; last_peak_cnt = 1; } else if (++last_peak_cnt == 11) sync_flag = PEAK_FOUND;
            break;
        case PEAK_FOUND:
            m_max = corr / 2; last_peak_cnt = 1; sync_flag = PEAK_VALID;
            break;
        case PEAK_VALID:
            if (corr > m_max) { m_max = corr; last_peak_cnt = 0; sync_flag = NO_PEAK_FOUND; } else if (++last_peak_cnt == 11) sync_flag = PEAK_VALIDED;
            break;
        default:
            sync_flag = BARKER_SYNCED;
        }
    };
    // ---- TSymTiming::Process on one 28-sample block held one sample per lane (symtiming.hpp:42-170)
    auto sym_timing = [&](int bre, int bim) {
        int idx = m_index;
        while (idx < 28) {                                              // Decimation: every 4th sample from the current phase
            int at = idx;
            if (idx < 0) { at = 0; m_index += 4; }
            idx += 4;
            chip_in(lane_of(bre, at), lane_of(bim, at));
        }
        if (m_index >= 4) m_index = 0;
        // AdjustTiming: energies of the four sampling phases over the block, early-late decision
        const int er = bre >> 3, ei = bim >> 3;
        uint32_t e = lane < 28 ? (uint32_t)(er * er) + (uint32_t)(ei * ei) : 0u;
        e += (uint32_t)__shfl_down((int)e, 16);                         // lanes 0..15 += lanes 16..31 (28..31 hold 0)
        e += (uint32_t)__shfl_down((int)e, 8);
        e += (uint32_t)__shfl_down((int)e, 4);
        const int s0 = lane_of((int)e, 0), s1 = lane_of((int)e, 1), s2 = lane_of((int)e, 2), s3 = lane_of((int)e, 3);
        const int mi = m_index;
        const int sm = mi == 0 ? s0 : mi == 1 ? s1 : mi == 2 ? s2 : s3;
        const int se = mi == 0 ? s3 : mi == 1 ? s0 : mi == 2 ? s1 : s2;
        const int sl = mi == 0 ? s1 : mi == 1 ? s2 : mi == 2 ? s3 : s0;
        if (se < sl) {
            if (sm < se) { m_index++; m_frag = 0; } else if (sm < sl) m_frag++;
        } else {
            if (sm < sl) { m_index--; m_frag = 0; } else if (sm < se) m_frag--;
        }
        if (m_frag >= 4) { m_index++; m_frag = -3; } else if (m_frag <= -4) { m_index--; m_frag = 3; }
    };

    uint32_t pos = 0, remain = cap.nsamples;
    uint32_t raw = 0;                                                   // the source's output burst: lanes beyond a partial call keep the previous call's sample
    int prev_re = 0, prev_im = 0;                                       // previous call, DC removed (what may still be queued in front of TSymTiming)
    int qoff = 0;                                                       // samples queued in front of TSymTiming (multiple of 4, < 28)
    while (nfr < A.max_frames) {
        // ---- TMemSamples::Process (memsource.hpp:87-114)
        bool ret = true;
        if (remain > 28) { if (lane < 28) raw = x[pos + lane]; pos += 28; remain -= 28; }
        else if (remain == 0) ret = false;
        else { if (lane < (int)remain) raw = x[pos + lane]; pos += remain; remain = 0; }
        if (ret) {
            const cpx r = unpack(raw);
            int first_queued = 0;                                       // first burst (of 7) of this call that goes to TSymTiming
            if (!power) {
                first_queued = 7;
                for (int i = 0; i < 7; i++) {                          // TDCRemove -> TBB11bRxSwitch -> TEnergyDetect -> TDCEstimator, burst by burst
                    const int vre = w16(r.re - dc_re), vim = w16(r.im - dc_im);
                    uint32_t en = (uint32_t)(((int)((uint32_t)(vre * vre) + (uint32_t)(vim * vim))) >> 5);
                    en += (uint32_t)__shfl_xor((int)en, 1); en += (uint32_t)__shfl_xor((int)en, 2);
                    const uint32_t ave = (uint32_t)lane_of((int)en, 4 * i);
                    avg_energy = avg_energy - win[0] + ave;             // the 8-entry window as a FIFO (same order as the circular buffer)
#pragma unroll
                    for (int k = 0; k < 7; k++) win[k] = win[k + 1];
                    win[7] = ave;
                    ecount++;
                    if (ecount >= 32) {
                        if (ecount >= 100) { error_code = E_CS_TIMEOUT; break; }
                        if (avg_energy >= thr) power = 1;
                    }
                    if (power) { first_queued = i + 1; break; }         // energy gating: this burst reaches neither estimator nor demodulator
                    int hr = vre >> 5, hi = vim >> 5;                   // TDCEstimator: hadd(shift_right(pi, 5)) in wrapping int16
                    hr += __shfl_xor(hr, 1); hi += __shfl_xor(hi, 1); hr += __shfl_xor(hr, 2); hi += __shfl_xor(hi, 2);
                    sdc_re = w16(sdc_re + w16(lane_of(hr, 4 * i))); sdc_im = w16(sdc_im + w16(lane_of(hi, 4 * i)));
                    if (update_cnt == 0) { dc_re = w16(dc_re + (sdc_re >> 2)); dc_im = w16(dc_im + (sdc_im >> 2)); update_cnt = 8; sdc_re = sdc_im = 0; }
                    update_cnt--;
                }
            }
            if (power && error_code != E_CS_TIMEOUT) {
                const int cre = w16(r.re - dc_re), cim = w16(r.im - dc_im);
                if (first_queued == 0) {                                // a whole call: the queue hands TSymTiming one block of 28
                    const int src = lane < qoff ? 28 - qoff + lane : lane - qoff;
                    const int a_re = __shfl(prev_re, src), a_im = __shfl(prev_im, src), b_re = __shfl(cre, src), b_im = __shfl(cim, src);
                    sym_timing(lane < qoff ? a_re : b_re, lane < qoff ? a_im : b_im);
                } else qoff = 4 * (7 - first_queued);                   // the call in which power was detected: its tail is queued
                prev_re = cre; prev_im = cim;
            }
        }
        // ---- MAC11b_Receive bookkeeping after the source call (fb11b_demod.cpp:31-70)
        if (error_code != 0) {
            const uint32_t err = error_code;
            if (err != E_CS_TIMEOUT) {
                if (lane == 0) {
                    Rx11bRow row; row.end_sample = pos; row.error_code = err; row.rate_kbps = rate_kbps; row.length = frame_length; row.crc32 = frame_crc32;
                    A.rows[(size_t)cap_i * A.max_frames + nfr] = row;
                }
                if (err == E_FRAME_OK || err == E_CRC32_FAIL) {
                    uint8_t* dst = A.mpdu + ((size_t)cap_i * A.max_frames + nfr) * kOutBuf;
                    const uint32_t n = min(frame_length, kOutBuf);
                    lds_order();
                    for (uint32_t i = lane; i < n; i += 64) dst[i] = s_out[i];
                }
                nfr++;
            }
            if (err == E_FRAME_OK || err == E_CRC32_FAIL) {             // "jump advance of the last CRC byte": Seek (memsource.hpp:116-150)
                uint32_t off = rate_kbps == 1000 ? 8 * 11 * 4 : rate_kbps == 2000 ? 4 * 11 * 4 : 0;
                off = min(off, remain); pos += off; remain -= off;
            }
            // pRxSource->Flush(): what is queued is padded with zero samples and pushed through (brick.h FlushPort);
            // the switch flushes the branch its state selects, the rate selector pads its current port only
            if (power) {
                if (qoff > 0) {
                    const int a_re = __shfl(prev_re, 28 - qoff + lane), a_im = __shfl(prev_im, 28 - qoff + lane);
                    sym_timing(lane < qoff ? a_re : 0, lane < qoff ? a_im : 0); qoff = 0;
                }
                if (rxrate <= RATE_2M && chip_n > 0) { const int sre = acc_re, sim = acc_im; chip_n = 0; acc_re = acc_im = 0; symbol_out(rxrate, sre, sim); }
            }
            graph_reset();
            continue;                                                   // the routine returns and is called again: rc is not looked at
        }
        if (!ret) break;
    }
    if (lane == 0) A.nframes[cap_i] = nfr;
}

}  // namespace sora

// ------------------------------------------------------------------------------------------------ host side (C ABI, include/sora_hip.h)
#include <vector>
#include <string.h>
#include "../../include/sora_hip.h"

using namespace sora;

struct sora_rx11b {
    sora_rx_cfg cfg{};
    hipStream_t stream = nullptr;
    CapDesc* d_caps = nullptr; Rx11bRow* d_rows = nullptr; uint32_t* d_nframes = nullptr; uint8_t* d_mpdu = nullptr;
    sora_complex16* d_iq_own = nullptr;
    const uint32_t* d_crc = nullptr;
    std::vector<sora_capture_desc> h_caps;
    uint32_t ncaps = 0; bool have_results = false;
};

#define HIPCHK11(call) do { hipError_t _e = (call); if (_e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, #call, (int)_e); } while (0)

static void rx11b_free(sora_rx11b_t* rx)
{
    if (!rx) return;
    if (rx->stream) { (void)hipStreamSynchronize(rx->stream); (void)hipStreamDestroy(rx->stream); }
    (void)hipFree(rx->d_caps); (void)hipFree(rx->d_rows); (void)hipFree(rx->d_nframes); (void)hipFree(rx->d_mpdu); (void)hipFree(rx->d_iq_own);
    delete rx;
}

int sora_rx11b_create(const sora_rx_cfg* cfg, sora_rx11b_t** out)
{
    if (!cfg || !out || cfg->struct_size != sizeof(sora_rx_cfg)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_create: bad cfg", 0);
    if (cfg->sample_rate_mhz != 44) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_create: the 802.11b graph takes 44 MHz samples (sample_rate_mhz = 44)", 0);
    if (cfg->max_captures == 0 || cfg->max_total_samples == 0 || cfg->max_frames_per_capture == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "zero capacity", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (cfg->device < 0 || cfg->device >= ndev) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "device ordinal out of range", 0);
    HIPCHK11(hipSetDevice(cfg->device));
    sora_rx11b_t* rx = new sora_rx11b();
    rx->cfg = *cfg;
    if (rx->cfg.cca_pwr_threshold == 0) rx->cfg.cca_pwr_threshold = 1000 * 1000;      // BB11bDemodCtx.init (fb11bdemod_config.hpp:95)
    rx->d_crc = sora_internal_crc_table(cfg->device);
    const size_t rows = (size_t)cfg->max_captures * cfg->max_frames_per_capture;
    hipError_t e = rx->d_crc ? hipSuccess : hipErrorUnknown;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&rx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&rx->d_caps, sizeof(CapDesc) * cfg->max_captures);
    if (e == hipSuccess) e = hipMalloc((void**)&rx->d_rows, sizeof(Rx11bRow) * rows);
    if (e == hipSuccess) e = hipMalloc((void**)&rx->d_nframes, 4 * (size_t)cfg->max_captures);
    if (e == hipSuccess) e = hipMalloc((void**)&rx->d_mpdu, rows * 4096);
    if (e != hipSuccess) { rx11b_free(rx); return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_rx11b_create: device allocation", (int)e); }
    *out = rx;
    return SORA_OK;
}

void sora_rx11b_destroy(sora_rx11b_t* rx) { if (rx) { (void)hipSetDevice(rx->cfg.device); rx11b_free(rx); } }

int sora_rx11b_process_dev(sora_rx11b_t* rx, const sora_complex16* d_iq, const sora_capture_desc* caps, size_t ncaps)
{
    if (!rx || (ncaps && (!d_iq || !caps))) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_process_dev: null argument", 0);
    if (ncaps > rx->cfg.max_captures) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11b_process_dev: more captures than max_captures", 0);
    HIPCHK11(hipSetDevice(rx->cfg.device));
    std::vector<CapDesc> h(ncaps);
    uint64_t total = 0;
    for (size_t i = 0; i < ncaps; i++) {
        if (caps[i].offset % 4 != 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "capture offset must be a multiple of 4 samples", 0);
        if (caps[i].nsamples % 28 != 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "capture length must be a whole number of 28-sample source bursts", 0);
        h[i].offset = caps[i].offset; h[i].nsamples = caps[i].nsamples; h[i].capture_id = caps[i].capture_id; h[i].slot_base = 0; h[i].nslots = 0;
        total += caps[i].nsamples;
    }
    if (total > rx->cfg.max_total_samples) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11b_process_dev: more samples than max_total_samples", 0);
    rx->h_caps.assign(caps, caps + ncaps); rx->ncaps = (uint32_t)ncaps; rx->have_results = true;
    if (ncaps == 0) return SORA_OK;
    HIPCHK11(hipMemcpyAsync(rx->d_caps, h.data(), sizeof(CapDesc) * ncaps, hipMemcpyHostToDevice, rx->stream));
    HIPCHK11(hipStreamSynchronize(rx->stream));                                       // h is a local
    Rx11bArgs A;
    A.iq = reinterpret_cast<const uint32_t*>(d_iq); A.caps = rx->d_caps; A.ncaps = (uint32_t)ncaps; A.thr = rx->cfg.cca_pwr_threshold;
    A.max_frames = rx->cfg.max_frames_per_capture; A.rows = rx->d_rows; A.nframes = rx->d_nframes; A.mpdu = rx->d_mpdu; A.crc = rx->d_crc;
    hipLaunchKernelGGL(k_rx11b, dim3((unsigned)((ncaps + 3) / 4)), dim3(256), 0, rx->stream, A);
    HIPCHK11(hipGetLastError());
    return SORA_OK;
}

int sora_rx11b_process(sora_rx11b_t* rx, const sora_complex16* h_iq, size_t nsamples, const sora_capture_desc* caps, size_t ncaps)
{
    if (!rx || (nsamples && !h_iq)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_process: null argument", 0);
    if (nsamples > rx->cfg.max_total_samples) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11b_process: more samples than max_total_samples", 0);
    HIPCHK11(hipSetDevice(rx->cfg.device));
    if (!rx->d_iq_own) HIPCHK11(hipMalloc((void**)&rx->d_iq_own, sizeof(sora_complex16) * (rx->cfg.max_total_samples + 64)));
    HIPCHK11(hipMemcpyAsync(rx->d_iq_own, h_iq, sizeof(sora_complex16) * nsamples, hipMemcpyHostToDevice, rx->stream));
    return sora_rx11b_process_dev(rx, rx->d_iq_own, caps, ncaps);
}

int sora_rx11b_results(sora_rx11b_t* rx, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (!rx || !nout) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_results: null argument", 0);
    *nout = 0;
    if (!rx->have_results) return sora_internal_fail(SORA_ERR_FAILED, "no process call to report", 0);
    if (rx->ncaps == 0) return SORA_OK;
    HIPCHK11(hipSetDevice(rx->cfg.device));
    HIPCHK11(hipStreamSynchronize(rx->stream));
    const uint32_t mf = rx->cfg.max_frames_per_capture;
    std::vector<Rx11bRow> rows((size_t)rx->ncaps * mf); std::vector<uint32_t> nfr(rx->ncaps);
    HIPCHK11(hipMemcpy(rows.data(), rx->d_rows, sizeof(Rx11bRow) * rows.size(), hipMemcpyDeviceToHost));
    HIPCHK11(hipMemcpy(nfr.data(), rx->d_nframes, 4 * (size_t)rx->ncaps, hipMemcpyDeviceToHost));
    size_t n = 0, moff = 0; int rc = SORA_OK;
    for (uint32_t c = 0; c < rx->ncaps; c++)
        for (uint32_t i = 0; i < nfr[c] && i < mf; i++) {
            const Rx11bRow& r = rows[(size_t)c * mf + i];
            if (n >= max_out) { rc = SORA_ERR_CAPACITY; continue; }
            sora_frame_result& o = out[n++];
            memset(&o, 0, sizeof(o));
            o.capture_id = rx->h_caps[c].capture_id; o.end_sample = r.end_sample; o.error_code = r.error_code; o.rate_kbps = r.rate_kbps;
            o.length = (uint16_t)r.length; o.crc32 = r.crc32; o.mpdu_offset = (uint32_t)moff;
            if (h_mpdu && (r.error_code == 1u || r.error_code == 0x80000006u)) {
                const size_t len = r.length < 4096 ? r.length : 4096;
                if (moff + len > mpdu_cap) { rc = SORA_ERR_CAPACITY; continue; }
                HIPCHK11(hipMemcpy(h_mpdu + moff, rx->d_mpdu + ((size_t)c * mf + i) * 4096, len, hipMemcpyDeviceToHost));
                moff += len;
            }
        }
    *nout = n;
    if (rc != SORA_OK) return sora_internal_fail(rc, "sora_rx11b_results: output buffer too small", 0);
    return SORA_OK;
}
