// k_rx11b.hip -- the 802.11b receive graph for gfx950: 44 MHz samples in, long preamble, 1 Mbps DBPSK,
// 2 Mbps DQPSK, 5.5 and 11 Mbps CCK payloads.  Reference: CreateDemodGraph (kernel/bb/demod11/fb11bdemod_config.hpp:122-172) driven by
// MAC11b_Receive (kernel/bb/demod11/fb11b_demod.cpp:27-76):
//   src -> TDCRemove -> TBB11bRxSwitch -+-> TEnergyDetect -> TDCEstimator                                  (carrier sense)
//                                        +-> TSymTiming -> TBarkerSync -> TBB11bRxRateSel -> TBB11bDespread ->
//       TSFDSync | TDBPSKDemap | TDQPSKDemap -> TDesc741 -> TBB11bPlcpSwitch -> TBB11bPlcpParser | TBB11bFrameSink
//                                                         TBB11bRxRateSel -> TCCK5P5Decoder | TCCK11Decoder -> TDesc741 (kernel/bb/Brick11/src/cck.hpp)
//
// Mapping: one wavefront per capture (captures are independent; a capture is one serial state machine because the
// early-late symbol timing, the Barker alignment and the differential demapper all carry state from sample to sample).
// A source call is 28 samples = 28 lanes: one coalesced 112-byte load, DC removal, the four sampling-phase energies of
// TSymTiming and the per-burst energies of TEnergyDetect are lane-parallel (cross-lane sums); the <= 8 chips a block
// yields are then pulled out with v_readlane and walk the chip/symbol/byte state machines as wave-uniform scalars.
// The 11-chip Barker correlation needs no chip queue: every operation of QuickBarkerDespread is a wrapping int16 add,
// so the symbol is accumulated chip by chip.  The harness's output buffer (stale bytes survive from frame to frame and
// end up in the reported FCS word and last MPDU byte) lives in LDS, 4 KiB per wave.
// That is the call-by-call path: it handles the alignment search and every source call that contains an EVENT (SFD found or given up,
// the PLCP header's last byte, the FCS compare, a reset).  Everything in between -- SFD search, header, payload of any rate -- goes through
// bulk_pass below: up to 64 source calls at once, ONE CALL PER LANE, because the only dependence from call to call is the early-late timing
// decision, which looks at nothing but the call's four phase energies.  Carrier sense stays call by call but
// evaluates a call's seven bursts together.
// HBM: 4 B per input sample read once; results are a few bytes per frame.  Bound: instruction issue (about 700 instructions per pass of
// 64 calls); 42 % of the HBM peak on the 1 Mbps bench batch.
// CCK: the chips of a block stay one per lane and are appended to a per-wave chip queue held in a register (lane j = j-th queued
// chip).  A code word is decoded across 16 lanes: lane 4m+s evaluates the hypothesis phi2 = m pi/2, phi3 = s-th of (0, pi/2, pi, 3pi/2)
// -- the reference's three "modules" of four correlations each (cck.hpp:268-745) are the quads of that grid -- and the reference's
// comparison tree (ties included) runs as two quad_perm exchanges plus four scalar compares.  phi4 and the DQPSK bits are scalar.
#include "kernels.h"

namespace sora {

namespace {
// the others: rx_types.h
constexpr uint32_t E_NOT_SUPPORTED = 0x80000003u, E_SFD_FAIL = 0x80000004u, E_SFD_TIMEOUT = 0x80000008u, E_SYNC_TIMEOUT = 0x80000009u;
enum { RATE_SYNC = 0, RATE_1M, RATE_2M, RATE_5P5M, RATE_11M };
enum { NO_PEAK_FOUND = 0, PEAK_FOUND, PEAK_VALID, PEAK_VALIDED, BARKER_SYNCED };
constexpr uint32_t kOutBuf = 4096;                                    // OUTPUTBUF_SIZE (fb11b_demod.cpp:21)

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int lane_of(int v, int l) { return __builtin_amdgcn_readlane(v, uni(l)); }
// lanes without a source read 0
template <int CTRL> __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// quad_perm [1,0,3,2], [2,3,0,1]: every lane of a quad gets its sum
__device__ __forceinline__ int quad_sum(int v) { v += dpp<0xB1>(v); return v + dpp<0x4E>(v); }
__device__ __forceinline__ int sum8(int v) { v = quad_sum(v); return v + dpp<0x104>(v); }                        // + row_shl:4: lanes 0..3 hold lanes 0..7
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ uint32_t dot_sign(int rre, int rim, int xre, int xim)          // (ulong)(ref.re*s.re + ref.im*s.im) >> 31
{ return ((uint32_t)(rre * xre) + (uint32_t)(rim * xim)) >> 31; }
}  // namespace

// Two instantiations.  CCK = false is the kernel for what nearly every capture holds (Barker-spread 1 / 2 Mbps frames): without the CCK decoders it
// fits 64 VGPRs and runs 8 waves per SIMD.  When a header announces 5.5 / 11 Mbps it abandons the capture and flags it; the CCK = true instantiation
// (125 VGPRs, 4 waves per SIMD) then redoes the flagged captures from their first sample.  Both write the same rows: the result is what one kernel
// with everything inlined gives (that single kernel cost every capture 7-13 % of its speed).
template <bool CCK>
__device__ __forceinline__ void rx11b_capture(const Rx11bArgs& A)
{
    __shared__ uint8_t s_out_all[4][kOutBuf];
    __shared__ int s_state_all[4][32];                                  // per wave: [0..19] TBarkerSync partial sums, [20..27] TEnergyDetect window
    // CRC-32 tables: [0] the byte table, [k] = the same after k more zero bytes ("slicing"), built below
    __shared__ uint32_t s_crc_all[8][256];
    uint32_t* const s_crc = s_crc_all[0];
    __shared__ int s_sym_all[4][2][64];                                 // per wave: despread sums of the symbols of one bulk pass (re, im)
    // per wave: the chips of one bulk pass over a CCK payload (queued ones first)
    __shared__ __attribute__((aligned(16))) uint32_t s_chip_all[CCK ? 4 : 1][CCK ? 544 : 4];
    s_crc[threadIdx.x] = A.crc[threadIdx.x];
    __syncthreads();
    { uint32_t t = s_crc[threadIdx.x]; for (int k = 1; k < 8; k++) { t = s_crc[t & 0xFF] ^ (t >> 8); s_crc_all[k][threadIdx.x] = t; } }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = uni((int)(threadIdx.x >> 6));      // wave-uniform values are told to be so: state then lives in SGPRs
    const uint32_t cap_i = blockIdx.x * 4 + wave;
    if (cap_i >= A.ncaps) return;
    if (CCK && uni((int)A.needs_cck[cap_i]) == 0) return;
    int cck_abort = 0;
    uint8_t* s_out = s_out_all[wave];
    int* const p_re = s_state_all[wave]; int* const p_im = p_re + 10; uint32_t* const win = reinterpret_cast<uint32_t*>(p_re + 20);
    if (lane < 32) p_re[lane] = 0;
    for (int i = lane; i < (int)kOutBuf / 4; i += 64) reinterpret_cast<uint32_t*>(s_out)[i] = 0;
    lds_order();
    const CapDesc cap = A.caps[cap_i];
    const uint32_t cap_n = (uint32_t)uni((int)cap.nsamples);
    const uint32_t* x = A.iq + (((uint64_t)(uint32_t)uni((int)(cap.offset >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)cap.offset));
    const uint32_t thr = A.thr;

    // ---- context facades: survive every reset except as noted (ieee80211facade.hpp:21-135, stdfacade.h:51-57)
    // NOTE on code shape: all of this state is wave-uniform and must stay in scalar registers.  Two constructs silently break that:
    // "if (c) a++; else b++;" and "if (c) a = 1; else b = 1;" are merged by the optimiser into a store through a SELECTED POINTER, which pins
    // both variables to private memory; loads from there count as divergent and drag every branch of the state machine into exec-masked
    // vector code (3x the instructions).  Hence the select-style updates below.  Check: opt -passes='print<uniformity>' on the device IR.
    uint32_t error_code = 0; int power = 0, rxrate = RATE_SYNC, plcp_data = 0;
    int dc_re = 0, dc_im = 0;                                          // CF_VecDC
    int last_re = 0, last_im = 0; uint32_t byte_reg = 0;              // CF_DifferentialDemap::last_symbol, CF_Descramber::byte_reg: never reset
    uint32_t frame_length = 0, rate_kbps = 0, frame_crc32 = 0;
    // ---- brick state
    uint32_t avg_energy = 0, ecount = 0;                                                  // TEnergyDetect (window in LDS)
    uint32_t update_cnt = 8; int sdc_re = 0, sdc_im = 0;                                  // TDCEstimator
    int m_index = 2, m_frag = 0;                                                          // TSymTiming
    int sync_flag = NO_PEAK_FOUND, last_peak_cnt = -1, m_max = 0, search_count = 0;      // TBarkerSync
    int chip_n = 0, acc_re = 0, acc_im = 0;                                               // the current port's despreader
    // TCCK5P5Decoder / TCCK11Decoder: queued chips (lane j = j-th chip, packed), is_even
    int cck_n = 0; uint32_t cck_even = 0; uint32_t cbuf = 0;
    int bit_one_found = 0; uint32_t word = 0; int bit_err_cnt = 0; uint32_t sync_cnt = 0; // TSFDSync
    int sym_n = 0; uint32_t sym_byte = 0; int ref_re = 0, ref_im = 0;                     // TDBPSKDemap / TDQPSKDemap burst in progress
    int hdr_n = 0; uint32_t hdr_lo = 0, hdr_hi = 0;                                       // TBB11bPlcpParser's 6-byte burst
    uint32_t byte_count = 0, crc32 = 0xFFFFFFFFu;                                         // TBB11bFrameSink
    uint32_t nfr = 0;

    // (every lambda below is force-inlined: a called closure would keep the state it captures in scratch memory)
    auto graph_reset = [&]() __attribute__((always_inline)) {                                          // BB11bDemodCtx.reset() + pRxSource->Reset()
        error_code = 0; power = 0; rxrate = RATE_SYNC; plcp_data = 0;
        avg_energy = 0; ecount = 0;
        update_cnt = 8; sdc_re = sdc_im = 0;
        m_index = 2; m_frag = 0;
        sync_flag = NO_PEAK_FOUND; last_peak_cnt = -1; m_max = 0; search_count = 0;
        lds_order(); if (lane < 32) p_re[lane] = 0; lds_order();
        chip_n = 0; acc_re = acc_im = 0; cck_n = 0; cck_even = 0;
        bit_one_found = 0; word = 0; bit_err_cnt = 0; sync_cnt = 0;
        sym_n = 0; sym_byte = 0; hdr_n = 0; hdr_lo = hdr_hi = 0;
        byte_count = 0; crc32 = 0xFFFFFFFFu;
    };

    // ---- TBB11bFrameSink (PHY_11b.hpp:700-740)
    auto frame_sink = [&](uint32_t b) __attribute__((always_inline)) {
        if (byte_count < (uint32_t)((int)frame_length - 4)) {
            if (lane == 0 && byte_count < kOutBuf) s_out[byte_count] = (uint8_t)b;
            byte_count++;
            crc32 = (uint32_t)uni((int)s_crc[(crc32 ^ b) & 0xFF]) ^ (crc32 >> 8);
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
    auto plcp_parser = [&]() __attribute__((always_inline)) {
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
        if (!CCK && uni(rxrate) > RATE_2M) { cck_abort = 1; error_code = E_NOT_SUPPORTED; }        // this capture goes to the CCK instantiation
    };
    // ---- TDesc741 (scramble.hpp:93-170) -> TBB11bPlcpSwitch (PHY_11b.hpp:459-519)
    auto byte_out = [&](uint32_t b) __attribute__((always_inline)) {
        uint32_t st = byte_reg & 0x7F, xx = b, o = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t o1 = (xx ^ st ^ (st >> 3)) & 1;
            st = (st >> 1) | ((xx & 1) << 6);
            o = (o >> 1) | (o1 << 7);
            xx >>= 1;
        }
        byte_reg = b >> 1;
        if (!uni(plcp_data)) {
            const uint32_t sh = o << (8 * (hdr_n & 3));                 // (no pointer select between the two words: that would pin them to memory)
            hdr_lo |= hdr_n < 4 ? sh : 0u; hdr_hi |= hdr_n < 4 ? 0u : sh;
            if (++hdr_n == 6) { plcp_parser(); hdr_n = 0; hdr_lo = hdr_hi = 0; }
        } else frame_sink(o);
    };
    // ---- TCCK11Decoder / TCCK5P5Decoder (cck.hpp): one code word = the chips in lanes 0..7 of `w` (packed COMPLEX16), wave-uniform result.
    // Lane 4m+s (m, s = 0..3): r = (-j)^m, q = j^s;  A1 = P1 + r P0, A2 = r P2 - P3, A3 = P5 + r P4, A4 = P7 - r P6;
    // Bx = (A2 + q A1) >> 2, By = (A4 + q A3) >> 2;  L = conj(Bx) By  (cck.hpp:268-303 and its three repetitions).
    struct CckLane { int lre, lim, l5; };
    auto cck_correlate = [&](uint32_t w, int first) __attribute__((always_inline)) {
        int pre[8], pim[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const int v = lane_of((int)w, first + i); pre[i] = (int)(short)v; pim[i] = v >> 16; }
        const int k = (4 - (lane >> 2)) & 3, sq = lane & 3;
        auto rot = [](int re, int im, int kk, int& ore, int& oim) __attribute__((always_inline)) {     // (re + j im) * j^kk, exact in 32 bits
            const int xr = (kk & 1) ? im : re, xi = (kk & 1) ? re : im;
            ore = ((kk + 1) & 2) ? -xr : xr; oim = (kk & 2) ? -xi : xi;
        };
        int t_re, t_im, a1r, a1i, a2r, a2i, a3r, a3i, a4r, a4i;
        rot(pre[0], pim[0], k, t_re, t_im); a1r = pre[1] + t_re; a1i = pim[1] + t_im;
        rot(pre[2], pim[2], k, t_re, t_im); a2r = t_re - pre[3]; a2i = t_im - pim[3];
        rot(pre[4], pim[4], k, t_re, t_im); a3r = pre[5] + t_re; a3i = pim[5] + t_im;
        rot(pre[6], pim[6], k, t_re, t_im); a4r = pre[7] - t_re; a4i = pim[7] - t_im;
        rot(a1r, a1i, sq, t_re, t_im); const int bxr0 = a2r + t_re, bxi0 = a2i + t_im;
        rot(a3r, a3i, sq, t_re, t_im); const int byr0 = a4r + t_re, byi0 = a4i + t_im;
        const int bxr = bxr0 >> 2, bxi = bxi0 >> 2, byr = byr0 >> 2, byi = byi0 >> 2;
        CckLane o;
        o.lre = (int)((uint32_t)(bxr * byr) + (uint32_t)(bxi * byi));
        o.lim = (int)((uint32_t)(bxr * byi) - (uint32_t)(bxi * byr));
        o.l5  = (int)((uint32_t)(bxr * byr) - (uint32_t)(((-bxi0) >> 2) * byi));     // 5.5 Mbps: B[0].im is negated BEFORE the shift (cck.hpp:94-98)
        return o;
    };
    // demap_dqpsk_bits (core/inc/soradsp.h:190-198): halves first
    auto cck_dqpsk = [&](int pos, int xre, int xim) __attribute__((always_inline)) {
        const int re = (int)((uint32_t)(last_re * xre) + (uint32_t)(last_im * xim)) >> 1, im = (int)((uint32_t)(last_re * xim) - (uint32_t)(last_im * xre)) >> 1;
        return ((((uint32_t)re + (uint32_t)im) >> 31) << pos) | ((((uint32_t)re - (uint32_t)im) >> 31) << (pos + 1));
    };
    auto cck11_decode = [&](uint32_t w) __attribute__((always_inline)) {                               // CCK11_DECODER (cck.hpp:255-763)
        const CckLane c = cck_correlate(w, 0);
        const int a1 = c.lre < 0 ? -c.lre : c.lre, a2 = c.lim < 0 ? -c.lim : c.lim;                   // (wrapping negation, as the reference's)
        const int sq = lane & 3;
        const uint32_t base = sq == 0 ? 0x00u : sq == 1 ? 0x30u : sq == 2 ? 0x10u : 0x20u;
        int M; uint32_t V;
        if (a1 > a2) { M = c.lre > 0 ? c.lre : -c.lre; V = base | (c.lre > 0 ? 0x00u : 0x40u); }
        else         { M = c.lim > 0 ? c.lim : -c.lim; V = base | (c.lim > 0 ? 0xC0u : 0x80u); }
        {   // "if (Max1 > Max2) 1 else 2", "if (Max3 > Max4) 3 else 4": the even lane keeps its own only when strictly larger
            const int oM = dpp<0xB1>(M); const uint32_t oV = (uint32_t)dpp<0xB1>((int)V);
            const bool other = (lane & 1) ? (oM > M) : !(M > oM);
            M = other ? oM : M; V = other ? oV : V;
        }
        {   // "if (Max34 > Module) Module = Max34": the pair (3,4) replaces the pair (1,2) only when strictly larger
            const int oM = dpp<0x4E>(M); const uint32_t oV = (uint32_t)dpp<0x4E>((int)V);
            const bool other = (lane & 2) ? !(M > oM) : (oM > M);
            M = other ? oM : M; V = other ? oV : V;
        }
        const int m1 = lane_of(M, 0), m2 = lane_of(M, 4), m3 = lane_of(M, 8), m4 = lane_of(M, 12);
        const uint32_t v1 = (uint32_t)lane_of((int)V, 0), v2 = (uint32_t)lane_of((int)V, 4) | 0x08u, v3 = (uint32_t)lane_of((int)V,
                8) | 0x04u, v4 = (uint32_t)lane_of((int)V, 12) | 0x0Cu;
        // "lable4" tests 3pi/2 against module 1, else pi against module 2
        uint32_t out = m1 > m2 ? (m1 > m4 ? v1 : v4) : (m2 > m3 ? v2 : v3);
        const int p7 = lane_of((int)w, 7); const int xre = (int)(short)p7, xim = p7 >> 16;
        out |= cck_dqpsk(0, xre, xim);
        out ^= (cck_even << 1) | cck_even; cck_even ^= 1u;
        last_re = xre; last_im = xim;
        return out & 0xFFu;
    };
    // TCCK5P5Decoder: even + odd half byte (cck.hpp:26-46, 70-206)
    auto cck5p5_decode = [&](uint32_t w) __attribute__((always_inline)) {
        uint32_t out = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const CckLane c = cck_correlate(w, 8 * half);
            const int l1 = lane_of(c.l5, 4), l2 = lane_of(c.l5, 12);                                  // phi2 = pi/2 and 3pi/2, phi3 = 0
            const int max1 = l1 > 0 ? l1 : -l1, max2 = l2 > 0 ? l2 : -l2;
            const uint32_t nib = max1 > max2 ? (l1 > 0 ? 0x0u : 0x8u) : (l2 > 0 ? 0x4u : 0xCu);
            out |= nib << (4 * half);
            const int p7 = lane_of((int)w, 8 * half + 7); const int xre = (int)(short)p7, xim = p7 >> 16;
            out |= cck_dqpsk(4 * half, xre, xim);
            last_re = xre; last_im = xim;
        }
        return (out ^ 0x30u) & 0xFFu;
    };
    // ---- TBB11bRxRateSel ports 3 / 4: chips k0 .. k0+n-1 of the block (lane k = chip k, packed) join the queue; a full burst is decoded
    auto cck_push = [&](uint32_t chips, int k0, int n) __attribute__((always_inline)) {
        if constexpr (CCK) {
        const uint32_t moved = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((lane - cck_n + k0) & 63), (int)chips);
        if (lane >= cck_n && lane < cck_n + n) cbuf = moved;
        cck_n += n;
        const int need = uni(rxrate) == RATE_5P5M ? 16 : 8;
        if (cck_n >= need) {
            const uint32_t b = need == 8 ? cck11_decode(cbuf) : cck5p5_decode(cbuf);
            cbuf = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((lane + need) & 63), (int)cbuf);
            cck_n -= need;
            byte_out(b);
        }
        }
    };
    // ---- one despread symbol into the brick the rate selector's port leads to
    auto symbol_out = [&](int port, int sre, int sim) __attribute__((always_inline)) {
        if (port == RATE_SYNC) {                                        // TSFDSync (sfd_sync.hpp:76-126)
            const uint32_t bit = dot_sign(last_re, last_im, sre, sim);
            last_re = sre; last_im = sim;
            byte_reg &= 0x7F;
            const uint32_t sbit = (bit ^ byte_reg ^ (byte_reg >> 3)) & 1;
            byte_reg = (byte_reg >> 1) | (bit << 6);
            word = ((word >> 1) | (sbit << 15)) & 0xFFFF;
            sync_cnt++;
            const bool ones = word == 0xFFFF, sfd = word == 0xF3A0, found = bit_one_found != 0;     // DOT11B_PLCP_LONG_PREAMBLE_SFD
            bit_one_found |= ones ? 1 : 0;                              // (selects, not "store 1 to one of two variables": that becomes a pointer
            rxrate = found && sfd ? RATE_1M : rxrate;                   //  select and pins both to memory)
            if (found && !sfd && !ones) { if (bit_err_cnt++ > 32) { error_code = E_SFD_FAIL; return; } }
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
    auto chip_in = [&](int cre, int cim) __attribute__((always_inline)) {
        if (sync_flag == BARKER_SYNCED) {
            const int port = uni(rxrate);                               // (a CCK rate never gets here: its blocks take the lane-parallel path of sym_timing)
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
        // (values read back from LDS are wave-uniform; saying so keeps everything derived from them -- the whole state machine -- scalar)
        const int o_re = w16(uni(p_re[0]) - sr), o_im = w16(uni(p_im[0]) - si);
#define P_SUB(d, s) p_re[d] = w16(uni(p_re[s]) - sr); p_im[d] = w16(uni(p_im[s]) - si);
#define P_ADD(d, s) p_re[d] = w16(uni(p_re[s]) + sr); p_im[d] = w16(uni(p_im[s]) + si);
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
    auto sym_timing = [&](uint32_t braw) __attribute__((always_inline)) {
        const cpx bu = unpack(braw);
        const int bre = w16(bu.re - dc_re), bim = w16(bu.im - dc_im);      // TDCRemove (dc.hpp:6-38): the estimate is frozen while demodulating
        if (sync_flag == BARKER_SYNCED) {
            // Decimation + rate selector + despreader for the whole block at once: chip k of the block in lane k.  A block
            // yields 6..8 chips and a symbol takes 11, so at most one symbol ends inside it; every operation of the
            // despreader is a wrapping add, so the two partial sums (before / after that boundary) are lane reductions.
            const int mi = m_index;
            const int cnt = mi < 0 ? 8 : (28 - mi + 3) >> 2;
            const int at = max(mi + 4 * lane, 0);                       // idx < 0 reads sample 0 (symtiming.hpp:72-75)
            const cpx cu = unpack((uint32_t)__shfl((int)braw, at));
            const int cre = w16(cu.re - dc_re), cim = w16(cu.im - dc_im);
            if (mi < 0) m_index += 4;
            if (uni(rxrate) > RATE_2M) cck_push(pack(mk(cre, cim)), 0, cnt);     // (uni: see the note at rxrate's declaration)
            else {
            int c = chip_n + lane; const bool first = c < 11; if (!first) c -= 11;
            int tre, tim;
            if (c == 1 || c == 4) { tre = neg16(cre) >> 4; tim = neg16(cim) >> 4; }
            else { tre = cre >> 4; tim = cim >> 4; if (c >= 8) { tre = -tre; tim = -tim; } }
            const bool valid = lane < cnt;
            int a_re = valid && first ? tre : 0, a_im = valid && first ? tim : 0, b_re = valid && !first ? tre : 0, b_im = valid && !first ? tim : 0;
            a_re = uni(sum8(a_re)); a_im = uni(sum8(a_im)); b_re = uni(sum8(b_re)); b_im = uni(sum8(b_im));
            if (chip_n + cnt >= 11) {
                const int port = uni(rxrate), sre = w16(acc_re + a_re), sim = w16(acc_im + a_im);
                acc_re = w16(b_re); acc_im = w16(b_im); chip_n = chip_n + cnt - 11;
                symbol_out(port, sre, sim);
                // the header has just announced a CCK rate: the chips after the symbol boundary are the first of the payload
                if (uni(rxrate) > RATE_2M) {
                    const int k0 = cnt - chip_n, n = chip_n;
                    chip_n = 0; acc_re = acc_im = 0;
                    cck_push(pack(mk(cre, cim)), k0, n);
                }
            } else { acc_re = w16(acc_re + a_re); acc_im = w16(acc_im + a_im); chip_n += cnt; }
            }
        } else {
            int idx = m_index;
            while (idx < 28) {                                          // Decimation: every 4th sample from the current phase
                int at = idx;
                if (idx < 0) { at = 0; m_index += 4; }
                idx += 4;
                chip_in(lane_of(bre, at), lane_of(bim, at));
            }
        }
        if (m_index >= 4) m_index = 0;
        // AdjustTiming: energies of the four sampling phases over the block, early-late decision
        const int er = bre >> 3, ei = bim >> 3;
        int e = lane < 28 ? (int)((uint32_t)(er * er) + (uint32_t)(ei * ei)) : 0;
        e += dpp<0x104>(e); e += dpp<0x108>(e);                         // row_shl:4, row_shl:8: lanes 0..3 of each 16-lane row hold that row's phase sums
        const int s0 = (int)((uint32_t)lane_of(e, 0) + (uint32_t)lane_of(e, 16)), s1 = (int)((uint32_t)lane_of(e, 1) + (uint32_t)lane_of(e, 17)),
                  s2 = (int)((uint32_t)lane_of(e, 2) + (uint32_t)lane_of(e, 18)), s3 = (int)((uint32_t)lane_of(e, 3) + (uint32_t)lane_of(e, 19));
        const int mi = m_index;
        const int sm = mi == 0 ? s0 : mi == 1 ? s1 : mi == 2 ? s2 : s3;
        const int se = mi == 0 ? s3 : mi == 1 ? s0 : mi == 2 ? s1 : s2;
        const int sl = mi == 0 ? s1 : mi == 1 ? s2 : mi == 2 ? s3 : s0;
        // (branch-free, as selects: "m_index++ else m_frag++" would become a pointer select that pins both to memory, and the scalar
        //  unit is what this kernel runs out of -- every avoided branch counts)
        const bool late_better = se < sl, a = sm < se, b = sm < sl;
        const bool jump = late_better ? a : b;
        const int di = late_better ? (a ? 1 : 0) : (b ? -1 : 0);
        const int df = late_better ? (!a && b ? 1 : 0) : (!b && a ? -1 : 0);
        int mf = jump ? 0 : m_frag + df;
        const int carry = mf >= 4 ? 1 : (mf <= -4 ? -1 : 0);
        m_index += di + carry; m_frag = carry > 0 ? -3 : (carry < 0 ? 3 : mf);
    };


    // ---- Runs of source calls in bulk.  Once the receiver is aligned to the Barker code -- SFD search, PLCP header, 1 / 2 Mbps payload; whole
    // source calls following one another -- the only serial dependence from call to call is the early-late timing loop, and that loop looks at
    // nothing but the four phase energies of each 28-sample call.  So up to 64 calls are taken in one pass, one call per LANE: every lane computes
    // its call's phase energies (v_dot2 on packed samples) and what AdjustTiming (symtiming.hpp:118-170) would decide at each sampling phase; the
    // recurrence over the calls is settled in runs by prefix sums; then the lanes pull their 6..8 chips and despread them into the (at most two)
    // symbols they belong to -- every step of QuickBarkerDespread is a wrapping int16 add, so partial sums from different calls simply add up
    // (LDS atomics) -- and the symbols are demapped one per lane against their left neighbours, the bits gathered with a ballot, descrambled with
    // two shifts and handed on as whole bytes.  A pass never contains an EVENT (SFD found or given up, the header's sixth byte, the FCS): header
    // and payload passes are sized to stop short of it, an SFD pass is cut back to the calls in front of the one the event falls into; that call,
    // like every other phase, goes through the call-by-call code.
    int* const sy_re = s_sym_all[wave][0]; int* const sy_im = s_sym_all[wave][1];
    auto writelane_u = [&](uint32_t& vec, uint32_t val, uint32_t ln) __attribute__((always_inline)) {
        vec = (uint32_t)lane == ln ? val : vec;
    };
    typedef short bs16x2_t __attribute__((ext_vector_type(2)));
    auto pk_sub = [](uint32_t a, uint32_t b) __attribute__((always_inline)) { return __builtin_bit_cast(uint32_t, (bs16x2_t)(__builtin_bit_cast(bs16x2_t,
            a) - __builtin_bit_cast(bs16x2_t, b))); };
    auto pk_add = [](uint32_t a, uint32_t b) __attribute__((always_inline)) { return __builtin_bit_cast(uint32_t, (bs16x2_t)(__builtin_bit_cast(bs16x2_t,
            a) + __builtin_bit_cast(bs16x2_t, b))); };
    auto pk_sra = [](uint32_t a, int n) __attribute__((always_inline)) { return __builtin_bit_cast(uint32_t, (bs16x2_t)(__builtin_bit_cast(bs16x2_t, a) >> (short)n)); };
    auto bulk_pass = [&](uint32_t& pos_, uint32_t& remain_, uint32_t& c_start_, uint32_t& c_stale_, uint32_t& p_start_, uint32_t& p_stale_,
            int qoff_) __attribute__((always_inline)) -> bool {
        const int port = uni(rxrate);
        // 0: TSFDSync, 1: the PLCP header's bytes, 2: a Barker payload's, 3: a CCK payload's
        const int mode = port == RATE_SYNC ? 0 : (!uni(plcp_data) ? 1 : port <= RATE_2M ? 2 : 3);
        if (mode == 1 && port != RATE_1M) return false;
        if (mode == 3 && !CCK) return false;
        const int spb = port == RATE_2M ? 4 : 8;                        // symbols per byte
        const int need = port == RATE_5P5M ? 16 : 8;                    // CCK: chips per byte
        int K = min(mode == 3 ? 62 : 64, (int)(remain_ / 28u));         // (62 calls: at most 63 code words with the queued chips)
        if (mode == 3) {
            if (byte_count + 1 >= frame_length) return false;
            const int chips_to_event = (int)(frame_length - 1 - byte_count) * need - cck_n;
            K = min(K, (chips_to_event - 1) / 8 - 1);
        } else if (mode != 0) {
            if (mode == 2 && byte_count + 1 >= frame_length) return false;
            const int R = mode == 2 ? (int)(frame_length - 1 - byte_count) : 6 - hdr_n;     // bytes up to and including the one that raises the event
            const int chips_to_event = 11 * (R * spb - sym_n) - chip_n;
            K = min(port == RATE_2M ? 32 : K, min(K, (chips_to_event - 1) / 8 - 1));        // (2 Mbps: 32 calls fill the 64-bit bit string)
        }
        if (K < 6) return false;
        const int m_index0 = m_index, m_frag0 = m_frag;
        const uint32_t base0 = pos_ - (uint32_t)qoff_;
        // ---- A. one call per lane: 28 samples, DC removed (TDCRemove: the estimate is frozen while demodulating), phase energies
        const bool act = lane < K;
        const uint32_t dcp = ((uint32_t)dc_re & 0xFFFFu) | ((uint32_t)dc_im << 16);
        uint32_t xs[28];
        {
            const uint4* src = reinterpret_cast<const uint4*>(x + base0 + 28u * (uint32_t)(act ? lane : 0));
#pragma unroll
            for (int q = 0; q < 7; q++) { const uint4 v = src[q]; xs[4 * q] = pk_sub(v.x, dcp); xs[4 * q + 1] = pk_sub(v.y, dcp); xs[4 * q + 2] = pk_sub(v.z,
                    dcp); xs[4 * q + 3] = pk_sub(v.w, dcp); }
        }
        int e0 = 0, e1 = 0, e2 = 0, e3 = 0;                            // sums over the block of |x >> 3|^2 per sampling phase, wrapping
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const bs16x2_t a0 = __builtin_bit_cast(bs16x2_t, pk_sra(xs[4 * q], 3)), a1 = __builtin_bit_cast(bs16x2_t, pk_sra(xs[4 * q + 1], 3));
            const bs16x2_t a2 = __builtin_bit_cast(bs16x2_t, pk_sra(xs[4 * q + 2], 3)), a3 = __builtin_bit_cast(bs16x2_t, pk_sra(xs[4 * q + 3], 3));
            e0 = __builtin_amdgcn_sdot2(a0, a0, e0, false); e1 = __builtin_amdgcn_sdot2(a1, a1, e1, false);
            e2 = __builtin_amdgcn_sdot2(a2, a2, e2, false); e3 = __builtin_amdgcn_sdot2(a3, a3, e3, false);
        }
        // ---- B. the timing recurrence.  Every lane first works out what AdjustTiming would decide for its call at each of the four sampling
        // phases (two 2-bit fields per phase: index step + 1, fraction step + 1); the serial part that is left is a scalar loop of a dozen
        // instructions per call: pick the field of the current phase, move (m_index, m_frag) on, leave lane k + 1 its decimation phase.
        uint32_t codes = 0;
#pragma unroll
        for (int mx = 0; mx < 4; mx++) {
            const int sm = mx == 0 ? e0 : mx == 1 ? e1 : mx == 2 ? e2 : e3;
            const int se = mx == 0 ? e3 : mx == 1 ? e0 : mx == 2 ? e1 : e2;
            const int sl = mx == 0 ? e1 : mx == 1 ? e2 : mx == 2 ? e3 : e0;
            const bool late_better = se < sl, a = sm < se, b = sm < sl;
            const int di = late_better ? (a ? 1 : 0) : (b ? -1 : 0);
            const int df = late_better ? (!a && b ? 1 : 0) : (!b && a ? -1 : 0);
            codes |= (uint32_t)((di + 1) | ((df + 1) << 2)) << (4 * mx);
        }
        // State: mx = m_index & 3 (what Decimation leaves: -1 -> 3 after its "idx < 0" step, 4 -> 0), M = m_frag + 3.  As long as the index
        // stays where it is every call just moves the fraction by its df, so a whole run of calls is settled by a prefix sum over the lanes:
        // the run ends at the first call that steps the index (its own decision, or the fraction reaching +-4).  That one call is done exactly
        // (F = M + (df + 1) is the new fraction + 4, or 4 after an index step; the 64-bit constant maps F = 0..8 to the wrapped fraction and
        // the carry + 1, 5 bits each), and the next run starts behind it.
        const unsigned long long kFrac = 0x6ull | (0x8ull << 5) | (0x9ull << 10) | (0xAull << 15) | (0xBull << 20) | (0xCull << 25) | (0xDull << 30) | (0xEull << 35) | (0x10ull << 40);
        const unsigned long long actm = K == 64 ? ~0ull : (1ull << K) - 1ull;
        uint32_t miv2, Mv;                                              // lane k: m_index + 2 and m_frag + 3 as call k finds them
        {
            uint32_t mx = (uint32_t)m_index & 3u, M = (uint32_t)(m_frag + 3), mraw = (uint32_t)(m_index + 2);
            int k0 = 0;
            miv2 = mraw; Mv = M;
            while (k0 < K) {
                miv2 = lane == k0 ? mraw : (lane > k0 ? mx + 2u : miv2);
                const uint32_t c = codes >> (4 * mx);
                const bool jump = (c & 3u) != 1u;
                const int step = lane >= k0 && !jump ? (int)((c >> 2) & 3u) - 1 : 0;
                int P = step;                                           // inclusive prefix sum (DPP, as scan_add in k_rx11n.hip)
                P += __builtin_amdgcn_update_dpp(0, P, 0x111, 0xF, 0xF, true); P += __builtin_amdgcn_update_dpp(0, P, 0x112, 0xF, 0xF, true);
                P += __builtin_amdgcn_update_dpp(0, P, 0x114, 0xF, 0xF, true); P += __builtin_amdgcn_update_dpp(0, P, 0x118, 0xF, 0xF, true);
                P += __builtin_amdgcn_update_dpp(0, P, 0x142, 0xA, 0xF, false); P += __builtin_amdgcn_update_dpp(0, P, 0x143, 0xC, 0xF, false);
                Mv = lane >= k0 ? (uint32_t)((int)M + P - step) : Mv;
                const unsigned long long evm = __ballot(lane >= k0 && (jump || (uint32_t)((int)M + P) > 6u)) & actm;
                if (evm == 0) { M = (uint32_t)((int)M + lane_of(P, K - 1)); mraw = mx + 2u; break; }
                const int ks = __builtin_ctzll(evm);
                const uint32_t ck = (uint32_t)lane_of((int)c, ks);
                const uint32_t d = ck & 3u, e = (ck >> 2) & 3u;
                const uint32_t Mb = (uint32_t)((int)M + lane_of(P, ks)) - (d == 1u ? e - 1u : 0u);      // the fraction call ks finds
                const uint32_t F = d == 1u ? Mb + e : 4u;
                const uint32_t t = (uint32_t)(kFrac >> (5 * F));
                M = t & 7u;
                mraw = mx + d + ((t >> 3) & 3u);                        // m_index + 2 = mx + (di + 1) + (carry + 1)
                mx = (mraw + 2u) & 3u;
                k0 = ks + 1;
            }
            m_frag = (int)M - 3; m_index = (int)mraw - 2;
        }
        const uint32_t miv = miv2 - 2u;
        // ---- C. chips of this lane's call and their partial despread sums
        const int mi = (int)miv;
        const int cnt = act ? (mi < 0 ? 8 : (28 - mi + 3) >> 2) : 0;
        int incl = cnt;                                                // inclusive prefix sum of the chip counts (DPP, as scan_add in k_rx11n.hip)
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true); incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true); incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false); incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
        const int total = lane_of(incl, K - 1);
        const int g0 = chip_n + incl - cnt;                            // chips before this call's first, counted from the start of the symbol in progress
        const int sid0 = (int)(((uint32_t)g0 * 5958u) >> 16);          // g0 / 11 for g0 < 2^13
        const int c0 = g0 - 11 * sid0;                                  // the first chip's place in the Barker code
        // chip j = the sample at offset max(mi + 4 j, 0): column mi & 3 of the block seen as 7 rows of 4, one row up for mi = 4, one row down
        // (and sample 0 first) for mi = -1.  (Selects spelled out on lane masks: left to itself the optimiser turns the choice into a computed
        // index and a 28-way select per chip.)
        auto vsel = [](unsigned long long m, uint32_t a, uint32_t b) __attribute__((always_inline)) { uint32_t d;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m)); return d; };
        const unsigned long long r0m = __ballot((mi & 1) != 0), r1m = __ballot((mi & 2) != 0), dnm = __ballot(mi < 0), upm = __ballot(mi == 4);
        uint32_t col[7], y[8];
#pragma unroll
        for (int q = 0; q < 7; q++) col[q] = vsel(r1m, vsel(r0m, xs[4 * q + 3], xs[4 * q + 2]), vsel(r0m, xs[4 * q + 1], xs[4 * q]));
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = col[min(j, 6)];
            v = vsel(dnm, j == 0 ? xs[0] : col[j - 1], v);
            if (j < 6) v = vsel(upm, col[j + 1], v);
            y[j] = v;
        }
        // QuickBarkerDespread's signs as bit strings over j: chips 1 and 4 negated before the >> 4, chips 8..10 after it; the chips in front of
        // the symbol boundary go to partA, the others to partB
        const uint32_t preB = 0x9012u >> c0, postB = 0x380700u >> c0, crossB = 0x3FF800u >> c0, validB = (1u << cnt) - 1u;
        const uint32_t aB = validB & ~crossB, bB = validB & crossB;
        const bool crossed = bB != 0;
        uint32_t partA = 0, partB = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t m1 = (uint32_t)((int)(preB << (31 - j)) >> 31), m2 = (uint32_t)((int)(postB << (31 - j)) >> 31);
            const uint32_t ma = (uint32_t)((int)(aB << (31 - j)) >> 31), mb = (uint32_t)((int)(bB << (31 - j)) >> 31);
            uint32_t t = pk_sra(pk_sub(y[j] ^ m1, m1), 4);             // (v ^ -1) - (-1) = -v in each half, wrapping
            t = pk_sub(t ^ m2, m2);
            partA = pk_add(partA, t & ma); partB = pk_add(partB, t & mb);
        }
        if constexpr (CCK) {
        if (mode == 3) {
            // ---- C3..F3. a CCK payload: the chips go to a buffer in LDS behind the queued ones, a code word (8 chips) per lane is decoded with the
            // reference's own sequence of comparisons (CCK11_DECODER cck.hpp:255-763: four phi2 hypotheses x four phi3 hypotheses, phi4 from
            // the larger component's sign; TCCK5P5Decoder cck.hpp:26-206: two hypotheses), the DQPSK pair against the previous word's last
            // chip, the odd-symbol rotation by lane parity, the descrambler a byte per lane against its left neighbour, and the frame sink's
            // CRC-32 folded eight bytes at a time.
            uint32_t* const chipbuf = s_chip_all[wave];
            lds_order();
            if (lane < cck_n) chipbuf[lane] = cbuf;
            const int gq = cck_n + incl - cnt;                          // this call's first chip in the buffer
#pragma unroll
            for (int j = 0; j < 8; j++) if (j < cnt) chipbuf[gq + j] = y[j];
            lds_order();
            const int T = cck_n + total, nw = T >> 3;                   // chips in the buffer, complete words
            const bool inw = lane < nw;
            int pre[8], pim[8];
            {
                const uint4 c0 = *reinterpret_cast<const uint4*>(chipbuf + 8 * (inw ? lane : 0)), c1 = *reinterpret_cast<const uint4*>(chipbuf + 8 * (inw ? lane : 0) + 4);
                const uint32_t cw[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
#pragma unroll
                for (int i = 0; i < 8; i++) { pre[i] = (int)(short)(cw[i] & 0xFFFFu); pim[i] = (int)cw[i] >> 16; }
            }
            auto rotc = [](int re, int im, int kk, int& ore, int& oim) __attribute__((always_inline)) {  // (re + j im) * j^kk, kk a constant
                const int xr = (kk & 1) ? im : re, xi = (kk & 1) ? re : im;
                ore = ((kk + 1) & 2) ? -xr : xr; oim = (kk & 2) ? -xi : xi;
            };
            // A1..A4 of hypothesis phi2 = m pi/2 (r = (-j)^m = j^k, k = (4 - m) & 3), then Bx, By and L of hypothesis phi3 = s-th of 0, pi/2, pi, 3pi/2
            auto corr = [&](int m, int sq, int& lre, int& lim, int& l5) __attribute__((always_inline)) {
                const int k = (4 - m) & 3;
                int t_re, t_im;
                rotc(pre[0], pim[0], k, t_re, t_im); const int a1r = pre[1] + t_re, a1i = pim[1] + t_im;
                rotc(pre[2], pim[2], k, t_re, t_im); const int a2r = t_re - pre[3], a2i = t_im - pim[3];
                rotc(pre[4], pim[4], k, t_re, t_im); const int a3r = pre[5] + t_re, a3i = pim[5] + t_im;
                rotc(pre[6], pim[6], k, t_re, t_im); const int a4r = pre[7] - t_re, a4i = pim[7] - t_im;
                rotc(a1r, a1i, sq, t_re, t_im); const int bxr0 = a2r + t_re, bxi0 = a2i + t_im;
                rotc(a3r, a3i, sq, t_re, t_im); const int byr0 = a4r + t_re, byi0 = a4i + t_im;
                const int bxr = bxr0 >> 2, bxi = bxi0 >> 2, byr = byr0 >> 2, byi = byi0 >> 2;
                lre = (int)((uint32_t)(bxr * byr) + (uint32_t)(bxi * byi));
                lim = (int)((uint32_t)(bxr * byi) - (uint32_t)(bxi * byr));
                l5  = (int)((uint32_t)(bxr * byr) - (uint32_t)(((-bxi0) >> 2) * byi));
            };
            const uint32_t p7 = (uint32_t)(pre[7] & 0xFFFF) | ((uint32_t)pim[7] << 16);
            // wave_shr:1: the previous word's last chip
            const uint32_t p7prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p7, 0x138, 0xF, 0xF, false);
            const int qre = lane == 0 ? last_re : (int)(short)(p7prev & 0xFFFFu), qim = lane == 0 ? last_im : (int)p7prev >> 16;
            const int dre = (int)((uint32_t)(qre * pre[7]) + (uint32_t)(qim * pim[7])) >> 1, dim = (int)((uint32_t)(qre * pim[7]) - (uint32_t)(qim * pre[7])) >> 1;
            // demap_dqpsk_bits (soradsp.h:190-198)
            const uint32_t dq = (((uint32_t)dre + (uint32_t)dim) >> 31) | ((((uint32_t)dre - (uint32_t)dim) >> 31) << 1);
            uint32_t raw; int nbytes;                                   // lane i: the i-th byte in front of the descrambler
            if (port == RATE_11M) {
                int Mm[4]; uint32_t Vm[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    int Ms[4]; uint32_t Vs[4];
#pragma unroll
                    for (int sq = 0; sq < 4; sq++) {
                        int lre, lim, l5; corr(m, sq, lre, lim, l5);
                        const int a1 = lre < 0 ? -lre : lre, a2 = lim < 0 ? -lim : lim;
                        const uint32_t base = sq == 0 ? 0x00u : sq == 1 ? 0x30u : sq == 2 ? 0x10u : 0x20u;
                        const bool first = a1 > a2;
                        Ms[sq] = first ? (lre > 0 ? lre : -lre) : (lim > 0 ? lim : -lim);
                        Vs[sq] = base | (first ? (lre > 0 ? 0x00u : 0x40u) : (lim > 0 ? 0xC0u : 0x80u));
                    }
                    // "if (Max1 > Max2) 1 else 2", "if (Max3 > Max4) 3 else 4"
                    const bool k01 = Ms[0] > Ms[1], k23 = Ms[2] > Ms[3];
                    const int M01 = k01 ? Ms[0] : Ms[1], M23 = k23 ? Ms[2] : Ms[3];
                    const uint32_t V01 = k01 ? Vs[0] : Vs[1], V23 = k23 ? Vs[2] : Vs[3];
                    const bool up = M23 > M01;                                                          // "if (Max34 > Module) Module = Max34"
                    Mm[m] = up ? M23 : M01; Vm[m] = (up ? V23 : V01) | (m == 0 ? 0x00u : m == 1 ? 0x08u : m == 2 ? 0x04u : 0x0Cu);
                }
                uint32_t out = Mm[0] > Mm[1] ? (Mm[0] > Mm[3] ? Vm[0] : Vm[3]) : (Mm[1] > Mm[2] ? Vm[1] : Vm[2]);   // "lable4"
                out |= dq;
                out ^= ((cck_even ^ (uint32_t)lane) & 1u) ? 3u : 0u;   // the extra pi of every other symbol
                raw = out & 0xFFu; nbytes = nw;
            } else {
                int lre, lim, l1, l2;
                corr(1, 0, lre, lim, l1); corr(3, 0, lre, lim, l2);     // phi2 = pi/2 and 3pi/2, phi3 = 0
                const int max1 = l1 > 0 ? l1 : -l1, max2 = l2 > 0 ? l2 : -l2;
                const uint32_t half = (max1 > max2 ? (l1 > 0 ? 0x0u : 0x8u) : (l2 > 0 ? 0x4u : 0xCu)) | dq;
                const uint32_t h0 = (uint32_t)__shfl((int)half, 2 * lane), h1 = (uint32_t)__shfl((int)half, 2 * lane + 1);
                raw = ((h0 | (h1 << 4)) ^ 0x30u) & 0xFFu; nbytes = nw >> 1;
            }
            // TDesc741 a byte per lane: the seven bits in front of it are the top of the byte to the left (byte_reg for the first)
            const uint32_t rprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x138, 0xF, 0xF, false);
            const uint32_t S15 = (raw << 7) | (lane == 0 ? (byte_reg & 0x7Fu) : (rprev >> 1));
            const uint32_t ob = ((S15 >> 7) ^ (S15 >> 3) ^ S15) & 0xFFu;
            if (nbytes > 0) {
                byte_reg = (uint32_t)lane_of((int)raw, nbytes - 1) >> 1;
                const uint32_t idx = byte_count + (uint32_t)lane;
                if (lane < nbytes && idx < kOutBuf) s_out[idx] = (uint8_t)ob;
                const uint32_t lim = (uint32_t)((int)frame_length - 4);
                const int n = byte_count >= lim ? 0 : (int)min((uint32_t)nbytes, lim - byte_count);     // bytes in front of the FCS
                // CRC-32 eight bytes at a time: every group's own contribution at once (byte i of a group of g through the table that carries it
                // over the g - 1 - i bytes behind it), then the register is carried over one group after the other
                const int glen = min(8, n - 8 * (lane >> 3));
                uint32_t v = lane < n ? s_crc_all[glen - 1 - (lane & 7)][ob] : 0u;
                v ^= (uint32_t)dpp<0xB1>((int)v); v ^= (uint32_t)dpp<0x4E>((int)v); v ^= (uint32_t)dpp<0x141>((int)v);
                for (int g = 0; 8 * g < n; g++) {
                    const int gl = min(8, n - 8 * g);
                    uint32_t z = lane < min(gl, 4) ? s_crc_all[gl - 1 - lane][(crc32 >> (8 * lane)) & 0xFFu] : 0u;
                    z ^= (uint32_t)dpp<0xB1>((int)z); z ^= (uint32_t)dpp<0x4E>((int)z);
                    crc32 = (uint32_t)lane_of((int)z, 0) ^ (gl < 4 ? crc32 >> (8 * gl) : 0u) ^ (uint32_t)lane_of((int)v, 8 * g);
                }
                byte_count += (uint32_t)nbytes;
                const int lw = nbytes * (need >> 3) - 1;                // the last word decoded
                last_re = lane_of(pre[7], lw); last_im = lane_of(pim[7], lw);
                if (port == RATE_11M) cck_even ^= (uint32_t)nbytes & 1u;
            }
            const int used = nbytes * need;
            lds_order();
            cbuf = lane < T - used ? chipbuf[used + lane] : 0u;
            cck_n = T - used;
            const uint32_t adv = 28u * (uint32_t)K;
            const uint32_t s0_ = c_start_;
            pos_ += adv; remain_ -= adv;
            c_start_ = s0_ + adv; c_stale_ = s0_ + adv - 28u; p_start_ = s0_ + adv - 28u; p_stale_ = s0_ + adv - 56u;
            return true;
        }
        }
        // ---- D. symbols: the carried partial sum, then every lane's parts (wrapping int16 sums: accumulated wide, wrapped when read)
        auto despread_sum = [&](int kc) __attribute__((always_inline)) {
            lds_order();
            sy_re[lane] = lane == 0 ? acc_re : 0; sy_im[lane] = lane == 0 ? acc_im : 0;
            lds_order();
            if (cnt > 0 && lane < kc) {
                atomicAdd(&sy_re[sid0], (int)(short)(partA & 0xFFFFu)); atomicAdd(&sy_im[sid0], (int)partA >> 16);
                if (crossed && sid0 + 1 < 64) { atomicAdd(&sy_re[sid0 + 1], (int)(short)(partB & 0xFFFFu)); atomicAdd(&sy_im[sid0 + 1], (int)partB >> 16); }
            }
            lds_order();
        };
        despread_sum(K);
        int Kc = K;                                                     // calls this pass commits
        int T = chip_n + total;
        int nsym = (int)(((uint32_t)T * 5958u) >> 16);                  // symbols completed in this pass
        int sre = w16(sy_re[lane]), sim = w16(sy_im[lane]);             // lane s: symbol s
        const int pre_i = lane > 0 ? lane - 1 : 0;
        int qre = w16(sy_re[pre_i]), qim = w16(sy_im[pre_i]);          // its left neighbour = the differential reference
        const unsigned long long below = (1ull << lane) - 1ull;
        if (mode == 0) {
            // ---- E0. TSFDSync (sfd_sync.hpp:76-126), a symbol per lane: DBPSK bit against the previous symbol, descrambled, the 16-bit window
            // compared with all-ones and with the SFD; "ones seen before" and the error count are prefix counts over lane masks
            if (lane == 0) { qre = last_re; qim = last_im; }
            const bool in = lane < nsym;
            const unsigned long long bits = __ballot(in && dot_sign(qre, qim, sre, sim));
            const unsigned long long S = (bits << 7) | (unsigned long long)(byte_reg & 0x7Fu);
            const unsigned long long O = (S >> 7) ^ (S >> 3) ^ S;
            const unsigned long long Bw = (unsigned long long)(word & 0xFFFFu) | (O << 16);      // word after symbol n = bits n + 1 .. n + 16
            const uint32_t wn = (uint32_t)(Bw >> (lane + 1)) & 0xFFFFu;
            const bool ones = wn == 0xFFFFu, sfd = wn == 0xF3A0u;
            const unsigned long long onesm = __ballot(in && ones);
            const bool found = bit_one_found != 0 || (onesm & below) != 0;
            const bool err = found && !sfd && !ones;
            const unsigned long long errm = __ballot(in && err);
            const bool fail = err && bit_err_cnt + (int)__popcll(errm & below) > 32;
            const bool late = sync_cnt + (uint32_t)lane + 1u > 128u + 16u;
            const unsigned long long evs = __ballot(in && ((found && sfd) || fail || late));
            if (evs != 0) {                                             // cut back to the calls in front of the one that completes the event's symbol
                const int gt = 11 * (__builtin_ctzll(evs) + 1) - 1;
                Kc = __builtin_ctzll(__ballot(act && g0 <= gt && gt < g0 + cnt));
                if (Kc == 0) { m_index = m_index0; m_frag = m_frag0; return false; }
                despread_sum(Kc);
                sre = w16(sy_re[lane]); sim = w16(sy_im[lane]);
                T = chip_n + lane_of(incl, Kc - 1); nsym = (int)(((uint32_t)T * 5958u) >> 16);
                m_index = (int)lane_of((int)miv2, Kc) - 2; m_frag = (int)lane_of((int)Mv, Kc) - 3;
            }
            const unsigned long long done = (1ull << nsym) - 1ull;      // (nsym <= 47)
            if (nsym > 0) { last_re = lane_of(sre, nsym - 1); last_im = lane_of(sim, nsym - 1); }
            byte_reg = (uint32_t)(S >> nsym) & 0x7Fu; word = (uint32_t)(Bw >> nsym) & 0xFFFFu;
            sync_cnt += (uint32_t)nsym;
            bit_one_found |= (onesm & done) != 0 ? 1 : 0;
            bit_err_cnt += (int)__popcll(errm & done);
        } else {
            // ---- E. bits (TDBPSKDemap / TDQPSKDemap, barkerspread.hpp:312-454) and bytes
            const int cin_re = sym_n == 0 ? last_re : ref_re, cin_im = sym_n == 0 ? last_im : ref_im;
            if (lane == 0) { qre = cin_re; qim = cin_im; }
            unsigned long long W; int nbits;
            if (port == RATE_1M) {
                const unsigned long long b0 = __ballot(lane < nsym && dot_sign(qre, qim, sre, sim));
                W = (unsigned long long)sym_byte | (b0 << sym_n); nbits = sym_n + nsym;
            } else {
                const int re = (int)((uint32_t)(qre * sre) + (uint32_t)(qim * sim)), im = (int)((uint32_t)(qre * sim) - (uint32_t)(qim * sre));
                unsigned long long b0 = __ballot(lane < nsym && ((((uint32_t)re + (uint32_t)im) >> 31) != 0));
                unsigned long long b1 = __ballot(lane < nsym && ((((uint32_t)re - (uint32_t)im) >> 31) != 0));
                auto spread = [](unsigned long long v) { v &= 0xFFFFFFFFull; v = (v | (v << 16)) & 0x0000FFFF0000FFFFull; v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
                                                         v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full; v = (v | (v << 2)) & 0x3333333333333333ull;
                                                             v = (v | (v << 1)) & 0x5555555555555555ull; return v; };
                W = (unsigned long long)sym_byte | ((spread(b0) | (spread(b1) << 1)) << (2 * sym_n)); nbits = 2 * (sym_n + nsym);
            }
            const int nbytes = nbits >> 3;
            // TDesc741 is self-synchronising -- out[n] = in[n] ^ in[n-4] ^ in[n-7] -- so the whole bit string is descrambled with two shifts
            const unsigned long long S = (W << 7) | (unsigned long long)(byte_reg & 0x7Fu);
            const unsigned long long O = (S >> 7) ^ (S >> 3) ^ S;
            if (nbytes > 0) {
                byte_reg = (uint32_t)(W >> (8 * nbytes - 7)) & 0x7Fu;
                // TBB11bPlcpSwitch -> TBB11bPlcpParser's burst (at most five bytes: the sixth is the event)
                if (mode == 1) {
                    for (int i = 0; i < nbytes; i++) {
                        const uint32_t sh = ((uint32_t)(O >> (8 * i)) & 0xFFu) << (8 * (hdr_n & 3));
                        hdr_lo |= hdr_n < 4 ? sh : 0u; hdr_hi |= hdr_n < 4 ? 0u : sh;
                        hdr_n++;
                    }
                } else {
                    const uint32_t ob = (uint32_t)(O >> (8 * (lane & 7))) & 0xFFu;  // lane i: the i-th byte for TBB11bFrameSink (at most 6 complete in a pass)
                    const uint32_t idx = byte_count + (uint32_t)lane;
                    if (lane < nbytes && idx < kOutBuf) s_out[idx] = (uint8_t)ob;
                    // CRC-32 of the bytes in front of the FCS, all at once: byte i of n goes through the table that also carries it over the
                    // n - 1 - i bytes behind it; the register's four bytes enter with the first four message bytes
                    const uint32_t lim = (uint32_t)((int)frame_length - 4);
                    const int n = byte_count >= lim ? 0 : (int)min((uint32_t)nbytes, lim - byte_count);
                    uint32_t v = 0;
                    if (lane < n) v = s_crc_all[n - 1 - lane][(ob ^ (lane < 4 ? crc32 >> (8 * lane) : 0u)) & 0xFFu];
                    v ^= (uint32_t)dpp<0xB1>((int)v); v ^= (uint32_t)dpp<0x4E>((int)v); v ^= (uint32_t)dpp<0x141>((int)v);      // xor over lanes 0..7
                    crc32 = (uint32_t)lane_of((int)v, 0) ^ (n < 4 ? crc32 >> (8 * n) : 0u);
                    byte_count += (uint32_t)nbytes;
                }
            }
            if (nsym > 0) {
                ref_re = lane_of(sre, nsym - 1); ref_im = lane_of(sim, nsym - 1);
                if (nbytes > 0) { const int ls = nbytes * spb - sym_n - 1; last_re = lane_of(sre, ls); last_im = lane_of(sim, ls); }
            }
            sym_n = sym_n + nsym - nbytes * spb; sym_byte = (uint32_t)(W >> (8 * nbytes)) & 0xFFu;
        }
        // ---- F. what the pass leaves behind
        chip_n = T - 11 * nsym;
        acc_re = w16(lane_of(sre, nsym)); acc_im = w16(lane_of(sim, nsym));
        const uint32_t adv = 28u * (uint32_t)Kc;
        const uint32_t s0_ = c_start_, st0_ = c_stale_;
        pos_ += adv; remain_ -= adv;
        c_start_ = s0_ + adv; c_stale_ = s0_ + adv - 28u; p_start_ = s0_ + adv - 28u; p_stale_ = Kc >= 2 ? s0_ + adv - 56u : st0_;
        return true;
    };

    uint32_t pos = 0, remain = cap_n;
    // TMemSamples appends 28 entries per call to its output queue; a call that finds fewer than 28 samples left (possible
    // after the Seek that follows a frame) leaves the tail of the previous call's burst in place (memsource.hpp:99-107).
    // Entry j of a call is therefore sample  j < take ? start + j : stale + j  of the capture; nothing is copied: the two
    // most recent calls are kept as (start, take, stale) and every consumer computes its own addresses.
    uint32_t c_start = 0, c_take = 28, c_stale = 0;                    // the current call
    uint32_t p_start = 0, p_take = 28, p_stale = 0;                    // the call before it (its last qoff entries may still be queued)
    int qoff = 0;                                                       // entries queued in front of TSymTiming (a multiple of 4, < 28)
    auto entry = [&](uint32_t start, uint32_t take, uint32_t stale,
            int j) __attribute__((always_inline)) { return (uint32_t)j < take ? start + (uint32_t)j : stale + (uint32_t)j; };
    // The next 28 samples are requested one source call ahead (the chain is latency-bound: a block's arithmetic must not
    // wait for its own HBM read).  pf_raw holds samples pf_base .. pf_base+27 when pf_ok; a Seek or a partial call simply misses.
    uint32_t pf_base = 0, pf_raw = lane < 28 && cap_n >= 28 ? x[lane] : 0u; bool pf_ok = cap_n >= 28;
    auto fetch28 = [&](uint32_t base, bool contiguous, uint32_t at) __attribute__((always_inline)) {
        const bool hit = contiguous && pf_ok && pf_base == base;        // wave-uniform
        const uint32_t v = hit ? pf_raw : (lane < 28 ? x[at] : 0u);
        pf_base = base + 28; pf_ok = contiguous && pf_base + 28 <= cap_n;
        if (pf_ok) pf_raw = lane < 28 ? x[pf_base + lane] : 0u;
        return v;
    };
    while (true) {                                                      // every frame is found and counted; rows / MPDUs only up to max_frames
        // All of the state is wave-uniform by construction, but the compiler's uniformity analysis loses that for the values
        // that live across the event handling below (3342 values of this kernel count as divergent without these lines, 614
        // with them -- the genuinely per-lane ones; tools/min_uniform_set.py found the smallest set that is needed).
#define U(v) v = (decltype(v))uni((int)v)
        U(last_re); U(last_im); U(byte_reg); U(frame_length); U(rate_kbps); U(frame_crc32); U(ref_re); U(ref_im); U(qoff); U(cck_n); U(cck_even);
#undef U
        // ---- TMemSamples::Process (memsource.hpp:87-114)
        const bool ret = remain != 0;
        if (ret) {
            p_start = c_start; p_take = c_take; p_stale = c_stale;
            c_stale = c_start; c_start = pos; c_take = remain > 28 ? 28u : remain;
            pos += c_take; remain -= c_take;
            if (!power) {
                // TDCRemove -> TBB11bRxSwitch -> TEnergyDetect -> TDCEstimator on the call's seven bursts (of four samples) at once: lane = sample.
                // The only thing that moves inside a call is the DC estimate, once at most (every 8th burst, after burst j0 = update_cnt): the
                // bursts are evaluated with the old and with the new estimate and each takes its own; window average, counter and threshold are
                // prefix sums over the bursts; what lies behind the burst that raises power (or gives up) is not committed.
                const cpx r = unpack(fetch28(c_start, c_take == 28, entry(c_start, c_take, c_stale, lane)));
                const int bq = lane >> 2;                               // this lane's burst
                // inclusive prefix over bursts of a per-burst value (same in a quad's lanes)
                auto burst_prefix = [&](int v) __attribute__((always_inline)) {
                    v = (int)((uint32_t)v + (uint32_t)dpp<0x114>(v)); v = (int)((uint32_t)v + (uint32_t)dpp<0x118>(v));     // row_shr:4, row_shr:8
                    // + row 0's total (its lane 15) in row 1
                    return (int)((uint32_t)v + (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));
                };
                const int j0 = (int)update_cnt;
                const int v1re = w16(r.re - dc_re), v1im = w16(r.im - dc_im);
                const int h1re = w16(quad_sum(v1re >> 5)), h1im = w16(quad_sum(v1im >> 5));      // TDCEstimator: hadd(shift_right(pi, 5)) in wrapping int16
                const int p1re = burst_prefix(h1re), p1im = burst_prefix(h1im);
                const int j0c = min(j0, 6);
                const int dcn_re = w16(dc_re + (w16(sdc_re + lane_of(p1re, 4 * j0c)) >> 2)), dcn_im = w16(dc_im + (w16(sdc_im + lane_of(p1im, 4 * j0c)) >> 2));
                const bool nw = bq > j0;                                // bursts behind the update see the new estimate
                const int vre = nw ? w16(r.re - dcn_re) : v1re, vim = nw ? w16(r.im - dcn_im) : v1im;
                const int q2re = w16(quad_sum(vre >> 5)), q2im = w16(quad_sum(vim >> 5));
                const int h2re = nw ? q2re : 0, h2im = nw ? q2im : 0;
                const int p2re = burst_prefix(h2re), p2im = burst_prefix(h2im);
                const uint32_t ave = (uint32_t)quad_sum((int)(uint32_t)(((int)((uint32_t)(vre * vre) + (uint32_t)(vim * vim))) >> 5));
                const uint32_t wold = lane < 8 ? win[lane] : 0u;        // the 8-entry window, oldest first
                const uint32_t wout = (uint32_t)__shfl((int)wold, bq);  // the entry burst bq pushes out (7 bursts never reach a new one)
                const uint32_t avg = avg_energy + (uint32_t)burst_prefix((int)(ave - wout));
                const uint32_t ecn = ecount + (uint32_t)bq + 1u;
                const unsigned long long evb = __ballot(lane < 28 && (lane & 3) == 0 && ecn >= 32u && (ecn >= 100u || avg >= thr));
                const int js = evb != 0 ? __builtin_ctzll(evb) >> 2 : 7; // the burst that ends carrier sensing, if any
                const int np = min(js + 1, 7), nd = min(js, 7);         // bursts through TEnergyDetect / through TDCEstimator
                avg_energy = (uint32_t)lane_of((int)avg, 4 * (np - 1));
                {                                                       // the window moves on by np entries
                    const uint32_t av8 = (uint32_t)__shfl((int)ave, 4 * (lane - 8));     // (a statement of its own: inside the select's arm it would run with
                    const uint32_t comb = lane < 8 ? wold : av8;                           //  lanes 0..7 switched off, and a permute reads 0 from such lanes)
                    const uint32_t wnew = (uint32_t)__shfl((int)comb, lane + np);
                    lds_order();
                    if (lane < 8) win[lane] = wnew;
                    lds_order();
                }
                if (js < 7) { if (ecount + (uint32_t)js + 1u >= 100u) error_code = E_CS_TIMEOUT; else power = 1; }
                ecount += (uint32_t)np;
                if (nd > 0) {
                    // the estimate was renewed after burst j0; bursts j0 + 1 .. nd - 1 went into the new sum
                    if (j0 < nd) {
                        dc_re = dcn_re; dc_im = dcn_im;
                        sdc_re = w16(lane_of(p2re, 4 * (nd - 1))); sdc_im = w16(lane_of(p2im, 4 * (nd - 1)));
                        update_cnt = (uint32_t)(8 - (nd - j0));
                    } else {
                        sdc_re = w16(sdc_re + lane_of(p1re, 4 * (nd - 1))); sdc_im = w16(sdc_im + lane_of(p1im, 4 * (nd - 1)));
                        update_cnt -= (uint32_t)nd;
                    }
                }
                const int first_queued = js + 1;
                if (power) qoff = 4 * (7 - first_queued);               // the tail of the call in which power came up is queued
            } else {                                                    // a whole call: the queue hands TSymTiming one block of 28
                const uint32_t at = lane < qoff ? entry(p_start, p_take, p_stale, 28 - qoff + lane) : entry(c_start, c_take, c_stale, lane - qoff);
                const bool contiguous = c_take == 28 && (qoff == 0 || (p_take == 28 && p_start + 28 == c_start));
                sym_timing(fetch28(c_start - (uint32_t)qoff, contiguous, at));
                // Steady state (aligned to the Barker code, no event pending, whole calls that follow one another): further source
                // calls are taken right here.  Same semantics as going round the outer loop -- MAC11b_Receive only looks at
                // error_code after a call -- but in a loop of its own the compiler keeps just this phase's state in registers.
                while (error_code == 0 && sync_flag == BARKER_SYNCED && remain >= 28 && c_take == 28 && c_start + 28 == pos) {
                    if ((uni(rxrate) <= RATE_2M || (CCK && uni(plcp_data))) && bulk_pass(pos, remain, c_start, c_stale, p_start, p_stale, qoff)) { p_take = 28;
                        pf_ok = false; continue; }
                    p_start = c_start; p_take = 28; p_stale = c_stale;
                    c_stale = c_start; c_start = pos; pos += 28; remain -= 28;
                    const uint32_t base = c_start - (uint32_t)qoff;
                    sym_timing(fetch28(base, true, base + (uint32_t)lane));
                }
            }
        }
        // ---- MAC11b_Receive bookkeeping after the source call (fb11b_demod.cpp:31-70)
        if (!CCK && uni(cck_abort)) { if (lane == 0) A.needs_cck[cap_i] = 1u; return; }
        if (error_code != 0) {
            const uint32_t err = error_code;
            if (err != E_CS_TIMEOUT) {
                if (lane == 0 && nfr < A.max_frames) {
                    Rx11bRow row; row.end_sample = pos; row.error_code = err; row.rate_kbps = rate_kbps; row.length = frame_length; row.crc32 = frame_crc32;
                    A.rows[(size_t)cap_i * A.max_frames + nfr] = row;
                }
                if ((err == E_FRAME_OK || err == E_CRC32_FAIL) && nfr < A.max_frames) {
                    uint8_t* dst = A.mpdu + ((size_t)cap_i * A.max_frames + nfr) * kOutBuf;
                    const uint32_t n = min(frame_length, kOutBuf);
                    lds_order();
                    for (uint32_t i = lane; i < n; i += 64) dst[i] = s_out[i];
                }
                nfr++;
            }
            if (err == E_FRAME_OK || err == E_CRC32_FAIL) {             // "jump advance of the last CRC byte": Seek (memsource.hpp:116-150)
                uint32_t off = rate_kbps == 1000 ? 8 * 11 * 4 : rate_kbps == 2000 ? 4 * 11 * 4 : rate_kbps == 5500 ? 8 * 2 * 4 : rate_kbps == 11000 ? 8 * 1 * 4 : 0;
                off = min(off, remain); pos += off; remain -= off;
            }
            // pRxSource->Flush(): what is queued is padded with zero samples and pushed through (brick.h FlushPort);
            // the switch flushes the branch its state selects, the rate selector pads its current port only
            if (power) {
                if (qoff > 0) {                                         // pad() fills with COMPLEX16() = 0: TDCRemove has already been applied upstream,
                    // so the padding must come out of sym_timing's subtraction as 0
                    const cpx q = unpack(lane < qoff ? x[entry(c_start, c_take, c_stale, 28 - qoff + lane)] : 0u);
                    const uint32_t padded = lane < qoff ? pack(mk(q.re, q.im)) : pack(mk(dc_re, dc_im));
                    sym_timing(padded); qoff = 0;
                }
                if (uni(rxrate) <= RATE_2M && chip_n > 0) { const int sre = acc_re, sim = acc_im; chip_n = 0; acc_re = acc_im = 0; symbol_out(uni(rxrate), sre, sim); }
                if constexpr (CCK) {
                if (uni(rxrate) > RATE_2M && cck_n > 0) {               // opin3/4().pad(): zero chips complete the burst, one more byte comes out
                    const uint32_t w = lane < cck_n ? cbuf : 0u; cck_n = 0;
                    byte_out(uni(rxrate) == RATE_5P5M ? cck5p5_decode(w) : cck11_decode(w));
                }
                }
            }
            graph_reset();
            continue;                                                   // the routine returns and is called again: rc is not looked at
        }
        if (!ret) break;
    }
    if (lane == 0) A.nframes[cap_i] = nfr;
}

__global__ void __launch_bounds__(256, 4) k_rx11b(Rx11bArgs A) { rx11b_capture<false>(A); }
__global__ void __launch_bounds__(256, 4) k_rx11b_cck(Rx11bArgs A) { rx11b_capture<true>(A); }

// how many captures the first pass handed over (the automatic pass plan's measurement): one block
__global__ void __launch_bounds__(1024) k_rx11b_count_flagged(const uint32_t* __restrict__ needs_cck, uint32_t ncaps, uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t n = 0;
    for (uint32_t i = threadIdx.x; i < ncaps; i += 1024) n += needs_cck[i] ? 1u : 0u;
    if (n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) out[0] = s_n;
}

}  // namespace sora

// ------------------------------------------------------------------------------------------------ host side (C ABI, include/sora_hip.h)
#include <vector>
#include <thread>
#include <string.h>
#include "../../include/sora_hip.h"

using namespace sora;

// A handle owns two slots (stream + result buffers), used in turn: process_dev waits only for the call before the previous one, so the last
// waves of call n and the first of call n + 1 share the chip (one wave per capture: a batch rarely fills the resident waves evenly).
// sora_rx11b_results reports the most recent call.
static constexpr int kSlots11b = 2;
struct Slot11b {
    hipStream_t stream = nullptr;
    CapDesc* d_caps = nullptr; Rx11bRow* d_rows = nullptr; uint32_t* d_nframes = nullptr; uint8_t* d_mpdu = nullptr; uint32_t* d_needs_cck = nullptr;
    std::vector<sora_capture_desc> h_caps;
    int ticket = 0;              // of the call this slot holds (0: none)
    hipEvent_t ev_done = nullptr; bool delivered = false, released = false;      // sora_rx11b_wait_any (kernels.h: slots_next / slots_poll)
    DenseStage dense;            // sora_rx11b_deliver_async
    std::vector<CapDesc> h_desc;                 // staging for the descriptor upload (kept until the slot's next call)
    uint32_t ncaps = 0;
    // automatic pass plan: how many captures the first pass handed to the CCK instantiation, counted on the device behind a two-pass call and
    // copied to page-locked memory (h_flagged[0] = count, valid once ev_flagged has completed)
    uint32_t* d_flagged = nullptr; uint32_t* h_flagged = nullptr; hipEvent_t ev_flagged = nullptr; bool flagged_pending = false;
};
struct sora_rx11b {
    sora_rx_cfg cfg{};
    Slot11b slot[kSlots11b]; int next = 0, last = 0, seq = 0;
    sora_complex16* d_iq_own = nullptr;
    const uint32_t* d_crc = nullptr;
    bool have_results = false;
    // sora_rx11b_set_single_pass: 0 = two passes, 1 = every capture straight through the CCK-capable instantiation, 2 (default) = automatic
    int  pass_plan = 2;
    bool auto_single = false;       // automatic plan: what the most recent measurement said (more than half of a call's captures carried CCK frames)
    uint32_t auto_calls = 0;        // ... and every 16th call of a single-pass run is a two-pass call again, to measure
};

#define HIPCHK11(call) do { hipError_t _e = (call); if (_e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, #call, (int)_e); } while (0)

static void rx11b_free(sora_rx11b_t* rx)
{
    if (!rx) return;
    for (Slot11b& S : rx->slot) {
        if (S.stream) { (void)hipStreamSynchronize(S.stream); (void)hipStreamDestroy(S.stream); }
        (void)hipFree(S.d_caps); (void)hipFree(S.d_rows); (void)hipFree(S.d_nframes); (void)hipFree(S.d_mpdu); (void)hipFree(S.d_needs_cck);
        (void)hipFree(S.d_flagged); if (S.h_flagged) (void)hipHostFree(S.h_flagged); if (S.ev_flagged) (void)hipEventDestroy(S.ev_flagged);
            if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        sora_internal_dense_free(&S.dense);
    }
    (void)hipFree(rx->d_iq_own);
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
    for (Slot11b& S : rx->slot) {
        if (e == hipSuccess) e = sora_internal_stream_create(&S.stream, (int)(&S - &rx->slot[0]));
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_caps, sizeof(CapDesc) * cfg->max_captures);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_rows, sizeof(Rx11bRow) * rows);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_nframes, 4 * (size_t)cfg->max_captures);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_needs_cck, 4 * (size_t)cfg->max_captures);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_flagged, 16);
        if (e == hipSuccess) e = hipHostMalloc((void**)&S.h_flagged, 16, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_flagged, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_mpdu, rows * 4096);
    }
    if (e != hipSuccess) { rx11b_free(rx); return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_rx11b_create: device allocation", (int)e); }
    *out = rx;
    return SORA_OK;
}

void* sora_rx11b_stream(sora_rx11b_t* rx) { return rx ? (void*)rx->slot[rx->last].stream : nullptr; }       // the stream of the most recent call
int sora_rx11b_synchronize(sora_rx11b_t* rx)
{
    if (!rx) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_synchronize: null handle", 0);
    HIPCHK11(hipSetDevice(rx->cfg.device));
    for (Slot11b& S : rx->slot) HIPCHK11(hipStreamSynchronize(S.stream));
    return SORA_OK;
}

void sora_rx11b_destroy(sora_rx11b_t* rx) { if (rx) { (void)hipSetDevice(rx->cfg.device); rx11b_free(rx); } }

int sora_rx11b_process_dev(sora_rx11b_t* rx, const sora_complex16* d_iq, const sora_capture_desc* caps, size_t ncaps)
{
    if (!rx || (ncaps && (!d_iq || !caps))) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_process_dev: null argument", 0);
    if (ncaps > rx->cfg.max_captures) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11b_process_dev: more captures than max_captures", 0);
    HIPCHK11(hipSetDevice(rx->cfg.device));
    rx->next = slots_next(rx->slot, kSlots11b);                                       // an unused slot, else a released call's, else the oldest call's
    Slot11b& S = rx->slot[rx->next];
    std::vector<CapDesc>& h = S.h_desc;
    HIPCHK11(hipStreamSynchronize(S.stream));                                         // the slot's previous call may still be reading d_caps / writing results
    h.resize(ncaps);
    uint64_t total = 0;
    for (size_t i = 0; i < ncaps; i++) {
        if (caps[i].offset % 4 != 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "capture offset must be a multiple of 4 samples", 0);
        if (caps[i].nsamples % 28 != 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "capture length must be a whole number of 28-sample source bursts", 0);
        h[i].offset = caps[i].offset; h[i].nsamples = caps[i].nsamples; h[i].capture_id = caps[i].capture_id; h[i].slot_base = 0; h[i].nslots = 0;
        total += caps[i].nsamples;
    }
    if (total > rx->cfg.max_total_samples) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11b_process_dev: more samples than max_total_samples", 0);
    S.h_caps.assign(caps, caps + ncaps); S.ncaps = (uint32_t)ncaps; rx->have_results = true;
    rx->last = rx->next;
    S.ticket = ++rx->seq; S.delivered = S.released = false;
    if (ncaps == 0) return SORA_OK;
    HIPCHK11(hipMemcpyAsync(S.d_caps, h.data(), sizeof(CapDesc) * ncaps, hipMemcpyHostToDevice, S.stream));
    Rx11bArgs A;
    A.iq = reinterpret_cast<const uint32_t*>(d_iq); A.caps = S.d_caps; A.ncaps = (uint32_t)ncaps; A.thr = rx->cfg.cca_pwr_threshold;
    A.max_frames = rx->cfg.max_frames_per_capture; A.rows = S.d_rows; A.nframes = S.d_nframes; A.mpdu = S.d_mpdu; A.crc = rx->d_crc; A.needs_cck = S.d_needs_cck;
#ifdef SORA_VARIANT_11B_ONE_KERNEL                                             // build variant (sora_amd.build.build_variant): every capture through the CCK instantiation
    constexpr bool one_kernel = true;
#else
    constexpr bool one_kernel = false;
#endif
    // The pass plan.  Automatic (default): a two-pass call also counts, on the device, how many captures its first pass handed over; once such a
    // count has come back (no waiting: the event is only queried) and says "more than half", the following calls go straight through the CCK
    // instantiation -- which decodes all four rates with identical rows -- except every 16th, which is a two-pass call again and measures.
    for (Slot11b& Q : rx->slot) {
        if (!Q.flagged_pending) continue;
        const hipError_t qe = hipEventQuery(Q.ev_flagged);
        if (qe == hipSuccess) { Q.flagged_pending = false; rx->auto_single = 2u * Q.h_flagged[0] > Q.h_flagged[1]; }
        // (hipErrorNotReady is sticky for hipGetLastError: the check at the end of this call must not see it -- ADVICE r4)
        else (void)hipGetLastError();
    }
    bool single = one_kernel || rx->pass_plan == 1;
    if (rx->pass_plan == 2 && rx->auto_single && (++rx->auto_calls & 15u) != 0u) single = true;
    HIPCHK11(hipMemsetAsync(S.d_needs_cck, single ? 1 : 0, 4 * ncaps, S.stream));
    if (!single) hipLaunchKernelGGL(k_rx11b, dim3((unsigned)((ncaps + 3) / 4)), dim3(256), 0, S.stream, A);
    if (!single && rx->pass_plan == 2 && !S.flagged_pending) {
        hipLaunchKernelGGL(k_rx11b_count_flagged, dim3(1), dim3(1024), 0, S.stream, (const uint32_t*)S.d_needs_cck, (uint32_t)ncaps, S.d_flagged);
        S.h_flagged[1] = (uint32_t)ncaps;
        HIPCHK11(hipMemcpyAsync(S.h_flagged, S.d_flagged, 4, hipMemcpyDeviceToHost, S.stream));
        HIPCHK11(hipEventRecord(S.ev_flagged, S.stream));
        S.flagged_pending = true;
    }
    // redoes the captures the first pass flagged (a wave of any other capture returns at once)
    hipLaunchKernelGGL(k_rx11b_cck, dim3((unsigned)((ncaps + 3) / 4)), dim3(256), 0, S.stream, A);
    HIPCHK11(hipGetLastError());
    return SORA_OK;
}

int sora_rx11b_process(sora_rx11b_t* rx, const sora_complex16* h_iq, size_t nsamples, const sora_capture_desc* caps, size_t ncaps)
{
    if (!rx || (nsamples && !h_iq)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_process: null argument", 0);
    if (nsamples > rx->cfg.max_total_samples) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11b_process: more samples than max_total_samples", 0);
    for (size_t i = 0; i < ncaps; i++)                                               // the buffer's size is known here: no descriptor may reach past it
        if (caps && (caps[i].offset > nsamples || caps[i].nsamples > nsamples - caps[i].offset)) return sora_internal_fail(SORA_ERR_INVALID_PARAM,
                "a capture descriptor reaches past the end of the sample buffer", 0);
    HIPCHK11(hipSetDevice(rx->cfg.device));
    if (!rx->d_iq_own) HIPCHK11(hipMalloc((void**)&rx->d_iq_own, sizeof(sora_complex16) * (rx->cfg.max_total_samples + 64)));
    for (Slot11b& S : rx->slot) HIPCHK11(hipStreamSynchronize(S.stream));             // one upload buffer: no call may still be reading it
    rx->next = slots_next(rx->slot, kSlots11b);                                       // (the slot process_dev is about to pick: nothing changes in between)
    HIPCHK11(hipMemcpyAsync(rx->d_iq_own, h_iq, sizeof(sora_complex16) * nsamples, hipMemcpyHostToDevice, rx->slot[rx->next].stream));
    return sora_rx11b_process_dev(rx, rx->d_iq_own, caps, ncaps);
}

static int slot11b_results(sora_rx11b_t* rx, Slot11b& S, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (S.ncaps == 0) return SORA_OK;
    HIPCHK11(hipSetDevice(rx->cfg.device));
    HIPCHK11(hipStreamSynchronize(S.stream));
    const uint32_t mf = rx->cfg.max_frames_per_capture;
    std::vector<Rx11bRow> rows((size_t)S.ncaps * mf); std::vector<uint32_t> nfr(S.ncaps);
    HIPCHK11(hipMemcpy(rows.data(), S.d_rows, sizeof(Rx11bRow) * rows.size(), hipMemcpyDeviceToHost));
    HIPCHK11(hipMemcpy(nfr.data(), S.d_nframes, 4 * (size_t)S.ncaps, hipMemcpyDeviceToHost));
    // MPDU bytes: one bulk copy of the per-frame slots that are in use when that is cheap, else frame by frame
    size_t used_rows = 0;
    for (uint32_t c = 0; c < S.ncaps; c++) used_rows += nfr[c] < mf ? nfr[c] : mf;
    std::vector<uint8_t> bulk;
    const size_t slots = (size_t)S.ncaps * mf;
    if (h_mpdu && used_rows > 16 && slots * 4096 <= ((size_t)1 << 30)) {
        bulk.resize(slots * 4096);
        HIPCHK11(hipMemcpy(bulk.data(), S.d_mpdu, bulk.size(), hipMemcpyDeviceToHost));
    }
    size_t n = 0, moff = 0; int rc = SORA_OK;
    for (uint32_t c = 0; c < S.ncaps; c++)
        for (uint32_t i = 0; i < nfr[c] && i < mf; i++) {
            const Rx11bRow& r = rows[(size_t)c * mf + i];
            if (n >= max_out) { rc = SORA_ERR_CAPACITY; continue; }
            sora_frame_result& o = out[n++];
            memset(&o, 0, sizeof(o));
            o.capture_id = S.h_caps[c].capture_id; o.end_sample = r.end_sample; o.error_code = r.error_code; o.rate_kbps = r.rate_kbps;
            o.length = (uint16_t)r.length; o.crc32 = r.crc32; o.mpdu_offset = (uint32_t)moff;
            if (i + 1 == mf && nfr[c] > mf) o.flags = SORA_ROW_TRUNCATED;             // more frames were found than the capture has rows
            if (h_mpdu && (r.error_code == 1u || r.error_code == 0x80000006u)) {
                const size_t len = r.length < 4096 ? r.length : 4096;
                if (moff + len > mpdu_cap) { rc = SORA_ERR_CAPACITY; continue; }
                if (!bulk.empty()) memcpy(h_mpdu + moff, bulk.data() + ((size_t)c * mf + i) * 4096, len);
                else HIPCHK11(hipMemcpy(h_mpdu + moff, S.d_mpdu + ((size_t)c * mf + i) * 4096, len, hipMemcpyDeviceToHost));
                moff += len;
            }
        }
    *nout = n;
    if (rc != SORA_OK) return sora_internal_fail(rc, "sora_rx11b_results: output buffer too small", 0);
    return SORA_OK;
}

int sora_rx11b_results(sora_rx11b_t* rx, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (!rx || !nout) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_results: null argument", 0);
    *nout = 0;
    if (!rx->have_results) return sora_internal_fail(SORA_ERR_FAILED, "no process call to report", 0);
    return slot11b_results(rx, rx->slot[rx->last], out, max_out, nout, h_mpdu, mpdu_cap);
}

// Tickets (as sora_rx_ticket / _wait / _results_of): every process call is addressable until kSlots11b further calls have reused its slot.
static Slot11b* slot11b_of(sora_rx11b_t* rx, int ticket)
{
    if (!rx || ticket <= 0) return nullptr;
    for (Slot11b& S : rx->slot) if (S.ticket == ticket) return &S;
    return nullptr;
}
static const char* const kStale11b = "stale ticket: its slot has been reused by a later process call (or the ticket was never issued)";
int sora_rx11b_ticket(sora_rx11b_t* rx) { return rx && rx->have_results ? rx->slot[rx->last].ticket : 0; }
int sora_rx11b_calls_in_flight(sora_rx11b_t* rx) { (void)rx; return kSlots11b; }
// Two passes (default): k_rx11b (90 VGPRs, the Barker rates) decodes every capture and hands the ones whose PLCP header announces 5.5 / 11 Mbps
// to k_rx11b_cck (128 VGPRs), which redoes them from their first sample.  A host that expects mostly CCK traffic skips the first pass:
// every capture goes straight through the CCK-capable instantiation (it decodes all four rates; identical rows).  Returns the previous setting.
int sora_rx11b_set_single_pass(sora_rx11b_t* rx, int enable)
{
    if (!rx) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_set_single_pass: null handle", 0);
    const int old = rx->pass_plan;
    if (enable >= 0) { rx->pass_plan = enable > 2 ? 2 : enable; rx->auto_single = false; rx->auto_calls = 0; }
    return old;
}
int sora_rx11b_wait(sora_rx11b_t* rx, int ticket)
{
    Slot11b* S = slot11b_of(rx, ticket);
    if (!S) return sora_internal_fail(SORA_ERR_INVALID_PARAM, kStale11b, 0);
    HIPCHK11(hipSetDevice(rx->cfg.device));
    HIPCHK11(hipStreamSynchronize(S->stream));
    if (S->delivered) S->released = true;
    return SORA_OK;
}
int sora_rx11b_wait_any(sora_rx11b_t* rx, int* ticket)
{
    if (!rx || !ticket) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_wait_any: null argument", 0);
    *ticket = 0;
    HIPCHK11(hipSetDevice(rx->cfg.device));
    for (unsigned spin = 0;; spin++) {
        bool pending; hipError_t err;
        Slot11b* S = slots_poll(rx->slot, kSlots11b, &pending, &err);
        if (err != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_rx11b_wait_any: hipEventQuery", (int)err);
        if (S) { const int t = S->ticket; const int rc = sora_rx11b_wait(rx, t); if (rc == SORA_OK) *ticket = t; return rc; }
        if (!pending) return sora_internal_fail(SORA_ERR_FAILED, "sora_rx11b_wait_any: no call with an enqueued delivery (sora_rx11b_deliver_async) is in flight", 0);
        if (spin > 64) std::this_thread::yield();
    }
}
void* sora_rx11b_stream_of(sora_rx11b_t* rx, int ticket) { Slot11b* S = slot11b_of(rx, ticket); return S ? (void*)S->stream : nullptr; }
int sora_rx11b_deliver_async(sora_rx11b_t* rx, int ticket, sora_frame_result* h_rows, size_t max_rows, uint32_t* h_counts, uint8_t* h_mpdu, size_t mpdu_cap)
{
    Slot11b* S = slot11b_of(rx, ticket);
    if (!S) return sora_internal_fail(SORA_ERR_INVALID_PARAM, kStale11b, 0);
    HIPCHK11(hipSetDevice(rx->cfg.device));
    const int rc = sora_internal_dense_deliver(&S->dense, S->d_rows, S->d_nframes, S->d_caps, nullptr, S->ncaps, rx->cfg.max_frames_per_capture, S->d_mpdu, S->stream,
                                               h_rows, max_rows, h_counts, h_mpdu, mpdu_cap);
    if (rc != SORA_OK) return rc;
    HIPCHK11(slots_mark_delivered(*S));
    return SORA_OK;
}

int sora_rx11b_results_of(sora_rx11b_t* rx, int ticket, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (!nout) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11b_results_of: null argument", 0);
    *nout = 0;
    Slot11b* S = slot11b_of(rx, ticket);
    if (!S) return sora_internal_fail(SORA_ERR_INVALID_PARAM, kStale11b, 0);
    return slot11b_results(rx, *S, out, max_out, nout, h_mpdu, mpdu_cap);
}
