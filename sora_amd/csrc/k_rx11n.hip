// k_rx11n.hip -- the reference's 802.11n 2x2 receive graph over a batch of independent two-chain captures:
// CreateDemodGraph11n (kernel/bb/demod11/fb11ndemod_config.hpp:166-257) driven as RxThread drives it (fb11n_demod.cpp:30-85).
//
//   TMemSamples2 -> TDownSample2 -> RxSwitch -> TCCA11n (MimoAutoCorr)                                   carrier sense
//                                            -> TFreqEstimator_11n -> TFreqComp_11n -> TFFT64 x4 -> TSisoChannelEst    L-LTF
//                                            -> TFreqComp_11n -> T11nDataSymbol -> TFFT64 x2 -> T11nSymSel
//       SIG:    TSisoChannelComp -> TMrcCombine -> T11nSigDemap -> T11aDeinterleaveBPSK -> T11nViterbiSig -> T11nSigParser
//       HT-LTF: TMimoChannelEst
//       DATA:   TMimoChannelComp -> TPilotTrack_11n -> T11nDemap* -> T11nDeinterleave*_S0/_S1 -> TStreamJoin -> TStreamConcat<2,1>
//               -> T11aViterbi<5000*8, 312, 192, 36> -> T11aDesc -> TBB11aFrameSink
//
// One wave per capture (four per workgroup); all control flow is wave-uniform, the 64 lanes are the 64 samples / carriers / trellis
// states of the step at hand.  The graph's queues reduce to positions in the 20 MHz stream:
//   * a source call brings 14 samples; a frame event is seen when the call that completed its last burst returns, the queues are
//     cleared and the stream restarts at the next call boundary;
//   * carrier sense runs in blocks of 64 samples: the moving sums are prefix sums over the lanes on top of the rings MimoAutoCorr
//     keeps, the peak counter walks the two ballots of its conditions;
//   * TFreqComp_11n's running phase is n * CFO - theta (mod 2^16) for the n-th sample after detection;
//   * samples past the end of the capture read as zero: that is the flush TMemSamples2 issues when it runs dry, which pads every
//     partly filled queue with zero items (see oracle/so_rx11n.c, flush_graph).
// HBM traffic = the samples, once (8 bytes per 40 MHz sample pair of the two chains, of which the even half is used).
#include "kernels.h"
#include "dev_winplan.h"
#include "dev_11n.h"
#include "../../include/sora_hip.h"
#include <type_traits>

namespace sora {

struct Rx11nArgs {
    const uint32_t* iq0; const uint32_t* iq1;    // packed COMPLEX16 @40 MHz, RX chain 0 / 1
    const CapDesc*  caps; uint32_t ncaps, max_frames;
    Rx11bRow*       rows;                          // [ncaps * max_frames]: end_sample = 40 MHz source position, rate_kbps = MCS index
    uint32_t*       nframes;                       // [ncaps]
    uint8_t*        mpdu;                          // [ncaps * max_frames][4096]
    Tables          T;
    const uint32_t* sincos; const short* atan;     // dsp_math tables
};

namespace {
constexpr uint32_t E_OK = 1u, E_PLCP = 0x80000005u, E_CRC = 0x80000006u;
enum { SYM_SIG = 2, SYM_HT_STF, SYM_HT_LTF, SYM_DATA };

struct WaveLds {
    uint32_t his[2][32]; int hcr[2][32], hci[2][32], he[2][32];     // MimoAutoCorr rings
    long long his_e[64];                                            // TCCA11n::his_moving_energy
    uint32_t buf[2][128];                                           // compensated samples in front of the FFTs
    uint32_t fft[4][64];                                            // FFT staging, one slice per 16-lane group
    uint32_t y[2][128];                                             // FFT output per chain (L-LTF: both halves; HT-LTF: both symbols)
    uint32_t ch[2][64]; uint32_t hinv[4][64];
    uint32_t sig[192];
    uint32_t xs[2][64];                                             // spatial streams after TMimoChannelComp
    uint8_t  soft[2][160];                                          // demapped soft values per stream (<= 104); [0] doubles as SIG scratch
    alignas(4) uint8_t joined[256];                                           // stream-parsed, de-interleaved soft values of one symbol
    uint8_t  sigsoft[144];
    uint8_t  dtab[208];                                             // joined position g (stream g & 1) <- soft[g & 1][dtab[g]], for this frame's N_BPSC
    unsigned long long dec[256];                                    // decision words of the last 256 trellis columns
    uint8_t  out[1536];                                             // decoded bytes (service field first)
};

__device__ __forceinline__ void wsync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }            // a value every lane holds alike -> SGPR
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ unsigned wave_min(unsigned v)                // minimum over the 64 lanes, in every lane: four DPP moves, two lane swaps
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));       // lane ^ 1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));       // lane ^ 2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true));      // row_ror:4
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true));      // row_ror:8: each row of 16 is done
    { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = min(r[0], r[1]); }
    { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = min(r[0], r[1]); }
    return v;
}
__device__ __forceinline__ int scan_add(int v, int)                    // inclusive prefix sum over the wave, wrapping: six DPP adds, no LDS round trips
{
    // inside each row of 16: row_shr:1, 2, 4, 8 (lanes shifted in read 0); then lane 15 of row r - 1 into rows 1 and 3, lane 31 into rows 2 and 3
    v = (int)((unsigned)v + (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true));
    v = (int)((unsigned)v + (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true));
    v = (int)((unsigned)v + (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true));
    v = (int)((unsigned)v + (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true));
    v = (int)((unsigned)v + (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));   // row_bcast:15, rows 1 and 3
    v = (int)((unsigned)v + (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));   // row_bcast:31, rows 2 and 3
    return v;
}
}  // namespace

#ifdef SORA_VARIANT_11N_MONO            // build variant only (sora_amd.build.build_variant("mono11n", ["SORA_VARIANT_11N_MONO"])): the round-1 one-kernel form, an A/B partner
__global__ void __launch_bounds__(256, 3) k_rx11n_mono(Rx11nArgs A)
{
    __shared__ WaveLds s_w[4];
    __shared__ uint8_t s_lut[6][256];
    __shared__ uint32_t s_crc[256];
    __shared__ uint32_t s_z[6 * 8 * 16];
    fill_demap_luts(s_lut);
    s_crc[threadIdx.x] = A.T.crc[threadIdx.x];
    for (int i = threadIdx.x; i < 6 * 8 * 16; i += 256) s_z[i] = A.T.crcz[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // wave-uniform by construction: keep it in an SGPR
    const uint32_t cap = blockIdx.x * 4 + wv;
    if (cap >= A.ncaps) return;                                              // whole waves leave; no block barrier below
    WaveLds& W = s_w[wv];
    const CapDesc cd = A.caps[cap];
    const uint32_t* iq[2] = { A.iq0 + cd.offset, A.iq1 + cd.offset };
    const uint32_t n20 = cd.nsamples / 2;                                    // 20 MHz samples (TDownSample2 keeps the even ones)
    auto fetch = [&](int r, uint32_t i) __attribute__((always_inline)) -> uint32_t { return i < n20 ? iq[r][2 * (size_t)i] : 0u; };
    const Fft64Tw tw = fft64_twiddles(A.T, lane & 15);
    auto nosync = []() __attribute__((always_inline)) { wsync(); };

    // MimoAutoCorr / TCCA11n state: lives for the whole capture (only the peak counter is reset between frames)
    for (int k = lane; k < 64; k += 64) { W.his[0][k & 31] = 0; W.his[1][k & 31] = 0; W.hcr[k >> 5][k & 31] = 0; W.hci[k >> 5][k & 31] = 0;
        W.he[k >> 5][k & 31] = 0; W.his_e[k] = 0x7FFFFFFFFFFFFFFFll; }
    wsync();
    int sr[2] = { 0, 0 }, si[2] = { 0, 0 }, se[2] = { 0, 0 };                // running sums
    int ring_pos = 0, his_index = 0;
    uint32_t origin = 0, nfr = 0;                                            // stream origin (20 MHz index), frames reported
    Rx11bRow* rows = A.rows + (size_t)cap * A.max_frames;

    while (origin < n20) {
        // ================================================================ carrier sense from `origin`
        const uint32_t nb_total = (n20 - origin + 3) / 4;                    // bursts TDownSample2 will deliver (the last one zero-padded)
        bool pf = false, timeout = false; int pc = 0, sense = 0;             // peak_found, peak_count, sense_count (_reset)
        int64_t det_at = -1;                                                 // sample (relative to origin) at which power was detected
        for (uint32_t base = 0; base < nb_total * 4 && det_at < 0; base += 64) {
            const int lim = (int)min(64u, nb_total * 4 - base);
            int pr[2], pi[2], pe[2], cre[2], cim[2], een[2]; uint32_t xr[2];
            const int slot = (ring_pos + lane) & 31;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                xr[r] = fetch(r, origin + base + lane);
                const cpx x = unpack(xr[r]);
                const uint32_t dl = (uint32_t)__shfl((int)xr[r], lane - 32);
                const cpx delayed = unpack(lane < 32 ? W.his[r][slot] : dl);
                int re, im; conj_mul32(x, delayed, re, im); re >>= 5; im >>= 5;
                const int ore = __shfl(re, lane - 32), oim = __shfl(im, lane - 32);
                const int e = sqnorm(x) >> 5, oe = __shfl(e, lane - 32);
                const int dre = (int)((unsigned)re - (unsigned)(lane < 32 ? W.hcr[r][slot] : ore));
                const int dim = (int)((unsigned)im - (unsigned)(lane < 32 ? W.hci[r][slot] : oim));
                const int den = (int)((unsigned)e - (unsigned)(lane < 32 ? W.he[r][slot] : oe));
                pr[r] = (int)((unsigned)sr[r] + (unsigned)scan_add(dre, lane)); pi[r] = (int)((unsigned)si[r] + (unsigned)scan_add(dim, lane));
                pe[r] = (int)((unsigned)se[r] + (unsigned)scan_add(den, lane));
                cre[r] = re; cim[r] = im; een[r] = e;
            }
            const int are = (int)((unsigned)(pr[0] >> 1) + (unsigned)(pr[1] >> 1)), aim = (int)((unsigned)(pi[0] >> 1) + (unsigned)(pi[1] >> 1));
            const long long acorr = (long long)((unsigned long long)((long long)are * are) + (unsigned long long)((long long)aim * aim));
            const int ev = (int)((unsigned)(pe[0] >> 1) + (unsigned)(pe[1] >> 1));
            const long long energy = (long long)ev * ev;
            const long long olde = W.his_e[(his_index + lane) & 63];
            // eb = energy / (olde + 1) > 5  <=>  olde + 1 <= energy / 6   (olde = LLONG_MAX: the sum wraps negative and eb is 0)
            // (den <= energy / 6  <=>  6 den <= energy; the first test keeps 6 den inside 63 bits)
            const bool cA = olde != 0x7FFFFFFFFFFFFFFFll && (olde + 1) <= (energy >> 2) && 6 * (olde + 1) <= energy && acorr > (energy >> 1);
            const bool cB = acorr < (energy >> 3);
            const unsigned long long bA = __ballot(cA), bB = __ballot(cB);
            int det = -1;
            const unsigned long long lmask = lim >= 64 ? ~0ull : ((1ull << lim) - 1);
            if (!pf && (bA & lmask) == 0) {
                // idle block: no sample starts a plateau; only the timeout bookkeeping moves, burst by burst
                for (int i = 3; i < lim; i += 4) {
                    sense += 4;
                    if (sense >= 84) timeout = true;
                    const uint32_t s4 = base + (uint32_t)i - 3;
                    if (timeout && (s4 + 3) / 14 != (s4 + 7) / 14) { timeout = false; sense = 0; }
                }
                pc = 0;
            } else if (pf && !timeout && (bB & lmask) == 0 && pc + lim <= 160) {
                pc += lim;                                                   // inside a plateau: every sample counts, nothing else changes
            } else
            for (int i = 0; i < lim; i++) {                                  // cca_11n.hpp:46-121, on the two ballots
                const bool a = (bA >> i) & 1, b = (bB >> i) & 1;
                if (!pf) { sense++; if (a) { sense = 0; pc++; pf = true; } else pc = 0; }
                else if (b) { const bool good = pc > 96 && pc < 160; pf = false; pc = 0; if (good) { det = i; break; } }
                else { pc++; if (pc > 160) { pf = false; pc = 0; } }
                if ((i & 3) == 3) {
                    // end of a burst: the carrier-sense timeout is raised here (cca_11n.hpp:124-127) and acted on by RxThread when the
                    // source call returns (ResetCarrierSense + scs->Reset, fb11n_demod.cpp:44-52) -- which clears the peak counter
                    // even if a plateau has begun in the bursts between
                    if (sense >= 84) timeout = true;
                    const uint32_t s4 = base + (uint32_t)i - 3;             // first sample of this burst, relative to origin
                    if (timeout && (s4 + 3) / 14 != (s4 + 7) / 14) { timeout = false; pf = false; pc = 0; sense = 0; }
                }
            }
            const int ne = det >= 0 ? det : lim;                             // samples recorded in his_moving_energy
            const int na = det >= 0 ? (det | 3) + 1 : lim;                   // samples MimoAutoCorr has taken (whole bursts)
            if (lane < ne) W.his_e[(his_index + lane) & 63] = energy;
            if (lane < na && lane >= na - 32) {
#pragma unroll
                for (int r = 0; r < 2; r++) { W.his[r][slot] = xr[r]; W.hcr[r][slot] = cre[r]; W.hci[r][slot] = cim[r]; W.he[r][slot] = een[r]; }
            }
#pragma unroll
            for (int r = 0; r < 2; r++) { sr[r] = __shfl(pr[r], na - 1); si[r] = __shfl(pi[r], na - 1); se[r] = __shfl(pe[r], na - 1); }
            his_index = (his_index + ne) & 63; ring_pos = (ring_pos + na) & 31;
            wsync();
            if (det >= 0) det_at = (int64_t)base + na;                       // first sample behind the detecting burst
        }
        if (det_at < 0) break;                                               // nothing (more) in this capture
        const uint32_t n_real = n20 - origin;                                // real samples of this segment
        const uint32_t l0 = (uint32_t)det_at;                                // L-LTF start, relative to origin
        if (l0 + 128 > ((n_real + 3) & ~3u)) break;                          // the L-LTF queue never fills: its flush is not an event
        // ================================================================ L-LTF: CFO, compensation, four FFTs, SISO channel
        int cfo;
        {
            int sre = 0, sim = 0;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                int re, im; conj_mul32(unpack(fetch(r, origin + l0 + lane)), unpack(fetch(r, origin + l0 + 64 + lane)), re, im);
                sre += re >> 7; sim += im >> 7;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { sre += __shfl_xor(sre, d); sim += __shfl_xor(sim, d); }
            cfo = uni(dsp_atan32(A.atan, sre, sim) >> 6);
        }
        int theta = 0;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int n = 64 * h + lane;
                const cpx cof = unpack(A.sincos[(unsigned)(n * cfo) & 0xFFFFu]);
                int re, im; mul32(unpack(fetch(r, origin + l0 + n)), cof, re, im);
                W.buf[r][n] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
            }
        wsync();
        {   // group g = 2 r + half
            const int g = lane >> 4, e = lane & 15; cpx x[4], yy[4];
#pragma unroll
            for (int m = 0; m < 4; m++) x[m] = unpack(W.buf[g >> 1][64 * (g & 1) + e + 16 * m]);
            fft64_group(x, yy, W.fft[g], e, tw, nosync);
#pragma unroll
            for (int q = 0; q < 4; q++) W.y[g >> 1][64 * (g & 1) + e + 16 * q] = pack(yy[q]);
        }
        wsync();
#pragma unroll
        for (int r = 0; r < 2; r++) {
            uint32_t o = 0;
            if (lane < 28 || lane >= 36) {
                const uint32_t* l = W.y[r] + (lane & ~3);
                const cpx a = siso_one(l, lane & 3, lane), b = siso_one(l + 64, lane & 3, lane);
                o = pack(mk((short)((short)(a.re + b.re) >> 1), (short)((short)(a.im + b.im) >> 1)));
            }
            W.ch[r][lane] = o;
        }
        wsync();
        // ================================================================ symbols
        int type = SYM_SIG, nsig = 0, nltf = 0;
        uint32_t err = 0, mcs = 0, ht_len = 0, code_rate = 0, frame_crc = 0;
        unsigned m = (lane == 0) ? 0u : 0x30u;                               // Viterbi metrics, lane = state
        uint32_t tr = 0, ob = 0, nout = 0, soft_n = 0, tr_end = 0;
        if (lane == 0) W.dec[0] = 0;
        // The data Viterbi keeps a ROTATING state-to-lane map: at trellis column c lane l holds state rotl6(l, c mod 6).  The two
        // predecessors j and j + 32 of states 2j and 2j + 1 then sit in lanes that differ in ONE lane bit (5 - c mod 6), so a step needs a
        // single exchange -- v_permlane32_swap, v_permlane16_swap or a DPP move -- instead of two ds_bpermute round trips.
        int ph = 0;                                                          // tr mod 6
        unsigned pbits = 0;                                                  // per column phase q: expected code bits (A0, B0, A1, B1) of this lane's state
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const unsigned n = (((unsigned)lane << q) | ((unsigned)lane >> (6 - q))) & 63u;
            pbits |= (unsigned)((__popc(n & 0155) & 1) | ((__popc(n & 0117) & 1) << 1) | ((__popc((64 | n) & 0155) & 1) << 2) | ((__popc((64 | n) & 0117) & 1) << 3)) << (4 * q);
        }
        uint32_t last_burst_end = 0;                                         // sample (relative) behind the burst that raised the event

        // one trellis step; which: 0 = (A,B), 1 = A only, 2 = B only (viterbi.hpp:166-187)
        // one trellis step from column phase PH (= tr mod 6, a compile-time constant); which: 0 = (A,B), 1 = A only, 2 = B only (viterbi.hpp:166-187)
        auto acs_c = [&](auto PHC, int which, int va, int vb, unsigned long long* dslot) __attribute__((always_inline)) {
            constexpr int PH = decltype(PHC)::value, Q = PH == 5 ? 0 : PH + 1;
            unsigned other;                                                  // the metric of the lane whose state differs in the top state bit
            if constexpr (PH == 0) { const auto r = __builtin_amdgcn_permlane32_swap(m, m, false, false); other = lane < 32 ? r[1] : r[0]; }
            else if constexpr (PH == 1) { const auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false); other = (lane & 16) ? r[0] : r[1]; }
            // row_ror:8 = lane ^ 8
            else if constexpr (PH == 2) other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xF, 0xF, true);
            // row_half_mirror, then quads reversed = lane ^ 4
            else if constexpr (PH == 3) other = (unsigned)__builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
            // quad_perm [2,3,0,1] = lane ^ 2
            else if constexpr (PH == 4) other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true);
            // quad_perm [1,0,3,2] = lane ^ 1
            else other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true);
            const bool hi = (lane >> (5 - PH)) & 1;                          // this lane holds predecessor j + 32 (and will hold successor 2j + 1)
            const unsigned m0 = hi ? other : m, m1 = hi ? m : other;
            const unsigned pb = pbits >> (4 * Q);
            unsigned b0 = 0, b1 = 0;                                         // bm(v, bit) = bit ? 2 (7 - v) : 2 v = 2 (v ^ (bit ? 7 : 0))
            if (which != 2) { b0 += 2 * ((unsigned)va ^ ((0u - (pb & 1)) & 7u)); b1 += 2 * ((unsigned)va ^ ((0u - ((pb >> 2) & 1)) & 7u)); }
            if (which != 1) { b0 += 2 * ((unsigned)vb ^ ((0u - ((pb >> 1) & 1)) & 7u)); b1 += 2 * ((unsigned)vb ^ ((0u - ((pb >> 3) & 1)) & 7u)); }
            const unsigned c0 = (m0 + b0) & 0xFE, c1 = ((m1 + b1) & 0xFF) | 1;
            m = min(c0, c1);
            tr++; ph = Q;
            const unsigned long long d = __ballot(m & 1);
            // the decision word of column tr; every lane stores the same word: no exec juggling in the step
            *dslot = d;
        };
        auto acs = [&](int which, int va, int vb) __attribute__((always_inline)) {      // the same from a run-time phase (symbol edges)
            switch (ph) {
            case 0: acs_c(std::integral_constant<int, 0>{}, which, va, vb, &W.dec[(tr + 1) & 255]); break;
            case 1: acs_c(std::integral_constant<int, 1>{}, which, va, vb, &W.dec[(tr + 1) & 255]); break;
            case 2: acs_c(std::integral_constant<int, 2>{}, which, va, vb, &W.dec[(tr + 1) & 255]); break;
            case 3: acs_c(std::integral_constant<int, 3>{}, which, va, vb, &W.dec[(tr + 1) & 255]); break;
            case 4: acs_c(std::integral_constant<int, 4>{}, which, va, vb, &W.dec[(tr + 1) & 255]); break;
            default: acs_c(std::integral_constant<int, 5>{}, which, va, vb, &W.dec[(tr + 1) & 255]); break;
            }
        };
        auto normalize = [&]() __attribute__((always_inline)) {
            const unsigned mn = wave_min(m);
            m = (m - (mn & 0xFE)) & 0xFF;
        };
        // Traceback (viterbicore.h:468-555) of `bits` bits behind `look` columns, appended to W.out
        auto traceback = [&](uint32_t bits, uint32_t look) __attribute__((always_inline)) {
            const unsigned st = (((unsigned)lane << ph) | ((unsigned)lane >> (6 - ph))) & 63u;     // the state this lane holds now
            const unsigned kmin = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min((m << 8) | (st << 2)));
            const unsigned smin = (kmin >> 2) & 0x3F;
            // The walk runs in LANE space: going back from column c to c - 1 the state loses its lowest bit and gains the decision as its
            // top bit -- under the rotating map that is ONE lane bit, e = 5 - (c - 1) mod 6, being replaced by the decision.
            unsigned L = ((smin >> ph) | (smin << (6 - ph))) & 63u;          // the lane that holds the arg-min state
            unsigned b = (kmin >> 8) & 1;                                    // its decision (mark) bit = the reference's pos bit 6
            unsigned e = (6 - ph) % 6;
            wsync();
            uint32_t col = tr;
            auto back = [&](unsigned long long d) __attribute__((always_inline)) {
                L = (L & ~(1u << e)) | (b << e);
                b = (unsigned)(d >> L) & 1u;
                e = e == 5 ? 0 : e + 1;
            };
            for (uint32_t i = 0; i < look; i++) { col--; back(uni64(W.dec[col & 255])); }
            uint32_t po = nout + (bits >> 3);
            for (uint32_t i = 0; i < bits >> 3; i++) {
                unsigned long long d[8];
#pragma unroll
                for (int j = 0; j < 8; j++) d[j] = uni64(W.dec[(col - 1 - j) & 255]);      // the eight columns of this byte do not depend on the walk
                unsigned oc = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) { oc = ((oc << 1) | b) & 0xFF; back(d[j]); }
                col -= 8; po--;
                if (lane == 0 && po < sizeof(W.out)) W.out[po] = (uint8_t)oc;
            }
            nout += bits >> 3; ob += bits;
        };
        // the check T11aViterbi makes after every puncture group (viterbi.hpp:189-231); returns true when the frame is complete
        auto vit_check = [&]() __attribute__((always_inline)) -> bool {
            if ((tr & 7) == 0) normalize();
            if (tr >= tr_end) { traceback(tr_end - ob - 6, tr - tr_end); return true; }
            if (tr >= ob + 192 + 36 + 6) { const uint32_t rem = (tr - (ob + 192 + 36 + 6)) % 8; traceback(192, 36 + rem); }
            return false;
        };
        // soft values [0, n) of W.joined (or zeros when pad) through the decoder; returns true when the frame is complete
        auto vit_run = [&](uint32_t n, bool pad) __attribute__((always_inline)) -> bool {
            uint32_t k = 0;
            // the symbol's soft values, four per lane, fetched from LDS once; a step reads them with v_readlane (no LDS round trip per step)
            const uint32_t jw = pad ? 0u : reinterpret_cast<const uint32_t*>(W.joined)[lane & 63];
            auto sv = [&](uint32_t i) __attribute__((always_inline)) -> int { return (int)(((uint32_t)__builtin_amdgcn_readlane((int)jw,
                    (int)(i >> 2)) >> (8 * (i & 3))) & 0xFFu); };
            using std::integral_constant;
            while (k < n) {
                // six steps with the exchange pattern known at compile time, where no trace-back can become due inside them (only the
                // normalisation, every eighth column, has to be looked after)
                const uint32_t due = min(tr_end, ob + 192 + 36 + 6);
                const bool room = (tr & 255) + 6 <= 255;                     // the six decision words do not wrap around the ring
                if (ph == 0 && room && tr + 6 < due && code_rate == 0 && k + 12 <= n) {
                    using IC0 = integral_constant<int, 0>; using IC1 = integral_constant<int, 1>; using IC2 = integral_constant<int, 2>;
                    using IC3 = integral_constant<int, 3>; using IC4 = integral_constant<int, 4>; using IC5 = integral_constant<int, 5>;
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)jw, (int)(k >> 2)), w1 = (uint32_t)__builtin_amdgcn_readlane((int)jw, (int)(k >> 2) + 1),
                                   w2 = (uint32_t)__builtin_amdgcn_readlane((int)jw, (int)(k >> 2) + 2);
                    unsigned long long* d0 = &W.dec[(tr & 255) + 1];
                    acs_c(IC0{}, 0, (int)(w0 & 255), (int)((w0 >> 8) & 255), d0);         if ((tr & 7) == 0) normalize();
                    acs_c(IC1{}, 0, (int)((w0 >> 16) & 255), (int)(w0 >> 24), d0 + 1);    if ((tr & 7) == 0) normalize();
                    acs_c(IC2{}, 0, (int)(w1 & 255), (int)((w1 >> 8) & 255), d0 + 2);     if ((tr & 7) == 0) normalize();
                    acs_c(IC3{}, 0, (int)((w1 >> 16) & 255), (int)(w1 >> 24), d0 + 3);    if ((tr & 7) == 0) normalize();
                    acs_c(IC4{}, 0, (int)(w2 & 255), (int)((w2 >> 8) & 255), d0 + 4);     if ((tr & 7) == 0) normalize();
                    acs_c(IC5{}, 0, (int)((w2 >> 16) & 255), (int)(w2 >> 24), d0 + 5);    if ((tr & 7) == 0) normalize();
                    k += 12;
                } else if (ph == 0 && room && tr + 6 < due && code_rate != 0 && k + 8 <= n) {
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)jw, (int)(k >> 2)), w1 = (uint32_t)__builtin_amdgcn_readlane((int)jw, (int)(k >> 2) + 1);
                    unsigned long long* d0 = &W.dec[(tr & 255) + 1];
                    acs_c(integral_constant<int, 0>{}, 0, (int)(w0 & 255), (int)((w0 >> 8) & 255), d0); acs_c(integral_constant<int, 1>{}, 1, (int)((w0 >> 16) & 255), 0, d0 + 1);
                    acs_c(integral_constant<int, 2>{}, 2, 0, (int)(w0 >> 24), d0 + 2);
                    if ((tr & 7) == 0) normalize();
                    acs_c(integral_constant<int, 3>{}, 0, (int)(w1 & 255), (int)((w1 >> 8) & 255), d0 + 3); acs_c(integral_constant<int, 4>{}, 1,
                            (int)((w1 >> 16) & 255), 0, d0 + 4);
                    acs_c(integral_constant<int, 5>{}, 2, 0, (int)(w1 >> 24), d0 + 5);
                    if ((tr & 7) == 0) normalize();
                    k += 8;
                } else {
                    acs(0, sv(k), sv(k + 1));
                    if (code_rate != 0) { acs(1, sv(k + 2), 0); acs(2, 0, sv(k + 3)); k += 4; } else k += 2;
                    if (vit_check()) return true;
                }
            }
            return false;
        };
        // T11aDesc + TBB11aFrameSink (scramble.hpp:319-349, PHY_11a.hpp:660-692) on W.out -> MPDU slot, error code
        auto finish_frame = [&]() __attribute__((always_inline)) {
            wsync();
            const bool has_row = nfr < A.max_frames;                         // frames past the row limit are decoded and counted, not stored
            uint8_t* mp = A.mpdu + ((size_t)cap * A.max_frames + (has_row ? nfr : 0u)) * 4096;
            const unsigned seed = W.out[1] >> 1;
            const unsigned phase = A.T.scr_phase[seed & 0x7F];
            uint8_t* bytes = reinterpret_cast<uint8_t*>(W.buf);              // 1024 bytes + W.fft behind it: 2048 >= 1500
            for (uint32_t i = lane; i < ht_len; i += 64) {
                const unsigned sb = phase == 255 ? 0u : A.T.scr_seq[(phase + 8u * i) % 127u];
                const unsigned o = W.out[2 + i] ^ sb;
                bytes[i] = (uint8_t)o; if (has_row) mp[i] = (uint8_t)o;
            }
            wsync();
            const int n = ht_len >= 4 ? (int)ht_len - 4 : 0;
            uint32_t crc;
            if (n >= 4) crc = crc32_wave(bytes, n, s_crc, s_z, lane);
            else { crc = 0xFFFFFFFFu; for (int i = 0; i < n; i++) crc = (crc >> 8) ^ s_crc[(bytes[i] ^ crc) & 0xFF]; }
            crc = (uint32_t)__builtin_amdgcn_readfirstlane((int)crc);
            uint32_t fcs = 0;
            if (ht_len >= 4) fcs = (uint32_t)bytes[ht_len - 4] | ((uint32_t)bytes[ht_len - 3] << 8) | ((uint32_t)bytes[ht_len - 2] << 16) | ((uint32_t)bytes[ht_len - 1] << 24);
            frame_crc = (uint32_t)uni((int)fcs);
            err = ((~crc) == frame_crc) ? E_OK : E_CRC;
            wsync();
        };
        // T11nSigDemap -> T11aDeinterleaveBPSK -> T11nViterbiSig -> T11nSigParser on W.sig
        auto decode_sig = [&]() __attribute__((always_inline)) {
            wsync();
            for (int g = lane; g < 144; g += 64) {
                const int s3 = g / 48, k = g - 48 * s3;
                int bin; if (k < 24) bin = 38 + k + (k >= 5) + (k >= 18); else { const int q = k - 24; bin = 1 + q + (q >= 6) + (q >= 19); }
                const cpx v = unpack(W.sig[64 * s3 + bin]);
                const int qv = s3 == 0 ? v.re : v.im;
                W.soft[0][g] = s_lut[0][min(max(qv, -128), 127) + 128];
            }
            wsync();
            for (int g = lane; g < 144; g += 64) { const int s3 = g / 48, kk = g - 48 * s3; W.sigsoft[g] = W.soft[0][48 * s3 + 3 * (kk & 15) + (kk >> 4)]; }
            wsync();
            const uint32_t lsig = (uint32_t)uni((int)(uint32_t)(viterbi_sig_wave<24>(W.sigsoft, reinterpret_cast<uint64_t*>(W.dec), lane) >> 6));
            wsync();
            const unsigned long long ht = uni64(viterbi_sig_wave<48>(W.sigsoft + 48, reinterpret_cast<uint64_t*>(W.dec), lane) >> 6);
            wsync();
            bool ok = false;
            do {
                const uint32_t sg = lsig & 0xFFFFFF;
                if (sg & 0xFC0010) break;
                if (__popc(sg) & 1) break;
                const uint32_t code = sg & 0xF;
                if (code < 8) break;                                         // BB11aParseDataRate: 0
                if (((sg >> 5) & 0xFFF) * 2 > 1500) break;
                uint32_t crc = 0xFF;
                for (int b = 0; b < 34; b++) { crc ^= (uint32_t)(ht >> b) & 1; crc = (crc & 1) ? (crc >> 1) ^ 0xE0 : crc >> 1; }
                if (((~crc) & 0xFF) != (uint32_t)((ht >> 34) & 0x3FFF)) break;
                const uint32_t mc = (uint32_t)ht & 0x7F;
                if (mc < 8 || mc >= 11) break;
                const uint32_t hl = (uint32_t)(ht >> 8) & 0xFFFF;
                if (hl > 1500) break;
                mcs = mc; ht_len = hl; code_rate = mc == 10 ? 2u : 0u;
                tr_end = hl * 8 + 16 + 6;
                ok = true;
            } while (0);
            // (two selects: `if (ok) a = ..; else b = ..;` would become a store through a selected pointer and pin both to scratch)
            type = ok ? (int)SYM_HT_STF : type; err = ok ? err : E_PLCP;
            // the Viterbi of the data field starts from a clean trellis (T11aViterbi::Reset at the frame reset)
            m = (lane == 0) ? 0u : 0x30u; tr = 0; ph = 0; ob = 0; nout = 0; soft_n = 0;
            if (lane == 0) W.dec[0] = 0;
            { const int nb = mcs == 8 ? 1 : 2; for (int g = lane; g < 104 * nb; g += 64) W.dtab[g] = (uint8_t)deint11n_index(nb, g & 1, g >> 1); }
            wsync();
        };

        uint32_t a = l0 + 128;                                               // start of the next symbol, relative to origin
        bool more = true;
        while (more) {
            const uint32_t n_pad = (n_real + 3) & ~3u;                       // the last burst is delivered zero-padded
            const bool at_end = a >= n_pad;
            bool do_sig = false, do_vit = false; uint32_t vit_n = 0;         // the SIG decoder and the Viterbi are entered from one place each (code size)
            if (at_end) {
                // ------------------------------------------------------- end of the capture: T11nSymSel::Flush on the empty symbol queue
                if (type == SYM_SIG && nsig > 0 && err == 0) {
                    for (int k = lane; k < 64 * (3 - nsig); k += 64) W.sig[64 * nsig + k] = 0;
                    nsig = 0; do_sig = true;
                } else if (type == SYM_DATA && err == 0 && (soft_n % 312) != 0) {
                    do_vit = true; vit_n = 312 - soft_n % 312;
                }
            } else {
            // ----------------------------------------------------------- one OFDM symbol of both chains: TFreqComp_11n, CP dropped, two FFTs
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint32_t n = a - l0 + 16 + lane;                       // samples since the L-LTF began
                const cpx cof = unpack(A.sincos[(unsigned)((int)n * cfo - theta) & 0xFFFFu]);
                int re, im; mul32(unpack(fetch(r, origin + a + 16 + lane)), cof, re, im);
                W.buf[r][lane] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
            }
            wsync();
            {
                const int g = lane >> 4, e = lane & 15; cpx x[4], yy[4];
#pragma unroll
                for (int q = 0; q < 4; q++) x[q] = unpack(W.buf[g & 1][e + 16 * q]);
                fft64_group(x, yy, W.fft[g], e, tw, nosync);
                if (g < 2) {
#pragma unroll
                    for (int q = 0; q < 4; q++) W.y[g][64 * (type == SYM_HT_LTF ? nltf : 0) + e + 16 * q] = pack(yy[q]);
                }
            }
            wsync();
            if (type == SYM_SIG) {
                int re, im;
                mul32(unpack(W.y[0][lane]), unpack(W.ch[0][lane]), re, im); const cpx x0 = mk(sat16(re >> 9), sat16(im >> 9));
                mul32(unpack(W.y[1][lane]), unpack(W.ch[1][lane]), re, im); const cpx x1 = mk(sat16(re >> 9), sat16(im >> 9));
                W.sig[64 * nsig + lane] = pack(mk((short)((short)(x0.re + x1.re) >> 1), (short)((short)(x0.im + x1.im) >> 1)));
                if (++nsig == 3) { nsig = 0; do_sig = true; }
            } else if (type == SYM_HT_STF) {
                type = SYM_HT_LTF;
            } else if (type == SYM_HT_LTF) {
                if (++nltf == 2) {
#pragma clang fp contract(off)
                    // TMimoChannelEst (channel_11n.hpp:329-443), as k_mimo_est11n_batch
                    nltf = 0; type = SYM_DATA;
                    const int i = lane, k = i < 32 ? i : i - 64;
                    const bool negate = !(k >= -28 && k <= 28 && kHtLtf[k + 28] == 1);
                    cpx hh[2][2];
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const cpx p = unpack(W.y[r][i]), q = unpack(W.y[r][i + 64]);
                        cpx d = sra(csubs(p, q), 1), s = sra(cadds(p, q), 1);
                        if (negate) { d = mk(neg16(d.re), neg16(d.im)); s = mk(neg16(s.re), neg16(s.im)); }
                        hh[r][0] = d; hh[r][1] = s;
                    }
                    const cf a00 = { (float)hh[0][0].re, (float)hh[0][0].im }, a01 = { (float)hh[0][1].re, (float)hh[0][1].im };
                    const cf a10 = { (float)hh[1][0].re, (float)hh[1][0].im }, a11 = { (float)hh[1][1].re, (float)hh[1][1].im };
                    const cf ad = cf_mul(a00, a11), bc = cf_mul(a01, a10);
                    const cf det = { ad.re - bc.re, ad.im - bc.im };
                    const float nn = ((det.re * det.re) + (det.im * det.im)) / 65536.0f;
                    const cf ds = { det.re, -det.im }, m01 = { -a01.re, -a01.im }, m10 = { -a10.re, -a10.im };
                    const cf r00 = cf_mul(a11, ds), r01 = cf_mul(m01, ds), r10 = cf_mul(m10, ds), r11 = cf_mul(a00, ds);
                    W.hinv[0][i] = pack(mk(cvtps_sat16(r00.re / nn), cvtps_sat16(r00.im / nn)));
                    W.hinv[1][i] = pack(mk(cvtps_sat16(r01.re / nn), cvtps_sat16(r01.im / nn)));
                    W.hinv[2][i] = pack(mk(cvtps_sat16(r10.re / nn), cvtps_sat16(r10.im / nn)));
                    W.hinv[3][i] = pack(mk(cvtps_sat16(r11.re / nn), cvtps_sat16(r11.im / nn)));
                }
            } else if (err == 0) {
                // TMimoChannelComp -> TPilotTrack_11n -> demap -> de-interleave -> stream parser -> Viterbi
                const cpx p = unpack(W.y[0][lane]), q = unpack(W.y[1][lane]);
                int ar, ai, br, bi;
                mul32(unpack(W.hinv[0][lane]), p, ar, ai); mul32(unpack(W.hinv[1][lane]), q, br, bi);
                W.xs[0][lane] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
                mul32(unpack(W.hinv[2][lane]), p, ar, ai); mul32(unpack(W.hinv[3][lane]), q, br, bi);
                W.xs[1][lane] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
                wsync();
                {   // TPilotTrack_11n: lane 4 s + k takes pilot k of stream s
                    const int k = lane & 3, sidx = (lane >> 2) & 1;
                    const int pbin = k == 0 ? 64 - 21 : k == 1 ? 64 - 7 : k == 2 ? 7 : 21;
                    const cpx v = unpack(W.xs[sidx][pbin]);
                    int th = dsp_atan16(A.atan, v.re, v.im);
                    th += __shfl_xor(th, 1); th += __shfl_xor(th, 2);
                    const int t0 = (int)(short)(uni(__shfl(th, 0)) >> 2), t1 = (int)(short)(uni(__shfl(th, 4)) >> 2);
                    theta = (int)(short)(theta + (int)(short)((t0 + t1) >> 1));
                }
                const int nb = mcs == 8 ? 1 : 2;
                if (lane < 52) {
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const cpx x = unpack(W.xs[s][data_bin(lane)]);
                        const int re = min(max(x.re, -128), 127) + 128, im = min(max(x.im, -128), 127) + 128;
                        if (nb == 1) W.soft[s][lane] = s_lut[0][re];
                        else { W.soft[s][2 * lane] = s_lut[0][re]; W.soft[s][2 * lane + 1] = s_lut[0][im]; }
                    }
                }
                wsync();
                for (int g = lane; g < 104 * nb; g += 64) W.joined[g] = W.soft[g & 1][W.dtab[g]];
                wsync();
                soft_n += 104 * nb;
                do_vit = true; vit_n = 104 * nb;
            }
            }
            if (do_sig) decode_sig();
            if (do_vit && vit_run(vit_n, at_end)) finish_frame();
            if (at_end) { last_burst_end = n_pad; more = false; }
            else { a += 80; if (err != 0) { last_burst_end = min(a, n_pad); more = false; } }
        }
        if (err == 0) break;                                                 // the capture ended inside a frame without an event
        // ================================================================ the event, as RxThread sees it after the source call returns
        const uint32_t abs_end = origin + last_burst_end;                    // 20 MHz index behind the last burst handed to the graph
        const uint32_t call = (abs_end - 1) / 14;                            // the call that delivered that burst's last sample
        const uint32_t next = min(14 * (call + 1), n20);
        if (lane == 0 && nfr < A.max_frames) {
            Rx11bRow r; r.end_sample = 2 * next; r.error_code = err; r.rate_kbps = err == E_PLCP ? 0u : mcs; r.length = err == E_PLCP ? 0u : ht_len;
                r.crc32 = err == E_PLCP ? 0u : frame_crc;
            rows[nfr] = r;
        }
        nfr++;
        origin = 14 * (call + 1);
    }
    if (lane == 0) A.nframes[cap] = nfr;
}
#endif


// ================================================================================================================================
// The same graph as a chain of kernels, the way the 802.11a path is built (k_scan -> k_frame -> k_viterbi -> k_finish):
//   k_scan11n     one wave per capture: carrier sense, L-LTF (CFO, four FFTs, TSisoChannelEst), the three SIG symbols and their decoder,
//                 RxThread's bookkeeping.  What follows the SIG field never feeds back into carrier sense -- the event of a frame is
//                 raised at its last data symbol, whose position the SIG field fixes (T11nSymSel counts remain_symbols down,
//                 PHY_11n.hpp:331; equivalently the Viterbi passes frame_length * 8 + 22 steps in that symbol) -- so the scan jumps
//                 over the data field and queues it as a job: (capture, L-LTF position, CFO, MCS, length, symbols, soft values incl.
//                 the zero padding of a flush at the end of the capture).
//   k_frame11n    one wave per queued frame: the two HT-LTF symbols -> TMimoChannelEst, then symbol by symbol (the pilot phase of
//                 symbol s rotates symbol s + 1: a serial chain) TFreqComp_11n, two FFTs, TMimoChannelComp, TPilotTrack_11n, demap,
//                 de-interleave, stream join; soft values leave as the 16-bit fields v << 9 the trellis kernel reads.
//   k_viterbi11n  k_rx.hip: the two-frames-per-wave trellis kernel of the 802.11a path with the 192 / 36 window schedule.
//   k_finish11n   T11aDesc + TBB11aFrameSink: descrambler phase table, parallel CRC-32; fills error code, FCS and the MPDU slot of the row.
struct N11Frame {                  // one queued data field
    uint32_t cap;                  // capture index
    uint32_t row;                  // index into rows[] / mpdu slots
    uint32_t l0;                   // 20 MHz index (in the capture) of the first L-LTF sample
    int32_t  cfo;
    uint32_t mcs, ht_len, code_rate;
    uint32_t nproc;                // data symbols that start inside the (padded) capture
    uint32_t nsoft;                // soft values handed to the decoder, zero padding of a final flush included
    uint32_t slot0;                // first symbol slot (global) of the frame's soft / decoded-byte storage
    uint32_t pad[6];
};
struct Scan11nArgs {
    const uint32_t* iq0; const uint32_t* iq1; const CapDesc* caps; uint32_t ncaps, max_frames;
    Rx11bRow* rows; uint32_t* nframes; Tables T; const uint32_t* sincos; const short* atan;
    N11Frame* frames;              // [3][nrows], by code-rate list, compacted
    VitJob*   jobs;                // [3][nrows], same order
    uint32_t* njobs;               // [3]
    uint32_t  nrows;
};
struct Frame11nArgs {
    const uint32_t* iq0; const uint32_t* iq1; const CapDesc* caps;
    const N11Frame* frames; const uint32_t* njobs; uint32_t nrows;
    Tables T; const uint32_t* sincos; const short* atan;
    uint8_t* soft;                 // [slots * 288] the frames' soft streams, one byte per value (VitJob::soft_bits = 8)
    VitJob*   jobs;                // [3][nrows]: k_frame11n fills in the pair stream's offset (the mate is known only after the scan)
    const uint8_t* vout;           // [slots * 32]
    Rx11bRow* rows; uint8_t* mpdu;
};

namespace {
struct ScanLds {
    uint32_t his[2][32]; int hcr[2][32], hci[2][32], he[2][32];
    long long his_e[64];
    uint32_t buf[2][128];
    uint32_t fft[4][64];
    uint32_t y[2][128];
    uint32_t ch[2][64];
    uint32_t sig[192];
    uint8_t  soft0[160];
    uint8_t  sigsoft[144];
    unsigned long long dec[52];
};
struct FrameLds {
    uint32_t buf[2][64];
    uint32_t fft[4][64];
    uint32_t y[2][128];
    uint32_t hinv[4][64];
    uint32_t xs[2][64];
    uint8_t  soft[2][160];
    alignas(4) uint8_t joined[256];
    uint8_t  dtab[208];
};
}  // namespace

// HT40 = false: the reference's 20 MHz graph (k_scan11n).  HT40 = true: the same front end on the legacy part of an HT-mixed 40 MHz frame
// (k_scan_ht40): the legacy preamble and HT-SIG are the 20 MHz waveforms sent on both halves of the channel, the upper one rotated by
// 90 degrees, so the even samples of x[n] j^n -- (-1)^m x[2m]: a sign per sample -- are (1 + j) times the 20 MHz legacy waveform and
// every brick up to T11nSigParser applies unchanged (oracle/py_ht40.py tx_frame / front_end_view; tests/test_ht40_preamble_model.py
// runs the restated reference receiver on it).  What differs behind the parser: MCS 8..14 at CBW 40 and lengths up to 4000 are
// accepted (PHY_11n.hpp:497 accepts 8..10 at 1500), and instead of queueing a 20 MHz data field the frame is recorded for the 40 MHz
// data-field kernels (k_ht40.hip) with what they need from here: position, CFO (per 40 MHz sample) and the noise variance, estimated
// from the difference of the two L-LTF symbols.  That part is this library's own definition: parity unpinned.
template <bool HT40>
__device__ __forceinline__ void scan11n_body(const Scan11nArgs& A, Ht40Found* found)
{
    __shared__ ScanLds s_w[4];
    __shared__ uint8_t s_lut[6][256];
    fill_demap_luts(s_lut);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t cap = blockIdx.x * 4 + wv;
    if (cap >= A.ncaps) return;
    ScanLds& W = s_w[wv];
    const CapDesc cd = A.caps[cap];
    const uint32_t* iq[2] = { A.iq0 + cd.offset, A.iq1 + cd.offset };
    const uint32_t n20 = cd.nsamples / 2;
    auto fetch = [&](int r, uint32_t i) __attribute__((always_inline)) -> uint32_t {
        if (i >= n20) return 0u;
        const uint32_t v = iq[r][2 * (size_t)i];
        if (HT40 && (i & 1u)) { const cpx c = unpack(v); return pack(mk(sat16(-c.re), sat16(-c.im))); }     // x[2m] (-1)^m
        return v;
    };
    const Fft64Tw tw = fft64_twiddles(A.T, lane & 15);
    auto nosync = []() __attribute__((always_inline)) { wsync(); };

    for (int k = lane; k < 64; k += 64) { W.his[0][k & 31] = 0; W.his[1][k & 31] = 0; W.hcr[k >> 5][k & 31] = 0; W.hci[k >> 5][k & 31] = 0;
        W.he[k >> 5][k & 31] = 0; W.his_e[k] = 0x7FFFFFFFFFFFFFFFll; }
    wsync();
    int sr[2] = { 0, 0 }, si[2] = { 0, 0 }, se[2] = { 0, 0 };
    int ring_pos = 0, his_index = 0;
    uint32_t origin = 0, nfr = 0;
    Rx11bRow* rows = A.rows + (size_t)cap * A.max_frames;

    while (origin < n20) {
        // ================================================================ carrier sense from `origin` (as k_rx11n_mono; cca_11n.hpp:46-127)
        const uint32_t nb_total = (n20 - origin + 3) / 4;
        bool pf = false, timeout = false; int pc = 0, sense = 0;
        int64_t det_at = -1;
        for (uint32_t base = 0; base < nb_total * 4 && det_at < 0; base += 64) {
            const int lim = (int)min(64u, nb_total * 4 - base);
            int pr[2], pi[2], pe[2], cre[2], cim[2], een[2]; uint32_t xr[2];
            const int slot = (ring_pos + lane) & 31;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                xr[r] = fetch(r, origin + base + lane);
                const cpx x = unpack(xr[r]);
                const uint32_t dl = (uint32_t)__shfl((int)xr[r], lane - 32);
                const cpx delayed = unpack(lane < 32 ? W.his[r][slot] : dl);
                int re, im; conj_mul32(x, delayed, re, im); re >>= 5; im >>= 5;
                const int ore = __shfl(re, lane - 32), oim = __shfl(im, lane - 32);
                const int e = sqnorm(x) >> 5, oe = __shfl(e, lane - 32);
                const int dre = (int)((unsigned)re - (unsigned)(lane < 32 ? W.hcr[r][slot] : ore));
                const int dim = (int)((unsigned)im - (unsigned)(lane < 32 ? W.hci[r][slot] : oim));
                const int den = (int)((unsigned)e - (unsigned)(lane < 32 ? W.he[r][slot] : oe));
                pr[r] = (int)((unsigned)sr[r] + (unsigned)scan_add(dre, lane)); pi[r] = (int)((unsigned)si[r] + (unsigned)scan_add(dim, lane));
                pe[r] = (int)((unsigned)se[r] + (unsigned)scan_add(den, lane));
                cre[r] = re; cim[r] = im; een[r] = e;
            }
            const int are = (int)((unsigned)(pr[0] >> 1) + (unsigned)(pr[1] >> 1)), aim = (int)((unsigned)(pi[0] >> 1) + (unsigned)(pi[1] >> 1));
            const long long acorr = (long long)((unsigned long long)((long long)are * are) + (unsigned long long)((long long)aim * aim));
            const int ev = (int)((unsigned)(pe[0] >> 1) + (unsigned)(pe[1] >> 1));
            const long long energy = (long long)ev * ev;
            const long long olde = W.his_e[(his_index + lane) & 63];
            const bool cA = olde != 0x7FFFFFFFFFFFFFFFll && (olde + 1) <= (energy >> 2) && 6 * (olde + 1) <= energy && acorr > (energy >> 1);
            const bool cB = acorr < (energy >> 3);
            const unsigned long long bA = __ballot(cA), bB = __ballot(cB);
            int det = -1;
            const unsigned long long lmask = lim >= 64 ? ~0ull : ((1ull << lim) - 1);
            if (!pf && (bA & lmask) == 0) {
                // idle block: per 4-sample burst j  sense += 4; sense >= 84 raises the time-out; a raised time-out is cleared, with the counter, by
                // the first burst that ends a 14-sample source call.  In closed form over the block's bursts (sense needs 21 bursts to reach 84:
                // at most one time-out is raised inside a block): lane j says whether burst j ends a source call, two find-first-bits do the rest.
                const int nbst = lim >> 2;
                const uint32_t X = (uint32_t)__ballot(((base + 4u * (uint32_t)lane + 3u) % 14u) >= 10u) & (nbst >= 32 ? 0xFFFFFFFFu : ((1u << nbst) - 1u));
                int j = 0;
                bool counted = false;
                if (timeout) {
                    if (X == 0u) { sense += 4 * nbst; counted = true; }
                    else { j = __builtin_ctz(X) + 1; sense = 0; timeout = false; }
                }
                if (!counted) {
                    const int r = nbst - j, k = max((84 - sense + 3) >> 2, 1);          // bursts left in the block; bursts until the counter reaches 84
                    if (k > r) sense += 4 * r;
                    else {
                        const int jt = j + k - 1;                                       // the burst that raises the time-out (it may clear it itself)
                        const uint32_t m = jt >= 32 ? 0u : (X >> jt) << jt;
                        if (m == 0u) { sense += 4 * r; timeout = true; }
                        else { sense = 4 * (nbst - 1 - __builtin_ctz(m)); timeout = false; }
                    }
                }
                pc = 0;
            } else if (pf && !timeout && (bB & lmask) == 0 && pc + lim <= 160) {
                pc += lim;
            } else
            for (int i = 0; i < lim; i++) {
                const bool a = (bA >> i) & 1, b = (bB >> i) & 1;
                if (!pf) { sense++; if (a) { sense = 0; pc++; pf = true; } else pc = 0; }
                else if (b) { const bool good = pc > 96 && pc < 160; pf = false; pc = 0; if (good) { det = i; break; } }
                else { pc++; if (pc > 160) { pf = false; pc = 0; } }
                if ((i & 3) == 3) {
                    if (sense >= 84) timeout = true;
                    const uint32_t s4 = base + (uint32_t)i - 3;
                    if (timeout && (s4 + 3) / 14 != (s4 + 7) / 14) { timeout = false; pf = false; pc = 0; sense = 0; }
                }
            }
            const int ne = det >= 0 ? det : lim;
            const int na = det >= 0 ? (det | 3) + 1 : lim;
            if (lane < ne) W.his_e[(his_index + lane) & 63] = energy;
            if (lane < na && lane >= na - 32) {
#pragma unroll
                for (int r = 0; r < 2; r++) { W.his[r][slot] = xr[r]; W.hcr[r][slot] = cre[r]; W.hci[r][slot] = cim[r]; W.he[r][slot] = een[r]; }
            }
#pragma unroll
            for (int r = 0; r < 2; r++) { sr[r] = __shfl(pr[r], na - 1); si[r] = __shfl(pi[r], na - 1); se[r] = __shfl(pe[r], na - 1); }
            his_index = (his_index + ne) & 63; ring_pos = (ring_pos + na) & 31;
            wsync();
            if (det >= 0) det_at = (int64_t)base + na;
        }
        if (det_at < 0) break;
        const uint32_t n_real = n20 - origin;
        const uint32_t n_pad = (n_real + 3) & ~3u;                           // the last burst is delivered zero-padded
        const uint32_t l0 = (uint32_t)det_at;
        if (l0 + 128 > n_pad) break;
        // ================================================================ L-LTF: CFO, compensation, four FFTs, SISO channel
        int cfo;
        {
            int sre = 0, sim = 0;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                int re, im; conj_mul32(unpack(fetch(r, origin + l0 + lane)), unpack(fetch(r, origin + l0 + 64 + lane)), re, im);
                sre += re >> 7; sim += im >> 7;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { sre += __shfl_xor(sre, d); sim += __shfl_xor(sim, d); }
            cfo = uni(dsp_atan32(A.atan, sre, sim) >> 6);
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int n = 64 * h + lane;
                const cpx cof = unpack(A.sincos[(unsigned)(n * cfo) & 0xFFFFu]);
                int re, im; mul32(unpack(fetch(r, origin + l0 + n)), cof, re, im);
                W.buf[r][n] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
            }
        wsync();
        {
            const int g = lane >> 4, e = lane & 15; cpx x[4], yy[4];
#pragma unroll
            for (int m = 0; m < 4; m++) x[m] = unpack(W.buf[g >> 1][64 * (g & 1) + e + 16 * m]);
            fft64_group(x, yy, W.fft[g], e, tw, nosync);
#pragma unroll
            for (int q = 0; q < 4; q++) W.y[g >> 1][64 * (g & 1) + e + 16 * q] = pack(yy[q]);
        }
        wsync();
        float noise_var = 0.0f;
        // the two L-LTF symbols differ by noise only: E|Y1 - Y2|^2 = 2 var(FFT<64> bin); an FFT<128> bin of the
        if (HT40) {
            // 40 MHz stream carries half that (same noise per sample, twice the 1/N), so noise_var = sum / (4 x 104 bins)
            float acc = 0.0f;
            if (lane != 0 && (lane < 27 || lane >= 38)) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const cpx p = unpack(W.y[r][lane]), q = unpack(W.y[r][64 + lane]);
                    const float dr = (float)(p.re - q.re), di = (float)(p.im - q.im);
                    acc += dr * dr + di * di;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
            noise_var = acc * (1.0f / 416.0f);
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            uint32_t o = 0;
            if (lane < 28 || lane >= 36) {
                const uint32_t* l = W.y[r] + (lane & ~3);
                const cpx a = siso_one(l, lane & 3, lane), b = siso_one(l + 64, lane & 3, lane);
                o = pack(mk((short)((short)(a.re + b.re) >> 1), (short)((short)(a.im + b.im) >> 1)));
            }
            W.ch[r][lane] = o;
        }
        wsync();
        // ================================================================ the SIG field: three symbols, theta = 0 (no pilot tracking before the data field)
        uint32_t err = 0, mcs = 0, ht_len = 0, code_rate = 0;
        bool sig_ok = false, decoded = false;
        uint32_t a = l0 + 128;
        int nsig = 0;
        bool at_end = false;
        for (int s3 = 0; s3 < 3; s3++) {
            if (a >= n_pad) { at_end = true; break; }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint32_t n = a - l0 + 16 + lane;
                const cpx cof = unpack(A.sincos[(unsigned)((int)n * cfo) & 0xFFFFu]);
                int re, im; mul32(unpack(fetch(r, origin + a + 16 + lane)), cof, re, im);
                W.buf[r][lane] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
            }
            wsync();
            {
                const int g = lane >> 4, e = lane & 15; cpx x[4], yy[4];
#pragma unroll
                for (int q = 0; q < 4; q++) x[q] = unpack(W.buf[g & 1][e + 16 * q]);
                fft64_group(x, yy, W.fft[g], e, tw, nosync);
                if (g < 2) {
#pragma unroll
                    for (int q = 0; q < 4; q++) W.y[g][e + 16 * q] = pack(yy[q]);
                }
            }
            wsync();
            {
                int re, im;
                mul32(unpack(W.y[0][lane]), unpack(W.ch[0][lane]), re, im); const cpx x0 = mk(sat16(re >> 9), sat16(im >> 9));
                mul32(unpack(W.y[1][lane]), unpack(W.ch[1][lane]), re, im); const cpx x1 = mk(sat16(re >> 9), sat16(im >> 9));
                W.sig[64 * nsig + lane] = pack(mk((short)((short)(x0.re + x1.re) >> 1), (short)((short)(x0.im + x1.im) >> 1)));
            }
            nsig++; a += 80;
            wsync();
        }
        // T11nSymSel::Flush: the missing symbols are zeros
        if (at_end && nsig > 0) { for (int k = lane; k < 64 * (3 - nsig); k += 64) W.sig[64 * nsig + k] = 0; }
        if (!at_end || nsig > 0) {
            // T11nSigDemap -> T11aDeinterleaveBPSK -> T11nViterbiSig -> T11nSigParser on W.sig
            decoded = true;
            wsync();
            for (int g = lane; g < 144; g += 64) {
                const int s3 = g / 48, k = g - 48 * s3;
                int bin; if (k < 24) bin = 38 + k + (k >= 5) + (k >= 18); else { const int q = k - 24; bin = 1 + q + (q >= 6) + (q >= 19); }
                const cpx v = unpack(W.sig[64 * s3 + bin]);
                const int qv = s3 == 0 ? v.re : v.im;
                W.soft0[g] = s_lut[0][min(max(qv, -128), 127) + 128];
            }
            wsync();
            for (int g = lane; g < 144; g += 64) { const int s3 = g / 48, kk = g - 48 * s3; W.sigsoft[g] = W.soft0[48 * s3 + 3 * (kk & 15) + (kk >> 4)]; }
            wsync();
            const uint32_t lsig = (uint32_t)uni((int)(uint32_t)(viterbi_sig_wave<24>(W.sigsoft, reinterpret_cast<uint64_t*>(W.dec), lane) >> 6));
            wsync();
            const unsigned long long ht = uni64(viterbi_sig_wave<48>(W.sigsoft + 48, reinterpret_cast<uint64_t*>(W.dec), lane) >> 6);
            wsync();
            do {
                const uint32_t sg = lsig & 0xFFFFFF;
                if (sg & 0xFC0010) break;
                if (__popc(sg) & 1) break;
                const uint32_t code = sg & 0xF;
                if (code < 8) break;
                if (((sg >> 5) & 0xFFF) * 2 > 1500) break;
                uint32_t crc = 0xFF;
                for (int b = 0; b < 34; b++) { crc ^= (uint32_t)(ht >> b) & 1; crc = (crc & 1) ? (crc >> 1) ^ 0xE0 : crc >> 1; }
                if (((~crc) & 0xFF) != (uint32_t)((ht >> 34) & 0x3FFF)) break;
                const uint32_t mc = (uint32_t)ht & 0x7F;
                const uint32_t hl = (uint32_t)(ht >> 8) & 0xFFFF;
                if (HT40) {
                    if (mc < 8 || mc > 14 || !((ht >> 7) & 1) || hl > 4000 || hl < 4) break;     // two streams, 40 MHz, a rate this library has a decoder for
                    mcs = mc; ht_len = hl; code_rate = (mc == 10 || mc == 12 || mc == 14) ? 2u : mc == 13 ? 1u : 0u;
                } else {
                    if (mc < 8 || mc >= 11) break;
                    if (hl > 1500) break;
                    mcs = mc; ht_len = hl; code_rate = mc == 10 ? 2u : 0u;
                }
                sig_ok = true;
            } while (0);
            if (!sig_ok) err = E_PLCP;
        }
        // ================================================================ what happens to the frame, without looking at its data field
        uint32_t last_burst_end = n_pad;                                     // (relative to origin) behind the burst that raises the event
        bool event = false, queue = false;
        uint32_t nproc = 0, nsoft = 0;
        if (decoded && err != 0) { event = true; last_burst_end = at_end ? n_pad : min(a, n_pad); }
        else if (HT40 && decoded && !at_end) {
            // HT-STF at a, HT-LTF 1 / 2 at a + 80 / a + 160, data symbol d at a + 240 + 80 d (20 MHz indices; 4 us symbols).  The frame is
            // recorded when all of it lies inside the capture; a frame the capture cuts off raises no event (as the 20 MHz graph behaves).
            const uint32_t nb = mcs == 8 ? 1u : mcs <= 10 ? 2u : mcs <= 12 ? 4u : 6u;
            const uint32_t ndbps = 108u * nb * (code_rate == 0 ? 1u : code_rate == 1 ? 2u : 3u) / (code_rate == 0 ? 2u : code_rate == 1 ? 3u : 4u);
            const uint32_t nsym = (16u + 8u * ht_len + 6u + ndbps - 1u) / ndbps;
            nproc = nsym;
            if (a + 240 + 80 * nsym <= n_real) { event = true; queue = true; last_burst_end = a + 240 + 80 * nsym; }
        }
        else if (decoded && !at_end) {
            // HT-STF at a, HT-LTF at a + 80 / a + 160, data symbol d at a + 240 + 80 d; a symbol is processed when it starts inside the
            // padded capture (its missing samples read as zero: the flush of the partly filled queues)
            const uint32_t tr_end = ht_len * 8 + 16 + 6;
            const uint32_t S = mcs == 8 ? 104u : 208u, sps = code_rate == 0 ? S / 2 : S / 4 * 3;     // soft values / trellis steps per symbol
            const uint32_t nsym = (tr_end + sps - 1) / sps;                  // the symbol in which the decoder passes tr_end
            const uint32_t a_data = a + 240;
            if (a_data < n_pad) nproc = min(nsym, (n_pad - a_data + 79) / 80);
            if (nproc == nsym) { event = true; queue = true; nsoft = nsym * S; last_burst_end = min(a_data + 80 * nsym, n_pad); }
            else if (a + 160 < n_pad && nproc > 0 && (nproc * S) % 312 != 0) {
                // the capture ends inside the data field: T11aViterbi's 312-value input burst is padded with zero soft values; an
                // event only if that takes the decoder past tr_end
                nsoft = (nproc * S + 311) / 312 * 312;
                const uint32_t steps = code_rate == 0 ? nsoft / 2 : nsoft / 4 * 3;
                if (steps >= tr_end) { event = true; queue = true; last_burst_end = n_pad; }
            }
        }
        if (!event) break;                                                   // the capture ended inside a frame without an event
        const uint32_t abs_end = origin + last_burst_end;
        const uint32_t call = (abs_end - 1) / 14;
        const uint32_t next = min(14 * (call + 1), n20);
        if (nfr < A.max_frames) {
            const uint32_t row = cap * A.max_frames + nfr;
            if (lane == 0) {
                Rx11bRow r; r.end_sample = 2 * next; r.error_code = queue ? 0u : err; r.rate_kbps = queue ? mcs : 0u; r.length = queue ? ht_len : 0u; r.crc32 = 0u;
                rows[nfr] = r;
                if (HT40) {
                    Ht40Found F; F.a20 = origin + a; F.mcs = queue ? mcs : 0u; F.ht_len = queue ? ht_len : 0u; F.cfo = cfo; F.noise_var = noise_var;
                    F.end_sample = 2 * next; F.error_code = queue ? 0u : err; F.nsym = queue ? nproc : 0u;
                    found[row] = F;
                } else if (queue) {
                    const uint32_t list = code_rate;
                    const uint32_t idx = atomicAdd(&A.njobs[list], 1u);
                    N11Frame F; F.cap = cap; F.row = row; F.l0 = origin + l0; F.cfo = cfo; F.mcs = mcs; F.ht_len = ht_len; F.code_rate = code_rate;
                    F.nproc = nproc; F.nsoft = nsoft; F.slot0 = cd.slot_base + (origin + a + 240) / 80;
                    for (int k = 0; k < 6; k++) F.pad[k] = 0;
                    A.frames[(size_t)list * A.nrows + idx] = F;
                    VitJob J; J.soft_off = F.slot0 * (uint32_t)kSoftPerSlot; J.soft_bits = 8; J.nsoft = nsoft; J.length = ht_len; J.dec_off = 0;
                        J.out_off = F.slot0 * (uint32_t)kOutPerSlot;
                    J.valid = 1; J.code_rate = code_rate;
                    A.jobs[(size_t)list * A.nrows + idx] = J;
                }
            }
        }
        nfr++;
        origin = 14 * (call + 1);
    }
    if (lane == 0) A.nframes[cap] = nfr;
}

__global__ void __launch_bounds__(256) k_scan11n(Scan11nArgs A) { scan11n_body<false>(A, nullptr); }
// The front end of the 40 MHz HT receiver (sora_ht40_process_captures_dev, k_ht40.hip): carrier sense, L-LTF, L-SIG / HT-SIG on the
// duplicated legacy preamble -> one Ht40Found record per event.
__global__ void __launch_bounds__(256) k_scan_ht40(Scan11nArgs A, Ht40Found* found) { scan11n_body<true>(A, found); }

}  // namespace sora
int sora_internal_scan_ht40(const uint32_t* iq0, const uint32_t* iq1, const sora::CapDesc* d_caps, uint32_t ncaps, uint32_t max_frames, sora::Rx11bRow* d_rows, uint32_t* d_nframes,
                            sora::Ht40Found* d_found, const sora::Tables& T, const uint32_t* sincos, const short* atan, hipStream_t st)
{
    using namespace sora;
    Scan11nArgs S{};
    S.iq0 = iq0; S.iq1 = iq1; S.caps = d_caps; S.ncaps = ncaps; S.max_frames = max_frames; S.rows = d_rows; S.nframes = d_nframes; S.T = T; S.sincos = sincos; S.atan = atan;
    S.frames = nullptr; S.jobs = nullptr; S.njobs = nullptr; S.nrows = ncaps * max_frames;
    hipLaunchKernelGGL(k_scan_ht40, dim3((ncaps + 3) / 4), dim3(256), 0, st, S, d_found);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "k_scan_ht40", (int)e);
}
namespace sora {

__global__ void __launch_bounds__(256) k_frame11n(Frame11nArgs A)
{
    __shared__ FrameLds s_w[4];
    __shared__ uint8_t s_lut[6][256];
    fill_demap_luts(s_lut);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const JobRef jr = locate_job(blockIdx.x * 4 + wv, A.njobs);
    if (!jr.ok) return;
    const N11Frame F = A.frames[(size_t)jr.list * A.nrows + jr.idx];
    FrameLds& W = s_w[wv];
    const CapDesc cd = A.caps[F.cap];
    const uint32_t* iq[2] = { A.iq0 + cd.offset, A.iq1 + cd.offset };
    const uint32_t n20 = cd.nsamples / 2;
    auto fetch = [&](int r, uint32_t i) __attribute__((always_inline)) -> uint32_t { return i < n20 ? iq[r][2 * (size_t)i] : 0u; };
    // the packed-arithmetic FFT<64> of k_frame (dev_arith.h): half the instructions of the unpacked one
    const Fft64TwPk tw = fft64_twiddles_pk(A.T, lane & 15);
    auto nosync = []() __attribute__((always_inline)) { wsync(); };
    const int cfo = uni(F.cfo);
    const uint32_t l0 = (uint32_t)uni((int)F.l0), mcs = (uint32_t)uni((int)F.mcs), nproc = (uint32_t)uni((int)F.nproc);
    const int nb = mcs == 8 ? 1 : 2;
    // joined position g = lane + 64 t (stream g & 1 = lane & 1) <- soft[lane & 1][dt[t]]: the lane's four de-interleaver entries stay in registers
    uint32_t dt[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { const int g = lane + 64 * t; dt[t] = g < 104 * nb ? (uint32_t)deint11n_index(nb, g & 1, g >> 1) : 0xFFFFFFFFu; }
    int theta = 0;
    // one symbol at 20 MHz index `pos` (its CP included): TFreqComp_11n (running phase n * CFO - theta, n counted from the L-LTF), two FFTs -> W.y[.][64 * half ..]
    auto symbol_fft = [&](uint32_t pos, uint32_t x0, uint32_t x1, int half) __attribute__((always_inline)) {
        const uint32_t n = pos - l0 + 16 + (uint32_t)lane;
        const cpx cof = unpack(A.sincos[(unsigned)((int)n * cfo - theta) & 0xFFFFu]);
        int re, im;
        mul32(unpack(x0), cof, re, im); W.buf[0][lane] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
        mul32(unpack(x1), cof, re, im); W.buf[1][lane] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
        wsync();
        const int g = lane >> 4, e = lane & 15; pcx x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = W.buf[g & 1][e + 16 * q];
        fft64_core_pk(x, W.fft[g], e, tw, nosync);                               // bin j at slot bitrev6(j) of W.fft[g]
        if (g < 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) W.y[g][64 * half + e + 16 * q] = W.fft[g][__brev((unsigned)(e + 16 * q)) >> 26];
        }
        wsync();
    };
    const uint32_t a_ltf = l0 + 128 + 320;                                   // L-LTF (128), three SIG symbols, HT-STF
    // ---- the two HT-LTF symbols -> TMimoChannelEst (channel_11n.hpp:329-443), as k_mimo_est11n_batch
    symbol_fft(a_ltf, fetch(0, a_ltf + 16 + lane), fetch(1, a_ltf + 16 + lane), 0);
    symbol_fft(a_ltf + 80, fetch(0, a_ltf + 96 + lane), fetch(1, a_ltf + 96 + lane), 1);
    {
#pragma clang fp contract(off)
        const int i = lane, k = i < 32 ? i : i - 64;
        const bool negate = !(k >= -28 && k <= 28 && kHtLtf[k + 28] == 1);
        cpx hh[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const cpx p = unpack(W.y[r][i]), q = unpack(W.y[r][i + 64]);
            cpx d = sra(csubs(p, q), 1), s = sra(cadds(p, q), 1);
            if (negate) { d = mk(neg16(d.re), neg16(d.im)); s = mk(neg16(s.re), neg16(s.im)); }
            hh[r][0] = d; hh[r][1] = s;
        }
        const cf a00 = { (float)hh[0][0].re, (float)hh[0][0].im }, a01 = { (float)hh[0][1].re, (float)hh[0][1].im };
        const cf a10 = { (float)hh[1][0].re, (float)hh[1][0].im }, a11 = { (float)hh[1][1].re, (float)hh[1][1].im };
        const cf ad = cf_mul(a00, a11), bc = cf_mul(a01, a10);
        const cf det = { ad.re - bc.re, ad.im - bc.im };
        const float nn = ((det.re * det.re) + (det.im * det.im)) / 65536.0f;
        const cf ds = { det.re, -det.im }, m01 = { -a01.re, -a01.im }, m10 = { -a10.re, -a10.im };
        const cf r00 = cf_mul(a11, ds), r01 = cf_mul(m01, ds), r10 = cf_mul(m10, ds), r11 = cf_mul(a00, ds);
        W.hinv[0][i] = pack(mk(cvtps_sat16(r00.re / nn), cvtps_sat16(r00.im / nn)));
        W.hinv[1][i] = pack(mk(cvtps_sat16(r01.re / nn), cvtps_sat16(r01.im / nn)));
        W.hinv[2][i] = pack(mk(cvtps_sat16(r10.re / nn), cvtps_sat16(r10.im / nn)));
        W.hinv[3][i] = pack(mk(cvtps_sat16(r11.re / nn), cvtps_sat16(r11.im / nn)));
    }
    wsync();
    // ---- the data symbols, in order
    const uint32_t a_data = a_ltf + 160;
    const uint32_t S = 104u * (uint32_t)nb;
    // the frame's soft stream, one byte per value (VitJob::soft_bits = 8), in its own symbol slots
    uint8_t* dst = A.soft + (size_t)F.slot0 * kSoftPerSlot;
    uint32_t nx0 = fetch(0, a_data + 16 + lane), nx1 = fetch(1, a_data + 16 + lane);     // the next symbol's samples are requested one symbol ahead
    for (uint32_t d = 0; d < nproc; d++) {
        const uint32_t pos = a_data + 80 * d;
        const uint32_t x0 = nx0, x1 = nx1;
        if (d + 1 < nproc) { nx0 = fetch(0, pos + 96 + lane); nx1 = fetch(1, pos + 96 + lane); }
        symbol_fft(pos, x0, x1, 0);
        // TMimoChannelComp -> TPilotTrack_11n -> demap -> de-interleave -> stream parser
        const cpx p = unpack(W.y[0][lane]), q = unpack(W.y[1][lane]);
        int ar, ai, br, bi;
        mul32(unpack(W.hinv[0][lane]), p, ar, ai); mul32(unpack(W.hinv[1][lane]), q, br, bi);
        W.xs[0][lane] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
        mul32(unpack(W.hinv[2][lane]), p, ar, ai); mul32(unpack(W.hinv[3][lane]), q, br, bi);
        W.xs[1][lane] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
        wsync();
        {
            const int k = lane & 3, sidx = (lane >> 2) & 1;
            const int pbin = k == 0 ? 64 - 21 : k == 1 ? 64 - 7 : k == 2 ? 7 : 21;
            const cpx v = unpack(W.xs[sidx][pbin]);
            int th = dsp_atan16(A.atan, v.re, v.im);
            th += __shfl_xor(th, 1); th += __shfl_xor(th, 2);
            const int t0 = (int)(short)(uni(__shfl(th, 0)) >> 2), t1 = (int)(short)(uni(__shfl(th, 4)) >> 2);
            theta = (int)(short)(theta + (int)(short)((t0 + t1) >> 1));
        }
        if (lane < 52) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const cpx x = unpack(W.xs[s][data_bin(lane)]);
                const int re = min(max(x.re, -128), 127) + 128, im = min(max(x.im, -128), 127) + 128;
                if (nb == 1) W.soft[s][lane] = s_lut[0][re];
                else { W.soft[s][2 * lane] = s_lut[0][re]; W.soft[s][2 * lane + 1] = s_lut[0][im]; }
            }
        }
        wsync();
        {
            const uint8_t* mine = W.soft[lane & 1];
            uint8_t* o = dst + (size_t)d * S + lane;
#pragma unroll
            for (int t = 0; t < 4; t++) if (dt[t] != 0xFFFFFFFFu) o[64 * t] = mine[dt[t]];
        }
        wsync();
    }
    for (uint32_t g = nproc * S + lane; g < F.nsoft; g += 64) dst[g] = 0;                     // the zero soft values of a flush at the end of the capture
}

// T11aDesc + TBB11aFrameSink (scramble.hpp:319-349, PHY_11a.hpp:660-692) on the decoded bytes of a queued frame -> MPDU slot, error code, FCS
__global__ void __launch_bounds__(256) k_finish11n(Frame11nArgs A)
{
    __shared__ uint32_t s_crc[256];
    __shared__ uint32_t s_z[6 * 8 * 16];
    __shared__ uint32_t s_bufs[4][1504 / 4 + 2];
    s_crc[threadIdx.x] = A.T.crc[threadIdx.x];
    for (int i = threadIdx.x; i < 6 * 8 * 16; i += 256) s_z[i] = A.T.crcz[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const JobRef jr = locate_job(blockIdx.x * 4 + wv, A.njobs);
    if (!jr.ok) return;
    const N11Frame F = A.frames[(size_t)jr.list * A.nrows + jr.idx];
    const uint8_t* dec = A.vout + (size_t)F.slot0 * kOutPerSlot;
    uint8_t* mp = A.mpdu + (size_t)F.row * 4096;
    uint8_t* bytes = reinterpret_cast<uint8_t*>(s_bufs[wv]);
    const uint32_t L = F.ht_len;
    const unsigned seed = dec[1] >> 1;
    const unsigned phase = A.T.scr_phase[seed & 0x7F];
    for (uint32_t i = lane; i < L; i += 64) {
        const unsigned sb = phase == 255 ? 0u : A.T.scr_seq[(phase + 8u * i) % 127u];
        const unsigned o = dec[2 + i] ^ sb;
        bytes[i] = (uint8_t)o; mp[i] = (uint8_t)o;
    }
    wsync();
    const int n = L >= 4 ? (int)L - 4 : 0;
    uint32_t crc;
    if (n >= 4) crc = crc32_wave(bytes, n, s_crc, s_z, lane);
    else { crc = 0xFFFFFFFFu; for (int i = 0; i < n; i++) crc = (crc >> 8) ^ s_crc[(bytes[i] ^ crc) & 0xFF]; }
    if (lane == 0) {
        uint32_t fcs = 0;
        if (L >= 4) fcs = (uint32_t)bytes[L - 4] | ((uint32_t)bytes[L - 3] << 8) | ((uint32_t)bytes[L - 2] << 16) | ((uint32_t)bytes[L - 1] << 24);
        Rx11bRow& r = A.rows[F.row];
        r.crc32 = fcs;
        r.error_code = ((~crc) == fcs) ? E_OK : E_CRC;
    }
}

}  // namespace sora

// ------------------------------------------------------------------------------------------------ host side (C ABI, include/sora_hip.h)
#include <vector>
#include <thread>
#include <string.h>
#include <stdlib.h>
using namespace sora;


struct Pipe11n {                         // one call in flight: a stream and every device array a call writes
    hipStream_t stream = nullptr;
    CapDesc* d_caps = nullptr; Rx11bRow* d_rows = nullptr; uint32_t* d_nframes = nullptr; uint8_t* d_mpdu = nullptr;
    N11Frame* d_frames = nullptr; VitJob* d_jobs = nullptr; uint32_t* d_njobs = nullptr; uint8_t* d_soft = nullptr; uint8_t* d_vout = nullptr;
    uint16_t* d_wvecs = nullptr; unsigned long long* d_wstats = nullptr; uint32_t wstride = 0;   // the window-parallel trellis's vectors and proof record (on its first use)
    std::vector<sora_capture_desc> h_caps; std::vector<CapDesc> h_desc;
    uint32_t ncaps = 0; bool have_results = false; int ticket = 0;
    DenseStage dense;                       // sora_rx11n_deliver_async
    hipEvent_t ev_done = nullptr; bool delivered = false, released = false;     // sora_rx11n_wait_any (the rules of sora_rx_wait_any, include/sora_hip.h)
};
struct sora_rx11n {
    sora_rx_cfg cfg{};
    sora_complex16* d_iq_own[2] = { nullptr, nullptr };
    Tables T{}; const uint32_t* sincos = nullptr; const short* atan = nullptr;
    // the staged chain (k_scan11n -> k_frame11n -> k_viterbi11n -> k_finish11n); the build variant SORA_VARIANT_11N_MONO runs the one-kernel form instead
#ifdef SORA_VARIANT_11N_MONO
    static constexpr bool mono = true;
#else
    static constexpr bool mono = false;
#endif
    // trellis kernel (sora_rx11n_set_trellis): 64 = k_viterbi11n (64 lanes per frame pair), 16 = k_viterbi16_11n, SORA_TRELLIS_WINDOWED = k_viterbi16w_11n + k_win_redo_11n
    // (round 6: the frame's 192-bit trace-back windows side by side, proven afterwards -- k_vitwin.hip), 0 = automatic: window-parallel while the handle holds few
    // frames in flight (one wave-slot per frame leaves the chip idle: a lone capture's 8000-step frame was 0.23 ms of a 0.37 ms call), k_viterbi11n above that
    int trellis = 0;
    uint64_t cap_slots = 0;
    static constexpr int kMaxDepth = 8;
    Pipe11n* pipes[kMaxDepth] = {};
    int depth = 1, cur = 0, next_ticket = 0; bool started = false;
};

#define HIPCHK11N(call) do { hipError_t _e = (call); if (_e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, #call, (int)_e); } while (0)

static void pipe11n_free(Pipe11n* p)
{
    if (!p) return;
    if (p->stream) { (void)hipStreamSynchronize(p->stream); (void)hipStreamDestroy(p->stream); }
    if (p->ev_done) (void)hipEventDestroy(p->ev_done);
    (void)hipFree(p->d_caps); (void)hipFree(p->d_rows); (void)hipFree(p->d_nframes); (void)hipFree(p->d_mpdu);
    (void)hipFree(p->d_frames); (void)hipFree(p->d_jobs); (void)hipFree(p->d_njobs); (void)hipFree(p->d_soft); (void)hipFree(p->d_vout);
    (void)hipFree(p->d_wvecs); (void)hipFree(p->d_wstats);
    sora_internal_dense_free(&p->dense);
    delete p;
}
static void rx11n_free(sora_rx11n_t* rx)
{
    if (!rx) return;
    for (Pipe11n* p : rx->pipes) pipe11n_free(p);
    (void)hipFree(rx->d_iq_own[0]); (void)hipFree(rx->d_iq_own[1]);
    delete rx;
}
static hipError_t pipe11n_create(sora_rx11n_t* rx, Pipe11n** out, int index = 0)
{
    const sora_rx_cfg* cfg = &rx->cfg;
    const size_t rows = (size_t)cfg->max_captures * cfg->max_frames_per_capture;
    Pipe11n* p = new Pipe11n();
    hipError_t e = sora_internal_stream_create(&p->stream, index);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_caps, sizeof(CapDesc) * cfg->max_captures);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_rows, sizeof(Rx11bRow) * rows);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_nframes, 4 * (size_t)cfg->max_captures);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_mpdu, rows * 4096);
    if (!rx->mono) {
        if (e == hipSuccess) e = hipMalloc((void**)&p->d_frames, 3 * sizeof(N11Frame) * rows);
        if (e == hipSuccess) e = hipMalloc((void**)&p->d_jobs, 3 * sizeof(VitJob) * rows);
        if (e == hipSuccess) e = hipMalloc((void**)&p->d_njobs, 16);
        if (e == hipSuccess) e = hipMalloc((void**)&p->d_soft, (size_t)rx->cap_slots * kSoftPerSlot + kSoftSlack);
        if (e == hipSuccess) e = hipMalloc((void**)&p->d_vout, (size_t)rx->cap_slots * kOutPerSlot + 256);
        // every array starts out defined: the decoder reads its soft stream in 12-step chunks (the tail of a frame's last chunk is read, never used)
        if (e == hipSuccess) {
            (void)hipMemsetAsync(p->d_frames, 0, 3 * sizeof(N11Frame) * rows, p->stream); (void)hipMemsetAsync(p->d_jobs, 0, 3 * sizeof(VitJob) * rows, p->stream);
            (void)hipMemsetAsync(p->d_soft, 0, (size_t)rx->cap_slots * kSoftPerSlot + kSoftSlack, p->stream); (void)hipMemsetAsync(p->d_vout, 0,
                    (size_t)rx->cap_slots * kOutPerSlot + 256, p->stream);
        }
    }
    if (e == hipSuccess) { (void)hipMemsetAsync(p->d_rows, 0, sizeof(Rx11bRow) * rows, p->stream); (void)hipMemsetAsync(p->d_nframes, 0,
            4 * (size_t)cfg->max_captures, p->stream); }
    if (e != hipSuccess) { pipe11n_free(p); return e; }
    *out = p;
    return hipSuccess;
}

int sora_rx11n_create(const sora_rx_cfg* cfg, sora_rx11n_t** out)
{
    if (!cfg || !out || cfg->struct_size != sizeof(sora_rx_cfg)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_create: bad cfg", 0);
    if (cfg->sample_rate_mhz != 40) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_create: the 802.11n graph takes 40 MHz samples (sample_rate_mhz = 40)", 0);
    if (cfg->max_captures == 0 || cfg->max_total_samples == 0 || cfg->max_frames_per_capture == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "zero capacity", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (cfg->device < 0 || cfg->device >= ndev) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "device ordinal out of range", 0);
    HIPCHK11N(hipSetDevice(cfg->device));
    sora_rx11n_t* rx = new sora_rx11n();
    rx->cfg = *cfg;
    if (!(sora_internal_tables(cfg->device, &rx->T) == SORA_OK && sora_internal_dsp_tables(&rx->sincos, &rx->atan) == SORA_OK)) { rx11n_free(rx);
        return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_rx11n_create: tables", 0); }
    if (!rx->mono) {
        // symbol slots: 80 samples at 20 MHz each, + 4 per capture (the decoder's padded last burst and its chunked reads may reach past the last symbol)
        rx->cap_slots = cfg->max_total_samples / 2 / 80 + 4 * (uint64_t)cfg->max_captures + 4;
        if (rx->cap_slots * (uint64_t)kSoftPerSlot * 2 >= (1ull << 32)) { rx11n_free(rx); return sora_internal_fail(SORA_ERR_CAPACITY,
                "sora_rx11n_create: max_total_samples exceeds the 32-bit slot geometry of one handle (split the batch over several handles)", 0); }
    }
    const hipError_t e = pipe11n_create(rx, &rx->pipes[0]);
    if (e != hipSuccess) { rx11n_free(rx); return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_rx11n_create: device allocation", (int)e); }
    *out = rx;
    return SORA_OK;
}

void* sora_rx11n_stream(sora_rx11n_t* rx) { return rx ? (void*)rx->pipes[rx->cur]->stream : nullptr; }
void sora_rx11n_destroy(sora_rx11n_t* rx) { if (rx) { (void)hipSetDevice(rx->cfg.device); rx11n_free(rx); } }

int sora_rx11n_set_depth(sora_rx11n_t* rx, int depth)
{
    if (!rx) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_set_depth: null handle", 0);
    const int prev = rx->depth;
    if (depth <= 0) return prev;
    if (depth > sora_rx11n::kMaxDepth) depth = sora_rx11n::kMaxDepth;
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    for (Pipe11n* p : rx->pipes) if (p) HIPCHK11N(hipStreamSynchronize(p->stream));
    for (int i = 0; i < depth; i++)
        if (!rx->pipes[i]) { const hipError_t e = pipe11n_create(rx, &rx->pipes[i], i);
            if (e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_rx11n_set_depth: device allocation", (int)e); }
    // a shrink keeps the most recent call addressable: its pipeline moves into the surviving range (the tickets of the pipelines that
    // fall outside it become stale, as the header says)
    if (rx->cur >= depth) { std::swap(rx->pipes[0], rx->pipes[rx->cur]); rx->cur = 0; }
    rx->depth = depth;
    return prev;
}

// frame rows in flight (depth x max_captures x max_frames_per_capture) up to which the automatic choice is the window-parallel trellis
constexpr long long kAutoWindowedRows11n = 2048;
static int trellis11n_for(const sora_rx11n_t* rx)
{
    if (rx->trellis) return rx->trellis;
    const long long rows = (long long)rx->depth * (long long)rx->cfg.max_captures * (long long)rx->cfg.max_frames_per_capture;
    return rows <= kAutoWindowedRows11n ? SORA_TRELLIS_WINDOWED : 64;
}
int sora_rx11n_set_trellis(sora_rx11n_t* rx, int lanes_per_pair)
{
    if (!rx) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_set_trellis: null handle", 0);
    const int old = rx->trellis;
    if (lanes_per_pair == 0 || lanes_per_pair == 16 || lanes_per_pair == 64 || lanes_per_pair == SORA_TRELLIS_WINDOWED) rx->trellis = lanes_per_pair;
    else if (lanes_per_pair > 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_set_trellis: 0 (automatic), 16 or 64 lanes per frame pair, or SORA_TRELLIS_WINDOWED", 0);
    return old;
}
int sora_rx11n_trellis(sora_rx11n_t* rx) { return rx ? trellis11n_for(rx) : SORA_ERR_INVALID_PARAM; }
// the window-parallel trellis's proof record since the handle was created (as sora_rx_window_stats)
int sora_rx11n_window_stats(sora_rx11n_t* rx, unsigned long long out[4])
{
    if (!rx || !out) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_window_stats: null argument", 0);
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    for (int i = 0; i < 4; i++) out[i] = 0;
    for (Pipe11n* p : rx->pipes) if (p && p->d_wstats) {
        unsigned long long v[4 * kWinStatBanks];
        HIPCHK11N(hipStreamSynchronize(p->stream));
        HIPCHK11N(hipMemcpy(v, p->d_wstats, sizeof v, hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < 4 * kWinStatBanks; i++) out[i & 3u] += v[i];
    }
    return SORA_OK;
}

static Pipe11n* pipe11n_of(sora_rx11n_t* rx, int ticket);
int sora_rx11n_deliver_async(sora_rx11n_t* rx, int ticket, sora_frame_result* h_rows, size_t max_rows, uint32_t* h_counts, uint8_t* h_mpdu, size_t mpdu_cap)
{
    Pipe11n* P = pipe11n_of(rx, ticket);
    if (!P || !P->have_results) return sora_internal_fail(SORA_ERR_INVALID_PARAM,
            "sora_rx11n_deliver_async: stale ticket (its pipeline has been reused by a later process call, or the ticket was never issued)", 0);
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    const int rc = sora_internal_dense_deliver(&P->dense, P->d_rows, P->d_nframes, P->d_caps, nullptr, P->ncaps, rx->cfg.max_frames_per_capture, P->d_mpdu, P->stream,
                                               h_rows, max_rows, h_counts, h_mpdu, mpdu_cap);
    if (rc != SORA_OK) return rc;
    if (!P->ev_done) HIPCHK11N(hipEventCreateWithFlags(&P->ev_done, hipEventDisableTiming));
    HIPCHK11N(hipEventRecord(P->ev_done, P->stream));
    P->delivered = true;
    return SORA_OK;
}

int sora_rx11n_synchronize(sora_rx11n_t* rx)
{
    if (!rx) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_synchronize: null handle", 0);
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    for (Pipe11n* p : rx->pipes) if (p) HIPCHK11N(hipStreamSynchronize(p->stream));
    return SORA_OK;
}

static Pipe11n* pipe11n_of(sora_rx11n_t* rx, int ticket)
{
    if (!rx || ticket <= 0) return nullptr;
    for (int i = 0; i < rx->depth; i++) if (rx->pipes[i] && rx->pipes[i]->ticket == ticket) return rx->pipes[i];
    return nullptr;
}
int sora_rx11n_ticket(sora_rx11n_t* rx) { return rx && rx->started ? rx->pipes[rx->cur]->ticket : 0; }
int sora_rx11n_wait(sora_rx11n_t* rx, int ticket)
{
    Pipe11n* p = pipe11n_of(rx, ticket);
    if (!p) return sora_internal_fail(SORA_ERR_INVALID_PARAM,
            "sora_rx11n_wait: stale ticket (its pipeline has been reused by a later process call, or the ticket was never issued)", 0);
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    HIPCHK11N(hipStreamSynchronize(p->stream));
    if (p->delivered) p->released = true;
    return SORA_OK;
}

int sora_rx11n_wait_any(sora_rx11n_t* rx, int* ticket)
{
    if (!rx || !ticket) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_wait_any: null argument", 0);
    *ticket = 0;
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    for (unsigned spin = 0;; spin++) {
        Pipe11n* done = nullptr; bool pending = false;
        for (int i = 0; i < rx->depth; i++) {
            Pipe11n* p = rx->pipes[i];
            if (!p || p->ticket == 0 || !p->delivered || p->released) continue;
            pending = true;
            const hipError_t q = hipEventQuery(p->ev_done);
            if (q == hipSuccess) { if (!done || p->ticket < done->ticket) done = p; }
            else if (q != hipErrorNotReady) { (void)hipGetLastError(); return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_rx11n_wait_any: hipEventQuery", (int)q); }
        }
        if (done) { HIPCHK11N(hipStreamSynchronize(done->stream)); done->released = true; *ticket = done->ticket; return SORA_OK; }
        if (!pending) return sora_internal_fail(SORA_ERR_FAILED, "sora_rx11n_wait_any: no call with an enqueued delivery (sora_rx11n_deliver_async) is in flight", 0);
        (void)hipGetLastError();
        if (spin > 64) std::this_thread::yield();
    }
}

// The pipeline of the next call: an unused one, else the released call with the oldest ticket, else the oldest call (plain rotation) -- sora_hip.cpp next_pipe()
static int next_pipe11n(const sora_rx11n_t* rx)
{
    if (!rx->started) return 0;
    int best = -1, best_rel = -1;
    for (int i = 0; i < rx->depth; i++) {
        const Pipe11n* p = rx->pipes[i];
        if (!p) continue;
        if (p->ticket == 0) return i;
        if (p->released && (best_rel < 0 || p->ticket < rx->pipes[best_rel]->ticket)) best_rel = i;
        if (best < 0 || p->ticket < rx->pipes[best]->ticket) best = i;
    }
    return best_rel >= 0 ? best_rel : best;
}

int sora_rx11n_process_dev(sora_rx11n_t* rx, const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_capture_desc* caps, size_t ncaps)
{
    if (!rx || (ncaps && (!d_iq0 || !d_iq1 || !caps))) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_process_dev: null argument", 0);
    if (ncaps > rx->cfg.max_captures) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11n_process_dev: more captures than max_captures", 0);
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    const int idx = next_pipe11n(rx);                                            // consecutive calls rotate over the pipelines; a released one first
    Pipe11n* P = rx->pipes[idx];
    std::vector<CapDesc> h(ncaps);                                               // validated first: a refused call leaves the handle's calls intact
    uint64_t total = 0, slots = 0;
    for (size_t i = 0; i < ncaps; i++) {
        if (caps[i].nsamples % 28 != 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "capture length must be a whole number of 28-sample source bursts", 0);
        h[i].offset = caps[i].offset; h[i].nsamples = caps[i].nsamples; h[i].capture_id = caps[i].capture_id;
        h[i].slot_base = (uint32_t)slots; h[i].nslots = caps[i].nsamples / 2 / 80 + 4;
        slots += h[i].nslots; total += caps[i].nsamples;
    }
    if (total > rx->cfg.max_total_samples || (!rx->mono && slots > rx->cap_slots)) return sora_internal_fail(SORA_ERR_CAPACITY,
            "sora_rx11n_process_dev: more samples than max_total_samples", 0);
    HIPCHK11N(hipStreamSynchronize(P->stream));                                  // the call that used this pipeline `depth` calls ago has finished
    P->h_desc.swap(h);
    P->h_caps.assign(caps, caps + ncaps); P->ncaps = (uint32_t)ncaps; P->have_results = true; P->ticket = ++rx->next_ticket; P->delivered = P->released = false;
    rx->cur = idx; rx->started = true;
    if (ncaps == 0) return SORA_OK;
    HIPCHK11N(hipMemcpyAsync(P->d_caps, P->h_desc.data(), sizeof(CapDesc) * ncaps, hipMemcpyHostToDevice, P->stream));
    Rx11nArgs A;
    A.iq0 = reinterpret_cast<const uint32_t*>(d_iq0); A.iq1 = reinterpret_cast<const uint32_t*>(d_iq1); A.caps = P->d_caps; A.ncaps = (uint32_t)ncaps;
    A.max_frames = rx->cfg.max_frames_per_capture; A.rows = P->d_rows; A.nframes = P->d_nframes; A.mpdu = P->d_mpdu; A.T = rx->T; A.sincos = rx->sincos; A.atan = rx->atan;
#ifdef SORA_VARIANT_11N_MONO
    hipLaunchKernelGGL(k_rx11n_mono, dim3((unsigned)((ncaps + 3) / 4)), dim3(256), 0, P->stream, A);
    HIPCHK11N(hipGetLastError());
    return SORA_OK;
#endif
    const uint32_t nrows = (uint32_t)ncaps * rx->cfg.max_frames_per_capture;
    HIPCHK11N(hipMemsetAsync(P->d_njobs, 0, 16, P->stream));
    Scan11nArgs S;
    S.iq0 = A.iq0; S.iq1 = A.iq1; S.caps = P->d_caps; S.ncaps = (uint32_t)ncaps; S.max_frames = A.max_frames; S.rows = P->d_rows; S.nframes = P->d_nframes;
    S.T = rx->T; S.sincos = rx->sincos; S.atan = rx->atan; S.frames = P->d_frames; S.jobs = P->d_jobs; S.njobs = P->d_njobs; S.nrows = nrows;
    hipLaunchKernelGGL(k_scan11n, dim3((unsigned)((ncaps + 3) / 4)), dim3(256), 0, P->stream, S);
    Frame11nArgs F;
    F.iq0 = A.iq0; F.iq1 = A.iq1; F.caps = P->d_caps; F.frames = P->d_frames; F.njobs = P->d_njobs; F.nrows = nrows; F.T = rx->T; F.sincos = rx->sincos; F.atan = rx->atan;
    F.soft = P->d_soft; F.jobs = P->d_jobs; F.vout = P->d_vout; F.rows = P->d_rows; F.mpdu = P->d_mpdu;
    hipLaunchKernelGGL(k_frame11n, dim3((nrows + 3) / 4), dim3(256), 0, P->stream, F);
    const int trellis = trellis11n_for(rx);
    if (trellis == SORA_TRELLIS_WINDOWED) {
        constexpr uint32_t kTarget = 16384, kLoneWaves = 3 * kWinLonePad / 8;
        if (!P->d_wvecs) {
            const uint64_t cap_rows = (uint64_t)rx->cfg.max_captures * rx->cfg.max_frames_per_capture;
            P->wstride = (uint32_t)(std::min<uint64_t>(std::max<uint64_t>(kTarget, cap_rows), (uint64_t)kWinMaxUnits * cap_rows) + cap_rows);
            HIPCHK11N(hipMalloc((void**)&P->d_wvecs, 3 * (size_t)kWinVecBytes * P->wstride));
            HIPCHK11N(hipMalloc((void**)&P->d_wstats, 4 * kWinStatBanks * sizeof(unsigned long long)));
            HIPCHK11N(hipMemsetAsync(P->d_wstats, 0, 4 * kWinStatBanks * sizeof(unsigned long long), P->stream));
        }
        const uint32_t units_max = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(kTarget, nrows), (uint64_t)kWinMaxUnits * nrows);
        hipLaunchKernelGGL(k_viterbi16w_11n, dim3((units_max + 7) / 8 + 3 + kLoneWaves), dim3(64), 0, P->stream, (const VitJob*)P->d_jobs, (const uint32_t*)P->d_njobs,
                nrows, kTarget, P->wstride, (const uint8_t*)P->d_soft, P->d_vout, P->d_wvecs);
        // the proof, and the serial decode of the pairs of frames that fail it (none, normally)
        hipLaunchKernelGGL(k_win_redo_11n, dim3((nrows / 2 + 3 + 3) / 4), dim3(256), 0, P->stream, (const VitJob*)P->d_jobs, (const uint32_t*)P->d_njobs,
                nrows, kTarget, P->wstride, (const uint16_t*)P->d_wvecs, (const uint8_t*)P->d_soft, P->d_vout, P->d_wstats);
    }
    else if (trellis == 16)
        hipLaunchKernelGGL(k_viterbi16_11n, dim3((nrows + 7) / 8 + 2), dim3(64), 0, P->stream, (const VitJob*)P->d_jobs, (const uint32_t*)P->d_njobs, 0u,
                nrows, (const uint8_t*)P->d_soft, P->d_vout);
    else
        hipLaunchKernelGGL(k_viterbi11n, dim3((nrows / 2 + 3 + 3) / 4), dim3(256), 0, P->stream, (const VitJob*)P->d_jobs, (const uint32_t*)P->d_njobs, 0u,
                nrows, (const uint8_t*)P->d_soft, P->d_vout);
    hipLaunchKernelGGL(k_finish11n, dim3((nrows + 3) / 4), dim3(256), 0, P->stream, F);
    HIPCHK11N(hipGetLastError());
    return SORA_OK;
}

int sora_rx11n_process(sora_rx11n_t* rx, const sora_complex16* h_iq0, const sora_complex16* h_iq1, size_t nsamples, const sora_capture_desc* caps, size_t ncaps)
{
    if (!rx || (nsamples && (!h_iq0 || !h_iq1))) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_process: null argument", 0);
    if (nsamples > rx->cfg.max_total_samples) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_rx11n_process: more samples than max_total_samples", 0);
    for (size_t i = 0; i < ncaps; i++)                                               // the buffer's size is known here: no descriptor may reach past it
        if (caps && (caps[i].offset > nsamples || caps[i].nsamples > nsamples - caps[i].offset)) return sora_internal_fail(SORA_ERR_INVALID_PARAM,
                "a capture descriptor reaches past the end of the sample buffer", 0);
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    // the handle's own sample buffers are shared by its pipelines: calls in flight read them
    for (Pipe11n* p : rx->pipes) if (p) HIPCHK11N(hipStreamSynchronize(p->stream));
    const sora_complex16* src[2] = { h_iq0, h_iq1 };
    for (int k = 0; k < 2; k++) {
        if (!rx->d_iq_own[k]) HIPCHK11N(hipMalloc((void**)&rx->d_iq_own[k], sizeof(sora_complex16) * (rx->cfg.max_total_samples + 64)));
        HIPCHK11N(hipMemcpy(rx->d_iq_own[k], src[k], sizeof(sora_complex16) * nsamples, hipMemcpyHostToDevice));
    }
    return sora_rx11n_process_dev(rx, rx->d_iq_own[0], rx->d_iq_own[1], caps, ncaps);
}

static int pipe11n_results(sora_rx11n_t* rx, Pipe11n* P, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    *nout = 0;
    if (!P->have_results) return sora_internal_fail(SORA_ERR_FAILED, "no process call to report", 0);
    if (P->ncaps == 0) return SORA_OK;
    HIPCHK11N(hipSetDevice(rx->cfg.device));
    HIPCHK11N(hipStreamSynchronize(P->stream));
    const uint32_t mf = rx->cfg.max_frames_per_capture;
    std::vector<Rx11bRow> rows((size_t)P->ncaps * mf); std::vector<uint32_t> nfr(P->ncaps);
    HIPCHK11N(hipMemcpy(rows.data(), P->d_rows, sizeof(Rx11bRow) * rows.size(), hipMemcpyDeviceToHost));
    HIPCHK11N(hipMemcpy(nfr.data(), P->d_nframes, 4 * (size_t)P->ncaps, hipMemcpyDeviceToHost));
    size_t used_rows = 0;
    for (uint32_t c = 0; c < P->ncaps; c++) used_rows += nfr[c] < mf ? nfr[c] : mf;
    std::vector<uint8_t> bulk;
    const size_t slots = (size_t)P->ncaps * mf;
    if (h_mpdu && used_rows > 16 && slots * 4096 <= ((size_t)1 << 30)) {
        bulk.resize(slots * 4096);
        HIPCHK11N(hipMemcpy(bulk.data(), P->d_mpdu, bulk.size(), hipMemcpyDeviceToHost));
    }
    size_t n = 0, moff = 0; int rc = SORA_OK;
    for (uint32_t c = 0; c < P->ncaps; c++)
        for (uint32_t i = 0; i < nfr[c] && i < mf; i++) {
            const Rx11bRow& r = rows[(size_t)c * mf + i];
            if (n >= max_out) { rc = SORA_ERR_CAPACITY; continue; }
            sora_frame_result& o = out[n++];
            memset(&o, 0, sizeof(o));
            o.capture_id = P->h_caps[c].capture_id; o.end_sample = r.end_sample; o.error_code = r.error_code; o.rate_kbps = r.rate_kbps;
            o.length = (uint16_t)r.length; o.crc32 = r.crc32; o.mpdu_offset = (uint32_t)moff;
            if (i + 1 == mf && nfr[c] > mf) o.flags = SORA_ROW_TRUNCATED;             // more frames were found than the capture has rows
            if (h_mpdu && (r.error_code == 1u || r.error_code == 0x80000006u)) {
                const size_t len = r.length < 4096 ? r.length : 4096;
                if (moff + len > mpdu_cap) { rc = SORA_ERR_CAPACITY; continue; }
                if (!bulk.empty()) memcpy(h_mpdu + moff, bulk.data() + ((size_t)c * mf + i) * 4096, len);
                else HIPCHK11N(hipMemcpy(h_mpdu + moff, P->d_mpdu + ((size_t)c * mf + i) * 4096, len, hipMemcpyDeviceToHost));
                moff += len;
            }
        }
    *nout = n;
    if (rc != SORA_OK) return sora_internal_fail(rc, "sora_rx11n_results: output buffer too small", 0);
    return SORA_OK;
}

int sora_rx11n_results(sora_rx11n_t* rx, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (!rx || !nout) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_results: null argument", 0);
    return pipe11n_results(rx, rx->pipes[rx->cur], out, max_out, nout, h_mpdu, mpdu_cap);
}

int sora_rx11n_results_of(sora_rx11n_t* rx, int ticket, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (!rx || !nout) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_rx11n_results_of: null argument", 0);
    Pipe11n* P = pipe11n_of(rx, ticket);
    if (!P) { *nout = 0; return sora_internal_fail(SORA_ERR_INVALID_PARAM,
            "sora_rx11n_results_of: stale ticket (its pipeline has been reused by a later process call, or the ticket was never issued)", 0); }
    return pipe11n_results(rx, P, out, max_out, nout, h_mpdu, mpdu_cap);
}
