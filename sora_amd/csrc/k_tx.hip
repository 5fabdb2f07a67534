// k_tx.hip -- 802.11a transmitter on the GPU: the modulation graph
//   TBB11aSrc -> T11aSc -> TBB11aMRSelect -> TConvEncode_{12,23,34} -> T11aInterleave* -> TMap11a* -> T11aAddPilot
//   -> TIFFTx -> TPackSample16to8 -> TModSink          (kernel/bb/demod11/fb11amod_config.hpp:74-110)
// plus the preamble source (kernel/bb/Brick11/src/preamble11a.hpp:19-140).  Output: COMPLEX8 at 40 MHz, what
// `demod11 -m` writes.  Every stage is data-parallel once restated:
//   * scrambler (scramble.hpp:237-251): the register sequence is a phase of one period-127 cycle -> two table reads
//   * convolutional encoder (conv_enc.hpp:6-14): coded bit = xor of five of the last seven input bits; the puncturing
//     patterns map a coded-bit index to (input bit, generator) in closed form
//   * interleaver (interleave.hpp:43-58): coded bit k -> position j(k), the table the receiver's de-interleaver reads
//   * FCS: the parallel CRC-32 of k_finish
// One 256-thread block per frame; eight OFDM symbols per pass, 32 lanes each (IFFT<128>: 4 points per lane).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace sora {

__device__ __constant__ uint8_t kLtsPos[64] = {              // LTS_Positive_table (ieee80211const.h:23-28)
    0,1,0,0,1,1,0,1,0,1,0,0,0,0,0,1, 1,0,0,1,0,1,0,1,1,1,1,0,0,0,0,0,
    0,0,0,0,0,0,1,1,0,0,1,1,0,1,0,1, 1,1,1,1,1,0,0,1,1,0,1,0,1,1,1,1 };
constexpr uint8_t kPilotSgnTx[128] = {         // pilot.hpp:10-28: 1 <=> polarity -1
    0,0,0,1,1,1,0,1, 1,1,1,0,0,1,0,1, 1,0,0,1,0,0,1,0, 0,0,0,0,0,1,0,0,
    0,1,0,0,1,1,0,0, 0,1,0,1,1,1,0,1, 0,1,1,0,1,1,0,0, 0,0,0,1,1,0,0,1,
    1,0,1,0,1,0,0,1, 1,1,0,0,1,1,1,1, 0,1,1,0,1,0,0,0, 0,1,0,1,0,1,0,1,
    1,1,1,1,0,1,0,0, 1,0,1,0,0,0,1,1, 0,1,1,1,0,0,0,1, 1,1,1,1,1,1,0,0 };

constexpr uint32_t pilot_word(int w) { uint32_t v = 0; for (int j = 0; j < 32; j++) v |= (uint32_t)kPilotSgnTx[32 * w + j] << j; return v; }   // bit n of word n >> 5 = kPilotSgnTx[n]
constexpr uint32_t kPilotW0 = pilot_word(0), kPilotW1 = pilot_word(1), kPilotW2 = pilot_word(2), kPilotW3 = pilot_word(3);
static_assert(kPilotW0 == 0x2049a7b8u && kPilotW3 == 0x3f8ec52fu, "pilot polarity words");
constexpr int kBpskMod = 10720;                              // mapper11a.hpp:8-11
__device__ __forceinline__ int kmod_of(int nb) { return nb == 1 ? kBpskMod : nb == 2 ? (int)(kBpskMod / 1.414) : nb == 4 ? (int)(kBpskMod / 3.162) : (int)(kBpskMod / 6.481); }
__device__ __forceinline__ int sat8(int v) { return min(max(v, -128), 127); }          // _mm_packs_epi16 (stdbrick.hpp:430)

// 160 time samples of one OFDM symbol from its 64 frequency bins (TIFFTx, fft.hpp:21-59): bins 0..31 -> 0..31, 32..63 ->
// 96..127 of a 128-point IFFT, >> 4, GI = last 32, first/last two samples halved, saturating 16 -> 8 bit pack.
// s_bins: 128 words (zero outside the 64 bins); 32 lanes, e = lane of the group.
// Round 6: the samples leave straight from where the IFFT's last stage put them -- time sample n at word brev7(n) (FFT128LUTMap), output sample i of the symbol is
// n = (i + 96) & 127 -- four per lane as one 8-byte store, the shift, the clamp (_mm_packs_epi16, stdbrick.hpp:430) and the byte pick on packed halves.  (Round 5
// un-reversed into a 160-word symbol buffer, copied the GI behind a barrier and stored two bytes per lane and pass.)
// Output sample i of a symbol is time sample n = (i + 96) & 127, which the IFFT's last stage left at word brev7(n).  A lane's four samples in a row, i = 4 e + k, sit at
// brev5((e + 24) & 31) + 32 brev2(k); its one sample of the last 32, i = 128 + e, at 4 brev5(e) + 3.
struct EmitPlan { uint32_t a4, a1; uint32_t sh01, shs; };
__device__ __forceinline__ EmitPlan emit_plan(int e)
{
    EmitPlan P;
    P.a4 = __brev((unsigned)((e + 24) & 31)) >> 27;
    P.a1 = 4u * (__brev((unsigned)e) >> 27) + 3u;
    P.sh01 = e == 0 ? 0x00050005u : 0x00040004u;                                 // samples 0, 1 ...
    P.shs = e >= 30 ? 0x00050005u : 0x00040004u;                                 // ... and 158, 159 are halved
    return P;
}
__device__ __forceinline__ uint32_t pk_sra_clamp8(uint32_t v, uint32_t sh)
{
    const s16x2_t lo = { (short)-128, (short)-128 }, hi = { (short)127, (short)127 };
    const s16x2_t x = __builtin_bit_cast(s16x2_t, v) >> __builtin_bit_cast(s16x2_t, sh);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(x, lo), hi));
}
template <typename SYNC>
__device__ __forceinline__ void ifft_emit(uint32_t* s_bins, int e, const Fft128Tw& tw, const EmitPlan& P, int8_t* out8, SYNC sync)
{
    pcx x[4];
    sync();
#pragma unroll
    for (int m = 0; m < 4; m++) x[m] = s_bins[e + 32 * m];
    ifft128_core_pk(x, s_bins, e, tw, sync);                                     // IFFT<128> on packed COMPLEX16 (bit-exact with fft128_core<true>)
    if (out8 == nullptr) return;                                                 // (a group past the last symbol only keeps the barriers company)
    if ((reinterpret_cast<uintptr_t>(out8) & 7u) == 0) {
        const uint32_t w0 = pk_sra_clamp8(s_bins[P.a4], P.sh01), w1 = pk_sra_clamp8(s_bins[P.a4 + 64], P.sh01);
        const uint32_t w2 = pk_sra_clamp8(s_bins[P.a4 + 32], 0x00040004u), w3 = pk_sra_clamp8(s_bins[P.a4 + 96], 0x00040004u), w4 = pk_sra_clamp8(s_bins[P.a1], P.shs);
        uint2 o;                                                                 // low bytes of (re0, im0, re1, im1)
        o.x = __builtin_amdgcn_perm(w1, w0, 0x06040200u); o.y = __builtin_amdgcn_perm(w3, w2, 0x06040200u);
        reinterpret_cast<uint2*>(out8)[e] = o;
        reinterpret_cast<uint16_t*>(out8)[128 + e] = (uint16_t)__builtin_amdgcn_perm(0u, w4, 0x0c0c0200u);
    } else {                                                                     // (a frame the caller placed at a sample offset that is not a multiple of four)
        for (int i = e; i < 160; i += 32) {
            const uint32_t w = pk_sra_clamp8(s_bins[__brev((unsigned)((i + 96) & 127)) >> 25], (i < 2 || i >= 158) ? 0x00050005u : 0x00040004u);
            reinterpret_cast<uint16_t*>(out8)[i] = (uint16_t)__builtin_amdgcn_perm(0u, w, 0x0c0c0200u);
        }
    }
}

// The 640-sample preamble (preamble11a.hpp:19-100), computed once per device into a table.
__global__ void __launch_bounds__(64) k_tx_preamble(int8_t* out8, Tables T)
{
    __shared__ uint32_t s_f[2][128];
    __shared__ uint32_t s_t[2][128];
    __shared__ uint32_t s_lut[640];
    const int g = threadIdx.x >> 5, e = threadIdx.x & 31;
    auto sync = []() { __syncthreads(); };
    for (int i = e; i < 128; i += 32) s_f[g][i] = 0;
    sync();
    if (g == 0 && e == 0) {                                                      // short training symbol: 12 carriers
        const int m = (int)(uint16_t)(1.0 * kBpskMod * 1.472);
        const int idx[12] = { 4, 8, 12, 16, 20, 24, 104, 108, 112, 116, 120, 124 };
        const int sg[12]  = { -1, -1, 1, 1, 1, 1, 1, -1, 1, -1, -1, 1 };
        for (int k = 0; k < 12; k++) { const int v = w16(sg[k] * m); s_f[0][idx[k]] = pack(mk(v, v)); }
    }
    if (g == 1) {                                                                // long training symbol
        for (int i = 1 + e; i <= 26; i += 32) s_f[1][i] = pack(mk(kLtsPos[i] ? kBpskMod : -kBpskMod, 0));
        for (int i = 64 - 26 + e; i < 64; i += 32) s_f[1][i + 64] = pack(mk(kLtsPos[i] ? kBpskMod : -kBpskMod, 0));
    }
    sync();
    cpx x[4], y[4];
#pragma unroll
    for (int m = 0; m < 4; m++) x[m] = unpack(s_f[g][e + 32 * m]);
    fft128_group<true>(x, y, s_f[g], e, T, sync);
#pragma unroll
    for (int q = 0; q < 4; q++) s_t[g][e + 32 * q] = pack(sra(y[q], 4));
    sync();
    // STS: 128 samples repeated periodically over 320; LTS: GI2 (last 64 of the symbol) + two copies of 128
    for (int i = threadIdx.x; i < 320; i += 64) s_lut[i] = s_t[0][i & 127];
    for (int i = threadIdx.x; i < 256; i += 64) s_lut[320 + 64 + i] = s_t[1][i & 127];
    for (int i = threadIdx.x; i < 64; i += 64) s_lut[320 + i] = s_t[1][64 + i];
    sync();
    for (int i = threadIdx.x; i < 640; i += 64) {
        cpx v = unpack(s_lut[i]);
        if (i == 0 || i == 1 || i == 318 || i == 319 || i == 320 || i == 321 || i == 638 || i == 639) v = sra(v, 1);
        out8[2 * i] = (int8_t)sat8(v.re); out8[2 * i + 1] = (int8_t)sat8(v.im);
    }
}

__global__ void __launch_bounds__(256, 8) k_tx11a(TxArgs A)
{
    __shared__ alignas(4) uint8_t s_data[2608];
    // generator outputs A (133) / B (171) of the whole data field, bit i of the stream = bit i & 31 of word i >> 5
    __shared__ uint32_t s_gab[2][656];
    uint32_t* const s_ga = s_gab[0]; uint32_t* const s_gb = s_gab[1];
    __shared__ uint32_t s_crc[256];
    __shared__ uint32_t s_z[6 * 8 * 16];
    __shared__ uint32_t s_bins[8][128];
    // the interleaver inverted: position -> coded bit of the symbol, for the frame's modulation and for the SIGNAL symbol (BPSK)
    __shared__ uint16_t s_inv[288 + 48];
    __shared__ uint32_t s_fcs;
    // interleaver positions of the frame's modulation, then of the SIGNAL symbol (BPSK)
    __shared__ uint16_t s_map[288 + 48];
    const uint32_t f = blockIdx.x;
    const int tid = threadIdx.x, g = tid >> 5, e = tid & 31;
    const Tables& T = A.T;
    const uint32_t L = A.len[f], kbps = A.rate[f];
    const uint8_t* mp = A.mpdu + A.off[f];
    int8_t* out = A.out8 + A.out_off[f] * 2;
    int nb, cr, nd, rc;
    switch (kbps) {                                                              // ieee80211a_cmn.h:65-149, ieee80211const.h:3-10
    case 6000:  nb = 1; cr = 0; nd = 24;  rc = 0xB; break;  case 9000:  nb = 1; cr = 2; nd = 36;  rc = 0xF; break;
    case 12000: nb = 2; cr = 0; nd = 48;  rc = 0xA; break;  case 18000: nb = 2; cr = 2; nd = 72;  rc = 0xE; break;
    case 24000: nb = 4; cr = 0; nd = 96;  rc = 0x9; break;  case 36000: nb = 4; cr = 2; nd = 144; rc = 0xD; break;
    case 48000: nb = 6; cr = 1; nd = 192; rc = 0x8; break;  default:    nb = 6; cr = 2; nd = 216; rc = 0xC; break;
    }
    // TBB11aSrc::Process (PHY_11a.hpp:132-202): SERVICE(2) + MPDU + FCS(4) + tail(1) + pad; rate 9 pads to two symbols
    const uint32_t ndp = kbps == 9000 ? (uint32_t)nd * 2 : (uint32_t)nd;
    const uint32_t dbytes = 2 + (L + 4) + 1;
    const uint32_t rem = (dbytes * 8) % ndp, pad_bits = rem ? ndp - rem : 0;
    const uint32_t nbytes = dbytes + (pad_bits + 7) / 8;
    const uint32_t nsym = nbytes * 8 / (uint32_t)nd;

    s_crc[tid] = T.crc[tid];
    for (int i = tid; i < 6 * 8 * 16; i += 256) s_z[i] = T.crcz[i];
    // (+ 8: the word-wise encoder reads up to 3 bytes past nbytes)
    for (uint32_t i = tid; i < nbytes + 8; i += 256) s_data[i] = (i >= 2 && i < 2 + L) ? mp[i - 2] : (uint8_t)0;
    __syncthreads();
    if (tid < 64) {                                                              // FCS of the MPDU (PHY_11a.hpp:87,160-170)
        uint32_t crc;
        if (L >= 4) crc = crc32_wave(s_data + 2, (int)L, s_crc, s_z, tid);
        else { crc = 0xFFFFFFFFu; for (uint32_t i = 0; i < L; i++) crc = (crc >> 8) ^ s_crc[(s_data[2 + i] ^ crc) & 0xFF]; }
        if (tid == 0) s_fcs = ~crc;
    }
    __syncthreads();
    if (tid < 4) s_data[2 + L + tid] = (uint8_t)(s_fcs >> (8 * tid));
    __syncthreads();
    {   // T11aSc (scramble.hpp:237-251): register = previous 8 output bits; the tail byte keeps only its two pad bits
        const unsigned s7 = A.seed[f] >> 1;
        const unsigned phase = T.scr_phase[s7];                                  // 255: the all-zero state stays zero
        for (uint32_t i = tid; i < nbytes; i += 256) {
            unsigned c = s_data[i] ^ (phase == 255 ? 0u : T.scr_seq[(phase + 8u * i) % 127u]);
            if (i == dbytes - 1) c &= 0xC0u;
            s_data[i] = (uint8_t)c;
        }
    }
    for (int i = tid; i < 640; i += 256) reinterpret_cast<uint16_t*>(out)[i] = reinterpret_cast<const uint16_t*>(A.preamble)[i];
    __syncthreads();
    // TConvEncode_* (conv_enc.hpp:6-14) 32 input bits at a time: A = x ^ x>>2 ^ x>>3 ^ x>>5 ^ x>>6, B = x ^ x>>1 ^ x>>2 ^ x>>3 ^ x>>6 over the bit
    // stream (x>>k = the bit k positions EARLIER: shifted in from the previous word; the encoder starts from state 0)
    {
        const uint32_t* dw = reinterpret_cast<const uint32_t*>(s_data);
        for (uint32_t w = tid; w < (nbytes + 3) / 4; w += 256) {
            const uint32_t X = dw[w], P = w ? dw[w - 1] : 0u;
            auto sh = [&](int k) { return (X << k) | (P >> (32 - k)); };
            const uint32_t x2 = sh(2), x3 = sh(3), x6 = sh(6);
            s_ga[w] = X ^ x2 ^ x3 ^ sh(5) ^ x6;
            s_gb[w] = X ^ sh(1) ^ x2 ^ x3 ^ x6;
        }
    }
    __syncthreads();

    // PLCP SIGNAL (ieee80211a_cmn.h:8-26): RATE, LENGTH, even parity
    uint32_t sig = (uint32_t)rc | ((L + 4) << 5);
    sig |= (uint32_t)(__popc(sig) & 1) << 17;
    // From here on every LDS slice is private to a 32-lane group (half a wave): a wave-level barrier orders what the groups of a wave
    // write and read, the waves of the block run free of each other (a block barrier per stage of every pass used to hold them together).
    for (int i = tid; i < 48 * nb; i += 256) s_map[i] = T.deint[(nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3) * 288 + i];
    if (tid < 48) s_map[288 + tid] = T.deint[tid];
    const Fft128Tw tw = fft128_twiddles(T, e);
    __syncthreads();
    for (int k = tid; k < 48 * nb; k += 256) s_inv[s_map[k]] = (uint16_t)k;
    if (tid < 48) s_inv[288 + s_map[288 + tid]] = (uint16_t)tid;
    __syncthreads();
    // Round 6: the mapper reads its bits where the encoder left them.  A symbol is 96 components (carrier c, I or Q; 48 for BPSK), three per lane of the symbol's 32:
    // component q = e + 32 t.  Its M bits sit at interleaved positions c N_BPSC + h M + m, i.e. are coded bits k = inverse(position) of the symbol, and coded bit k of
    // a symbol is generator `which` at input bit (s - 1) N_DBPS + il -- (il, which) follow from k and the puncturing pattern and do NOT depend on the symbol (N_CBPS is a
    // whole number of puncture periods): nine bits per entry, three entries per component, held in one register per component.  (Round 5 wrote every coded bit into a byte
    // array at its interleaved position and read the bytes back: 9 + 9 LDS byte accesses and their address arithmetic per lane and symbol.)
    const int M = nb == 1 ? 1 : nb / 2;
    auto entry_of = [&](int k) -> uint32_t {                                    // coded bit k of a data symbol -> il | which << 8
        int il, which;
        if (cr == 0) { il = k >> 1; which = k & 1; }
        else if (cr == 1) { const int q3 = k / 3, r = k - 3 * q3; il = 2 * q3 + (r == 2); which = r == 1; }
        else { const int q4 = k >> 2, r = k & 3; il = 3 * q4 + (r == 2 ? 1 : r == 3 ? 2 : 0); which = r & 1; }
        return (uint32_t)il | ((uint32_t)which << 8);
    };
    // (registers, not a packed word: the loop below neither unpacks nor recomputes anything that depends on the lane alone)
    // bit offset within s_gab of each of the component's bits at symbol 0: input bit within the symbol + 656 * 32 for generator B (a whole number of words)
    uint32_t il[3][3], ES[2] = { 0, 0 };
    uint32_t cw[3];                                                              // byte address of the component's 16-bit half in the symbol's bins
    // TMap11a* + T11aAddPilot (mapper11a.hpp, pilot.hpp:76-118): carriers in the order -26..-1, +1..+26 without pilots; TIFFTx: bins 32..63 go to 96..127
    auto bin_of = [](int c) { int bin; if (c < 24) { bin = 38 + c; if (bin >= 43) bin++; if (bin >= 57) bin++; } else { bin = 1 + (c - 24); if (bin >= 7) bin++; if (bin >= 21) bin++; }
                              return bin < 32 ? bin : bin + 64; };
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int q = e + 32 * t;
#pragma unroll
        for (int m = 0; m < 3; m++) il[t][m] = 0;
        if (nb == 1) {
            if (t < 2 && q < 48) { const uint32_t en = entry_of(s_inv[q]); il[t][0] = (en & 255u) + (en >> 8) * (656u * 32u); }
            cw[t] = (uint32_t)bin_of(t < 2 && q < 48 ? q : 0) * 4u;
        } else {
            const int c = q >> 1, h = q & 1;
#pragma unroll
            for (int m = 0; m < 3; m++) if (m < M) { const uint32_t en = entry_of(s_inv[c * nb + h * M + m]); il[t][m] = (en & 255u) + (en >> 8) * (656u * 32u); }
            cw[t] = (uint32_t)bin_of(c) * 4u + 2u * (uint32_t)h;
        }
        if (t < 2 && q < 48) { const int k = s_inv[288 + q]; ES[t] = (uint32_t)(k >> 1) | ((uint32_t)(k & 1) << 8); }
    }
    const EmitPlan plan = emit_plan(e);
    const int kmod = kmod_of(nb), lvl0 = -((1 << M) - 1) * kmod, kmod2 = 2 * kmod;
    auto sync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    const uint32_t total = 1 + nsym;                                             // SIGNAL + data symbols
    const uint32_t* const gab = &s_gab[0][0];
    char* const bins = reinterpret_cast<char*>(s_bins[g]);
    for (uint32_t s0 = 0; s0 < total; s0 += 8) {
        const uint32_t s = s0 + (uint32_t)g;
        const bool active = s < total;
        const bool is_sig = s == 0;
        for (int i = e; i < 128; i += 32) s_bins[g][i] = 0;
        if (active) {
            auto gen_bit = [&](uint32_t idx) -> uint32_t { return (gab[idx >> 5] >> (idx & 31u)) & 1u; };      // (idx >= 656 * 32: generator B's stream)
            if (is_sig) {
                // the SIGNAL symbol: rate 1/2 over the 24 header bits (encoder state 0), BPSK
                const uint32_t A_ = sig ^ (sig << 2) ^ (sig << 3) ^ (sig << 5) ^ (sig << 6), B_ = sig ^ (sig << 1) ^ (sig << 2) ^ (sig << 3) ^ (sig << 6);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int c = e + 32 * t;
                    if (c < 48) { const unsigned bit = (((ES[t] >> 8) ? B_ : A_) >> (ES[t] & 255u)) & 1u; s_bins[g][bin_of(c)] = pack(mk(bit ? kBpskMod : -kBpskMod, 0)); }
                }
            } else {
                const uint32_t ibase = (s - 1u) * (uint32_t)nd;
                if (nb == 1) {
#pragma unroll
                    for (int t = 0; t < 2; t++)
                        if (e + 32 * t < 48) *reinterpret_cast<uint32_t*>(bins + cw[t]) = pack(mk(gen_bit(ibase + il[t][0]) ? kBpskMod : -kBpskMod, 0));
                } else {
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        unsigned v = 0;                                         // the component's bits, first-transmitted = MSB (InitQamMapLut's reversal, mapper11a.hpp:16-43)
#pragma unroll
                        for (int m = 0; m < 3; m++) if (m < M) v |= gen_bit(ibase + il[t][m]) << (M - 1 - m);
                        unsigned bb = v ^ (v >> 1); bb ^= bb >> 2;                // Gray -> binary (M <= 3)
                        *reinterpret_cast<uint16_t*>(bins + cw[t]) = (uint16_t)((int)bb * kmod2 + lvl0);
                    }
                }
            }
            if (e < 4) {
                // pilot.hpp:10-28 as four words, bit n = 1 <=> polarity -1 at index n; m_PilotIndex 127 -> 0 after SIGNAL (pilot.hpp:66-69)
                const unsigned pidx = is_sig ? 127u : (unsigned)((s - 1) % 127u);
                const uint32_t pw = pidx < 64 ? (pidx < 32 ? kPilotW0 : kPilotW1) : (pidx < 96 ? kPilotW2 : kPilotW3);
                const int p = (pw >> (pidx & 31u)) & 1u ? -kBpskMod : kBpskMod;
                const int bin = e == 0 ? 7 : e == 1 ? 21 : e == 2 ? 64 - 7 : 64 - 21;
                s_bins[g][bin < 32 ? bin : bin + 64] = pack(mk(e == 1 ? -p : p, 0));
            }
        }
        ifft_emit(s_bins[g], e, tw, plan, active ? out + 2 * (640 + 160 * (size_t)s) : (int8_t*)nullptr, sync);
        sync();
    }
}

}  // namespace sora
