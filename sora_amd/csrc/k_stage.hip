// k_stage.hip -- stand-alone, batched versions of the remaining receive-path bricks, one C entry point each
// (include/sora_hip.h), so every stage can be parity-tested on its own and sit behind a BRICK adapter:
//   k_lts_batch       T11aLTS                                        (channel_11a.hpp:33-230)
//   k_symfront_batch  T11aDataSymbol + TFreqCompensation + TFFT64 + TChannelEqualization
//   k_ptrack_batch    TPhaseCompensate + TPilotTrack, symbol by symbol (freqoffset.hpp:14-66, pilot.hpp:121-269)
//   k_fft128_batch    FFT<128>   (core/inc/fft_r4dif.h: 128 = 4 x 32, 32 = 4 x 8, 8-point terminal stage)
//   k_fft64_batch     TFFT64, k_demap_batch T11aDemap<N>, k_deint_batch T11aDeinterleave*
//
// The per-symbol bricks (FFT, symbol front end, demap, de-interleave, ingest) are STREAMING kernels: a few hundred bytes in,
// a few hundred out, a few hundred integer operations per symbol -- bound by HBM.  They share one shape: a block owns tiles
// of consecutive symbols; every global access is a 16-byte-per-lane coalesced load or store of a contiguous tile (the
// symbol-internal reshuffling -- 4 points per lane for the butterflies, carrier order, the interleaver permutation --
// happens in LDS); and a thread issues the loads of ALL its tiles before it touches the first, so that every CU keeps
// >= 64 KB of reads in flight (6.3 TB/s x ~2 us of latency = ~50 KB per CU is what the chip needs to stay busy).
// Pointers given to these entry points must be 16-byte aligned (any hipMalloc'd buffer is).
#include "kernels.h"

namespace sora {

__device__ __constant__ uint8_t kLtsSeqS[64] = {        // LTS_Sequence_11a (channel_11a.hpp:13-18)
    0,1,0,0,1,1,0,1,0,1,0,0,0,0,0,1, 1,0,0,1,0,1,0,1,1,1,1,0,0,0,0,0,
    0,0,0,0,0,0,1,1,0,0,1,1,0,1,0,1, 1,1,1,1,1,0,0,1,1,0,1,0,1,1,1,1 };
__device__ __constant__ uint8_t kPilotSgnS[128] = {     // pilot.hpp:10-28: 1 <=> polarity -1
    0,0,0,1,1,1,0,1, 1,1,1,0,0,1,0,1, 1,0,0,1,0,0,1,0, 0,0,0,0,0,1,0,0,
    0,1,0,0,1,1,0,0, 0,1,0,1,1,1,0,1, 0,1,1,0,1,1,0,0, 0,0,0,1,1,0,0,1,
    1,0,1,0,1,0,0,1, 1,1,0,0,1,1,1,1, 0,1,1,0,1,0,0,0, 0,1,0,1,0,1,0,1,
    1,1,1,1,0,1,0,0, 1,0,1,0,0,0,1,1, 0,1,1,1,0,0,0,1, 1,1,1,1,1,1,0,0 };

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = (int)((unsigned)v + (unsigned)__shfl_xor(v, o));
    return v;
}

// sora_lts11a_ctx: { int16 cfo_est; int16 reserved; COMPLEX16 freq[64]; COMPLEX16 chan[64]; } = 129 words
__global__ void __launch_bounds__(64) k_lts_batch(const uint32_t* in, uint32_t* ctx, uint32_t n, Tables T)
{
    __shared__ uint32_t s_fft[64];
    __shared__ uint32_t s_x[144];
    const uint32_t f = blockIdx.x;
    if (f >= n) return;
    const int lane = threadIdx.x;
    auto sync = []() { __syncthreads(); };
    for (int i = lane; i < 144; i += 64) s_x[i] = in[(size_t)f * 144 + i];
    sync();
    cpx x1 = sra(unpack(s_x[8 + lane]), 1);                                    // skip_cp = 8; first 64 samples >> 1 (:211-216)
    cpx x2 = unpack(s_x[8 + 64 + lane]);
    int re, im; conj_mul32(x2, x1, re, im);                                    // FreqOffsetEstimate<16> (dspalg.hpp:226-243)
    const int sum_re = wave_sum_i(re >> 5), sum_im = wave_sum_i(im >> 5);
    const int cfo = uatan2(T, sum_im, sum_re) >> 6;          // size_t divisor: unsigned division = floor (dspalg.hpp:242)
    const cpx fc = rot_coeff(T, w16(lane * cfo));                               // BuildFrequencyShiftCoeffs<64> (dspalg.hpp:200-208)
    uint32_t* o = ctx + (size_t)f * 129;
    if (lane == 0) o[0] = (uint32_t)cfo & 0xFFFFu;
    o[1 + lane] = pack(fc);
    cpx xs = mul_q15(x1, fc);                                                   // FrequencyShift (:120)
    sync();
    s_x[lane] = pack(xs);
    sync();
    cpx Y[4];
    {
        const int e = lane & 15;
        cpx xin[4];
#pragma unroll
        for (int m = 0; m < 4; m++) xin[m] = unpack(s_x[e + 16 * m]);
        fft64_group(xin, Y, s_fft, e, T, sync);
    }
    cpx Yb = (lane >> 4) == 0 ? Y[0] : (lane >> 4) == 1 ? Y[1] : (lane >> 4) == 2 ? Y[2] : Y[3];
    uint32_t coef = 0;
    if (!(lane >= 28 && lane < 36)) {                                           // _channel_estimation (:125-178)
        const int e = sqnorm(Yb) >> 8;
        const cpx L = mk(kLtsSeqS[lane] ? 1600 : -1600, 0);
        int cre, cim; conj_mul32(L, Yb, cre, cim);
        int rre = 0, rim = 0;
        if (e != 0) { rre = cre / e; rim = cim / e; }
        coef = pack(mk(w16(rre), w16(rim)));
    }
    o[65 + lane] = coef;
}

constexpr int kFftTiles = 4;                                                 // tiles of 16 symbols per 256-thread block

// T11aDataSymbol -> TFreqCompensation -> TFFT64 -> TChannelEqualization.  16 lanes per symbol, 16 symbols per tile; the 64
// samples behind the cyclic prefix arrive as one 16-byte load per lane (samples 4e..4e+3), go through the group's LDS
// slice to the 4-points-per-lane layout of the butterflies, and the equalised bins 4e..4e+3 leave as one 16-byte store.
__global__ void __launch_bounds__(256) k_symfront_batch(const uint32_t* __restrict__ in, const uint32_t* __restrict__ ctx,
        const uint32_t* __restrict__ ctx_index, uint32_t* __restrict__ eq, uint32_t n, Tables T)
{
    __shared__ uint32_t s_all[16][64];
    const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    uint32_t* s = s_all[g];
    uint4 v[kFftTiles]; uint32_t ci[kFftTiles];
#pragma unroll
    for (int t = 0; t < kFftTiles; t++) {
        const uint32_t i = (blockIdx.x * kFftTiles + t) * 16 + g;
        v[t] = i < n ? reinterpret_cast<const uint4*>(in)[(size_t)i * 20 + 2 + e] : uint4{0, 0, 0, 0};   // sample i*80 + 8 + 4e (skip_cp = 8)
        ci[t] = (i < n && ctx_index) ? ctx_index[i] : 0u;
    }
#pragma unroll
    for (int t = 0; t < kFftTiles; t++) {
        const uint32_t i = (blockIdx.x * kFftTiles + t) * 16 + g;
        const uint32_t* c = ctx + (size_t)ci[t] * 129;
        uint32_t fq[4], ch[4];                                                       // the frame's FreqCoeffs / ChannelCoeffs this lane needs (L2-resident)
#pragma unroll
        for (int m = 0; m < 4; m++) { fq[m] = c[1 + e + 16 * m]; ch[m] = c[65 + 4 * e + m]; }
        wave_lds_sync();
        reinterpret_cast<uint4*>(s)[e] = v[t];
        wave_lds_sync();
        pcx x[4];
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = pk_cmul<15>(pk_sra(s[e + 16 * m], 1), pk_tw_mul(fq[m]));   // >>1, x FreqCoeffs (channel_11a.hpp:643-644)
        fft64_core_pk(x, s, e, W, wave_lds_sync);
        uint32_t o[4];
        const unsigned r = __brev((unsigned)e) >> 28;                                // bin 4e+q sits at slot bitrev6(4e+q) = bitrev4(e) + 16 bitrev2(q)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int bin = 4 * e + q;
            o[q] = (bin >= 28 && bin < 36) ? 0u : pk_cmul<8>(s[r + 16u * ((q & 1) * 2 + (q >> 1))], pk_tw_mul(ch[q]));   // channel_11a.hpp:548-574
        }
        if (i < n) reinterpret_cast<uint4*>(eq)[(size_t)i * 16 + e] = uint4{o[0], o[1], o[2], o[3]};
    }
}

// The three one-multiply bricks of the symbol chain on their own (in the receive path they are fused into k_frame / k_sym_front / k_sym_back):
//   KIND 0  TFreqCompensation      out = ((in >> 1) x FreqCoeffs) >> 15, wrapping pack                  (channel_11a.hpp:642-644, dspalg.hpp:219-224)
//   KIND 1  TChannelEqualization   out = (in x ChannelCoeffs) >> 8, wrapping pack; bins 28 .. 35 = 0     (channel_11a.hpp:548-574)
//   KIND 2  TPhaseCompensate       out = (in x CompCoeffs) >> 15, wrapping pack                          (freqoffset.hpp:28-30)
// Symbol i multiplies by the 64 coefficients at coef + cstride x cindex[i] + coff words (cindex == nullptr: set 0): a frame's coefficients serve all its symbols, so
// a symbol is 256 bytes in and 256 out.  16 lanes per symbol, one 16-byte load and store per lane, every load of a thread's tiles in flight before the first product.
template <int KIND>
__global__ void __launch_bounds__(256) k_cmul64_batch(const uint32_t* __restrict__ in, const uint32_t* __restrict__ coef, uint32_t cstride, uint32_t coff,
        const uint32_t* __restrict__ cindex,
                                                      uint32_t* __restrict__ out, uint32_t n)
{
    constexpr int kTiles = 8;
    const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
    uint4 v[kTiles]; uint32_t ci[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; t++) {
        const uint32_t i = (blockIdx.x * kTiles + t) * 16 + g;
        v[t] = i < n ? reinterpret_cast<const uint4*>(in)[(size_t)i * 16 + e] : uint4{0, 0, 0, 0};
        ci[t] = (i < n && cindex) ? cindex[i] : 0u;
    }
#pragma unroll
    for (int t = 0; t < kTiles; t++) {
        const uint32_t i = (blockIdx.x * kTiles + t) * 16 + g;
        // (a context's coefficient arrays start on a word, not on 16 bytes: four word loads, L2-resident)
        const uint32_t* c = coef + (size_t)ci[t] * cstride + coff + 4 * e;
        const uint32_t x[4] = { v[t].x, v[t].y, v[t].z, v[t].w };
        // bins 28 .. 35 = lanes 7 and 8 of the symbol: zero (channel_11a.hpp:545-546)
        const uint32_t keep = (e == 7 || e == 8) ? 0u : 0xFFFFFFFFu;
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const PkTw w = pk_tw_mul(c[q]);
            if (KIND == 0)      o[q] = pk_cmul<15>(pk_sra(x[q], 1), w);
            // (a mask, not a branch: the select on 4e + q made the compiler fetch every coefficient under its own exec mask)
            else if (KIND == 1) o[q] = pk_cmul<8>(x[q], w) & keep;
            else                o[q] = pk_cmul<15>(x[q], w);
        }
        if (i < n) reinterpret_cast<uint4*>(out)[(size_t)i * 16 + e] = uint4{o[0], o[1], o[2], o[3]};
    }
}
template __global__ void k_cmul64_batch<0>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t*, uint32_t);
template __global__ void k_cmul64_batch<1>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t*, uint32_t);
template __global__ void k_cmul64_batch<2>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t*, uint32_t);

// TFFT64: the same shape without the context.
__global__ void __launch_bounds__(256) k_fft64_batch(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, Tables T)
{
    __shared__ uint32_t s_all[16][64];
    const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
    const Fft64TwPk W = fft64_twiddles_pk(T, e);
    uint32_t* s = s_all[g];
    uint4 v[kFftTiles];
#pragma unroll
    for (int t = 0; t < kFftTiles; t++) {
        const uint32_t i = (blockIdx.x * kFftTiles + t) * 16 + g;
        v[t] = i < n ? reinterpret_cast<const uint4*>(in)[(size_t)i * 16 + e] : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int t = 0; t < kFftTiles; t++) {
        const uint32_t i = (blockIdx.x * kFftTiles + t) * 16 + g;
        wave_lds_sync();
        reinterpret_cast<uint4*>(s)[e] = v[t];
        wave_lds_sync();
        pcx x[4];
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = s[e + 16 * m];
        fft64_core_pk(x, s, e, W, wave_lds_sync);
        const unsigned r = __brev((unsigned)e) >> 28;
        if (i < n) reinterpret_cast<uint4*>(out)[(size_t)i * 16 + e] = uint4{s[r], s[r + 32], s[r + 16], s[r + 48]};
    }
}

// T11aDemap<NB>::Filter (demapper11a.hpp:10-79 over DemapperCore, demapper.h:16-45): a tile of 32 symbols is loaded into
// LDS as it lies in memory, every thread demaps (symbol, carrier) items out of LDS through the step tables (LDS too) and
// writes the NB soft values of the carrier into the tile's output image, which leaves with 16-byte stores.
constexpr int kDmSyms = 32;
template <int NB>
__global__ void __launch_bounds__(256) k_demap_batch(const uint32_t* __restrict__ in, uint8_t* __restrict__ soft, uint32_t n, Tables T)
{
    constexpr int NCB = 48 * NB;
    __shared__ uint32_t s_in[kDmSyms * 64];
    __shared__ uint32_t s_out[kDmSyms * NCB / 4];
    __shared__ uint8_t  s_lut[1024];
    const int tid = threadIdx.x;
    const uint32_t s0 = blockIdx.x * kDmSyms;
    const int ns = (int)min((uint32_t)kDmSyms, n - s0);
    const uint4* in4 = reinterpret_cast<const uint4*>(in) + (size_t)s0 * 16;
    const uint4 a = tid < ns * 16 ? in4[tid] : uint4{0, 0, 0, 0};
    const uint4 b = tid + 256 < ns * 16 ? in4[tid + 256] : uint4{0, 0, 0, 0};
    reinterpret_cast<uint32_t*>(s_lut)[tid] = reinterpret_cast<const uint32_t*>(T.demap)[tid];
    reinterpret_cast<uint4*>(s_in)[tid] = a; reinterpret_cast<uint4*>(s_in)[tid + 256] = b;
    __syncthreads();
    uint16_t* o16 = reinterpret_cast<uint16_t*>(s_out);
    uint8_t* o8 = reinterpret_cast<uint8_t*>(s_out);
    for (int it = tid; it < ns * 48; it += 256) {
        const int sym = it / 48, k = it - sym * 48;
        const cpx v = unpack(s_in[sym * 64 + carrier_bin48(k)]);
        int re = v.re >> 4, im = v.im >> 4;                                       // demap_limit<64> (demapper.h:141-151)
        re = min(max(re, -128), 127); im = min(max(im, -128), 127);
        const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
        const int at = sym * NCB + k * NB;                                        // byte position of the carrier's first soft value
        if (NB == 1) o8[at] = s_lut[ur];
        else if (NB == 2) o16[at >> 1] = (uint16_t)(s_lut[ur] | (s_lut[ui] << 8));
        else if (NB == 4) s_out[at >> 2] = (uint32_t)s_lut[ur] | ((uint32_t)s_lut[256 + ur] << 8) | ((uint32_t)s_lut[ui] << 16) | ((uint32_t)s_lut[256 + ui] << 24);
        else {
            o16[(at >> 1)]     = (uint16_t)(s_lut[ur] | (s_lut[512 + ur] << 8));
            o16[(at >> 1) + 1] = (uint16_t)(s_lut[768 + ur] | (s_lut[ui] << 8));
            o16[(at >> 1) + 2] = (uint16_t)(s_lut[512 + ui] | (s_lut[768 + ui] << 8));
        }
    }
    __syncthreads();
    uint4* o4 = reinterpret_cast<uint4*>(soft + (size_t)s0 * NCB);                 // 48 NB bytes per symbol: every tile starts 16-byte aligned
    for (int k = tid; k < ns * NCB / 16; k += 256) o4[k] = reinterpret_cast<const uint4*>(s_out)[k];
}

// T11aDeinterleave{BPSK,QPSK,QAM16,QAM64} (deinterleaver.hpp): out[k] = in[j(k)] within a symbol.  A tile of 32 symbols is
// loaded into LDS with 16-byte loads; a thread gathers the 16 bytes of one output quad-word through the map and stores it.
template <int NB>
__global__ void __launch_bounds__(256) k_deint_batch(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n, Tables T)
{
    constexpr int NCB = 48 * NB, QW = NCB / 16, TILE_QW = kDmSyms * QW, PER = (TILE_QW + 255) / 256;
    __shared__ uint32_t s_in[kDmSyms * NCB / 4];
    __shared__ uint16_t s_map[NCB];
    const int tid = threadIdx.x;
    const uint32_t s0 = blockIdx.x * kDmSyms;
    const int ns = (int)min((uint32_t)kDmSyms, n - s0), nqw = ns * QW;
    const uint4* in4 = reinterpret_cast<const uint4*>(in + (size_t)s0 * NCB);
    uint4 v[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) v[j] = tid + 256 * j < nqw ? in4[tid + 256 * j] : uint4{0, 0, 0, 0};
    constexpr int di = NB == 1 ? 0 : NB == 2 ? 1 : NB == 4 ? 2 : 3;
    for (int i = tid; i < NCB; i += 256) s_map[i] = T.deint[di * 288 + i];
#pragma unroll
    for (int j = 0; j < PER; j++) if (tid + 256 * j < TILE_QW) reinterpret_cast<uint4*>(s_in)[tid + 256 * j] = v[j];
    __syncthreads();
    const uint8_t* b = reinterpret_cast<const uint8_t*>(s_in);
    uint4* o4 = reinterpret_cast<uint4*>(out + (size_t)s0 * NCB);
    for (int q = tid; q < nqw; q += 256) {
        const int sym = q / QW, w = q - sym * QW;
        const uint8_t* src = b + sym * NCB;
        const uint16_t* m = s_map + 16 * w;
        uint32_t r[4];
#pragma unroll
        for (int d = 0; d < 4; d++)
            r[d] = (uint32_t)src[m[4 * d]] | ((uint32_t)src[m[4 * d + 1]] << 8) | ((uint32_t)src[m[4 * d + 2]] << 16) | ((uint32_t)src[m[4 * d + 3]] << 24);
        o4[q] = uint4{r[0], r[1], r[2], r[3]};
    }
}
template __global__ void k_demap_batch<1>(const uint32_t*, uint8_t*, uint32_t, Tables);
template __global__ void k_demap_batch<2>(const uint32_t*, uint8_t*, uint32_t, Tables);
template __global__ void k_demap_batch<4>(const uint32_t*, uint8_t*, uint32_t, Tables);
template __global__ void k_demap_batch<6>(const uint32_t*, uint8_t*, uint32_t, Tables);
template __global__ void k_deint_batch<1>(const uint8_t*, uint8_t*, uint32_t, Tables);
template __global__ void k_deint_batch<2>(const uint8_t*, uint8_t*, uint32_t, Tables);
template __global__ void k_deint_batch<4>(const uint8_t*, uint8_t*, uint32_t, Tables);
template __global__ void k_deint_batch<6>(const uint8_t*, uint8_t*, uint32_t, Tables);

// sora_track11a_state: { int16 cfo_comp, sfo_comp, cfo_tracker, sfo_tracker; uint32 symbol_count; COMPLEX16 comp[64]; } = 67 words
// PHASE = false: TPilotTrack alone (its input is TPhaseCompensate's output)
template <bool PHASE>
__global__ void __launch_bounds__(64) k_ptrack_batch(const uint32_t* eq, const uint32_t* first, const uint32_t* nsym, uint32_t* state, uint32_t* out, uint32_t nframes, Tables T)
{
    const uint32_t f = blockIdx.x;
    if (f >= nframes) return;
    const int lane = threadIdx.x;
    uint32_t* st = state + (size_t)f * 67;
    int cfo_comp = (int)(short)(st[0] & 0xFFFF), sfo_comp = (int)st[0] >> 16;
    int cfo_tr = (int)(short)(st[1] & 0xFFFF), sfo_tr = (int)st[1] >> 16;
    unsigned symbol_count = st[2];
    cpx comp = unpack(st[3 + lane]);                                            // CompCoeffs[lane]
    const int c = lane < 32 ? lane : lane - 64;                                 // signed carrier number of bin `lane`
    const bool data_bin = (lane >= 1 && lane <= 26) || lane >= 38;              // bins _build_coeff writes (pilot.hpp:138-164)
    const uint32_t s0 = first[f], ns = nsym[f];
    for (uint32_t s = 0; s < ns; s++) {
        const cpx in = unpack(eq[(size_t)(s0 + s) * 64 + lane]);
        const cpx pc = PHASE ? mul_q15(in, comp) : in;                           // TPhaseCompensate: rep_mul<16>
        const uint32_t pk = pack(pc);
        const cpx p43 = unpack((uint32_t)__shfl((int)pk, 43)), p57 = unpack((uint32_t)__shfl((int)pk, 57));
        const cpx p7 = unpack((uint32_t)__shfl((int)pk, 7)),   p21 = unpack((uint32_t)__shfl((int)pk, 21));
        int th1 = uatan2(T, p43.im, p43.re), th2 = uatan2(T, p57.im, p57.re);
        int th3 = uatan2(T, p7.im, p7.re),   th4 = uatan2(T, -p21.im, -p21.re);
        if (kPilotSgnS[symbol_count & 127]) { th1 = w16(th1 + 0x8000); th2 = w16(th2 + 0x8000); th3 = w16(th3 + 0x8000); th4 = w16(th4 + 0x8000); }
        symbol_count++; if (symbol_count >= 127) symbol_count = 0;
        const int avg = w16((th1 + th2 + th3 + th4) / 4);
        const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
        cpx o = mk(0, 0);                                                        // bins 0, 27..37: not produced by _rotate / undefined in the reference
        if (data_bin) o = mul_q15(pc, rot_coeff(T, w16(avg + c * del)));
        out[(size_t)(s0 + s) * 64 + lane] = pack(o);
        cfo_tr = w16(cfo_tr + (avg >> 2)); sfo_tr = w16(sfo_tr + (del >> 2));
        cfo_comp = w16(cfo_comp + avg + cfo_tr); sfo_comp = w16(sfo_comp + del + sfo_tr);
        if (data_bin) comp = rot_coeff(T, w16(cfo_comp + c * sfo_comp));         // _build_coeff(CompCoeffs, CFO_comp, SFO_comp)
    }
    st[3 + lane] = pack(comp);
    if (lane == 0) {
        st[0] = ((uint32_t)cfo_comp & 0xFFFFu) | ((uint32_t)sfo_comp << 16);
        st[1] = ((uint32_t)cfo_tr & 0xFFFFu) | ((uint32_t)sfo_tr << 16);
        st[2] = symbol_count;
    }
}
template __global__ void k_ptrack_batch<true>(const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t, Tables);
template __global__ void k_ptrack_batch<false>(const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t, Tables);

// ---------------------------------------------------------------------------------------------------------------
// FFT<128>: 32 lanes per transform, 4 points per lane, 8 transforms per tile, 4 tiles per 256-thread block (fft128_core,
// dev_arith.h); points 4e..4e+3 in and out as one 16-byte access per lane, twiddles in registers.
__global__ void __launch_bounds__(256) k_fft128_batch(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, Tables T)
{
    __shared__ uint32_t s_all[8][128];
    const int g = threadIdx.x >> 5, e = threadIdx.x & 31;
    const Fft128TwPk W = fft128_twiddles_pk(T, e);
    uint32_t* s = s_all[g];
    uint4 v[kFftTiles];
#pragma unroll
    for (int t = 0; t < kFftTiles; t++) {
        const uint32_t i = (blockIdx.x * kFftTiles + t) * 8 + g;
        v[t] = i < n ? reinterpret_cast<const uint4*>(in)[(size_t)i * 32 + e] : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int t = 0; t < kFftTiles; t++) {
        const uint32_t i = (blockIdx.x * kFftTiles + t) * 8 + g;
        wave_lds_sync();
        reinterpret_cast<uint4*>(s)[e] = v[t];
        wave_lds_sync();
        pcx x[4];
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = s[e + 32 * m];
        fft128_core_pk(x, s, e, W, wave_lds_sync);
        const unsigned r = __brev((unsigned)e) >> 27;                               // point 4e+q sits at slot bitrev7(4e+q) = bitrev5(e) + 32 bitrev2(q)
        if (i < n) reinterpret_cast<uint4*>(out)[(size_t)i * 32 + e] = uint4{s[r], s[r + 64], s[r + 32], s[r + 96]};
    }
}

// ------------------------------------------------------------------------------------------------
// k_ingest: capture ingest in one streaming pass, one output sample per thread (HBM bound: ~4.6 B read + 4 B written
// per 40 MHz sample for a 44 MHz RX_BLOCK dump):
//   RX_BLOCK de-framing   128-byte block = 16-byte descriptor + 28 COMPLEX16 (brickutil.h:40-55, _rx_manager.h:96-137)
//   14 -> 16-bit sign fix (int16)(raw << 2)   (the fixture's samples are 14-bit two's complement, zero-extended)
//   TDownSample44_40      Down44to40::Resample (44MTo40M.hpp:62-123): period 11 -> 10, out[10p] = x[11p],
//                         out[10p+k] = (x[11p+k] R[k] + x[11p+k+1] L[k+1]) >> 7 -- stateless per period, so parallel
//   TDownSample2          out[j] = in[2j] (samples.hpp:36-39)
__device__ __constant__ int kLinR[11] = { 1, 115, 102, 90, 77, 64, 51, 38, 26, 13, 0 };      // 44MTo40M.hpp:35-39
__device__ __constant__ int kLinL[11] = { 0, 0, 13, 26, 38, 51, 64, 77, 90, 102, 115 };

__global__ void __launch_bounds__(256) k_ingest(const uint8_t* __restrict__ raw, uint32_t* __restrict__ out, uint64_t m0, uint64_t n_out, unsigned flags)
{
    const uint64_t m = m0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_out) return;
    auto X = [&](uint64_t i) -> cpx {                                            // sample i of the de-framed, sign-fixed stream
        const uint64_t off = (flags & 1u) ? (i / 28) * 128 + 16 + (i % 28) * 4 : i * 4;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(raw + off);
        cpx x = unpack(w);
        if (flags & 2u) { x.re = w16(x.re << 2); x.im = w16(x.im << 2); }
        return x;
    };
    const uint64_t m1 = (flags & 8u) ? 2 * m : m;
    cpx s;
    if (flags & 4u) {
        const uint64_t p = m1 / 10; const int k = (int)(m1 % 10);
        if (k == 0) s = X(11 * p);
        else {
            const cpx a = X(11 * p + k), b = X(11 * p + k + 1);
            s = mk(w16((a.re * kLinR[k] + b.re * kLinL[k + 1]) >> 7), w16((a.im * kLinR[k] + b.im * kLinL[k + 1]) >> 7));
        }
    } else s = X(m1);
    out[m] = pack(s);
}

// The common case (a 44 MHz dump: de-frame + resample, optionally sign fix / decimate) tiled so that every byte moves in
// 16-byte coalesced accesses: a tile is 55 RX_BLOCKs = 1540 input samples = 140 resampler periods = 1400 output samples
// (lcm(28, 11) x 5); the 7040 raw bytes are staged in LDS, the index arithmetic is 32-bit and local.
constexpr int kTileBlocks = 55, kTileRaw = kTileBlocks * 128, kTileOut = 1400;

__global__ void __launch_bounds__(256) k_ingest_tile(const uint8_t* __restrict__ raw, uint32_t* __restrict__ out, unsigned flags, uint32_t tiles)
{
    __shared__ uint32_t s_raw[kTileRaw / 4];
    __shared__ uint32_t s_out[kTileOut];
    constexpr int NQ = kTileRaw / 16;                                              // 440 quad-words per tile: two per thread (the second for 184 threads)
    const int tid = threadIdx.x;
    const bool dec = (flags & 8u) != 0, fix = (flags & 2u) != 0;
    const int nout = dec ? kTileOut / 2 : kTileOut;
    uint32_t tile = blockIdx.x;
    if (tile >= tiles) return;
    // Which two input samples and which weights an output takes depends only on its index inside the tile: worked out once per thread (six outputs
    // each), so that the per-tile work is two LDS reads, the sign fix as one packed shift, and the interpolation as two 16x16+16x16 dot products
    // (v_dot2_i32_i16 wraps like the reference's pmaddwd; k = 0 is x * 128 >> 7 = x).
    constexpr int PER = (kTileOut + 255) / 256;
    int ia[PER], ib[PER]; uint32_t wrl[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const int m = tid + 256 * j, m1 = dec ? 2 * m : m;
        const int p = m1 / 10, k = m1 - 10 * p;
        const int a = 11 * p + k, b = k == 0 ? a : a + 1;
        ia[j] = (a / 28) * 32 + 4 + (a % 28); ib[j] = (b / 28) * 32 + 4 + (b % 28);
        const int R = k == 0 ? 128 : kLinR[k], L = k == 0 ? 0 : kLinL[k + 1];
        wrl[j] = ((uint32_t)R & 0xFFFFu) | ((uint32_t)L << 16);
        if (m >= nout) { ia[j] = ib[j] = 4; }
    }
    const uint4* src = reinterpret_cast<const uint4*>(raw + (size_t)tile * kTileRaw);
    uint4 r0 = src[tid], r1 = tid + 256 < NQ ? src[tid + 256] : uint4{0, 0, 0, 0};
    for (;;) {
        reinterpret_cast<uint4*>(s_raw)[tid] = r0;
        if (tid + 256 < NQ) reinterpret_cast<uint4*>(s_raw)[tid + 256] = r1;
        const uint32_t next = tile + gridDim.x;                                    // the next tile's reads are in flight while this one is computed and stored
        if (next < tiles) {
            const uint4* nsrc = reinterpret_cast<const uint4*>(raw + (size_t)next * kTileRaw);
            r0 = nsrc[tid]; if (tid + 256 < NQ) r1 = nsrc[tid + 256];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int m = tid + 256 * j;
            pcx a = s_raw[ia[j]], b = s_raw[ib[j]];
            if (fix) { a = __builtin_bit_cast(pcx, (s16x2_t)(__builtin_bit_cast(s16x2_t, a) << (short)2)); b = __builtin_bit_cast(pcx,
                    (s16x2_t)(__builtin_bit_cast(s16x2_t, b) << (short)2)); }
            const pcx re2 = (a & 0xFFFFu) | (b << 16), im2 = (a >> 16) | (b & 0xFFFF0000u);          // (a.re, b.re), (a.im, b.im)
            const int vr = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, re2), __builtin_bit_cast(s16x2_t, wrl[j]), 0, false);
            const int vi = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, im2), __builtin_bit_cast(s16x2_t, wrl[j]), 0, false);
            if (m < nout) s_out[m] = (((uint32_t)vr >> 7) & 0xFFFFu) | (((uint32_t)vi << 9) & 0xFFFF0000u);
        }
        __syncthreads();
        uint4* dst = reinterpret_cast<uint4*>(out + (size_t)tile * nout);          // nout * 4 bytes is a multiple of 16
        for (int i = tid; i < nout / 4; i += 256) dst[i] = reinterpret_cast<const uint4*>(s_out)[i];
        if (next >= tiles) break;
        tile = next;
        __syncthreads();                                                           // s_raw / s_out are rewritten by the next round
    }
}

}  // namespace sora
