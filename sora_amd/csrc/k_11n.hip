// k_11n.hip -- 802.11n 2x2, stage level: the bricks of the reference's 11n receive graph that exist here
// so far, as batched per-stage entry points with the brick port shapes (the whole-path graph is not built yet):
//   sora_hip_demap11n         T11nDemap{BPSK,QPSK,QAM16,QAM64}        kernel/bb/Brick11/src/demapper11n.hpp:89-309 (dsp_demap.h)
//   sora_hip_deinterleave11n  T11nDeinterleave{...}_S0 / _S1           kernel/bb/Brick11/src/deinterleaver_11n.hpp:4-1618
// Both are pure gathers: one coalesced read of a symbol, table look-ups out of LDS, one coalesced write -- HBM-bound.
// The soft-value tables of dsp_demap.h are step functions of the limited coordinate v in [-128,127]; they are regenerated
// from their run lengths (value, count from v = -128 upward).  The de-interleavers are the standard HT interleaver
// (N_COL 13, N_ROW 4 N_BPSC, N_ROT 11) inverted, computed per element instead of the reference's unrolled tables.
#include "kernels.h"
#include "dev_11n.h"
#include "../../include/sora_hip.h"

namespace sora {


// one wave per symbol, four symbols per block; lane l < 52 = data carrier l
__global__ void __launch_bounds__(256) k_demap11n_batch(const uint32_t* in, uint8_t* soft, int nb, uint32_t n)
{
    __shared__ uint8_t s_lut[6][256];
    fill_demap_luts(s_lut);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t sym = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sym >= n || lane >= 52) return;
    const cpx x = unpack(in[(size_t)sym * 64 + data_bin(lane)]);
    const int re = min(max(x.re, -128), 127) + 128, im = min(max(x.im, -128), 127) + 128;      // demap_limit
    uint8_t* o = soft + ((size_t)sym * 52 + lane) * nb;
    switch (nb) {
    case 1: o[0] = s_lut[0][re]; break;
    case 2: o[0] = s_lut[0][re]; o[1] = s_lut[0][im]; break;
    case 4: o[0] = s_lut[1][re]; o[1] = s_lut[2][re]; o[2] = s_lut[1][im]; o[3] = s_lut[2][im]; break;
    default: o[0] = s_lut[3][re]; o[1] = s_lut[4][re]; o[2] = s_lut[5][re]; o[3] = s_lut[3][im]; o[4] = s_lut[4][im]; o[5] = s_lut[5][im];
    }
}

// one thread per output soft value
__global__ void __launch_bounds__(256) k_deint11n_batch(const uint8_t* in, uint8_t* out, int nb, int iss, uint32_t n)
{
    const uint32_t per = 52u * nb;
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)n * per) return;
    const uint32_t sym = (uint32_t)(g / per); const int k = (int)(g - (uint64_t)sym * per);
    const int j = deint11n_index(nb, iss, k);
    out[g] = in[(uint64_t)sym * per + j];
}

// ---- TMimoChannelEst (channel_11n.hpp:329-443): one thread per carrier, 64 per frame

__global__ void __launch_bounds__(256) k_mimo_est11n_batch(const uint32_t* ltf0, const uint32_t* ltf1, uint32_t* h, uint32_t* hinv, uint32_t nframes)
{
#pragma clang fp contract(off)
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6); const int i = threadIdx.x & 63;
    if (f >= nframes) return;
    const int k = i < 32 ? i : i - 64;
    const bool negate = !(k >= -28 && k <= 28 && kHtLtf[k + 28] == 1);             // _80211n_HTLTFMask: every bin whose HT-LTF value is not +1
    cpx hh[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t* l = (r ? ltf1 : ltf0) + (size_t)f * 128;
        const cpx a = unpack(l[i]), b = unpack(l[i + 64]);
        cpx d = sra(csubs(a, b), 1), s = sra(cadds(a, b), 1);                      // P-matrix combination of the two HT-LTFs
        if (negate) { d = mk(neg16(d.re), neg16(d.im)); s = mk(neg16(s.re), neg16(s.im)); }
        hh[r][0] = d; hh[r][1] = s;
        h[((size_t)f * 2 + r) * 128 + i] = pack(d); h[((size_t)f * 2 + r) * 128 + 64 + i] = pack(s);
    }
    // 2x2 inverse x 2^16 in single precision, operation for operation as brick/inc/sora_matrix.h:134-148,305-313
    const cf a00 = { (float)hh[0][0].re, (float)hh[0][0].im }, a01 = { (float)hh[0][1].re, (float)hh[0][1].im };
    const cf a10 = { (float)hh[1][0].re, (float)hh[1][0].im }, a11 = { (float)hh[1][1].re, (float)hh[1][1].im };
    const cf ad = cf_mul(a00, a11), bc = cf_mul(a01, a10);
    const cf det = { ad.re - bc.re, ad.im - bc.im };
    const float n = ((det.re * det.re) + (det.im * det.im)) / 65536.0f;
    const cf ds = { det.re, -det.im }, m01 = { -a01.re, -a01.im }, m10 = { -a10.re, -a10.im };
    const cf r00 = cf_mul(a11, ds), r01 = cf_mul(m01, ds), r10 = cf_mul(m10, ds), r11 = cf_mul(a00, ds);
    uint32_t* o = hinv + (size_t)f * 256;
    o[i]       = pack(mk(cvtps_sat16(r00.re / n), cvtps_sat16(r00.im / n)));
    o[64 + i]  = pack(mk(cvtps_sat16(r01.re / n), cvtps_sat16(r01.im / n)));
    o[128 + i] = pack(mk(cvtps_sat16(r10.re / n), cvtps_sat16(r10.im / n)));
    o[192 + i] = pack(mk(cvtps_sat16(r11.re / n), cvtps_sat16(r11.im / n)));
}

// ---- TMimoChannelComp (channel_11n.hpp:445-521): x = (Hinv y) >> 9, saturating pack; one thread per carrier
__global__ void __launch_bounds__(256) k_mimo_comp11n_batch(const uint32_t* hinv, const uint32_t* frame_index, const uint32_t* y0, const uint32_t* y1,
                                                            uint32_t* x0, uint32_t* x1, uint32_t nsym)
{
    const uint32_t sidx = blockIdx.x * 4 + (threadIdx.x >> 6); const int i = threadIdx.x & 63;
    if (sidx >= nsym) return;
    const uint32_t* hi = hinv + (size_t)(frame_index ? frame_index[sidx] : 0u) * 256;
    const cpx a = unpack(y0[(size_t)sidx * 64 + i]), b = unpack(y1[(size_t)sidx * 64 + i]);
    int ar, ai, br, bi;
    mul32(unpack(hi[i]), a, ar, ai); mul32(unpack(hi[64 + i]), b, br, bi);
    x0[(size_t)sidx * 64 + i] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
    mul32(unpack(hi[128 + i]), a, ar, ai); mul32(unpack(hi[192 + i]), b, br, bi);
    x1[(size_t)sidx * 64 + i] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
}

// ---- TSisoChannelEst (channel_11n.hpp:33-231): one thread per (frame, RX chain, carrier).  The rounding term is added lane for lane as
// the reference's vectors line up: component c of carrier j of a group of four gets |x[(2j + c) mod 4]|^2 >> 1.
__global__ void __launch_bounds__(256) k_siso_est11n_batch(const uint32_t* l0, const uint32_t* l1, uint32_t* ch, uint32_t nframes)
{
    const uint32_t f = blockIdx.x * 2 + (threadIdx.x >> 7); const int r = (threadIdx.x >> 6) & 1, i = threadIdx.x & 63;
    if (f >= nframes) return;
    uint32_t out = 0;
    if (i < 28 || i >= 36) {
        const uint32_t* l = (r ? l1 : l0) + (size_t)f * 128 + (i & ~3);
        const cpx a = siso_one(l, i & 3, i), b = siso_one(l + 64, i & 3, i);
        out = pack(mk((short)((short)(a.re + b.re) >> 1), (short)((short)(a.im + b.im) >> 1)));
    }
    ch[(size_t)f * 128 + r * 64 + i] = out;
}

// ---- TSisoChannelComp (channel_11n.hpp:233-297) + TMrcCombine (PHY_11n.hpp:362-398): x_r = sat((y_r * c_r) >> 9), mrc = (x_0 + x_1) >> 1
__global__ void __launch_bounds__(256) k_siso_comp11n_batch(const uint32_t* ch, const uint32_t* frame_index, const uint32_t* y0, const uint32_t* y1,
                                                            uint32_t* x0, uint32_t* x1, uint32_t* mrc, uint32_t nsym)
{
    const uint32_t sidx = blockIdx.x * 4 + (threadIdx.x >> 6); const int i = threadIdx.x & 63;
    if (sidx >= nsym) return;
    const uint32_t* c = ch + (size_t)(frame_index ? frame_index[sidx] : 0u) * 128;
    int re, im;
    mul32(unpack(y0[(size_t)sidx * 64 + i]), unpack(c[i]), re, im);      const cpx a = mk(sat16(re >> 9), sat16(im >> 9));
    mul32(unpack(y1[(size_t)sidx * 64 + i]), unpack(c[64 + i]), re, im); const cpx b = mk(sat16(re >> 9), sat16(im >> 9));
    if (x0) x0[(size_t)sidx * 64 + i] = pack(a);
    if (x1) x1[(size_t)sidx * 64 + i] = pack(b);
    if (mrc) mrc[(size_t)sidx * 64 + i] = pack(mk((short)((short)(a.re + b.re) >> 1), (short)((short)(a.im + b.im) >> 1)));
}

// ---- T11nSigDemap (demapper11n.hpp:6-87): three symbols per frame, L-SIG on I, HT-SIG 1/2 on Q; one wave per symbol, lane < 48 = carrier
__global__ void __launch_bounds__(256) k_sig_demap11n_batch(const uint32_t* sym, uint8_t* soft, uint32_t nframes)
{
    __shared__ uint8_t s_lut[6][256];
    fill_demap_luts(s_lut);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);                  // symbol number = 3 * frame + s
    if (g >= nframes * 3u || lane >= 48) return;
    const int s = (int)(g % 3u);
    int bin;                                                                 // carriers -26..-1 then 1..26 without the pilots
    if (lane < 24) bin = 38 + lane + (lane >= 5) + (lane >= 18); else { const int m = lane - 24; bin = 1 + m + (m >= 6) + (m >= 19); }
    const cpx v = unpack(sym[(size_t)g * 64 + bin]);
    const int q = s == 0 ? v.re : v.im;
    soft[(size_t)g * 48 + lane] = s_lut[0][min(max(q, -128), 127) + 128];
}

// ---- T11aDeinterleaveBPSK x3 -> T11nViterbiSig (viterbi.hpp:51-99) -> T11nSigParser (PHY_11n.hpp:432-513): one wave per frame,
// lane = trellis state (Viterbi_sig11, viterbicore.h:36-261, over NB = 24 and NB = 48 steps)
__global__ void __launch_bounds__(256) k_sig_decode11n_batch(const uint8_t* soft, uint32_t* rec, uint32_t nframes)
{
    __shared__ uint8_t s_soft[4][144];
    __shared__ uint64_t s_dec[4][50];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t f = blockIdx.x * 4 + w;
    if (f >= nframes) return;                                                // whole waves leave; no block-wide barrier below
    for (int k = lane; k < 144; k += 64) { const int s3 = k / 48, kk = k - 48 * s3; s_soft[w][k] = soft[(size_t)f * 144 + 48 * s3 + 3 * (kk & 15) + (kk >> 4)]; }
    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
    const uint32_t lsig = (uint32_t)(viterbi_sig_wave<24>(s_soft[w], s_dec[w], lane) >> 6);
    __builtin_amdgcn_wave_barrier();
    const uint64_t ht = viterbi_sig_wave<48>(s_soft[w] + 48, s_dec[w], lane) >> 6;
    if (lane != 0) return;
    uint32_t fl[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    bool ok = false;
    do {
        const uint32_t sig = lsig & 0xFFFFFF;
        if (sig & 0xFC0010) break;
        if (__popc(sig) & 1) break;
        const uint32_t code = sig & 0xF;                                     // ieee80211a_cmn.h:97-107
        fl[1] = code == 0x8 ? 48000u : code == 0x9 ? 24000u : code == 0xA ? 12000u : code == 0xB ? 6000u : code == 0xC ? 54000u
              : code == 0xD ? 36000u : code == 0xE ? 18000u : code == 0xF ? 9000u : 0u;
        if (fl[1] == 0) break;
        fl[2] = ((sig >> 5) & 0xFFF) * 2;
        if (fl[2] > 1500) break;
        uint32_t crc = 0xFF;                                                 // CalcCRC8(ip, 4, 2): reflected, polynomial 0xE0, over HT-SIG bits 0..33
        for (int b = 0; b < 34; b++) { crc ^= (uint32_t)(ht >> b) & 1; crc = (crc & 1) ? (crc >> 1) ^ 0xE0 : crc >> 1; }
        if (((~crc) & 0xFF) != (uint32_t)((ht >> 34) & 0x3FFF)) break;      // compared in int: bits 42.. (always 0 after the >> 6) included
        fl[3] = (uint32_t)ht & 0x7F;
        if (fl[3] < 8 || fl[3] >= 11) break;
        fl[4] = (uint32_t)(ht >> 8) & 0xFFFF;
        if (fl[4] > 1500) break;
        fl[5] = fl[3] == 10 ? 2u : 0u;                                       // CR_34 : CR_12
        const uint32_t nd = 52u * (fl[3] - 7);                               // DOT11N_RATE_PARAMS[8..10].ndbps
        fl[6] = fl[7] = (fl[4] * 8 + 22 + nd - 1) / nd + 4;
        fl[2] = fl[4]; fl[8] = 3;                                            // frame_length = ht_frame_length; SYMBOL_HT_STF
        ok = true;
    } while (0);
    if (!ok) fl[0] = 0x80000005u;                                            // E_ERROR_PLCP_HEADER_FAIL
    uint32_t* o = rec + (size_t)f * 12;
    for (int i = 0; i < 9; i++) o[i] = fl[i];
    o[9] = lsig & 0xFFFFFF; o[10] = (uint32_t)ht; o[11] = (uint32_t)(ht >> 32);
}

// ---- dsp_math (Brick11/src/dsp_math.h:96-213): arctangent through a 4097-entry table, exactly as the reference indexes it

// TFreqEstimator_11n (freqoffset_11n.hpp:42-160): one wave per frame, lane = sample of the first L-LTF half, both RX chains
__global__ void __launch_bounds__(256) k_cfo_est11n_batch(const uint32_t* l0, const uint32_t* l1, short* state, uint32_t nframes, const short* atan_tab)
{
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6); const int i = threadIdx.x & 63;
    if (f >= nframes) return;
    int re, im, sre, sim;
    conj_mul32(unpack(l0[(size_t)f * 128 + i]), unpack(l0[(size_t)f * 128 + 64 + i]), re, im); sre = re >> 7; sim = im >> 7;
    conj_mul32(unpack(l1[(size_t)f * 128 + i]), unpack(l1[(size_t)f * 128 + 64 + i]), re, im); sre += re >> 7; sim += im >> 7;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sre += __shfl_xor(sre, d); sim += __shfl_xor(sim, d); }       // wrapping 32-bit sums: order does not matter
    if (i < 8) {
        const int delta = dsp_atan32(atan_tab, sre, sim) >> 6;
        short* st = state + (size_t)f * 24;
        st[i] = (short)(i * delta); st[8 + i] = (short)(delta << 3); st[16 + i] = 0;
    }
}

// TFreqComp_11n (freqoffset_11n.hpp:162-280): frame f owns nbursts[f] bursts of 8 samples from sample first[f] of both chains;
// one thread per sample; the running phase is delta0 + burst * step - theta in wrapping int16
__global__ void __launch_bounds__(256) k_freq_comp11n_batch(const uint32_t* in0, const uint32_t* in1, uint32_t* out0, uint32_t* out1, const uint32_t* first,
                                                            const uint32_t* nbursts, const short* state, const uint32_t* sincos)
{
    const uint32_t f = blockIdx.y, nb = nbursts[f];
    const short* st = state + (size_t)f * 24;
    for (uint32_t m = blockIdx.x * 256 + threadIdx.x; m < nb * 8; m += gridDim.x * 256) {
        const int k = m & 7, b = m >> 3;
        const int ph = (int)(short)(st[k] + (short)(b * st[8 + k]) - st[16 + k]);
        const cpx cof = unpack(sincos[(unsigned)ph & 0xFFFFu]);
        const size_t at = (size_t)first[f] + m;
        int re, im;
        mul32(unpack(in0[at]), cof, re, im); out0[at] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
        mul32(unpack(in1[at]), cof, re, im); out1[at] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
    }
}
__global__ void k_freq_comp11n_advance(const uint32_t* nbursts, short* state, uint32_t nframes)       // vfo_delta_i += nbursts * vfo_step_i
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nframes * 8) return;
    short* st = state + (size_t)(t >> 3) * 24; const int k = t & 7;
    st[k] = (short)(st[k] + (short)(nbursts[t >> 3] * st[8 + k]));
}

// TPilotTrack_11n (pilot_11n.hpp:84-141): one wave per frame, 64 symbols per pass: per-symbol mean pilot phase, then a wrapping prefix sum
__global__ void __launch_bounds__(256) k_pilot_track11n_batch(const uint32_t* x0, const uint32_t* x1, const uint32_t* first, const uint32_t* nsym,
                                                              short* state, short* theta_out, uint32_t nframes, const short* atan_tab)
{
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6); const int lane = threadIdx.x & 63;
    if (f >= nframes) return;
    short* st = state + (size_t)f * 24;
    int base[8];
#pragma unroll
    for (int k = 0; k < 8; k++) base[k] = st[16 + k];
    const uint32_t n = nsym[f], s0 = first[f];
    int run = 0;                                                           // sum of the increments so far (the same for all 8 entries)
    for (uint32_t p = 0; p < n; p += 64) {
        const uint32_t s = p + lane;
        int inc = 0;
        if (s < n) {
            int t[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t* x = (c ? x1 : x0) + (size_t)(s0 + s) * 64;
                const cpx a = unpack(x[64 - 21]), b = unpack(x[64 - 7]), d = unpack(x[7]), e = unpack(x[21]);
                t[c] = (int)(short)((dsp_atan16(atan_tab, a.re, a.im) + dsp_atan16(atan_tab, b.re, b.im) + dsp_atan16(atan_tab, d.re,
                        d.im) + dsp_atan16(atan_tab, e.re, e.im)) >> 2);
            }
            inc = (int)(short)((t[0] + t[1]) >> 1);
        }
        int pre = inc;                                                     // inclusive prefix sum over the pass (wrapping in the end)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(pre, d); if (lane >= d) pre += v; }
        if (theta_out && s < n)
#pragma unroll
            for (int k = 0; k < 8; k++) theta_out[(size_t)(s0 + s) * 8 + k] = (short)(base[k] + run + pre);
        run += __shfl(pre, 63);
    }
    if (lane < 8) st[16 + lane] = (short)(base[lane] + run);
}

}  // namespace sora

using namespace sora;

int sora_hip_demap11n(const sora_complex16* d_in, uint8_t* d_soft, int n_bpsc, size_t n, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_in || !d_soft || !(n_bpsc == 1 || n_bpsc == 2 || n_bpsc == 4 || n_bpsc == 6)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_demap11n: bad argument", 0);
    if (n == 0) return SORA_OK;
    hipLaunchKernelGGL(k_demap11n_batch, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_in), d_soft, n_bpsc, (uint32_t)n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "k_demap11n_batch", (int)e);
}

int sora_hip_deinterleave11n(const uint8_t* d_in, uint8_t* d_out, int n_bpsc, int spatial_stream, size_t n, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_in || !d_out || !(n_bpsc == 1 || n_bpsc == 2 || n_bpsc == 4 || n_bpsc == 6) || spatial_stream < 0 || spatial_stream > 1)
        return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_deinterleave11n: bad argument", 0);
    if (n == 0) return SORA_OK;
    const uint64_t total = (uint64_t)n * 52 * n_bpsc;
    hipLaunchKernelGGL(k_deint11n_batch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n_bpsc, spatial_stream, (uint32_t)n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "k_deint11n_batch", (int)e);
}

int sora_hip_mimo_est11n(const sora_complex16* d_ltf0, const sora_complex16* d_ltf1, sora_complex16* d_h, sora_complex16* d_hinv, size_t nframes, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_ltf0 || !d_ltf1 || !d_h || !d_hinv) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_mimo_est11n: null argument", 0);
    if (nframes == 0) return SORA_OK;
    hipLaunchKernelGGL(k_mimo_est11n_batch, dim3((unsigned)((nframes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_ltf0),
                       reinterpret_cast<const uint32_t*>(d_ltf1), reinterpret_cast<uint32_t*>(d_h), reinterpret_cast<uint32_t*>(d_hinv), (uint32_t)nframes);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "k_mimo_est11n_batch", (int)e);
}

int sora_hip_mimo_comp11n(const sora_complex16* d_hinv, const uint32_t* d_frame_index, const sora_complex16* d_y0, const sora_complex16* d_y1,
                          sora_complex16* d_x0, sora_complex16* d_x1, size_t nsym, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_hinv || !d_y0 || !d_y1 || !d_x0 || !d_x1) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_mimo_comp11n: null argument", 0);
    if (nsym == 0) return SORA_OK;
    hipLaunchKernelGGL(k_mimo_comp11n_batch, dim3((unsigned)((nsym + 3) / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_hinv), d_frame_index,
                       reinterpret_cast<const uint32_t*>(d_y0), reinterpret_cast<const uint32_t*>(d_y1), reinterpret_cast<uint32_t*>(d_x0), reinterpret_cast<uint32_t*>(d_x1), (uint32_t)nsym);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "k_mimo_comp11n_batch", (int)e);
}

// ---- dsp_math tables (dsp_math.h:215-245): generated once per device with the C library, as the reference generates them at start-up
#include <cmath>
#include <mutex>
#include <vector>
// the two tables on the host (dsp_math.h:215-245)
void sora_internal_dsp_host_tables(std::vector<uint32_t>& sc, std::vector<short>& at)
{
    sc.resize(65536); at.resize(4097);
    for (unsigned i = 0; i < 65536; i++) {
        const double r = (double)i * 2.0 * M_PI / 65535.0;
        const short c = (short)(cos(r) * 32767.5), s = (short)(sin(r) * 32767.5);
        sc[i] = ((uint32_t)(uint16_t)c) | ((uint32_t)(uint16_t)s << 16);
    }
    for (int i = 0; i <= 4096; i++) at[i] = (short)(atan((double)i / 4096.0) / (M_PI / 4.0) * 8192);
}
namespace {
struct DspTables { uint32_t* sincos = nullptr; short* atan = nullptr; };
std::mutex g_dsp_mutex;
const DspTables* dsp_tables()
{
    static DspTables tabs[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    // handles may be created from different threads: built once per device, published whole
    std::lock_guard<std::mutex> lock(g_dsp_mutex);
    DspTables& T = tabs[dev];
    if (!T.sincos) {
        std::vector<uint32_t> sc; std::vector<short> at;
        sora_internal_dsp_host_tables(sc, at);
        if (sora_internal_pin_table("dsp_sincos", sc.data(), sc.size() * 4) != SORA_OK || sora_internal_pin_table("dsp_atan", at.data(), at.size() * 2) != SORA_OK) return nullptr;
        uint32_t* d_sc = nullptr; short* d_at = nullptr;
        if (hipMalloc((void**)&d_sc, sc.size() * 4) != hipSuccess || hipMalloc((void**)&d_at, at.size() * 2) != hipSuccess ||
            hipMemcpy(d_sc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_at, at.data(), at.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_sc); (void)hipFree(d_at);
            return nullptr;
        }
        T.atan = d_at; T.sincos = d_sc;                                         // (sincos last: it is what marks the entry as built)
    }
    return &T;
}
int launch_result(const char* what) { const hipError_t e = hipGetLastError(); return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, what, (int)e); }
}  // namespace

int sora_internal_dsp_tables(const uint32_t** sincos, const short** atan)       // current device; used by k_rx11n.hip
{
    const DspTables* T = dsp_tables();
    if (!T) return SORA_ERR_HARDWARE_FAILED;
    *sincos = T->sincos; *atan = T->atan; return SORA_OK;
}

int sora_hip_cfo_est11n(const sora_complex16* d_lltf0, const sora_complex16* d_lltf1, int16_t* d_state, size_t nframes, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_lltf0 || !d_lltf1 || !d_state) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_cfo_est11n: null argument", 0);
    if (nframes == 0) return SORA_OK;
    const DspTables* T = dsp_tables(); if (!T) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "dsp_math table upload failed", 0);
    hipLaunchKernelGGL(k_cfo_est11n_batch, dim3((unsigned)((nframes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_lltf0),
                       reinterpret_cast<const uint32_t*>(d_lltf1), d_state, (uint32_t)nframes, T->atan);
    return launch_result("k_cfo_est11n_batch");
}

int sora_hip_freq_comp11n(const sora_complex16* d_in0, const sora_complex16* d_in1, sora_complex16* d_out0, sora_complex16* d_out1,
                          const uint32_t* d_first, const uint32_t* d_nbursts, int16_t* d_state, size_t nframes, size_t max_bursts, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_in0 || !d_in1 || !d_out0 || !d_out1 || !d_first || !d_nbursts || !d_state) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_freq_comp11n: null argument", 0);
    if (nframes == 0 || max_bursts == 0) return SORA_OK;
    if (nframes > 65535) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_hip_freq_comp11n: at most 65535 frames per call", 0);
    const DspTables* T = dsp_tables(); if (!T) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "dsp_math table upload failed", 0);
    const unsigned gx = (unsigned)((max_bursts * 8 + 255) / 256);
    hipLaunchKernelGGL(k_freq_comp11n_batch, dim3(gx < 64 ? gx : 64, (unsigned)nframes), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_in0),
                       reinterpret_cast<const uint32_t*>(d_in1), reinterpret_cast<uint32_t*>(d_out0), reinterpret_cast<uint32_t*>(d_out1), d_first, d_nbursts, d_state, T->sincos);
    hipLaunchKernelGGL(k_freq_comp11n_advance, dim3((unsigned)((nframes * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_nbursts, d_state, (uint32_t)nframes);
    return launch_result("k_freq_comp11n_batch");
}

int sora_hip_pilot_track11n(const sora_complex16* d_x0, const sora_complex16* d_x1, const uint32_t* d_first, const uint32_t* d_nsym, int16_t* d_state,
                            int16_t* d_theta, size_t nframes, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_x0 || !d_x1 || !d_first || !d_nsym || !d_state) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_pilot_track11n: null argument", 0);
    if (nframes == 0) return SORA_OK;
    const DspTables* T = dsp_tables(); if (!T) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "dsp_math table upload failed", 0);
    hipLaunchKernelGGL(k_pilot_track11n_batch, dim3((unsigned)((nframes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_x0),
                       reinterpret_cast<const uint32_t*>(d_x1), d_first, d_nsym, d_state, d_theta, (uint32_t)nframes, T->atan);
    return launch_result("k_pilot_track11n_batch");
}

int sora_hip_siso_est11n(const sora_complex16* d_lltf0, const sora_complex16* d_lltf1, sora_complex16* d_ch, size_t nframes, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_lltf0 || !d_lltf1 || !d_ch) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_siso_est11n: null argument", 0);
    if (nframes == 0) return SORA_OK;
    hipLaunchKernelGGL(k_siso_est11n_batch, dim3((unsigned)((nframes + 1) / 2)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_lltf0),
                       reinterpret_cast<const uint32_t*>(d_lltf1), reinterpret_cast<uint32_t*>(d_ch), (uint32_t)nframes);
    return launch_result("k_siso_est11n_batch");
}

int sora_hip_siso_comp11n(const sora_complex16* d_ch, const uint32_t* d_frame_index, const sora_complex16* d_y0, const sora_complex16* d_y1,
                          sora_complex16* d_x0, sora_complex16* d_x1, sora_complex16* d_mrc, size_t nsym, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_ch || !d_y0 || !d_y1 || !(d_x0 || d_x1 || d_mrc)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_siso_comp11n: null argument", 0);
    if (nsym == 0) return SORA_OK;
    hipLaunchKernelGGL(k_siso_comp11n_batch, dim3((unsigned)((nsym + 3) / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_ch), d_frame_index,
                       reinterpret_cast<const uint32_t*>(d_y0), reinterpret_cast<const uint32_t*>(d_y1), reinterpret_cast<uint32_t*>(d_x0),
                       reinterpret_cast<uint32_t*>(d_x1), reinterpret_cast<uint32_t*>(d_mrc), (uint32_t)nsym);
    return launch_result("k_siso_comp11n_batch");
}

int sora_hip_sig_demap11n(const sora_complex16* d_sym, uint8_t* d_soft, size_t nframes, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_sym || !d_soft) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_sig_demap11n: null argument", 0);
    if (nframes == 0) return SORA_OK;
    hipLaunchKernelGGL(k_sig_demap11n_batch, dim3((unsigned)((nframes * 3 + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const uint32_t*>(d_sym), d_soft, (uint32_t)nframes);
    return launch_result("k_sig_demap11n_batch");
}

int sora_hip_sig_decode11n(const uint8_t* d_soft, uint32_t* d_rec, size_t nframes, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_soft || !d_rec) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_sig_decode11n: null argument", 0);
    if (nframes == 0) return SORA_OK;
    hipLaunchKernelGGL(k_sig_decode11n_batch, dim3((unsigned)((nframes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_soft, d_rec, (uint32_t)nframes);
    return launch_result("k_sig_decode11n_batch");
}
