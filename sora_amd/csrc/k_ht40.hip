// k_ht40.hip -- the data field of an 802.11n HT-mixed 40 MHz, two-stream frame (BASELINE.json configs[3]: "2x2 MIMO 40 MHz HT, 128-pt FFT,
// MMSE MIMO detect, dual Viterbi"), SURVEY.md section 8(f)1 extension.  PARITY UNPINNED: the reference has no 40 MHz receive graph -- its
// 802.11n graph is 20 MHz, zero-forcing, one decoder, MCS 8-10 (kernel/bb/Brick11/src/PHY_11n.hpp:497, channel_11n.hpp:423-433).  What IS
// pinned is every piece the reference does have, used here unchanged:
//   FFT<128>                      core/inc/fft_r4dif.h (fft128_group, pinned by tests/golden/ref_vectors.npz)
//   TFreqComp_11n                 freqoffset_11n.hpp:162-280: sat((x * sincos(n cfo - theta)) >> 15), dsp_math tables
//   TMimoChannelEst               channel_11n.hpp:329-443: P-matrix combination of the two HT-LTFs; with noise_var = 0 the weights are its
//                                 2x2 inverse x 2^16, operation for operation (bit-exact with sora_hip_mimo_est11n on the same carriers)
//   TMimoChannelComp              channel_11n.hpp:445-521: x = sat((W y) >> 9)
//   TPilotTrack_11n               pilot_11n.hpp:84-141: theta += mean pilot phase (here over 6 pilots per stream, sum / 6)
//   T11nDemap*                    demapper11n.hpp:89-309 and the dsp_demap.h tables (dev_11n.h)
//   T11aViterbi<.., 192, 36>      k_viterbi11n, one decoder per spatial stream: the two streams of a frame are the two halves of one wave
//   T11aDesc + TBB11aFrameSink    descrambler phase table + parallel CRC-32
// New, from IEEE 802.11n-2009 clause 20: the 40 MHz carrier plan (108 data + 6 pilots per stream), the HT-LTF sequence for 40 MHz, the HT
// interleaver parameters for 40 MHz (N_COL 18, N_ROW 6 N_BPSC, N_ROT 29), per-stream encoding, and the MMSE weights
// W = diag((W' H)_ss)^-1 W', W' = (H^H H + s2 I)^-1 H^H (unbiased MMSE) in single precision (documented tolerance: +-1 LSB of the int16 weight against a float64 evaluation,
// tests/test_gpu_ht40.py).  The model the tests generate captures with is oracle/py_ht40.py.
#include <vector>
#include <thread>
#include <algorithm>
#include <string.h>
#include "kernels.h"
#include "dev_11n.h"
#include "../../include/sora_hip.h"

namespace sora {

struct Ht40Frame {
    uint64_t offset;            // 40 MHz samples from the iq bases to the first sample (CP) of HT-LTF 1
    uint32_t nsym, nb, code_rate;
    uint32_t length[2];
    int32_t  cfo;               // phase per 40 MHz sample, 65536 = 2 pi (TFreqComp_11n's vfo_step convention)
    float    noise_var;         // per carrier, in LSB^2 of the FFT output; 0 = zero forcing
    // bytes from the soft base: stream 0's soft values, one byte each (VitJob::soft_bits = 8); stream 1's follow at + round_up(values per stream, 32)
    uint32_t soft_off;
    uint32_t pad[4];
};
struct Ht40Args {
    const uint32_t* iq0; const uint32_t* iq1; const Ht40Frame* frames; uint32_t nframes;
    Tables T; const uint32_t* sincos; const short* atan;
    uint8_t* soft;
    uint32_t* w_out;            // optional [nframes][4][128]: the detection weights (tests)
    const uint32_t* plan;       // raw-capture calls: {frames, events, soft bytes, error} written by k_ht40_plan (nframes above is then only a bound)
};
struct Ht40Job { uint32_t out_off, length, row, pad; };
struct Ht40FinishArgs { const Ht40Job* jobs; uint32_t njobs; const uint8_t* vout; uint8_t* mpdu; Rx11bRow* rows; Tables T; const uint32_t* plan; };

namespace {
static __constant__ int8_t kHtLtf40[117] = {    // carriers -58..58 (IEEE 802.11n-2009 eq. 20-24)
    1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1,
    -1, -1, -1, 1, 0, 0, 0, -1, 1, 1, -1,
    1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1 };
__device__ __forceinline__ void wsync40() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__device__ __forceinline__ int data_bin40(int c)          // data carrier c (0..107) -> FFT bin: -58..-2 then 2..58 without +-11, +-25, +-53
{
    int k;
    if (c < 54) { k = -58 + c; if (k >= -53) k++; if (k >= -25) k++; if (k >= -11) k++; }
    else { k = 2 + (c - 54); if (k >= 11) k++; if (k >= 25) k++; if (k >= 53) k++; }
    return k & 127;
}
__device__ __forceinline__ int deint40_index(int nb, int iss, int k)     // HT interleaver for 40 MHz: where coded bit k of stream iss sits in the symbol
{
    const int s = nb / 2 > 1 ? nb / 2 : 1, nrow = 6 * nb, np = 108 * nb;
    const int i = nrow * (k % 18) + k / 18;
    int j = s * (i / s) + (i + np - (18 * i) / np) % s;
    if (iss > 0) j = ((j - ((iss * 2) % 3 + 3 * (iss / 3)) * 29 * nb) % np + np) % np;
    return j;
}
struct Ht40Lds {
    uint32_t buf[2][128];
    uint32_t fft[2][128];
    uint32_t y[2][2][128];          // [HT-LTF symbol / scratch][chain][bin]
    uint32_t w[4][128];
    uint32_t xs[2][128];
    uint8_t  soft[2][656];
    uint16_t dtab[2][648];
};
}  // namespace

__global__ void __launch_bounds__(256) k_ht40_frame(Ht40Args A)
{
    __shared__ Ht40Lds s_w[4];
    __shared__ uint8_t s_lut[6][256];
    fill_demap_luts(s_lut);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t f = blockIdx.x * 4 + wv;
    if (f >= (A.plan ? min(A.nframes, A.plan[0]) : A.nframes)) return;
    Ht40Lds& W = s_w[wv];
    const Ht40Frame F = A.frames[f];
    const uint32_t* iq[2] = { A.iq0 + F.offset, A.iq1 + F.offset };
    const int nb = (int)F.nb, cfo = F.cfo;
    const int ncb = 108 * nb;
    for (int k = lane; k < ncb; k += 64) { W.dtab[0][k] = (uint16_t)deint40_index(nb, 0, k); W.dtab[1][k] = (uint16_t)deint40_index(nb, 1, k); }
    int theta = 0;
    auto nosync = []() __attribute__((always_inline)) { wsync40(); };
    const Fft128TwPk twpk = fft128_twiddles_pk(A.T, lane & 31);
    // one 160-sample symbol starting at sample `pos` of the frame: TFreqComp_11n, cyclic prefix dropped, FFT<128> per chain (32 lanes each) -> W.y[slot]
    auto symbol_fft = [&](uint32_t pos, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t n = pos + 32 + 64 * h + (uint32_t)lane;
            const cpx cof = unpack(A.sincos[(unsigned)((int)n * cfo - theta) & 0xFFFFu]);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                int re, im; mul32(unpack(iq[r][n]), cof, re, im);
                W.buf[r][64 * h + lane] = pack(mk(sat16(re >> 15), sat16(im >> 15)));
            }
        }
        wsync40();
        const int r = lane >> 5, e = lane & 31; pcx x[4];
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = W.buf[r][e + 32 * m];
        // the packed-arithmetic FFT<128> of k_fft128_batch (dev_arith.h): point j at slot bitrev7(j)
        fft128_core_pk(x, W.fft[r], e, twpk, nosync);
#pragma unroll
        for (int q = 0; q < 4; q++) W.y[slot][r][e + 32 * q] = W.fft[r][__brev((unsigned)(e + 32 * q)) >> 25];
        wsync40();
    };
    symbol_fft(0, 0);
    symbol_fft(160, 1);
    // ---- channel matrix per carrier (TMimoChannelEst's combination) and detection weights
#pragma unroll
    for (int hb = 0; hb < 2; hb++) {
#pragma clang fp contract(off)
        const int i = lane + 64 * hb, k = i < 64 ? i : i - 128;
        const int ltf = (k >= -58 && k <= 58) ? (int)kHtLtf40[k + 58] : 0;
        const bool negate = ltf != 1;
        cpx hh[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const cpx p = unpack(W.y[0][r][i]), q = unpack(W.y[1][r][i]);
            cpx d = sra(csubs(p, q), 1), s = sra(cadds(p, q), 1);
            if (negate) { d = mk(neg16(d.re), neg16(d.im)); s = mk(neg16(s.re), neg16(s.im)); }
            hh[r][0] = d; hh[r][1] = s;
        }
        const cf a00 = { (float)hh[0][0].re, (float)hh[0][0].im }, a01 = { (float)hh[0][1].re, (float)hh[0][1].im };
        const cf a10 = { (float)hh[1][0].re, (float)hh[1][0].im }, a11 = { (float)hh[1][1].re, (float)hh[1][1].im };
        cf w00, w01, w10, w11;
        if (F.noise_var == 0.0f) {
            // zero forcing, exactly as TMimoChannelEst computes the inverse (brick/inc/sora_matrix.h:134-148,305-313)
            const cf ad = cf_mul(a00, a11), bc = cf_mul(a01, a10);
            const cf det = { ad.re - bc.re, ad.im - bc.im };
            const float nn = ((det.re * det.re) + (det.im * det.im)) / 65536.0f;
            const cf ds = { det.re, -det.im }, m01 = { -a01.re, -a01.im }, m10 = { -a10.re, -a10.im };
            const cf r00 = cf_mul(a11, ds), r01 = cf_mul(m01, ds), r10 = cf_mul(m10, ds), r11 = cf_mul(a00, ds);
            w00 = { r00.re / nn, r00.im / nn }; w01 = { r01.re / nn, r01.im / nn }; w10 = { r10.re / nn, r10.im / nn }; w11 = { r11.re / nn, r11.im / nn };
        } else {
            // MMSE: W = (H^H H + s2 I)^-1 H^H = adj(G) H^H / det(G), G Hermitian
            auto cj = [](cf a) { return cf{ a.re, -a.im }; };
            auto n2 = [](cf a) { return (a.re * a.re) + (a.im * a.im); };
            const float g00 = (n2(a00) + n2(a10)) + F.noise_var, g11 = (n2(a01) + n2(a11)) + F.noise_var;
            const cf t0 = cf_mul(cj(a00), a01), t1 = cf_mul(cj(a10), a11);
            const cf g01 = { t0.re + t1.re, t0.im + t1.im };
            const float det = ((g00 * g11) - n2(g01)) / 65536.0f;
            // g11 conj(h_r0) - g01 conj(h_r1)
            auto row0 = [&](cf hc0, cf hc1) { const cf u = cf_mul(g01, hc1); return cf{ ((g11 * hc0.re) - u.re) / det, ((g11 * hc0.im) - u.im) / det }; };
            // g00 conj(h_r1) - g10 conj(h_r0)
            auto row1 = [&](cf hc0, cf hc1) { const cf u = cf_mul(cj(g01), hc0); return cf{ ((g00 * hc1.re) - u.re) / det, ((g00 * hc1.im) - u.im) / det }; };
            w00 = row0(cj(a00), cj(a01)); w01 = row0(cj(a10), cj(a11));
            w10 = row1(cj(a00), cj(a01)); w11 = row1(cj(a10), cj(a11));
            // unbiased: row s is divided by (W H)_ss = the gain the stream's own symbol comes out with, so that the constellation sits where
            // the demapper's fixed tables expect it (at s2 = 0 that gain is 1)
            const cf e0 = cf_mul(w00, a00), e1 = cf_mul(w01, a10), e2 = cf_mul(w10, a01), e3 = cf_mul(w11, a11);
            const float beta0 = (e0.re + e1.re) / 65536.0f, beta1 = (e2.re + e3.re) / 65536.0f;
            w00 = { w00.re / beta0, w00.im / beta0 }; w01 = { w01.re / beta0, w01.im / beta0 };
            w10 = { w10.re / beta1, w10.im / beta1 }; w11 = { w11.re / beta1, w11.im / beta1 };
        }
        W.w[0][i] = pack(mk(cvtps_sat16(w00.re), cvtps_sat16(w00.im))); W.w[1][i] = pack(mk(cvtps_sat16(w01.re), cvtps_sat16(w01.im)));
        W.w[2][i] = pack(mk(cvtps_sat16(w10.re), cvtps_sat16(w10.im))); W.w[3][i] = pack(mk(cvtps_sat16(w11.re), cvtps_sat16(w11.im)));
        if (A.w_out) {
#pragma unroll
            for (int m = 0; m < 4; m++) A.w_out[((size_t)f * 4 + m) * 128 + i] = W.w[m][i];
        }
    }
    wsync40();
    // ---- data symbols, in order (the pilot phase of symbol d rotates symbol d + 1)
    uint8_t* dst = A.soft + F.soft_off;
    const uint32_t per_pad = (F.nsym * 108u * F.nb + 31u) / 32u * 32u;         // stream 1 starts here (ht40_submit lays the job records out the same way)
    for (uint32_t d = 0; d < F.nsym; d++) {
        symbol_fft(320 + 160 * d, 0);
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {                                     // TMimoChannelComp
            const int i = lane + 64 * hb;
            const cpx p = unpack(W.y[0][0][i]), q = unpack(W.y[0][1][i]);
            int ar, ai, br, bi;
            mul32(unpack(W.w[0][i]), p, ar, ai); mul32(unpack(W.w[1][i]), q, br, bi);
            W.xs[0][i] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
            mul32(unpack(W.w[2][i]), p, ar, ai); mul32(unpack(W.w[3][i]), q, br, bi);
            W.xs[1][i] = pack(mk(sat16((int)((unsigned)ar + (unsigned)br) >> 9), sat16((int)((unsigned)ai + (unsigned)bi) >> 9)));
        }
        wsync40();
        {   // pilot phases: lane 8 s + k (k < 6) takes pilot k of stream s
            const int k = lane & 7, sidx = (lane >> 3) & 1;
            const int pk = k == 0 ? -53 : k == 1 ? -25 : k == 2 ? -11 : k == 3 ? 11 : k == 4 ? 25 : 53;
            const cpx v = unpack(W.xs[sidx][pk & 127]);
            int th = (k < 6) ? dsp_atan16(A.atan, v.re, v.im) : 0;
            th += __shfl_xor(th, 1); th += __shfl_xor(th, 2); th += __shfl_xor(th, 4);
            const int t0 = (int)(short)(__builtin_amdgcn_readfirstlane(__shfl(th, 0)) / 6), t1 = (int)(short)(__builtin_amdgcn_readfirstlane(__shfl(th, 8)) / 6);
            theta = (int)(short)(theta + (int)(short)((t0 + t1) >> 1));
        }
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {                                     // T11nDemap*: I bits then Q bits per carrier
            const int c = lane + 64 * hb;
            if (c < 108) {
                const int bin = data_bin40(c);
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const cpx x = unpack(W.xs[s][bin]);
                    const int re = min(max(x.re, -128), 127) + 128, im = min(max(x.im, -128), 127) + 128;
                    uint8_t* o = W.soft[s] + c * nb;
                    switch (nb) {
                    case 1: o[0] = s_lut[0][re]; break;
                    case 2: o[0] = s_lut[0][re]; o[1] = s_lut[0][im]; break;
                    case 4: o[0] = s_lut[1][re]; o[1] = s_lut[2][re]; o[2] = s_lut[1][im]; o[3] = s_lut[2][im]; break;
                    default: o[0] = s_lut[3][re]; o[1] = s_lut[4][re]; o[2] = s_lut[5][re]; o[3] = s_lut[3][im]; o[4] = s_lut[4][im]; o[5] = s_lut[5][im];
                    }
                }
            }
        }
        wsync40();
        // de-interleave both streams, each into its own byte stream (one decoder wave takes stream 0 in its low halves and stream 1 in its high halves)
        for (int g = lane; g < ncb; g += 64) {
            dst[(size_t)d * ncb + g] = W.soft[0][W.dtab[0][g]];
            dst[per_pad + (size_t)d * ncb + g] = W.soft[1][W.dtab[1][g]];
        }
        wsync40();
    }
}

__global__ void __launch_bounds__(256) k_ht40_finish(Ht40FinishArgs A)
{
    __shared__ uint32_t s_crc[256];
    __shared__ uint32_t s_z[6 * 8 * 16];
    __shared__ uint32_t s_bufs[4][4096 / 4 + 2];
    s_crc[threadIdx.x] = A.T.crc[threadIdx.x];
    for (int i = threadIdx.x; i < 6 * 8 * 16; i += 256) s_z[i] = A.T.crcz[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const uint32_t j = blockIdx.x * 4 + wv;
    if (j >= (A.plan ? min(A.njobs, 2u * A.plan[0]) : A.njobs)) return;
    const Ht40Job J = A.jobs[j];
    const uint8_t* dec = A.vout + J.out_off;
    uint8_t* mp = A.mpdu + (size_t)J.row * 4096;
    uint8_t* bytes = reinterpret_cast<uint8_t*>(s_bufs[wv]);
    const uint32_t L = J.length;
    const unsigned seed = dec[1] >> 1;
    const unsigned phase = A.T.scr_phase[seed & 0x7F];
    for (uint32_t i = lane; i < L; i += 64) {
        const unsigned sb = phase == 255 ? 0u : A.T.scr_seq[(phase + 8u * i) % 127u];
        const unsigned o = dec[2 + i] ^ sb;
        bytes[i] = (uint8_t)o; mp[i] = (uint8_t)o;
    }
    wsync40();
    const int n = L >= 4 ? (int)L - 4 : 0;
    uint32_t crc;
    if (n >= 4) crc = crc32_wave(bytes, n, s_crc, s_z, lane);
    else { crc = 0xFFFFFFFFu; for (int i = 0; i < n; i++) crc = (crc >> 8) ^ s_crc[(bytes[i] ^ crc) & 0xFF]; }
    if (lane == 0) {
        uint32_t fcs = 0;
        if (L >= 4) fcs = (uint32_t)bytes[L - 4] | ((uint32_t)bytes[L - 3] << 8) | ((uint32_t)bytes[L - 2] << 16) | ((uint32_t)bytes[L - 1] << 24);
        Rx11bRow r; r.end_sample = 0; r.rate_kbps = 0; r.length = L; r.crc32 = fcs; r.error_code = ((~crc) == fcs) ? 1u : 0x80000006u;
        A.rows[J.row] = r;
    }
}

// ---- raw-capture calls: what the front end found (k_scan_ht40: per capture a count and up to `mf` Ht40Found records in time order) -> the data
// field's tables, on the device, so that the call is one uninterrupted chain of kernels (the first version read the records back and built the
// tables on the host: one host wait per call).  One block: per capture the number of recorded frames, their soft bytes and their number per
// code rate; five exclusive prefix sums over the captures; then every capture writes its frames' descriptors, the two decoder jobs of each
// (neighbours in their code-rate list: one wave decodes both streams), the finish jobs and the rows' templates, in (capture, time) order.
__device__ __forceinline__ uint32_t ht40_ndbps_dev(uint32_t nb, uint32_t cr) { return 108u * nb * (cr == 0 ? 1u : cr == 1 ? 2u : 3u) / (cr == 0 ? 2u : cr == 1 ? 3u : 4u); }
struct Ht40Geom { uint32_t nb, cr, nsym, per, per_pad; };
__device__ __forceinline__ Ht40Geom ht40_geom(const Ht40Found& F)
{
    Ht40Geom G;
    G.nb = F.mcs == 8 ? 1u : F.mcs <= 10 ? 2u : F.mcs <= 12 ? 4u : 6u;
    G.cr = (F.mcs == 10 || F.mcs == 12 || F.mcs == 14) ? 2u : F.mcs == 13 ? 1u : 0u;
    const uint32_t nd = ht40_ndbps_dev(G.nb, G.cr);
    G.nsym = (16u + 8u * F.ht_len + 6u + nd - 1u) / nd;                          // sora_ht40_symbols(len, len, nb, cr)
    G.per = G.nsym * 108u * G.nb; G.per_pad = (G.per + 31u) / 32u * 32u;
    return G;
}
__global__ void __launch_bounds__(1024) k_ht40_plan(const CapDesc* __restrict__ caps, uint32_t ncaps, uint32_t mf, const uint32_t* __restrict__ nfr,
        const Ht40Found* __restrict__ found,
                                                    uint32_t max_frames, uint64_t max_soft, uint32_t vout_stride,
                                                    Ht40Frame* __restrict__ frames, VitJob* __restrict__ jobs, uint32_t stride, uint32_t* __restrict__ njobs, Ht40Job* __restrict__ fjobs,
                                                    sora_frame_result* __restrict__ tmpl, uint32_t* __restrict__ plan, uint32_t* __restrict__ evbase, uint32_t* __restrict__ evn)
{
    // Besides the data field's tables: the call's EVENT table for sora_ht40_deliver_async, in (capture, time) order like sora_ht40_results_of -- event e has evn[e] rows (two for a
    // recorded frame, one for a header that failed), template rows tmpl[2 e + k], and evbase[e] = the row of the decoder's row table its rows start at (0xFFFFFFFF: no frame).
    __shared__ uint32_t s_v[6][1024];
    __shared__ uint32_t s_base[6];
    __shared__ uint32_t s_err;
    const uint32_t t = threadIdx.x;
    if (t < 6) s_base[t] = 0;
    if (t == 0) s_err = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < ncaps; c0 += 1024) {
        const uint32_t c = c0 + t;
        const uint32_t n = c < ncaps ? min(nfr[c], mf) : 0u;
        uint32_t mine[6] = { 0, 0, 0, 0, 0, n };                                 // frames, soft bytes, frames of code rate 0 / 1 / 2, events
        for (uint32_t i = 0; i < n; i++) {
            const Ht40Found& F = found[(size_t)c * mf + i];
            if (F.error_code != 0) continue;
            const Ht40Geom G = ht40_geom(F);
            mine[0]++; mine[1] += 2u * G.per_pad; mine[2 + G.cr]++;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) s_v[k][t] = mine[k];
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {                                // Hillis-Steele inclusive scans of the six counters
            uint32_t add[6];
#pragma unroll
            for (int k = 0; k < 6; k++) add[k] = t >= o ? s_v[k][t - o] : 0u;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 6; k++) s_v[k][t] += add[k];
            __syncthreads();
        }
        uint32_t at[6];
#pragma unroll
        for (int k = 0; k < 6; k++) at[k] = s_base[k] + s_v[k][t] - mine[k];
        for (uint32_t i = 0; i < n; i++) {
            const Ht40Found& F = found[(size_t)c * mf + i];
            const uint32_t e = at[5]++;                                          // the event's index in the call
            const uint16_t tflag = (uint16_t)((i + 1 == mf && nfr[c] > mf) ? SORA_ROW_TRUNCATED : 0);
            // a header that failed: one row, no frame (what sora_ht40_results_of reports for it)
            if (F.error_code != 0) {
                sora_frame_result o;
                o.capture_id = caps[c].capture_id; o.start_sample = 0; o.end_sample = F.end_sample; o.error_code = F.error_code; o.rate_kbps = 0;
                o.length = 0; o.nsym = 0; o.crc32 = 0; o.cfo_est = 0; o.flags = tflag; o.mpdu_offset = 0;
                tmpl[2u * e] = o; evbase[e] = 0xFFFFFFFFu; evn[e] = 1u;
                continue;
            }
            const Ht40Geom G = ht40_geom(F);
            const uint32_t fi = at[0], soft_off = at[1], pos = 2u * at[2 + G.cr];
            at[0]++; at[1] += 2u * G.per_pad; at[2 + G.cr]++;
            evbase[e] = 2u * fi; evn[e] = 2u;
            // (reported by wait / results: SORA_ERR_CAPACITY)
            if (fi >= max_frames || (uint64_t)soft_off + 2u * G.per_pad > max_soft || !(F.noise_var >= 0.0f)) { s_err = 1u; evn[e] = 0u; continue; }
            Ht40Frame H;
            H.offset = caps[c].offset + 2ull * F.a20 + 160ull;                   // HT-STF is 4 us = 160 samples @40 MHz; HT-LTF 1 follows
            // one HT-SIG LENGTH: each stream carries its own PSDU of that length
            H.nsym = G.nsym; H.nb = G.nb; H.code_rate = G.cr; H.length[0] = H.length[1] = F.ht_len;
            H.cfo = F.cfo / 2;                                                   // per 20 MHz sample -> per 40 MHz sample
            H.noise_var = F.noise_var; H.soft_off = soft_off; H.pad[0] = H.pad[1] = H.pad[2] = H.pad[3] = 0;
            frames[fi] = H;
#pragma unroll
            for (uint32_t k = 0; k < 2; k++) {
                VitJob J;
                J.soft_off = soft_off + k * G.per_pad; J.soft_bits = 8; J.nsoft = G.per; J.length = F.ht_len; J.dec_off = 0;
                    J.out_off = (2u * fi + k) * vout_stride; J.valid = 1; J.code_rate = G.cr;
                jobs[(size_t)G.cr * stride + pos + k] = J;
                fjobs[2u * fi + k] = Ht40Job{ J.out_off, F.ht_len, 2u * fi + k, 0u };
                sora_frame_result o;
                o.capture_id = caps[c].capture_id; o.start_sample = k; o.end_sample = F.end_sample; o.error_code = 0; o.rate_kbps = F.mcs;
                o.length = 0; o.nsym = (uint16_t)G.nsym; o.crc32 = 0; o.cfo_est = 0; o.flags = tflag; o.mpdu_offset = 0;
                tmpl[2u * e + k] = o;
            }
        }
        __syncthreads();
        if (t == 1023) {
#pragma unroll
            for (int k = 0; k < 6; k++) s_base[k] += s_v[k][1023];
        }
        __syncthreads();
    }
    // (a batch that does not fit is not decoded at all: the job lists would have holes)
    if (t == 0) {
        const bool bad = s_err != 0;
        plan[0] = bad ? 0u : s_base[0]; plan[1] = s_base[0]; plan[2] = s_base[1]; plan[3] = s_err; plan[4] = bad ? 0u : s_base[5];
        njobs[0] = bad ? 0u : 2u * s_base[2]; njobs[1] = bad ? 0u : 2u * s_base[3]; njobs[2] = bad ? 0u : 2u * s_base[4]; njobs[3] = 0;
    }
}

}  // namespace sora

// ------------------------------------------------------------------------------------------------ host side (C ABI, include/sora_hip.h)
using namespace sora;

// A handle owns kHt40Slots independent slots (stream + every intermediate), used round-robin: a call waits only for the call that used its
// slot kHt40Slots calls ago, so the host's part of call n + 1 (job tables, four small copies) and the tail of call n's kernels overlap call n + 1's
// kernels.  sora_ht40_results reports the most recent call.
static constexpr int kHt40Slots = 8;
// frame: index into the call's described frames, -1 = header failed
struct Ht40Event { uint32_t capture_id, end_sample, error_code, mcs, length, nsym; int frame; bool truncated; };
struct Ht40Slot {
    hipStream_t stream = nullptr;
    Ht40Frame* d_frames = nullptr; VitJob* d_jobs = nullptr; uint32_t* d_njobs = nullptr; Ht40Job* d_fjobs = nullptr;
    uint8_t* d_soft = nullptr; uint8_t* d_vout = nullptr; uint8_t* d_mpdu = nullptr; Rx11bRow* d_rows = nullptr;
    std::vector<sora_ht40_frame> h_frames; uint32_t nframes = 0;
    int ticket = 0;             // of the call this slot holds (0: none)
    hipEvent_t ev_done = nullptr; bool delivered = false, released = false;     // sora_ht40_wait_any (kernels.h: slots_next / slots_poll)
    // sora_ht40_process_captures_dev: the front end's arrays (grow-only) and what it found in this slot's call
    CapDesc* d_caps = nullptr; size_t caps_bytes = 0; Rx11bRow* d_scanrows = nullptr; size_t scanrows_bytes = 0;
    uint32_t* d_nfr = nullptr; size_t nfr_bytes = 0; Ht40Found* d_found = nullptr; size_t found_bytes = 0;
    bool capture_mode = false; uint32_t capture_mf = 0; std::vector<Ht40Event> events;
    // ... planned on the device (k_ht40_plan): the records come back asynchronously into page-locked memory and are turned into `events` when the call is collected
    uint32_t* d_plan = nullptr;
    // the call's event table (k_ht40_plan -> sora_ht40_deliver_async)
    uint32_t* d_evbase = nullptr; size_t evbase_bytes = 0; uint32_t* d_evn = nullptr; size_t evn_bytes = 0;
    sora_frame_result* d_evtmpl = nullptr; size_t evtmpl_bytes = 0; uint32_t bound_events = 0;
    // descriptor calls: page-locked staging of {frames, job lists, finish jobs, counts} for asynchronous uploads
    void* h_stage = nullptr;
    void* h_pin = nullptr; size_t pin_bytes = 0;                                // {plan[4], CapDesc[ncaps] (upload), nfr[ncaps], Ht40Found[ncaps * mf]}
    uint32_t* h_plan = nullptr; CapDesc* h_capsup = nullptr; uint32_t* h_nfr = nullptr; Ht40Found* h_found = nullptr;
    std::vector<sora_capture_desc> h_caps; bool events_pending = false; uint32_t bound_frames = 0; bool plan_error = false;
    DenseStage dense;           // sora_ht40_deliver_async
    std::vector<sora_frame_result> h_tmpl;                                      // what the rows of this call carry besides the decoder's verdict
};
struct sora_ht40 {
    int device = 0; uint32_t max_frames = 0; uint64_t max_soft = 0;
    Tables T{}; const uint32_t* sincos = nullptr; const short* atan = nullptr;
    Ht40Slot slot[kHt40Slots]; int next = 0, last = 0, seq = 0; bool have_results = false;
    // trellis kernel: 1 = k_viterbi16_11n (default: the handle keeps eight calls in flight), 0 =
    // k_viterbi11n (64 lanes per stream pair; the faster one for a call alone) -- sora_ht40_set_trellis
    int lanes16 = 1;
};

#define HIPCHK40(call) do { hipError_t _e = (call); if (_e != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, #call, (int)_e); } while (0)
static constexpr uint32_t kVoutStride = 4352;

static void ht40_free(sora_ht40_t* rx)
{
    if (!rx) return;
    for (Ht40Slot& S : rx->slot) {
        if (S.stream) { (void)hipStreamSynchronize(S.stream); (void)hipStreamDestroy(S.stream); }
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        (void)hipFree(S.d_frames); (void)hipFree(S.d_jobs); (void)hipFree(S.d_njobs); (void)hipFree(S.d_fjobs); (void)hipFree(S.d_soft);
        (void)hipFree(S.d_vout); (void)hipFree(S.d_mpdu); (void)hipFree(S.d_rows);
        sora_internal_dense_free(&S.dense);
        (void)hipFree(S.d_caps); (void)hipFree(S.d_scanrows); (void)hipFree(S.d_nfr); (void)hipFree(S.d_found);
        (void)hipFree(S.d_plan); (void)hipFree(S.d_evbase); (void)hipFree(S.d_evn); (void)hipFree(S.d_evtmpl); if (S.h_pin) (void)hipHostFree(S.h_pin);
            if (S.h_stage) (void)hipHostFree(S.h_stage);
    }
    delete rx;
}

static uint32_t ht40_ndbps(uint32_t nb, uint32_t cr) { return 108u * nb * (cr == 0 ? 1u : cr == 1 ? 2u : 3u) / (cr == 0 ? 2u : cr == 1 ? 3u : 4u); }

uint32_t sora_ht40_symbols(uint32_t length0, uint32_t length1, uint32_t n_bpsc, uint32_t code_rate)
{
    if (!(n_bpsc == 1 || n_bpsc == 2 || n_bpsc == 4 || n_bpsc == 6) || code_rate > 2) return 0;
    const uint32_t L = length0 > length1 ? length0 : length1, nd = ht40_ndbps(n_bpsc, code_rate);
    return (16u + 8u * L + 6u + nd - 1) / nd;
}

int sora_ht40_create(int device, uint32_t max_frames, uint64_t max_soft_values, sora_ht40_t** out)
{
    if (!out || max_frames == 0 || max_soft_values == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_create: bad argument", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (device < 0 || device >= ndev) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "device ordinal out of range", 0);
    if (max_soft_values * 2 + 4096 + 1024 >= (1ull << 32)) return sora_internal_fail(SORA_ERR_CAPACITY,
            "sora_ht40_create: max_soft_values exceeds the 32-bit offsets of one handle", 0);
    HIPCHK40(hipSetDevice(device));
    sora_ht40_t* rx = new sora_ht40();
    rx->device = device; rx->max_frames = max_frames; rx->max_soft = max_soft_values;
    const size_t nj = 2 * (size_t)max_frames;
    hipError_t e = (sora_internal_tables(device, &rx->T) == SORA_OK && sora_internal_dsp_tables(&rx->sincos, &rx->atan) == SORA_OK) ? hipSuccess : hipErrorUnknown;
    for (Ht40Slot& S : rx->slot) {
        if (e == hipSuccess) e = sora_internal_stream_create(&S.stream, (int)(&S - &rx->slot[0]));
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_frames, sizeof(Ht40Frame) * max_frames);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_jobs, 3 * sizeof(VitJob) * nj);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_njobs, 16);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_fjobs, sizeof(Ht40Job) * nj);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_soft, max_soft_values * 2 + 4096 + 1024);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_vout, nj * kVoutStride + 256);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_mpdu, nj * 4096);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_rows, sizeof(Rx11bRow) * nj);
        if (e == hipSuccess) e = hipMalloc((void**)&S.d_plan, 32);
        if (e == hipSuccess) e = hipHostMalloc(&S.h_stage, sizeof(Ht40Frame) * max_frames + 3 * sizeof(VitJob) * nj + sizeof(Ht40Job) * nj + 64, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMemset(S.d_soft, 0, max_soft_values * 2 + 4096 + 1024);
        if (e == hipSuccess) e = hipMemset(S.d_vout, 0, nj * kVoutStride + 256);
    }
    if (e != hipSuccess) { ht40_free(rx); return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_ht40_create: device allocation / tables", (int)e); }
    *out = rx;
    return SORA_OK;
}

void  sora_ht40_destroy(sora_ht40_t* rx) { if (rx) { (void)hipSetDevice(rx->device); ht40_free(rx); } }
void* sora_ht40_stream(sora_ht40_t* rx) { return rx ? (void*)rx->slot[rx->last].stream : nullptr; }         // the stream of the most recent call
int sora_ht40_set_trellis(sora_ht40_t* rx, int lanes_per_pair)
{
    if (!rx) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_set_trellis: null handle", 0);
    const int old = rx->lanes16 ? 16 : 64;
    if (lanes_per_pair == 16 || lanes_per_pair == 64) rx->lanes16 = lanes_per_pair == 16;
    else if (lanes_per_pair >= 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_set_trellis: 16 or 64 lanes per stream pair", 0);
    return old;
}

int sora_ht40_synchronize(sora_ht40_t* rx)
{
    if (!rx) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_synchronize: null handle", 0);
    HIPCHK40(hipSetDevice(rx->device));
    for (Ht40Slot& S : rx->slot) HIPCHK40(hipStreamSynchronize(S.stream));
    return SORA_OK;
}

// the data field of `nframes` described frames on slot S (its stream is idle): descriptors -> device, k_ht40_frame, the trellis kernel, k_ht40_finish
static int ht40_submit(sora_ht40_t* rx, Ht40Slot& S, const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_ht40_frame* frames, size_t nframes,
        sora_complex16* d_weights)
{
    if (nframes > rx->max_frames) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_process_dev: more frames than max_frames", 0);
    // (the tables are built in the slot's page-locked staging area -- the slot's previous call has finished -- and uploaded asynchronously)
    Ht40Frame* hf = reinterpret_cast<Ht40Frame*>(S.h_stage);
    VitJob* hj = reinterpret_cast<VitJob*>(hf + rx->max_frames);
    Ht40Job* fj = reinterpret_cast<Ht40Job*>(hj + 3 * 2 * (size_t)rx->max_frames);
    uint32_t* nj_stage = reinterpret_cast<uint32_t*>(fj + 2 * (size_t)rx->max_frames);
    uint32_t nj[4] = { 0, 0, 0, 0 };
    uint64_t soft = 0;
    const size_t stride = 2 * (size_t)rx->max_frames;
    for (size_t i = 0; i < nframes; i++) {
        const sora_ht40_frame& s = frames[i];
        const uint32_t nsym = sora_ht40_symbols(s.length[0], s.length[1], s.n_bpsc, s.code_rate);
        if (nsym == 0 || s.length[0] > 4000 || s.length[1] > 4000 || !(s.noise_var >= 0.0f)) return sora_internal_fail(SORA_ERR_INVALID_PARAM,
                "sora_ht40_process_dev: bad frame descriptor (n_bpsc 1/2/4/6, code_rate 0..2, PSDU <= 4000 bytes, noise_var >= 0)", 0);
        Ht40Frame& F = hf[i];
        F.offset = s.offset; F.nsym = nsym; F.nb = s.n_bpsc; F.code_rate = s.code_rate; F.length[0] = s.length[0]; F.length[1] = s.length[1]; F.cfo = s.cfo;
            F.noise_var = s.noise_var;
        const uint64_t per = (uint64_t)nsym * 108 * s.n_bpsc;                       // soft values per stream
        F.soft_off = (uint32_t)soft;
        for (int k = 0; k < 2; k++) {
            // the two streams of a frame are neighbours in their list: one wave decodes both
            VitJob& J = hj[s.code_rate * stride + nj[s.code_rate]++];
            J.soft_off = F.soft_off + (uint32_t)k * (uint32_t)((per + 31) / 32 * 32); J.soft_bits = 8; J.nsoft = (uint32_t)per; J.length = s.length[k];
                J.dec_off = 0; J.out_off = (uint32_t)((2 * i + k) * kVoutStride); J.valid = 1; J.code_rate = s.code_rate;
            fj[2 * i + k] = Ht40Job{ J.out_off, s.length[k], (uint32_t)(2 * i + k), 0 };
        }
        soft += 2 * ((per + 31) / 32 * 32);                                         // bytes: one per soft value, both streams
        F.pad[0] = F.pad[1] = F.pad[2] = F.pad[3] = 0;
    }
    if (soft > rx->max_soft) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_process_dev: more soft values than max_soft_values", 0);
    S.h_frames.assign(frames, frames + nframes); S.nframes = (uint32_t)nframes; rx->have_results = true;
    rx->last = rx->next;
    S.ticket = ++rx->seq; S.delivered = S.released = false;
    if (nframes == 0) return SORA_OK;
    for (int r = 0; r < 4; r++) nj_stage[r] = nj[r];
    HIPCHK40(hipMemcpyAsync(S.d_frames, hf, sizeof(Ht40Frame) * nframes, hipMemcpyHostToDevice, S.stream));
    for (int r = 0; r < 3; r++)                                                   // (only the filled part of each code-rate list)
        if (nj[r]) HIPCHK40(hipMemcpyAsync(S.d_jobs + r * stride, hj + r * stride, sizeof(VitJob) * nj[r], hipMemcpyHostToDevice, S.stream));
    HIPCHK40(hipMemcpyAsync(S.d_njobs, nj_stage, 16, hipMemcpyHostToDevice, S.stream));
    HIPCHK40(hipMemcpyAsync(S.d_fjobs, fj, sizeof(Ht40Job) * 2 * nframes, hipMemcpyHostToDevice, S.stream));
    Ht40Args A;
    A.iq0 = reinterpret_cast<const uint32_t*>(d_iq0); A.iq1 = reinterpret_cast<const uint32_t*>(d_iq1); A.frames = S.d_frames; A.nframes = (uint32_t)nframes;
    A.T = rx->T; A.sincos = rx->sincos; A.atan = rx->atan; A.soft = S.d_soft; A.w_out = reinterpret_cast<uint32_t*>(d_weights); A.plan = nullptr;
    hipLaunchKernelGGL(k_ht40_frame, dim3((unsigned)((nframes + 3) / 4)), dim3(256), 0, S.stream, A);
    const uint32_t njobs = 2 * (uint32_t)nframes;
    if (rx->lanes16)
        hipLaunchKernelGGL(k_viterbi16_11n, dim3((njobs + 7) / 8 + 2), dim3(64), 0, S.stream, (const VitJob*)S.d_jobs, (const uint32_t*)S.d_njobs, 0u,
                (uint32_t)stride, (const uint8_t*)S.d_soft, S.d_vout);
    else
        hipLaunchKernelGGL(k_viterbi11n, dim3((njobs / 2 + 3 + 3) / 4), dim3(256), 0, S.stream, (const VitJob*)S.d_jobs, (const uint32_t*)S.d_njobs, 0u,
                (uint32_t)stride, (const uint8_t*)S.d_soft, S.d_vout);
    Ht40FinishArgs Fi; Fi.jobs = S.d_fjobs; Fi.njobs = njobs; Fi.vout = S.d_vout; Fi.mpdu = S.d_mpdu; Fi.rows = S.d_rows; Fi.T = rx->T; Fi.plan = nullptr;
    hipLaunchKernelGGL(k_ht40_finish, dim3((njobs + 3) / 4), dim3(256), 0, S.stream, Fi);
    HIPCHK40(hipGetLastError());
    return SORA_OK;
}

int sora_ht40_process_dev(sora_ht40_t* rx, const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_ht40_frame* frames, size_t nframes, sora_complex16* d_weights)
{
    if (!rx || (nframes && (!d_iq0 || !d_iq1 || !frames))) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_process_dev: null argument", 0);
    HIPCHK40(hipSetDevice(rx->device));
    rx->next = slots_next(rx->slot, kHt40Slots);                                  // an unused slot, else a released call's, else the oldest call's
    Ht40Slot& S = rx->slot[rx->next];
    HIPCHK40(hipStreamSynchronize(S.stream));                                     // the call that used this slot kHt40Slots calls ago
    S.events.clear(); S.capture_mode = false; S.events_pending = false; S.plan_error = false;
    return ht40_submit(rx, S, d_iq0, d_iq1, frames, nframes, d_weights);
}

// ---- the same receiver on RAW CAPTURES (BASELINE configs[3] as every other handle takes its input): two-chain 40 MHz captures in, the
// front end (k_scan_ht40 in k_rx11n.hip: the reference's 20 MHz carrier sense / L-LTF / SIG bricks on the duplicated legacy preamble)
// finds the frames, parses HT-SIG and estimates CFO and noise variance; k_ht40_plan turns its records into the data field's tables on the
// device, and the data-field kernels follow in the same stream: no host wait inside a call (the first version read the records back and
// planned on the host).  The records travel to page-locked host memory behind the kernels and become the call's events when it is
// collected.  Rows: per event in (capture, time) order -- a recorded frame reports two rows (start_sample = spatial stream 0 / 1,
// rate_kbps = MCS, end_sample = the 40 MHz source position of the event), a header that fails one row with SORA_E_PLCP_HEADER_FAIL.
int sora_ht40_process_captures_dev(sora_ht40_t* rx, const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_capture_desc* caps, size_t ncaps,
        uint32_t max_frames_per_capture)
{
    if (!rx || (ncaps && (!d_iq0 || !d_iq1 || !caps)) || max_frames_per_capture == 0) return sora_internal_fail(SORA_ERR_INVALID_PARAM,
            "sora_ht40_process_captures_dev: bad argument", 0);
    if ((uint64_t)ncaps * max_frames_per_capture >= (1ull << 31)) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_process_captures_dev: too many rows", 0);
    HIPCHK40(hipSetDevice(rx->device));
    rx->next = slots_next(rx->slot, kHt40Slots);                                  // an unused slot, else a released call's, else the oldest call's
    Ht40Slot& S = rx->slot[rx->next];
    HIPCHK40(hipStreamSynchronize(S.stream));                                     // the call that used this slot kHt40Slots calls ago
    const uint32_t mf = max_frames_per_capture;
    const size_t nrows = ncaps * (size_t)mf;
    for (size_t i = 0; i < ncaps; i++) {
        if (caps[i].offset & 3) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "capture offset must be a multiple of 4 samples", 0);
        if (caps[i].nsamples % 28) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "capture length must be a whole number of 28-sample source bursts", 0);
    }
    auto grow = [](void** p, size_t* have, size_t need) -> bool { if (*have >= need) return true; if (*p) (void)hipFree(*p); *p = nullptr; *have = 0;
        if (hipMalloc(p, need) != hipSuccess) return false; *have = need; return true; };
    if (ncaps && (!grow((void**)&S.d_caps, &S.caps_bytes, sizeof(CapDesc) * ncaps) || !grow((void**)&S.d_scanrows, &S.scanrows_bytes, sizeof(Rx11bRow) * nrows) ||
                  !grow((void**)&S.d_nfr, &S.nfr_bytes, 4 * ncaps) || !grow((void**)&S.d_found, &S.found_bytes, sizeof(Ht40Found) * nrows) ||
                  !grow((void**)&S.d_evbase, &S.evbase_bytes, 4 * nrows) || !grow((void**)&S.d_evn, &S.evn_bytes, 4 * nrows) ||
                  !grow((void**)&S.d_evtmpl, &S.evtmpl_bytes, sizeof(sora_frame_result) * 2 * nrows)))
        return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_ht40_process_captures_dev: device allocation", 0);
    const size_t o_caps = 64, o_nfr = o_caps + ((sizeof(CapDesc) * ncaps + 63) & ~(size_t)63), o_found = o_nfr + ((4 * ncaps + 63) & ~(size_t)63), need = o_found + sizeof(Ht40Found) * nrows + 64;
    if (S.pin_bytes < need) {
        if (S.h_pin) { (void)hipHostFree(S.h_pin); S.h_pin = nullptr; S.pin_bytes = 0; }
        if (hipHostMalloc(&S.h_pin, need, hipHostMallocDefault) != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED,
                "sora_ht40_process_captures_dev: page-locked host memory", 0);
        S.pin_bytes = need;
    }
    S.h_plan = reinterpret_cast<uint32_t*>(S.h_pin); S.h_capsup = reinterpret_cast<CapDesc*>((uint8_t*)S.h_pin + o_caps);
    S.h_nfr = reinterpret_cast<uint32_t*>((uint8_t*)S.h_pin + o_nfr); S.h_found = reinterpret_cast<Ht40Found*>((uint8_t*)S.h_pin + o_found);
    for (size_t i = 0; i < ncaps; i++) { CapDesc& h = S.h_capsup[i]; h.offset = caps[i].offset; h.nsamples = caps[i].nsamples;
        h.capture_id = caps[i].capture_id; h.slot_base = 0; h.nslots = 0; }
    S.h_caps.assign(caps, caps + ncaps);
    S.events.clear(); S.capture_mode = true; S.capture_mf = mf; S.events_pending = true; S.plan_error = false; S.nframes = 0;
    S.bound_frames = (uint32_t)std::min<uint64_t>(nrows, rx->max_frames); S.bound_events = (uint32_t)nrows;
    rx->have_results = true; rx->last = rx->next;
    S.ticket = ++rx->seq; S.delivered = S.released = false;
    S.h_plan[0] = S.h_plan[1] = S.h_plan[2] = S.h_plan[3] = 0;
    if (ncaps == 0) { S.events_pending = false; return SORA_OK; }
    HIPCHK40(hipMemcpyAsync(S.d_caps, S.h_capsup, sizeof(CapDesc) * ncaps, hipMemcpyHostToDevice, S.stream));
    HIPCHK40(hipMemsetAsync(S.d_nfr, 0, 4 * ncaps, S.stream));
    { const int rc = sora_internal_scan_ht40(reinterpret_cast<const uint32_t*>(d_iq0), reinterpret_cast<const uint32_t*>(d_iq1), S.d_caps, (uint32_t)ncaps, mf,
            S.d_scanrows, S.d_nfr, S.d_found,
                                             rx->T, rx->sincos, rx->atan, S.stream); if (rc) return rc; }
    const size_t stride = 2 * (size_t)rx->max_frames;
    hipLaunchKernelGGL(k_ht40_plan, dim3(1), dim3(1024), 0, S.stream, (const CapDesc*)S.d_caps, (uint32_t)ncaps, mf, (const uint32_t*)S.d_nfr, (const Ht40Found*)S.d_found,
                       rx->max_frames, (uint64_t)rx->max_soft, kVoutStride, S.d_frames, S.d_jobs, (uint32_t)stride, S.d_njobs, S.d_fjobs, S.d_evtmpl, S.d_plan, S.d_evbase, S.d_evn);
    HIPCHK40(hipMemcpyAsync(S.h_nfr, S.d_nfr, 4 * ncaps, hipMemcpyDeviceToHost, S.stream));
    HIPCHK40(hipMemcpyAsync(S.h_found, S.d_found, sizeof(Ht40Found) * nrows, hipMemcpyDeviceToHost, S.stream));
    HIPCHK40(hipMemcpyAsync(S.h_plan, S.d_plan, 16, hipMemcpyDeviceToHost, S.stream));
    // the kernels are launched for the most frames there can be and stop at the planned count
    const uint32_t bf = S.bound_frames, bj = 2 * bf;
    Ht40Args A;
    A.iq0 = reinterpret_cast<const uint32_t*>(d_iq0); A.iq1 = reinterpret_cast<const uint32_t*>(d_iq1); A.frames = S.d_frames; A.nframes = bf;
    A.T = rx->T; A.sincos = rx->sincos; A.atan = rx->atan; A.soft = S.d_soft; A.w_out = nullptr; A.plan = S.d_plan;
    hipLaunchKernelGGL(k_ht40_frame, dim3((bf + 3) / 4), dim3(256), 0, S.stream, A);
    if (rx->lanes16)
        hipLaunchKernelGGL(k_viterbi16_11n, dim3((bj + 7) / 8 + 2), dim3(64), 0, S.stream, (const VitJob*)S.d_jobs, (const uint32_t*)S.d_njobs, 0u,
                (uint32_t)stride, (const uint8_t*)S.d_soft, S.d_vout);
    else
        hipLaunchKernelGGL(k_viterbi11n, dim3((bj / 2 + 3 + 3) / 4), dim3(256), 0, S.stream, (const VitJob*)S.d_jobs, (const uint32_t*)S.d_njobs, 0u,
                (uint32_t)stride, (const uint8_t*)S.d_soft, S.d_vout);
    Ht40FinishArgs Fi; Fi.jobs = S.d_fjobs; Fi.njobs = bj; Fi.vout = S.d_vout; Fi.mpdu = S.d_mpdu; Fi.rows = S.d_rows; Fi.T = rx->T; Fi.plan = S.d_plan;
    hipLaunchKernelGGL(k_ht40_finish, dim3((bj + 3) / 4), dim3(256), 0, S.stream, Fi);
    HIPCHK40(hipGetLastError());
    return SORA_OK;
}

// a raw-capture call whose stream has been waited for: the front end's records (in page-locked memory by now) -> the call's events
static int ht40_collect_events(Ht40Slot& S)
{
    if (!S.events_pending)
        return S.plan_error ? sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_process_captures_dev: the captures hold more frames / soft values than the handle was created for "
                                                                    "(max_frames, max_soft_values)", 0) : SORA_OK;
    S.events_pending = false;
    S.plan_error = S.h_plan[3] != 0;
    S.events.clear();
    const uint32_t mf = S.capture_mf;
    uint32_t nf = 0;
    for (size_t c = 0; c < S.h_caps.size(); c++) {
        const uint32_t n = std::min(S.h_nfr[c], mf);
        for (uint32_t i = 0; i < n; i++) {
            const Ht40Found& F = S.h_found[c * mf + i];
            Ht40Event E; E.capture_id = S.h_caps[c].capture_id; E.end_sample = F.end_sample; E.error_code = F.error_code; E.mcs = F.mcs; E.length = F.ht_len;
                E.nsym = F.nsym; E.frame = -1;
            E.truncated = (i + 1 == mf && S.h_nfr[c] > mf);
            if (F.error_code == 0) E.frame = (int)nf++;                            // (the order k_ht40_plan numbers the frames in)
            S.events.push_back(E);
        }
    }
    S.nframes = S.h_plan[0];
    if (S.plan_error || nf != S.h_plan[1]) { S.plan_error = true; S.nframes = 0; return sora_internal_fail(SORA_ERR_CAPACITY,
            "sora_ht40_process_captures_dev: the captures hold more frames / soft values than the handle was created for (max_frames, max_soft_values)", 0); }
    return SORA_OK;
}

static int ht40_slot_results(sora_ht40_t* rx, Ht40Slot& S, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (S.capture_mode) {                                                        // rows per event of the front end, in (capture, time) order
        HIPCHK40(hipSetDevice(rx->device));
        HIPCHK40(hipStreamSynchronize(S.stream));
        { const int rc = ht40_collect_events(S); if (rc) return rc; }
        const size_t nj = 2 * (size_t)S.nframes;
        std::vector<Rx11bRow> rows(nj);
        if (nj) HIPCHK40(hipMemcpy(rows.data(), S.d_rows, sizeof(Rx11bRow) * nj, hipMemcpyDeviceToHost));
        std::vector<uint8_t> bulk;
        if (h_mpdu && nj) { bulk.resize(nj * 4096); HIPCHK40(hipMemcpy(bulk.data(), S.d_mpdu, bulk.size(), hipMemcpyDeviceToHost)); }
        size_t n = 0, moff = 0;
        for (const Ht40Event& E : S.events) {
            for (int k = 0; k < (E.frame >= 0 ? 2 : 1); k++) {
                if (n >= max_out) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_results: output buffer too small", 0);
                sora_frame_result& o = out[n++];
                memset(&o, 0, sizeof(o));
                o.capture_id = E.capture_id; o.end_sample = E.end_sample; o.error_code = E.error_code; o.flags = E.truncated ? SORA_ROW_TRUNCATED : 0;
                    o.mpdu_offset = (uint32_t)moff;
                if (E.frame >= 0) {
                    const Rx11bRow& r = rows[2 * (size_t)E.frame + k];
                    o.start_sample = (uint32_t)k; o.rate_kbps = E.mcs; o.nsym = (uint16_t)E.nsym; o.error_code = r.error_code; o.length = (uint16_t)r.length; o.crc32 = r.crc32;
                    if (h_mpdu) {
                        if (moff + r.length > mpdu_cap) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_results: MPDU buffer too small", 0);
                        memcpy(h_mpdu + moff, bulk.data() + (2 * (size_t)E.frame + k) * 4096, r.length); moff += r.length;
                    }
                }
            }
        }
        *nout = n;
        return SORA_OK;
    }
    if (S.nframes == 0) return SORA_OK;
    HIPCHK40(hipSetDevice(rx->device));
    HIPCHK40(hipStreamSynchronize(S.stream));
    const size_t nj = 2 * (size_t)S.nframes;
    if (nj > max_out) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_results: two rows per frame are reported", 0);
    std::vector<Rx11bRow> rows(nj);
    HIPCHK40(hipMemcpy(rows.data(), S.d_rows, sizeof(Rx11bRow) * nj, hipMemcpyDeviceToHost));
    std::vector<uint8_t> bulk;
    if (h_mpdu) { bulk.resize(nj * 4096); HIPCHK40(hipMemcpy(bulk.data(), S.d_mpdu, bulk.size(), hipMemcpyDeviceToHost)); }
    size_t moff = 0;
    for (size_t j = 0; j < nj; j++) {
        sora_frame_result& o = out[j];
        memset(&o, 0, sizeof(o));
        o.capture_id = S.h_frames[j / 2].frame_id; o.start_sample = (uint32_t)(j & 1);                  // start_sample carries the spatial stream
        o.error_code = rows[j].error_code; o.length = (uint16_t)rows[j].length; o.crc32 = rows[j].crc32; o.rate_kbps = S.h_frames[j / 2].n_bpsc * 10 + S.h_frames[j / 2].code_rate;
        o.nsym = (uint16_t)sora_ht40_symbols(S.h_frames[j / 2].length[0], S.h_frames[j / 2].length[1], S.h_frames[j / 2].n_bpsc, S.h_frames[j / 2].code_rate);
        o.mpdu_offset = (uint32_t)moff;
        if (h_mpdu) {
            if (moff + rows[j].length > mpdu_cap) return sora_internal_fail(SORA_ERR_CAPACITY, "sora_ht40_results: MPDU buffer too small", 0);
            memcpy(h_mpdu + moff, bulk.data() + j * 4096, rows[j].length); moff += rows[j].length;
        }
    }
    *nout = nj;
    return SORA_OK;
}

int sora_ht40_results(sora_ht40_t* rx, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (!rx || !nout || !out) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_results: null argument", 0);
    *nout = 0;
    if (!rx->have_results) return sora_internal_fail(SORA_ERR_FAILED, "no process call to report", 0);
    return ht40_slot_results(rx, rx->slot[rx->last], out, max_out, nout, h_mpdu, mpdu_cap);
}

// Tickets (as sora_rx_ticket / _wait / _results_of): every process call is addressable until kHt40Slots further calls have reused its slot.
static Ht40Slot* ht40_slot_of(sora_ht40_t* rx, int ticket)
{
    if (!rx || ticket <= 0) return nullptr;
    for (Ht40Slot& S : rx->slot) if (S.ticket == ticket) return &S;
    return nullptr;
}
static const char* const kStaleHt40 = "stale ticket: its slot has been reused by a later process call (or the ticket was never issued)";
int sora_ht40_ticket(sora_ht40_t* rx) { return rx && rx->have_results ? rx->slot[rx->last].ticket : 0; }
int sora_ht40_calls_in_flight(sora_ht40_t* rx) { (void)rx; return kHt40Slots; }
int sora_ht40_wait(sora_ht40_t* rx, int ticket)
{
    Ht40Slot* S = ht40_slot_of(rx, ticket);
    if (!S) return sora_internal_fail(SORA_ERR_INVALID_PARAM, kStaleHt40, 0);
    HIPCHK40(hipSetDevice(rx->device));
    HIPCHK40(hipStreamSynchronize(S->stream));
    if (S->delivered) S->released = true;
    return S->capture_mode ? ht40_collect_events(*S) : SORA_OK;                  // (a raw-capture call that outgrew the handle's capacity says so here)
}
int sora_ht40_wait_any(sora_ht40_t* rx, int* ticket)
{
    if (!rx || !ticket) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_wait_any: null argument", 0);
    *ticket = 0;
    HIPCHK40(hipSetDevice(rx->device));
    for (unsigned spin = 0;; spin++) {
        bool pending; hipError_t err;
        Ht40Slot* S = slots_poll(rx->slot, kHt40Slots, &pending, &err);
        if (err != hipSuccess) return sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "sora_ht40_wait_any: hipEventQuery", (int)err);
        // (the ticket is reported even when its call outgrew the handle's capacity)
        if (S) { const int t = S->ticket; const int rc = sora_ht40_wait(rx, t); *ticket = t; return rc; }
        if (!pending) return sora_internal_fail(SORA_ERR_FAILED, "sora_ht40_wait_any: no call with an enqueued delivery (sora_ht40_deliver_async) is in flight", 0);
        if (spin > 64) std::this_thread::yield();
    }
}
void* sora_ht40_stream_of(sora_ht40_t* rx, int ticket) { Ht40Slot* S = ht40_slot_of(rx, ticket); return S ? (void*)S->stream : nullptr; }
int sora_ht40_deliver_async(sora_ht40_t* rx, int ticket, sora_frame_result* h_rows, size_t max_rows, uint32_t* h_counts, uint8_t* h_mpdu, size_t mpdu_cap)
{
    Ht40Slot* S = ht40_slot_of(rx, ticket);
    if (!S) return sora_internal_fail(SORA_ERR_INVALID_PARAM, kStaleHt40, 0);
    HIPCHK40(hipSetDevice(rx->device));
    // raw captures: the event table (rows per event, template rows, source rows) and the event count were written by
    if (S->capture_mode) {
        // k_ht40_plan; the host knows only the bound ncaps x max_frames_per_capture.  The table is the one sora_ht40_results_of reports: a header that failed is a row, the
        // last row a full capture could hold carries SORA_ROW_TRUNCATED (round 4; before, only decoded frames were delivered).
        const int rc = sora_internal_dense_deliver(&S->dense, S->d_rows, S->d_evn, nullptr, nullptr, S->bound_events, 2, S->d_mpdu, S->stream,
                                                   h_rows, max_rows, h_counts, h_mpdu, mpdu_cap, S->d_evtmpl, S->d_plan + 4, S->d_evbase);
        if (rc != SORA_OK) return rc;
        HIPCHK40(slots_mark_delivered(*S));
        return SORA_OK;
    }
    S->h_tmpl.resize(2 * (size_t)S->nframes);
    for (size_t j = 0; j < S->h_tmpl.size(); j++) {                             // (the same fields sora_ht40_results fills in on the host)
        sora_frame_result& o = S->h_tmpl[j]; const sora_ht40_frame& f = S->h_frames[j / 2];
        memset(&o, 0, sizeof(o));
        o.capture_id = f.frame_id; o.start_sample = (uint32_t)(j & 1); o.rate_kbps = f.n_bpsc * 10 + f.code_rate;
        o.nsym = (uint16_t)sora_ht40_symbols(f.length[0], f.length[1], f.n_bpsc, f.code_rate);
    }
    // two rows per frame, always: "captures" = frames, max_frames_per_capture = 2, no per-capture counts.  (The template is read by an
    // asynchronous copy: it lives in the slot until the slot's next call.)
    const int rc = sora_internal_dense_deliver(&S->dense, S->d_rows, nullptr, nullptr, S->h_tmpl.data(), S->nframes, 2, S->d_mpdu, S->stream,
                                               h_rows, max_rows, h_counts, h_mpdu, mpdu_cap);
    if (rc != SORA_OK) return rc;
    HIPCHK40(slots_mark_delivered(*S));
    return SORA_OK;
}

int sora_ht40_results_of(sora_ht40_t* rx, int ticket, sora_frame_result* out, size_t max_out, size_t* nout, uint8_t* h_mpdu, size_t mpdu_cap)
{
    if (!nout || !out) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_ht40_results_of: null argument", 0);
    *nout = 0;
    Ht40Slot* S = ht40_slot_of(rx, ticket);
    if (!S) return sora_internal_fail(SORA_ERR_INVALID_PARAM, kStaleHt40, 0);
    return ht40_slot_results(rx, *S, out, max_out, nout, h_mpdu, mpdu_cap);
}
