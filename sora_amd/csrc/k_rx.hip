// k_rx.hip -- batched per-symbol / per-frame kernels of the 802.11a receive path (gfx950).
//
//   k_sym_front   T11aDataSymbol -> TFreqCompensation -> TFFT64 -> TChannelEqualization   (parallel over symbols)
//   k_track       TPhaseCompensate/TPilotTrack loop-carried state on the 4 pilots          (serial per frame)
//   k_demap       TPhaseCompensate + TPilotTrack rotation + T11aDemap<N> + T11aDeinterleave (parallel over symbols)
//   k_viterbi<CR> T11aViterbi<5000*8,48,256,24>: 64-state ACS, wave64 = 64 states          (one wave per frame)
//   k_traceback   TViterbiCore::Traceback for every window of the schedule                 (one thread per window)
//   k_finish      T11aDesc + TBB11aFrameSink (descramble, CRC-32, FRAME_OK / CRC32_FAIL)   (one thread per frame)
// plus the stand-alone stage kernels behind the per-stage C entry points.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace sora {

__device__ __constant__ uint8_t kPilotSgn[128] = {        // pilot.hpp:10-28: 1 <=> polarity -1
    0,0,0,1,1,1,0,1, 1,1,1,0,0,1,0,1, 1,0,0,1,0,0,1,0, 0,0,0,0,0,1,0,0,
    0,1,0,0,1,1,0,0, 0,1,0,1,1,1,0,1, 0,1,1,0,1,1,0,0, 0,0,0,1,1,0,0,1,
    1,0,1,0,1,0,0,1, 1,1,0,0,1,1,1,1, 0,1,1,0,1,0,0,0, 0,1,0,1,0,1,0,1,
    1,1,1,1,0,1,0,0, 1,0,1,0,0,0,1,1, 0,1,1,1,0,0,0,1, 1,1,1,1,1,1,0,0 };

__device__ __forceinline__ int carrier_bin48(int k)       // demap order -26..-1,+1..+26 without pilots (demapper11a.hpp:20-37)
{
    if (k < 24) { int b = 38 + k; if (b >= 43) b++; if (b >= 57) b++; return b; }
    int b = 1 + (k - 24); if (b >= 7) b++; if (b >= 21) b++; return b;
}

// ------------------------------------------------------------------------------------------------
// k_sym_front: 16 lanes per OFDM symbol, 16 symbols per 256-thread block.
//   algorithmic bytes per symbol: 256 read (64 samples) + 256 written (64 equalised bins)
__global__ void __launch_bounds__(256) k_sym_front(RxArgs A)
{
    __shared__ uint32_t s_all[16][64];
    const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
    const uint32_t slot = blockIdx.x * 16 + g;
    int fr = -1, sym = 0;
    if (slot < A.total_slots) { fr = A.slot_frame[slot]; sym = A.slot_sym[slot]; }
    const bool active = fr >= 0 && sym > 0;
    cpx x[4], Y[4];
    const FrameCtx* fx = A.fctx + (active ? fr : 0);
    if (active) {
        const FrameRow& r = A.frames[fr];
        const uint32_t* iq = A.iq + A.caps[r.capture].offset;
        const uint32_t p0 = r.data_start + 80u * (uint32_t)sym + 8u;           // skip_cp = 8 (PHY_11a.hpp:365,394)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int n = e + 16 * m;
            cpx s = sra(unpack(iq[(size_t)(p0 + n) * A.str]), 1);             // TFreqCompensation: >>1 (channel_11a.hpp:643)
            x[m] = mul_q15(s, unpack(fx->freq[n]));                            //   x FreqCoeffs (:644)
        }
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) x[m] = mk(0, 0);
    }
    fft64_group(x, Y, s_all[g], e, A.T, []() { __syncthreads(); });           // TFFT64
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; q++) {                                          // TChannelEqualization (channel_11a.hpp:548-574)
            const int bin = e + 16 * q;
            cpx o = mk(0, 0);
            if (!(bin >= 28 && bin < 36)) {
                int re, im; mul32(Y[q], unpack(fx->chan[bin]), re, im);
                o = mk(w16(re >> 8), w16(im >> 8));
            }
            A.eq[(size_t)slot * 64 + bin] = pack(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_track: the loop-carried part of TPhaseCompensate/TPilotTrack (freqoffset.hpp:28-30, pilot.hpp:166-233):
// only the 4 pilot bins take part.  One thread per frame, sequential over its data symbols.
__global__ void __launch_bounds__(64) k_track(RxArgs A)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= A.nrows) return;
    const FrameRow r = A.frames[f];
    VitJob J; J.valid = 0; J.soft_off = 0; J.nsoft = 0; J.length = 0; J.dec_off = 0; J.out_off = 0; J.code_rate = 0; J.pad = 0;
    if (!r.valid || r.error_code != 0) { A.jobs[f] = J; return; }
    J.valid = 1; J.soft_off = r.slot0 * (uint32_t)kSoftPerSlot; J.nsoft = (uint32_t)r.nsym * 48u * r.nbpsc; J.length = r.length;
    J.dec_off = r.slot0 * (uint32_t)kDecPerSlot; J.out_off = r.slot0 * (uint32_t)kOutPerSlot; J.code_rate = r.code_rate;
    A.jobs[f] = J;
    const Tables& T = A.T;
    int cfo_comp = r.cfo_comp, sfo_comp = r.sfo_comp, cfo_tr = r.cfo_tracker, sfo_tr = r.sfo_tracker;
    unsigned symbol_count = 0;                                                 // 127 -> 0 after the SIGNAL symbol
    for (uint32_t s = 1; s <= r.nsym; s++) {
        const uint32_t slot = r.slot0 + s;
        const uint32_t* eq = A.eq + (size_t)slot * 64;
        // CompCoeffs at carriers -21, -7, +7, +21: th = CFO_comp + c*SFO_comp (pilot.hpp:138-164)
        cpx p43 = mul_q15(unpack(eq[43]), rot_coeff(T, w16(cfo_comp - 21 * sfo_comp)));
        cpx p57 = mul_q15(unpack(eq[57]), rot_coeff(T, w16(cfo_comp - 7 * sfo_comp)));
        cpx p7  = mul_q15(unpack(eq[7]),  rot_coeff(T, w16(cfo_comp + 7 * sfo_comp)));
        cpx p21 = mul_q15(unpack(eq[21]), rot_coeff(T, w16(cfo_comp + 21 * sfo_comp)));
        int th1 = uatan2(T, p43.im, p43.re), th2 = uatan2(T, p57.im, p57.re);
        int th3 = uatan2(T, p7.im, p7.re),   th4 = uatan2(T, -p21.im, -p21.re);
        if (kPilotSgn[symbol_count]) { th1 = w16(th1 + 0x8000); th2 = w16(th2 + 0x8000); th3 = w16(th3 + 0x8000); th4 = w16(th4 + 0x8000); }
        symbol_count++; if (symbol_count >= 127) symbol_count = 0;
        const int avg = w16((th1 + th2 + th3 + th4) / 4);
        const int del = w16(((th3 - th1) / 28 + (th4 - th2) / 28) >> 1);
        TrackRec tr; tr.cfo_comp = (int16_t)cfo_comp; tr.sfo_comp = (int16_t)sfo_comp; tr.avg = (int16_t)avg; tr.del = (int16_t)del;
        A.track[slot] = tr;
        cfo_tr = w16(cfo_tr + (avg >> 2)); sfo_tr = w16(sfo_tr + (del >> 2));
        cfo_comp = w16(cfo_comp + avg + cfo_tr); sfo_comp = w16(sfo_comp + del + sfo_tr);
    }
}

// ------------------------------------------------------------------------------------------------
// k_demap: one wave per data symbol (4 symbols per block): 48 data carriers -> phase compensation ->
// pilot rotation -> soft demap -> de-interleave -> contiguous per-frame soft stream.
//   algorithmic bytes per symbol: 256 read + N_CBPS written
__global__ void __launch_bounds__(256) k_demap(RxArgs A)
{
    __shared__ uint8_t s_soft[4][288];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 4 + w;
    int fr = -1, sym = 0;
    if (slot < A.total_slots) { fr = A.slot_frame[slot]; sym = A.slot_sym[slot]; }
    const bool active = fr >= 0 && sym > 0;
    const Tables& T = A.T;
    int nb = 1; uint32_t slot0 = 0;
    if (active) {
        const FrameRow& r = A.frames[fr];
        nb = r.nbpsc; slot0 = r.slot0;
        if (lane < 48) {
            const TrackRec tr = A.track[slot];
            const int bin = carrier_bin48(lane);
            const int c = bin < 32 ? bin : bin - 64;
            cpx v = unpack(A.eq[(size_t)slot * 64 + bin]);
            v = mul_q15(v, rot_coeff(T, w16(tr.cfo_comp + c * tr.sfo_comp)));      // TPhaseCompensate
            v = mul_q15(v, rot_coeff(T, w16(tr.avg + c * tr.del)));                // TPilotTrack::_rotate
            int re = v.re >> 4, im = v.im >> 4;                                     // demap_limit<64> (demapper.h:141-151)
            re = min(max(re, -128), 127); im = min(max(im, -128), 127);
            const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
            uint8_t* o = s_soft[w] + lane * nb;                                     // DemapperCore::Demap<N_BPSC> (demapper.h:16-45)
            if (nb == 1) { o[0] = T.demap[ur]; }
            else if (nb == 2) { o[0] = T.demap[ur]; o[1] = T.demap[ui]; }
            else if (nb == 4) { o[0] = T.demap[ur]; o[1] = T.demap[256 + ur]; o[2] = T.demap[ui]; o[3] = T.demap[256 + ui]; }
            else { o[0] = T.demap[ur]; o[1] = T.demap[512 + ur]; o[2] = T.demap[768 + ur];
                   o[3] = T.demap[ui]; o[4] = T.demap[512 + ui]; o[5] = T.demap[768 + ui]; }
        }
    }
    __syncthreads();
    if (active) {
        const int ncbps = 48 * nb;
        const int di = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3;
        const uint16_t* map = T.deint + di * 288;
        uint8_t* dst = A.soft + (size_t)slot0 * kSoftPerSlot + (size_t)(sym - 1) * ncbps;
        for (int k4 = lane; k4 < ncbps / 4; k4 += 64) {                             // T11aDeinterleave* : out[k] = in[j(k)]
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= (uint32_t)s_soft[w][map[4 * k4 + b]] << (8 * b);
            reinterpret_cast<uint32_t*>(dst)[k4] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_viterbi<CR>: forward add-compare-select of the K=7 (133,171) code exactly as TViterbiCore does it
// (viterbicore.h:293-465): 8-bit WRAPPING path metrics, decision kept in the metric LSB (&0xFE / |1),
// unsigned minimum, normalisation whenever (trellis_index & 7) == 0 after a puncture group, and the
// window schedule of T11aViterbi<..,256,24>::Process (viterbi.hpp:189-214).  lane n = state n; the new
// state n is reached from n>>1 (decision 0) or 32+(n>>1) (decision 1) -> two ds_bpermute per step.
// Branch metric of soft value v for expected bit c is c ? 2*(7-v) : 2*v = (2v) ^ (c ? 14 : 0)  (VIT_MA/VIT_MB,
// viterbilut.h:50-185); expected bits: parity((br<<6 | n) & 0155) for A, & 0117 for B.
// The 64 decisions of a column are one wave ballot; 64 columns are gathered lane-wise and stored as
// one coalesced 512-byte write.  At every scheduled trace-back the arg-min state (tie-break:
// metric<<8 | state<<2, viterbicore.h:479-524) is recorded for k_traceback.
struct VitCore {
    unsigned m;             // path metric of state `lane` (low 8 bits)
    int idx0, idx1;         // ds_bpermute byte addresses of the two predecessors
    unsigned mA0, mB0, mA1, mB1;
};

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}

template <int WHICH>    // 0: (A,B)  1: A only  2: B only
__device__ __forceinline__ uint64_t acs_step(VitCore& V, unsigned a2, unsigned b2)
{
    const unsigned m0 = (unsigned)__builtin_amdgcn_ds_bpermute(V.idx0, (int)V.m);
    const unsigned m1 = (unsigned)__builtin_amdgcn_ds_bpermute(V.idx1, (int)V.m);
    unsigned bm0, bm1;
    if (WHICH == 0)      { bm0 = (a2 ^ V.mA0) + (b2 ^ V.mB0); bm1 = (a2 ^ V.mA1) + (b2 ^ V.mB1); }
    else if (WHICH == 1) { bm0 = a2 ^ V.mA0; bm1 = a2 ^ V.mA1; }
    else                 { bm0 = b2 ^ V.mB0; bm1 = b2 ^ V.mB1; }
    const unsigned c0 = (m0 + bm0) & 0xFEu;
    const unsigned c1 = ((m1 + bm1) & 0xFFu) | 1u;
    const bool d = c1 < c0;
    V.m = d ? c1 : c0;
    return __ballot(d);
}

template <int CR>
__device__ __forceinline__ void viterbi_forward(const VitJob& J, const uint8_t* soft_base, uint64_t* dec_base, uint32_t* tbk, uint32_t* nwin_out)
{
    const int lane = threadIdx.x;
    const uint32_t* soft = reinterpret_cast<const uint32_t*>(soft_base + J.soft_off);
    uint64_t* dec = dec_base + J.dec_off;
    const uint32_t nsoft = J.nsoft;
    const uint32_t tr_end = J.length * 8u + 16u + 6u;

    VitCore V;
    V.m = lane == 0 ? 0u : 0x30u;                                              // ALL_INIT0 / ALL_INIT (viterbilut.h:22-30)
    V.idx0 = (lane >> 1) * 4; V.idx1 = (32 + (lane >> 1)) * 4;
    V.mA0 = (__popc(lane & 0155) & 1) ? 14u : 0u;          V.mB0 = (__popc(lane & 0117) & 1) ? 14u : 0u;
    V.mA1 = (__popc((64 | lane) & 0155) & 1) ? 14u : 0u;   V.mB1 = (__popc((64 | lane) & 0117) & 1) ? 14u : 0u;

    constexpr int GB = CR == 0 ? 2 : CR == 2 ? 4 : 3;                           // soft bytes per puncture group (CR: 0=1/2, 1=2/3, 2=3/4)
    constexpr int GS = CR == 0 ? 1 : CR == 2 ? 3 : 2;                           // trellis steps per group
    constexpr int BLK = CR == 1 ? 192 : 256;                                    // soft bytes fetched per refill
    constexpr int GPB = BLK / GB;                                               // groups per refill

    uint32_t tr = 0, ob = 0, nw = 0;
    uint64_t mydec = 0;                                                         // decisions of column (64*j + lane)
    if (lane == 0) dec[0] = 0;
    bool done = false;
    const uint32_t ngroups = nsoft / GB;
    for (uint32_t g0 = 0; g0 < ngroups && !done; g0 += GPB) {
        const uint32_t widx = (g0 * GB) / 4 + (uint32_t)lane;
        uint32_t wv = 0;
        if (widx * 4 < nsoft && lane < BLK / 4) wv = soft[widx];
        const uint32_t gend = min((uint32_t)GPB, ngroups - g0);
        for (uint32_t g = 0; g < gend; g++) {
            unsigned a2, b2, c2 = 0, d2 = 0;
            if (CR == 0) {
                uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)wv, (int)(g >> 1));
                w >>= (g & 1) * 16;
                a2 = (w & 0xFF) * 2; b2 = ((w >> 8) & 0xFF) * 2;
            } else if (CR == 2) {
                uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)wv, (int)g);
                a2 = (w & 0xFF) * 2; b2 = ((w >> 8) & 0xFF) * 2; c2 = ((w >> 16) & 0xFF) * 2; d2 = (w >> 24) * 2;
            } else {
                const uint32_t bo = g * 3;
                uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)wv, (int)(bo >> 2));
                uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)wv, (int)min((bo >> 2) + 1, 63u));
                uint64_t ww = (((uint64_t)w1 << 32) | w0) >> ((bo & 3) * 8);
                a2 = (unsigned)(ww & 0xFF) * 2; b2 = (unsigned)((ww >> 8) & 0xFF) * 2; c2 = (unsigned)((ww >> 16) & 0xFF) * 2;
            }
            // ---- the puncture group (viterbi.hpp:167-187)
#pragma unroll
            for (int s = 0; s < GS; s++) {
                uint64_t bal;
                if (s == 0) bal = acs_step<0>(V, a2, b2);
                else if (s == 1) bal = acs_step<1>(V, c2, 0);
                else bal = acs_step<2>(V, 0, d2);
                tr++;
                if ((tr & 63) == (uint32_t)lane) mydec = bal;
                if ((tr & 63) == 63) dec[tr - 63 + lane] = mydec;               // columns tr-63 .. tr (lane j holds column with (col&63)==j)
            }
            if ((tr & 7) == 0) {                                                // Normalize (viterbicore.h:444-465)
                const unsigned mn = wave_min_u32(V.m) & 0xFEu;
                V.m = (V.m - mn) & 0xFFu;
            }
            // ---- trace-back schedule (viterbi.hpp:196-214)
            uint32_t cnt = 0, look = 0;
            if (tr >= tr_end) { cnt = tr_end - ob - 6; look = tr - tr_end; }
            else if (tr >= ob + 256 + 24 + 6) { look = 24 + (tr - (ob + 256 + 24 + 6)) % 8; cnt = 256; }
            if (cnt) {
                const unsigned kmin = wave_min_u32((V.m << 8) | ((unsigned)lane << 2));
                if (lane == 0) {
                    tbk[nw * 3 + 0] = tr;
                    tbk[nw * 3 + 1] = look | (cnt << 16);
                    tbk[nw * 3 + 2] = ((kmin >> 2) & 0x3F) | (((kmin >> 8) & 1) << 6) | (ob << 8);
                }
                nw++; ob += cnt;
                if (tr >= tr_end) { done = true; break; }
            }
        }
    }
    // flush the partially gathered decision columns
    {
        const uint32_t basecol = tr & ~63u;
        if ((tr & 63) != 63 && (uint32_t)lane <= (tr & 63)) dec[basecol + lane] = mydec;
    }
    if (lane == 0) *nwin_out = nw;
}

__global__ void __launch_bounds__(64) k_viterbi(const VitJob* jobs, uint32_t njobs, const uint8_t* soft, uint64_t* dec, uint32_t* tbk, uint32_t* nwin)
{
    const uint32_t f = blockIdx.x;
    if (f >= njobs) return;
    const VitJob J = jobs[f];
    if (!J.valid) return;
    uint32_t* t = tbk + (size_t)f * kMaxWindows * 3;
    if (J.code_rate == 0)      viterbi_forward<0>(J, soft, dec, t, nwin + f);
    else if (J.code_rate == 1) viterbi_forward<1>(J, soft, dec, t, nwin + f);
    else                       viterbi_forward<2>(J, soft, dec, t, nwin + f);
}

// ------------------------------------------------------------------------------------------------
// k_traceback: TViterbiCore::Traceback (viterbicore.h:468-555) -- every window of a frame is an
// independent walk through the stored decisions, one thread each.  Emits the bytes LSB-first in time.
__global__ void __launch_bounds__(128) k_traceback(const VitJob* jobs, uint32_t njobs, const uint64_t* dec_base, const uint32_t* tbk, const uint32_t* nwin, uint8_t* out_base)
{
    const uint32_t f = blockIdx.x;
    if (f >= njobs) return;
    const VitJob J = jobs[f];
    if (!J.valid) return;
    const uint32_t nw = nwin[f];
    const uint64_t* dec = dec_base + J.dec_off;
    uint8_t* out = out_base + J.out_off;
    for (uint32_t w = threadIdx.x; w < nw; w += blockDim.x) {
        const uint32_t* t = tbk + ((size_t)f * kMaxWindows + w) * 3;
        uint32_t col = t[0]; const uint32_t look = t[1] & 0xFFFF, cnt = t[1] >> 16;
        int pos = (int)(t[2] & 0x7F); const uint32_t ob = t[2] >> 8;
        for (uint32_t i = 0; i < look; i++) {
            col--; pos = (pos >> 1) & 0x3F;
            pos |= (int)((dec[col] >> pos) & 1) << 6;
        }
        uint8_t* po = out + (ob >> 3) + (cnt >> 3);
        for (uint32_t i = 0; i < (cnt >> 3); i++) {
            unsigned oc = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                oc = (oc << 1) | ((unsigned)(pos >> 6) & 1);
                col--; pos = (pos >> 1) & 0x3F;
                pos |= (int)((dec[col] >> pos) & 1) << 6;
            }
            *--po = (uint8_t)oc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_finish: T11aDesc (scramble.hpp:267-353) + TBB11aFrameSink (PHY_11a.hpp:607-702).
__global__ void __launch_bounds__(64) k_finish(RxArgs A)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= A.nrows) return;
    FrameRow& r = A.frames[f];
    if (!r.valid || r.error_code != 0) return;
    const Tables& T = A.T;
    const uint8_t* dec = A.vout + (size_t)r.slot0 * kOutPerSlot;
    uint8_t* mp = A.mpdu + (size_t)r.slot0 * kOutPerSlot;
    const uint32_t L = r.length;
    unsigned reg = dec[1] >> 1;                                                  // byte 0 dropped, byte 1 seeds the register
    uint32_t crc = 0xFFFFFFFFu, fcs = 0;
    for (uint32_t i = 0; i < L; i++) {
        reg = T.scr[reg & 0x7F];
        const unsigned o = dec[2 + i] ^ reg;
        reg >>= 1;
        mp[i] = (uint8_t)o;
        if (i + 4 < L) crc = (crc >> 8) ^ T.crc[(o ^ crc) & 0xFF];
        else fcs |= o << (8 * (i + 4 - L));
    }
    r.crc32 = fcs;
    r.error_code = ((~crc) == fcs) ? E_FRAME_OK : E_CRC32_FAIL;
}

// ================================================================================================
// stand-alone stage kernels (per-stage C entry points)
__global__ void __launch_bounds__(256) k_fft64_batch(const uint32_t* in, uint32_t* out, uint32_t n, Tables T)
{
    __shared__ uint32_t s_all[16][64];
    const int g = threadIdx.x >> 4, e = threadIdx.x & 15;
    const uint32_t i = blockIdx.x * 16 + g;
    cpx x[4], Y[4];
#pragma unroll
    for (int m = 0; m < 4; m++) x[m] = i < n ? unpack(in[(size_t)i * 64 + e + 16 * m]) : mk(0, 0);
    fft64_group(x, Y, s_all[g], e, T, []() { __syncthreads(); });
    if (i < n) {
#pragma unroll
        for (int q = 0; q < 4; q++) out[(size_t)i * 64 + e + 16 * q] = pack(Y[q]);
    }
}

__global__ void __launch_bounds__(64) k_demap_batch(const uint32_t* in, uint8_t* soft, int nb, uint32_t n, Tables T)
{
    const uint32_t i = blockIdx.x; const int lane = threadIdx.x;
    if (i >= n || lane >= 48) return;
    const int bin = carrier_bin48(lane);
    cpx v = unpack(in[(size_t)i * 64 + bin]);
    int re = v.re >> 4, im = v.im >> 4;
    re = min(max(re, -128), 127); im = min(max(im, -128), 127);
    const unsigned ur = (unsigned)re & 0xFF, ui = (unsigned)im & 0xFF;
    uint8_t* o = soft + (size_t)i * 48 * nb + lane * nb;
    if (nb == 1) { o[0] = T.demap[ur]; }
    else if (nb == 2) { o[0] = T.demap[ur]; o[1] = T.demap[ui]; }
    else if (nb == 4) { o[0] = T.demap[ur]; o[1] = T.demap[256 + ur]; o[2] = T.demap[ui]; o[3] = T.demap[256 + ui]; }
    else { o[0] = T.demap[ur]; o[1] = T.demap[512 + ur]; o[2] = T.demap[768 + ur]; o[3] = T.demap[ui]; o[4] = T.demap[512 + ui]; o[5] = T.demap[768 + ui]; }
}

__global__ void __launch_bounds__(64) k_deint_batch(const uint8_t* in, uint8_t* out, int nb, uint32_t n, Tables T)
{
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const int ncbps = 48 * nb, di = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : 3;
    for (int k = threadIdx.x; k < ncbps; k += 64) out[(size_t)i * ncbps + k] = in[(size_t)i * ncbps + T.deint[di * 288 + k]];
}

__global__ void __launch_bounds__(64) k_make_vitjobs(VitJob* jobs, const uint32_t* soft_off, const uint32_t* nsoft, const uint16_t* flen,
                                                      const uint32_t* out_off, const uint32_t* dec_off, int code_rate, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    VitJob J; J.soft_off = soft_off[i]; J.nsoft = nsoft[i]; J.length = flen[i]; J.dec_off = dec_off[i]; J.out_off = out_off[i];
    J.valid = 1; J.code_rate = (uint32_t)code_rate; J.pad = 0;
    jobs[i] = J;
}

}  // namespace sora
