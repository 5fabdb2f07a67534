// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (not shipped, not linked into libsora_hip.so).
//
// Thin extern "C" exports around the REFERENCE's own arithmetic headers, compiled from the sources
// where they lie under /root/reference by oracle/build_ref.sh (which patches MSVC-only constructs in
// a scratch directory that is deleted after the compile).  The resulting oracle/_ref/libsora_ref.so
// is used by tests/ to pin the plain-C restatement in oracle/sora_oracle.c stage by stage, and by
// bench.py's cpu_baseline leg ("kind": "reference") for the SSE kernels.
//
// Reference headers pulled in (kernel/...):
//   core/inc/vector128.h, complex.h, fft_r4dif.h, ifft_r4dif.h, fft_lut_*.h, intalg.h(+lut), CRC32.h
//   bb/Brick11/src/viterbicore.h, viterbilut.h, demapper.h
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "const.h"          // generated stub (build_ref.sh)
#include "vector128.h"
#include "operator_repeater.h"
#include "fft_r4dif.h"
#include "ifft_r4dif.h"
#include "intalg.h"
#include "CRC32.h"
#include "viterbicore.h"
#include "demapper.h"
#include "44MTo40M.hpp"

#define EXPORT extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------- FFT / IFFT (fft_r4dif.h:133-141)
EXPORT void ref_fft64(const int16_t* in, int16_t* out)   { A16 vcs t[16], o[16]; memcpy(t, in, 256); FFT<64>(t, o);   memcpy(out, o, 256); }
EXPORT void ref_ifft64(const int16_t* in, int16_t* out)  { A16 vcs t[16], o[16]; memcpy(t, in, 256); IFFT<64>(t, o);  memcpy(out, o, 256); }
EXPORT void ref_fft128(const int16_t* in, int16_t* out)  { A16 vcs t[32], o[32]; memcpy(t, in, 512); FFT<128>(t, o);  memcpy(out, o, 512); }
EXPORT void ref_ifft128(const int16_t* in, int16_t* out) { A16 vcs t[32], o[32]; memcpy(t, in, 512); IFFT<128>(t, o); memcpy(out, o, 512); }

// ---------------------------------------------------------------- vector128.h primitives on one vcs (4 COMPLEX16)
EXPORT void ref_vcs_mul(const int16_t* a, const int16_t* b, int16_t* r)              // vcs mul(a,b): >>15, wrapping pack (:1201-1211)
{ A16 vcs x, y, z; memcpy(&x, a, 16); memcpy(&y, b, 16); z = mul(x, y); memcpy(r, &z, 16); }
EXPORT void ref_vcs_mul32(const int16_t* a, const int16_t* b, int32_t* re, int32_t* im) // mul(vi&,vi&,a,b) (:1075-1081)
{ A16 vcs x, y; A16 vi r, i; memcpy(&x, a, 16); memcpy(&y, b, 16); mul(r, i, x, y); memcpy(re, &r, 16); memcpy(im, &i, 16); }
EXPORT void ref_vcs_conj_mul32(const int16_t* a, const int16_t* b, int32_t* re, int32_t* im) // conj_mul (:1038-1044)
{ A16 vcs x, y; A16 vi r, i; memcpy(&x, a, 16); memcpy(&y, b, 16); conj_mul(r, i, x, y); memcpy(re, &r, 16); memcpy(im, &i, 16); }
EXPORT void ref_vcs_mul_shift(const int16_t* a, const int16_t* b, int n, int16_t* r)  // mul_shift (:1235-1246)
{ A16 vcs x, y, z; memcpy(&x, a, 16); memcpy(&y, b, 16); z = mul_shift(x, y, n); memcpy(r, &z, 16); }
EXPORT void ref_vcs_conj_mul_shift(const int16_t* a, const int16_t* b, int n, int16_t* r)
{ A16 vcs x, y, z; memcpy(&x, a, 16); memcpy(&y, b, 16); z = conj_mul_shift(x, y, n); memcpy(r, &z, 16); }
EXPORT void ref_vcs_sqnorm(const int16_t* a, int32_t* r)
{ A16 vcs x; A16 vi z; memcpy(&x, a, 16); z = SquaredNorm(x); memcpy(r, &z, 16); }
EXPORT void ref_vcs_pack(const int32_t* re, const int32_t* im, int16_t* r)
{ A16 vi a, b; A16 vcs z; memcpy(&a, re, 16); memcpy(&b, im, 16); pack(z, a, b); memcpy(r, &z, 16); }

// ---------------------------------------------------------------- three one-line bricks on one symbol (64 COMPLEX16), the reference's own primitives in the bricks' own order
// TFreqCompensation::Process (channel_11a.hpp:642-644): rep_shift_right<16>(pi, pi, 1); FrequencyShift<16>(po, pi, FreqCoeffs) = rep_mul<16> (dspalg.hpp:219-224)
EXPORT void ref_freq_comp64(const int16_t* in, const int16_t* coeffs, int16_t* out)
{ A16 vcs x[16], c[16], o[16]; memcpy(x, in, 256); memcpy(c, coeffs, 256); rep_shift_right<16>(x, x, 1); rep_mul<16>(o, x, c); memcpy(out, o, 256); }
// TChannelEqualization::_channel_equalize (channel_11a.hpp:548-574): mul -> >> norm_shift (8) -> pack, vcs 7 and 8 (bins 28..35) zero
EXPORT void ref_channel_equalize64(const int16_t* in, const int16_t* coeffs, int16_t* out)
{
    A16 vcs x[16], c[16], o[16]; memcpy(x, in, 256); memcpy(c, coeffs, 256);
    vi re, im;
    set_zero(o[7]); set_zero(o[8]);
    for (int i = 0; i < 7; i++)  { mul(re, im, x[i], c[i]); re = shift_right(re, 8); im = shift_right(im, 8); pack(o[i], re, im); }
    for (int i = 9; i < 16; i++) { mul(re, im, x[i], c[i]); re = shift_right(re, 8); im = shift_right(im, 8); pack(o[i], re, im); }
    memcpy(out, o, 256);
}
// TPhaseCompensate::_phase_compensate (freqoffset.hpp:28-30): rep_mul<16>(output, input, CompCoeffs)
EXPORT void ref_phase_comp64(const int16_t* in, const int16_t* coeffs, int16_t* out)
{ A16 vcs x[16], c[16], o[16]; memcpy(x, in, 256); memcpy(c, coeffs, 256); rep_mul<16>(o, x, c); memcpy(out, o, 256); }

// ---------------------------------------------------------------- intalg.h
EXPORT int16_t ref_usin(int16_t r) { return usin(r); }
EXPORT int16_t ref_ucos(int16_t r) { return ucos(r); }
EXPORT int16_t ref_uatan2(int y, int x) { return uatan2(y, x); }
EXPORT const int16_t* ref_usin_lut(void) { return usin_lut; }
EXPORT const int16_t* ref_ucos_lut(void) { return ucos_lut; }
EXPORT const int16_t* ref_uatan2_lut(void) { return &uatan2_lut[0][0]; }

// ---------------------------------------------------------------- CRC32.h
EXPORT uint32_t ref_crc32(const uint8_t* p, uint32_t n)
{ ULONG crc = 0xFFFFFFFF; for (uint32_t i = 0; i < n; i++) CalcCRC32Incremental(p[i], &crc); return ~crc; }

// ---------------------------------------------------------------- demapper.h
EXPORT void ref_demap_limit64(const int16_t* in, int16_t* out)
{ A16 COMPLEX16 a[64], b[64]; memcpy(a, in, 256); demap_limit<64>(a, b); memcpy(out, b, 256); }
// Restates only the carrier walk of demapper11a.hpp:20-37 around the reference's DemapperCore LUTs.
EXPORT void ref_demap11a(const int16_t* limited, int nbpsc, uint8_t* out)
{
    const COMPLEX16* in = (const COMPLEX16*)limited;
    for (int pass = 0; pass < 2; pass++) {
        int lo = pass ? 1 : 64 - 26, hi = pass ? 27 : 64;
        for (int i = lo; i < hi; i++) {
            if (i == 64 - 21 || i == 64 - 7 || i == 7 || i == 21) continue;
            switch (nbpsc) {
            case 1: DemapperCore::DemapBPSK(in[i], out); break;
            case 2: DemapperCore::DemapQPSK(in[i], out); break;
            case 4: DemapperCore::DemapQAM16(in[i], out); break;
            case 6: DemapperCore::DemapQAM64(in[i], out); break;
            }
            out += nbpsc;
        }
    }
}

// ---------------------------------------------------------------- viterbicore.h
EXPORT uint32_t ref_viterbi_sig(const uint8_t* soft48)
{
    static A16 vub trellis[4 * 49];
    uint32_t out = 0;
    Viterbi_sig11(trellis, (const char*)soft48, (char*)&out);
    return out >> 6;                                       // viterbi.hpp:38-39
}

typedef TViterbiCore<5000 * 8> RefCore;                    // same TRELLIS_MAX as fb11ademod_config.hpp:176
EXPORT void* ref_vit_new(void)
{ void* p = NULL; if (posix_memalign(&p, 64, sizeof(RefCore))) return NULL; return new (p) RefCore(); }
EXPORT void ref_vit_free(void* h) { free(h); }
EXPORT void ref_vit_reset(void* h) { ((RefCore*)h)->Reset(); }
EXPORT uint32_t ref_vit_index(void* h) { return ((RefCore*)h)->trellis_index(); }
EXPORT void ref_vit_acs2(void* h, uint32_t a, uint32_t b) { ((RefCore*)h)->BranchACS((const vub*)VIT_MA, a, (const vub*)VIT_MB, b); }
EXPORT void ref_vit_acs1a(void* h, uint32_t a) { ((RefCore*)h)->BranchACS((const vub*)VIT_MA, a); }
EXPORT void ref_vit_acs1b(void* h, uint32_t b) { ((RefCore*)h)->BranchACS((const vub*)VIT_MB, b); }
EXPORT void ref_vit_normalize(void* h) { ((RefCore*)h)->Normalize(); }
EXPORT void ref_vit_traceback(void* h, uint8_t* out, uint32_t bits, uint32_t lookahead) { ((RefCore*)h)->Traceback((char*)out, bits, lookahead); }

// The frame-level schedule of T11aViterbi<5000*8,48,256,24>::Filter::Process (viterbi.hpp:148-235) driven
// over the REFERENCE core: used as the "reference SSE Viterbi" leg of the CPU baseline and as a pin.
// code_rate: 0 = 1/2, 1 = 2/3, 2 = 3/4 (ieee80211const.h:14-20).  Returns number of bytes written.
EXPORT int ref_vit_decode_frame(void* h, const uint8_t* soft, uint32_t nsoft, int code_rate, uint32_t frame_length, uint8_t* out)
{
    RefCore& v = *(RefCore*)h;
    const uint32_t DEPTH = 256, LOOK = 24, PREFIX = 6;
    uint8_t buf[DEPTH / 8 + 1];
    uint32_t ob = 0; int nout = 0;
    v.Reset();
    const uint8_t* p = soft; const uint8_t* end = soft + nsoft;
    while (p < end) {
        if (code_rate == 0)      { ref_vit_acs2(h, p[0], p[1]); p += 2; }
        else if (code_rate == 2) { ref_vit_acs2(h, p[0], p[1]); ref_vit_acs1a(h, p[2]); ref_vit_acs1b(h, p[3]); p += 4; }
        else                     { ref_vit_acs2(h, p[0], p[1]); ref_vit_acs1a(h, p[2]); p += 3; }
        uint32_t tr = v.trellis_index();
        if ((tr & 7) == 0) v.Normalize();
        uint32_t cnt = 0, look = 0;
        const uint32_t tr_end = frame_length * 8 + 16 + PREFIX;
        if (tr >= tr_end) { cnt = tr_end - ob - PREFIX; look = tr - tr_end; }
        else if (tr >= ob + DEPTH + LOOK + PREFIX) { uint32_t remain = (tr - (ob + DEPTH + LOOK + PREFIX)) % 8; cnt = DEPTH; look = LOOK + remain; }
        if (cnt) {
            v.Traceback((char*)buf, cnt, look);
            ob += cnt;
            memcpy(out + nout, buf, cnt >> 3); nout += cnt >> 3;
            if (tr >= tr_end) break;
        }
    }
    return nout;
}

// TDownSample44_40 (sampling.hpp:35-66) over the REFERENCE resampler (44MTo40M.hpp:62-123): feed whole 28-sample
// RX blocks, collect the 28-sample output blocks exactly as the brick forwards them.  Returns samples written.
EXPORT int ref_down44to40(const int16_t* in_iq, uint32_t nblocks, int16_t* out_iq, uint32_t max_out)
{
    Down44to40 r;
    uint32_t n = 0;
    for (uint32_t b = 0; b < nblocks; b++) {
        SignalBlock blk;
        memcpy(&blk, in_iq + (size_t)b * 56, 112);
        r.Resample(blk);
        COMPLEX16* p = r.GetOutStream(28);
        if (p) {
            if (n + 28 > max_out) break;
            memcpy(out_iq + (size_t)n * 2, p, 112);
            n += 28;
        }
    }
    return (int)n;
}
