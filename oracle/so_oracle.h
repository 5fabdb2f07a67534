/* so_oracle.h -- CPU oracle for the 802.11a RX PHY hot path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C, scalar restatement of the reference's SSE brick chain
 *   kernel/bb/demod11/fb11ademod_config.hpp:168-233 (graph) + fb11a_demod.cpp:29-81 (frame loop)
 * written from the algorithm, bit-exact by construction and pinned against
 *   (1) the reference's own arithmetic headers compiled into oracle/_ref/libsora_ref.so, and
 *   (2) the reference's only IQ fixture kernel/test-data/fsample-6.dmp (MPDU sha256, FCS).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef SO_ORACLE_H
#define SO_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int16_t re, im; } so_c16;

/* error codes: kernel/bb/Brick11/src/ieee80211facade.hpp:10-19, kernel/brick/inc/stdfacade.h:9-11 */
#define SO_E_SUCCESS          0x00000000u
#define SO_E_FRAME_OK         0x00000001u
#define SO_E_PLCP_HEADER_FAIL 0x80000005u
#define SO_E_CRC32_FAIL       0x80000006u
#define SO_E_CS_TIMEOUT       0x80000007u
#define SO_E_NOT_SUPPORTED    0x80000003u
#define SO_E_SFD_FAIL         0x80000004u
#define SO_E_SFD_TIMEOUT      0x80000008u
#define SO_E_SYNC_TIMEOUT     0x80000009u
#define SO_E_FAILED           0x8000FFFFu

/* code rates: ieee80211const.h:14-20 */
enum { SO_CR_12 = 0, SO_CR_23 = 1, SO_CR_34 = 2 };

void so_init(void);                                   /* builds all LUTs; idempotent */

/* ---- LUTs (core/inc/intalg.h + intalglut.h restated as closed forms, see so_lut.c) ---- */
const int16_t* so_usin_lut(void);                     /* [65536] */
const int16_t* so_ucos_lut(void);                     /* [65536] */
const int16_t* so_uatan2_lut(void);                   /* [256*256] */
int16_t so_uatan2(int y, int x);                      /* intalg.h:100-113 */
const uint8_t* so_demap_lut(int which);               /* 0 bpsk, 1 qam16_2, 2 qam64_2, 3 qam64_3; [256] each */
const int16_t* so_twiddle(int n, int k);              /* wFFTLUT<n>_<k>: n/4 complex (re,im) pairs; k=1,2,3 (n=8: k=1) */
const so_c16*  so_sts_pattern(void);                  /* [16][16] cca.hpp:268-277 */

/* ---- primitives ---- */
void so_fft64(const so_c16* in, so_c16* out);         /* FFT<64>  core/inc/fft_r4dif.h */
void so_ifft64(const so_c16* in, so_c16* out);        /* IFFT<64> core/inc/ifft_r4dif.h */
void so_fft128(const so_c16* in, so_c16* out);
void so_ifft128(const so_c16* in, so_c16* out);
so_c16 so_mul_q15(so_c16 a, so_c16 b);                /* vcs mul(a,b)  vector128.h:1201-1211 */
uint32_t so_crc32(const uint8_t* p, uint32_t n);      /* core/inc/CRC32.h */

/* ---- per-frame receive context: the CF_* facades of ieee80211facade.hpp:166-257 ---- */
typedef struct {
    int16_t  CFO_est;
    so_c16   FreqCoeffs[64];
    so_c16   ChannelCoeffs[64];
    int16_t  CFO_comp, SFO_comp;
    so_c16   CompCoeffs[64];
    int16_t  CFO_tracker, SFO_tracker;
    uint32_t symbol_count;
} so_rx11a_ctx;

void so_rx11a_ctx_reset(so_rx11a_ctx* c);                                   /* fb11ademod_config.hpp:68-95 */
void so_lts(so_rx11a_ctx* c, const so_c16 in144[144]);                       /* T11aLTS  channel_11a.hpp:206-229 */
void so_sym_front(const so_rx11a_ctx* c, const so_c16 in80[80], so_c16 eq[64]);/* T11aDataSymbol..TChannelEqualization */
void so_sym_track(so_rx11a_ctx* c, const so_c16 eq[64], so_c16 out[64]);     /* TPhaseCompensate + TPilotTrack */
/* ... and the five bricks of those two, one by one */
void so_freq_comp(const so_rx11a_ctx* c, const so_c16 in[64], so_c16 out[64]);   /* TFreqCompensation (in = the 64 samples behind the cyclic prefix) */
void so_equalize(const so_rx11a_ctx* c, const so_c16 in[64], so_c16 out[64]);    /* TChannelEqualization */
void so_phase_comp(const so_rx11a_ctx* c, const so_c16 in[64], so_c16 out[64]);  /* TPhaseCompensate */
void so_pilot_track(so_rx11a_ctx* c, const so_c16 pc[64], so_c16 out[64]);       /* TPilotTrack (pc = TPhaseCompensate's output) */
void so_demap(int nbpsc, const so_c16 in[64], uint8_t* soft);                /* T11aDemap<N_BPSC> -> 48*nbpsc soft */
void so_deinterleave(int nbpsc, const uint8_t* in, uint8_t* out);            /* T11aDeinterleave* */
uint32_t so_viterbi_sig(const uint8_t soft48[48]);                           /* Viterbi_sig11 + >>6 */
int  so_parse_plcp(uint32_t sig, uint32_t* rate_kbps, uint16_t* length, uint16_t* code_rate, uint16_t* nsym); /* 1 ok */
/* T11aViterbi<5000*8,48,256,24> schedule over a whole frame; returns bytes written (frame_length+2) */
int  so_viterbi_frame(const uint8_t* soft, uint32_t nsoft, int code_rate, uint32_t frame_length, uint8_t* out);
int  so_viterbi_frame_ex(const uint8_t* soft, uint32_t nsoft, int code_rate, uint32_t frame_length, uint8_t* out, uint32_t depth, uint32_t lookahead);   /* depth <= 256 */
/* T11aDesc + TBB11aFrameSink: in = frame_length+2 decoded bytes; mpdu gets frame_length bytes; returns error code */
uint32_t so_desc_sink(const uint8_t* dec, uint32_t frame_length, uint8_t* mpdu, uint32_t* crc_in_frame);

/* ---- whole-capture receive (the offline harness, fb11a_demod.cpp:88-120) ---- */
typedef struct {
    uint32_t start_sample;   /* 20 MHz-rate index of the first sample handed to T11aLTS */
    uint32_t end_sample;     /* one past the last 20 MHz-rate sample consumed by the frame */
    uint32_t error_code;     /* SO_E_FRAME_OK / SO_E_CRC32_FAIL / SO_E_PLCP_HEADER_FAIL */
    uint32_t rate_kbps;
    uint16_t length;         /* PLCP LENGTH (MPDU incl. FCS) */
    uint16_t nsym;           /* data symbols (without SIGNAL) */
    uint32_t crc32;          /* FCS as found in the frame */
    int16_t  cfo_est;
    uint16_t reserved;
    uint32_t mpdu_offset;    /* into mpdu_buf */
} so_frame_result;

typedef struct {             /* optional intermediates of the FIRST decoded frame (NULL members are skipped) */
    so_rx11a_ctx* ctx_after_lts;
    so_c16*  eq;             /* [nsym+1][64] equalised symbols (SIGNAL first) */
    so_c16*  tracked;        /* [nsym+1][64] after pilot tracking */
    uint8_t* soft;           /* data-symbol soft bits, de-interleaved, nsym*ncbps */
    uint8_t* decoded;        /* length+2 Viterbi output bytes */
    uint32_t cap_syms;       /* capacity of eq/tracked in symbols */
    uint32_t cap_soft;
    uint32_t n_syms, n_soft; /* filled */
} so_trace;

/* sample_rate_mhz: 40 (dump rate; TDownSample2 first), 20 (already decimated: the even samples) or 44 (the 40 MHz
 * stream of TDownSample44_40 under the 44 MHz graph's reset semantics, see so_rx11a.c).
 * Returns number of frame results written. */
int so_rx11a_capture(const so_c16* iq, uint32_t nsamples, int sample_rate_mhz,
                     so_frame_result* res, int max_res, uint8_t* mpdu_buf, uint32_t mpdu_cap, so_trace* trace);

/* 802.11b receive graph (so_rx11b.c): one 44 MHz capture; end_sample = source position (44 MHz samples) at which the
 * harness sees the event, rate_kbps/length/crc32 as the reference's CF_11bRxVector holds them (crc32: 3 FCS bytes + 1 stale). */
int so_rx11b_capture(const so_c16* iq, uint32_t nsamples, so_frame_result* res, int max_res, uint8_t* mpdu_buf, uint32_t mpdu_cap);

/* 802.11n stage bricks (so_11n.c): T11nDemap* and T11nDeinterleave*_S{0,1}; return the number of soft values (52 * nbpsc) */
int so_demap11n(int nbpsc, const so_c16 in[64], uint8_t* out);
int so_deinterleave11n(int nbpsc, int stream, const uint8_t* in, uint8_t* out);
int so_deinterleave11n_index(int nbpsc, int stream, int k);
const uint8_t* so_demap11n_lut(int which);
/* TMimoChannelEst / TMimoChannelComp (channel_11n.hpp:329-521); ltf_r = the two HT-LTF symbols of RX chain r after the FFT */
void so_mimo_est11n(const so_c16 ltf0[128], const so_c16 ltf1[128], so_c16 h[2][128], so_c16 hinv[2][128]);
/* dsp_math (dsp_math.h) and the bricks built on it; state[24] = vfo_delta_i | vfo_step_i | vfo_theta_i (8 int16 each) */
const so_c16* so_dsp_sincos_table(void);
const int16_t* so_dsp_atan_table(void);
int16_t so_dsp_atan16(int16_t x, int16_t y);
int16_t so_dsp_atan32(int32_t x, int32_t y);
int16_t so_cfo_est11n(const so_c16 l0[128], const so_c16 l1[128], int16_t state[24]);
void so_freq_comp11n(int16_t state[24], const so_c16* in0, const so_c16* in1, so_c16* out0, so_c16* out1, int nbursts);
void so_pilot_track11n(int16_t theta[8], const so_c16 x0[64], const so_c16 x1[64]);
void so_siso_est11n(const so_c16 l0[128], const so_c16 l1[128], so_c16 ch[2][64]);
void so_siso_comp11n(const so_c16 ch[2][64], const so_c16 y0[64], const so_c16 y1[64], so_c16 x0[64], so_c16 x1[64]);
void so_mrc11n(const so_c16 a[64], const so_c16 b[64], so_c16 out[64]);
void so_sig_demap11n(const so_c16 sym[192], uint8_t soft[144]);
/* 802.11n receive graph (so_rx11n.c): two 40 MHz captures -> events (rate_kbps = MCS index) */
int so_rx11n_capture(const so_c16* iq0, const so_c16* iq1, uint32_t nsamples, so_frame_result* res, int max_res, uint8_t* mpdu_buf, uint32_t mpdu_cap);
size_t so_autocorr11n_size(void);
void so_autocorr11n_reset(void* state);
void so_autocorr11n_burst(void* state, const so_c16 x0[4], const so_c16 x1[4], int64_t acorr[4], int64_t energy[4]);
int so_cca11n(const so_c16* iq0, const so_c16* iq1, uint32_t nbursts, uint32_t skip, uint32_t* detect, int max_detect);
void so_viterbi_sig_bits(const uint8_t* soft, int nbits, uint8_t* out);
/* fields[9] = error_code, data_rate_kbps, frame_length, ht_frame_mcs, ht_frame_length, code_rate, total_symbols, remain_symbols, symbol_type */
int so_sig_decode11n(const uint8_t soft[144], uint8_t out9[9], uint32_t fields[9]);
void so_mimo_comp11n(const so_c16 hinv[2][128], const so_c16 y0[64], const so_c16 y1[64], so_c16 x0[64], so_c16 x1[64]);

/* RX_BLOCK dump de-framing (brick/inc/brickutil.h:20-58); raw14: apply the (int16)(x<<2) sign fix. */
int so_load_dump(const uint8_t* file, uint32_t file_bytes, so_c16* out, uint32_t max_samples, int raw14);
/* capture ingest (so_ingest.c): TDownSample44_40 / Down44to40 (sampling.hpp:35-66, 44MTo40M.hpp:62-123), TDownSample2 (samples.hpp:9-47) */
int so_down44to40(const so_c16* in, uint32_t n_in, so_c16* out, uint32_t max_out);
int so_downsample2(const so_c16* in, uint32_t n_in, so_c16* out, uint32_t max_out);

/* ---- transmitter (test-vector generator), fb11amod_config.hpp:74-110 ---- */
/* mpdu_nofcs: MPDU without FCS (FCS appended).  out8: COMPLEX8 samples @40 MHz (preamble 640 + 160/symbol).
 * Returns number of complex samples, or <0.  scramble_seed: CF_ScramblerSeed (harness uses 0xFF). */
int so_tx11a(const uint8_t* mpdu_nofcs, uint32_t len, uint32_t rate_kbps, uint8_t scramble_seed,
             int8_t* out8, uint32_t max_samples);

#ifdef __cplusplus
}
#endif
#endif
