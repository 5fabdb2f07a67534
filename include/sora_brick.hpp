// sora_brick.hpp -- BRICK-shaped adapters over the C ABI of sora_hip.h.
//
// The reference's operator API is the compile-time BRICK protocol (kernel/brick/inc/brick.h:151-475):
// a brick declares DEFINE_IPORT(TYPE,BURST)/DEFINE_OPORT(TYPE,BURST), implements
//     template<class T_IPIN> bool Process(T_IPIN& ipin)   -- while(ipin.check_read()){peek; append; pop; Next()->Process(opin());}
// plus Reset() and Flush(), and reports errors through bool + CF_Error::error_code.
// These adapters keep exactly that shape -- same verbs, same port TYPEs, burst = 64*N / 48*N_BPSC*N soft values
// for a batch of N symbols -- so a graph built with CREATE_BRICK_* can put `THipFFT64<N>` where `TFFT64`
// (kernel/bb/Brick11/src/fft.hpp:108-135) sits.  The pin-queue contract is reduced to the three calls the
// reference's idiom uses (check_read / peek / pop on the input pin, append on the output pin): any TPinQueue
// (kernel/brick/inc/pinqueue.h:104-183) satisfies it.  The queue buffers must be HBM-resident (device pointers):
// the adapters move no data across PCIe.
//
// Header-only, C++17, no dependency beyond sora_hip.h; nothing here is required by the C host path.
#pragma once
#include <cstddef>
#include <cstdint>
#include "sora_hip.h"

namespace sora_brick {

struct CF_Error { uint32_t error_code = 0; };                 // kernel/brick/inc/stdfacade.h:14-18

template <class TYPE, size_t BURST> struct port_traits { using type = TYPE; static constexpr size_t burst = BURST; };

// Minimal device-resident pin: one burst in flight (the N:N specialisation of TPinQueue, pinqueue.h:152-183).
template <class T, size_t BURST>
class DevicePin {
public:
    explicit DevicePin(T* d_buf = nullptr) : buf_(d_buf), cnt_(0) {}
    void bind(T* d_buf) { buf_ = d_buf; }
    bool check_read() const { return cnt_ > 0; }
    const T* peek() const { return buf_; }
    void pop() { cnt_ = 0; }
    T* append() { cnt_ = BURST; return buf_; }
    void clear() { cnt_ = 0; }
private:
    T* buf_; size_t cnt_;
};

// Common part of a filter brick: next-brick pointer, context, stream.
template <class T_CTX, class T_NEXT>
class HipFilter {
public:
    HipFilter(T_CTX& ctx, T_NEXT* next, void* stream = nullptr) : ctx_(ctx), next_(next), stream_(stream) {}
    void Reset() { if (next_) next_->Reset(); }
    void Flush() { if (next_) next_->Flush(); }
protected:
    bool raise(int rc) { if (rc != SORA_OK) { ctx_.error_code = (uint32_t)rc; return false; } return true; }
    T_CTX& ctx_; T_NEXT* next_; void* stream_;
};

// TFFT64 (fft.hpp:108-135): IPORT COMPLEX16 x 64  ->  OPORT COMPLEX16 x 64, batched over N symbols.
template <size_t N, class T_CTX, class T_NEXT>
class THipFFT64 : public HipFilter<T_CTX, T_NEXT> {
public:
    using iport_traits = port_traits<sora_complex16, 64 * N>;
    using oport_traits = port_traits<sora_complex16, 64 * N>;
    THipFFT64(T_CTX& ctx, T_NEXT* next, sora_complex16* d_out, void* stream = nullptr)
        : HipFilter<T_CTX, T_NEXT>(ctx, next, stream), opin_(d_out) {}
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            const sora_complex16* in = ipin.peek();
            sora_complex16* out = opin_.append();
            if (!this->raise(sora_hip_fft64(in, out, N, this->stream_))) return false;
            ipin.pop();
            if (this->next_ && !this->next_->Process(opin_)) return false;
        }
        return true;
    }
    DevicePin<sora_complex16, 64 * N>& opin() { return opin_; }
private:
    DevicePin<sora_complex16, 64 * N> opin_;
};

// T11aDemap<N_BPSC>::Filter (demapper11a.hpp:10-79): IPORT COMPLEX16 x 64 -> OPORT uchar x 48*N_BPSC.
template <int N_BPSC, size_t N, class T_CTX, class T_NEXT>
class THip11aDemap : public HipFilter<T_CTX, T_NEXT> {
public:
    using iport_traits = port_traits<sora_complex16, 64 * N>;
    using oport_traits = port_traits<uint8_t, 48 * N_BPSC * N>;
    THip11aDemap(T_CTX& ctx, T_NEXT* next, uint8_t* d_out, void* stream = nullptr)
        : HipFilter<T_CTX, T_NEXT>(ctx, next, stream), opin_(d_out) {}
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            uint8_t* out = opin_.append();
            if (!this->raise(sora_hip_demap11a(ipin.peek(), out, N_BPSC, N, this->stream_))) return false;
            ipin.pop();
            if (this->next_ && !this->next_->Process(opin_)) return false;
        }
        return true;
    }
    DevicePin<uint8_t, 48 * N_BPSC * N>& opin() { return opin_; }
private:
    DevicePin<uint8_t, 48 * N_BPSC * N> opin_;
};

// T11aDeinterleave{BPSK,QPSK,QAM16,QAM64} (deinterleaver.hpp): IPORT uchar x N_CBPS -> OPORT uchar x N_CBPS.
template <int N_BPSC, size_t N, class T_CTX, class T_NEXT>
class THip11aDeinterleave : public HipFilter<T_CTX, T_NEXT> {
public:
    using iport_traits = port_traits<uint8_t, 48 * N_BPSC * N>;
    using oport_traits = port_traits<uint8_t, 48 * N_BPSC * N>;
    THip11aDeinterleave(T_CTX& ctx, T_NEXT* next, uint8_t* d_out, void* stream = nullptr)
        : HipFilter<T_CTX, T_NEXT>(ctx, next, stream), opin_(d_out) {}
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            uint8_t* out = opin_.append();
            if (!this->raise(sora_hip_deinterleave11a(ipin.peek(), out, N_BPSC, N, this->stream_))) return false;
            ipin.pop();
            if (this->next_ && !this->next_->Process(opin_)) return false;
        }
        return true;
    }
    DevicePin<uint8_t, 48 * N_BPSC * N>& opin() { return opin_; }
private:
    DevicePin<uint8_t, 48 * N_BPSC * N> opin_;
};

// TSink that terminates a test graph (TDropAny analogue).
template <class T_CTX>
class TDrop {
public:
    explicit TDrop(T_CTX&) {}
    void Reset() {}
    void Flush() {}
    template <class T_IPIN> bool Process(T_IPIN& ipin) { while (ipin.check_read()) ipin.pop(); return true; }
};

// TDownSample44_40 (sampling.hpp:35-66) / TDownSample2 (samples.hpp:9-47) as one filter over a whole capture: the
// input pin holds NIN COMPLEX16 (a multiple of 28), the output pin what sora_hip_ingest_count says comes out of it.
template <size_t NIN, unsigned FLAGS, class T_CTX, class T_NEXT>
class THipResample : public HipFilter<T_CTX, T_NEXT> {
    static_assert(NIN % 28 == 0 && !(FLAGS & SORA_INGEST_RXBLOCK), "whole RX blocks of plain samples");
public:
    static constexpr size_t NOUT = NIN;                          // capacity; the produced count is returned by produced()
    using iport_traits = port_traits<sora_complex16, NIN>;
    using oport_traits = port_traits<sora_complex16, NOUT>;
    THipResample(T_CTX& ctx, T_NEXT* next, sora_complex16* d_out, void* stream = nullptr)
        : HipFilter<T_CTX, T_NEXT>(ctx, next, stream), opin_(d_out) {}
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            sora_complex16* out = opin_.append();
            if (!this->raise(sora_hip_ingest(ipin.peek(), NIN * sizeof(sora_complex16), FLAGS, out, NOUT, &produced_, this->stream_))) return false;
            ipin.pop();
            if (this->next_ && !this->next_->Process(opin_)) return false;
        }
        return true;
    }
    size_t produced() const { return produced_; }
    DevicePin<sora_complex16, NOUT>& opin() { return opin_; }
private:
    DevicePin<sora_complex16, NOUT> opin_;
    size_t produced_ = 0;
};

// ------------------------------------------------------------------------------------------------
// The remaining stage bricks.  They all have the shape spelt out above (peek / append / call / pop / Next()->Process), so they are
// one template: IPORT IT x IB -> OPORT OT x OB with the C entry point supplied as a callable `int(const IT*, OT*, void* stream)`.
template <class IT, size_t IB, class OT, size_t OB, class CALL, class T_CTX, class T_NEXT>
class THipStage : public HipFilter<T_CTX, T_NEXT> {
public:
    using iport_traits = port_traits<IT, IB>;
    using oport_traits = port_traits<OT, OB>;
    THipStage(T_CTX& ctx, T_NEXT* next, OT* d_out, CALL call, void* stream = nullptr)
        : HipFilter<T_CTX, T_NEXT>(ctx, next, stream), opin_(d_out), call_(call) {}
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            OT* out = opin_.append();
            if (!this->raise(call_(ipin.peek(), out, this->stream_))) return false;
            ipin.pop();
            if (this->next_ && !this->next_->Process(opin_)) return false;
        }
        return true;
    }
    DevicePin<OT, OB>& opin() { return opin_; }
private:
    DevicePin<OT, OB> opin_; CALL call_;
};

// FFT<128> (core/inc/fft_r4dif.h): IPORT COMPLEX16 x 128 -> OPORT COMPLEX16 x 128, N transforms per burst.
struct CallFFT128 { size_t n; int operator()(const sora_complex16* in, sora_complex16* out, void* st) const { return sora_hip_fft128(in, out, n, st); } };
template <size_t N, class T_CTX, class T_NEXT>
struct THipFFT128 : THipStage<sora_complex16, 128 * N, sora_complex16, 128 * N, CallFFT128, T_CTX, T_NEXT> {
    THipFFT128(T_CTX& ctx, T_NEXT* next, sora_complex16* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 128 * N, sora_complex16, 128 * N, CallFFT128, T_CTX, T_NEXT>(ctx, next, d_out, CallFFT128{ N }, stream) {}
};

// T11aLTS (channel_11a.hpp:33-230): IPORT COMPLEX16 x 144.  The reference brick is a sink that fills the context facades
// CF_CFOffset / CF_FreqCompensate / CF_Channel_11a; here they are the device record `d_ctx` (sora_lts11a_ctx) the later bricks are bound to.
template <class T_CTX>
class THip11aLTS {
public:
    using iport_traits = port_traits<sora_complex16, 144>;
    THip11aLTS(T_CTX& ctx, sora_lts11a_ctx* d_ctx, void* stream = nullptr) : ctx_(ctx), d_ctx_(d_ctx), stream_(stream) {}
    void Reset() {}
    void Flush() {}
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            const int rc = sora_hip_lts11a(ipin.peek(), d_ctx_, 1, stream_);
            if (rc != SORA_OK) { ctx_.error_code = (uint32_t)rc; return false; }
            ipin.pop();
        }
        return true;
    }
private:
    T_CTX& ctx_; sora_lts11a_ctx* d_ctx_; void* stream_;
};

// T11aDataSymbol -> TFreqCompensation -> TFFT64 -> TChannelEqualization (PHY_11a.hpp:361-430, channel_11a.hpp:532-653) as one brick:
// IPORT COMPLEX16 x 80 -> OPORT COMPLEX16 x 64, N symbols per burst, bound to the frame's T11aLTS record.
struct CallSymFront11a { const sora_lts11a_ctx* d_ctx; size_t n; int operator()(const sora_complex16* in, sora_complex16* out,
        void* st) const { return sora_hip_symfront11a(in, d_ctx, nullptr, out, n, st); } };
template <size_t N, class T_CTX, class T_NEXT>
struct THip11aSymFront : THipStage<sora_complex16, 80 * N, sora_complex16, 64 * N, CallSymFront11a, T_CTX, T_NEXT> {
    THip11aSymFront(T_CTX& ctx, T_NEXT* next, const sora_lts11a_ctx* d_ctx, sora_complex16* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 80 * N, sora_complex16, 64 * N, CallSymFront11a, T_CTX, T_NEXT>(ctx, next, d_out, CallSymFront11a{ d_ctx, N }, stream) {}
};

// The three one-multiply bricks of the symbol chain, each where its SSE brick sits (IPORT COMPLEX16 x 64 -> OPORT COMPLEX16 x 64, N symbols per burst):
//   THipFreqCompensation     TFreqCompensation     (channel_11a.hpp:614-653)   bound to the frame's T11aLTS record (CF_FreqCompensate::Coeffs)
//   THipChannelEqualization  TChannelEqualization  (channel_11a.hpp:534-604)   bound to the same record (CF_Channel_11a::ChannelCoeffs)
//   THipPhaseCompensate      TPhaseCompensate      (freqoffset.hpp:16-66)      bound to the tracker's state (CF_PhaseCompensate::CompCoeffs = sora_track11a_state::comp)
struct CallFreqComp11a { const sora_lts11a_ctx* d_ctx; size_t n; int operator()(const sora_complex16* in, sora_complex16* out,
        void* st) const { return sora_hip_freq_comp11a(in, d_ctx, nullptr, out, n, st); } };
struct CallEqualize11a { const sora_lts11a_ctx* d_ctx; size_t n; int operator()(const sora_complex16* in, sora_complex16* out,
        void* st) const { return sora_hip_equalize11a(in, d_ctx, nullptr, out, n, st); } };
struct CallPhaseComp11a { const sora_track11a_state* d_state; size_t n; int operator()(const sora_complex16* in, sora_complex16* out,
        void* st) const { return sora_hip_phase_comp11a(in, d_state, nullptr, out, n, st); } };
template <size_t N, class T_CTX, class T_NEXT>
struct THipFreqCompensation : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallFreqComp11a, T_CTX, T_NEXT> {
    THipFreqCompensation(T_CTX& ctx, T_NEXT* next, const sora_lts11a_ctx* d_ctx, sora_complex16* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallFreqComp11a, T_CTX, T_NEXT>(ctx, next, d_out, CallFreqComp11a{ d_ctx, N }, stream) {}
};
template <size_t N, class T_CTX, class T_NEXT>
struct THipChannelEqualization : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallEqualize11a, T_CTX, T_NEXT> {
    THipChannelEqualization(T_CTX& ctx, T_NEXT* next, const sora_lts11a_ctx* d_ctx, sora_complex16* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallEqualize11a, T_CTX, T_NEXT>(ctx, next, d_out, CallEqualize11a{ d_ctx, N }, stream) {}
};
template <size_t N, class T_CTX, class T_NEXT>
struct THipPhaseCompensate : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallPhaseComp11a, T_CTX, T_NEXT> {
    THipPhaseCompensate(T_CTX& ctx, T_NEXT* next, const sora_track11a_state* d_state, sora_complex16* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallPhaseComp11a, T_CTX, T_NEXT>(ctx, next, d_out, CallPhaseComp11a{ d_state, N }, stream) {}
};

// TPhaseCompensate -> TPilotTrack (freqoffset.hpp:14-66, pilot.hpp:121-269): IPORT COMPLEX16 x 64 -> OPORT COMPLEX16 x 64, N symbols of ONE
// frame per burst; CF_PhaseCompensate / CF_PilotTrack live in `d_state` and carry over from burst to burst.  d_first / d_nsym: device words
// holding 0 and N (the C entry point takes frame tables).
struct CallPilotTrack11a {
    const uint32_t* d_first; const uint32_t* d_nsym; sora_track11a_state* d_state;
    int operator()(const sora_complex16* in, sora_complex16* out, void* st) const { return sora_hip_pilot_track11a(in, d_first, d_nsym, d_state, out, 1, st); }
};
template <size_t N, class T_CTX, class T_NEXT>
struct THip11aPilotTrack : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallPilotTrack11a, T_CTX, T_NEXT> {
    THip11aPilotTrack(T_CTX& ctx, T_NEXT* next, const uint32_t* d_first, const uint32_t* d_nsym, sora_track11a_state* d_state, sora_complex16* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 64 * N, sora_complex16, 64 * N, CallPilotTrack11a, T_CTX, T_NEXT>(ctx, next, d_out, CallPilotTrack11a{ d_first, d_nsym, d_state }, stream) {}
};

// T11aViterbi<5000*8,48,256,24> (viterbi.hpp:103-237): IPORT uchar x 48*N_BPSC (one symbol's soft values) -> OPORT uchar x frame_length + 2.
// Like the reference brick it consumes symbol bursts and knows the frame from the context (SetFrame = CF_11aRxVector: length, code rate,
// symbols); the trellis is run when the last symbol has arrived.  d_frame: device scratch of nsym x 48*N_BPSC soft values + 16 bytes of tables.
template <int N_BPSC, class T_CTX, class T_NEXT>
class THip11aViterbi : public HipFilter<T_CTX, T_NEXT> {
public:
    using iport_traits = port_traits<uint8_t, 48 * N_BPSC>;
    THip11aViterbi(T_CTX& ctx, T_NEXT* next, uint8_t* d_frame, uint8_t* d_out, void* stream = nullptr)
        : HipFilter<T_CTX, T_NEXT>(ctx, next, stream), d_soft_(d_frame + 16), d_tab_(d_frame), d_out_(d_out) {}
    void SetFrame(uint16_t frame_length, int code_rate, uint32_t nsym) { len_ = frame_length; cr_ = code_rate; nsym_ = nsym; got_ = 0; }
    void Reset() { got_ = 0; HipFilter<T_CTX, T_NEXT>::Reset(); }
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            if (got_ < nsym_ && sora_hip_memcpy_d2d(d_soft_ + (size_t)got_ * 48 * N_BPSC, ipin.peek(), 48 * N_BPSC,
                    this->stream_) != SORA_OK) return this->raise(SORA_ERR_HARDWARE_FAILED);
            ipin.pop();
            if (++got_ == nsym_) {
                const uint32_t tab[4] = { 0u, nsym_ * 48u * (uint32_t)N_BPSC, (uint32_t)len_, 0u };   // soft_off, nsoft, frame_len (uint16), out_off
                if (sora_hip_memcpy_h2d(d_tab_, tab, sizeof(tab)) != SORA_OK) return this->raise(SORA_ERR_HARDWARE_FAILED);
                const uint32_t* t = reinterpret_cast<const uint32_t*>(d_tab_);
                if (!this->raise(sora_hip_viterbi11a(d_soft_, t, t + 1, reinterpret_cast<const uint16_t*>(t + 2), cr_, d_out_, t + 3, 1, this->stream_))) return false;
                decoded_ = true;
            }
        }
        return true;
    }
    bool decoded() const { return decoded_; }
    const uint8_t* output() const { return d_out_; }
private:
    uint8_t* d_soft_; uint8_t* d_tab_; uint8_t* d_out_;
    uint16_t len_ = 0; int cr_ = 0; uint32_t nsym_ = 0, got_ = 0; bool decoded_ = false;
};

// 802.11n stage bricks (one burst = N symbols / frames), each forwarding to its entry point:
//   T11nDemap* (demapper11n.hpp:89-309), T11nDeinterleave*_S0/_S1 (deinterleaver_11n.hpp), TMimoChannelComp (channel_11n.hpp:445-521)
struct CallDemap11n { int nb; size_t n; int operator()(const sora_complex16* in, uint8_t* out, void* st) const { return sora_hip_demap11n(in, out, nb, n, st); } };
template <int N_BPSC, size_t N, class T_CTX, class T_NEXT>
struct THip11nDemap : THipStage<sora_complex16, 64 * N, uint8_t, 52 * N_BPSC * N, CallDemap11n, T_CTX, T_NEXT> {
    THip11nDemap(T_CTX& ctx, T_NEXT* next, uint8_t* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 64 * N, uint8_t, 52 * N_BPSC * N, CallDemap11n, T_CTX, T_NEXT>(ctx, next, d_out, CallDemap11n{ N_BPSC, N }, stream) {}
};
struct CallDeint11n { int nb, ss; size_t n; int operator()(const uint8_t* in, uint8_t* out, void* st) const { return sora_hip_deinterleave11n(in, out, nb, ss, n, st); } };
template <int N_BPSC, int SPATIAL_STREAM, size_t N, class T_CTX, class T_NEXT>
struct THip11nDeinterleave : THipStage<uint8_t, 52 * N_BPSC * N, uint8_t, 52 * N_BPSC * N, CallDeint11n, T_CTX, T_NEXT> {
    THip11nDeinterleave(T_CTX& ctx, T_NEXT* next, uint8_t* d_out, void* stream = nullptr)
        : THipStage<uint8_t, 52 * N_BPSC * N, uint8_t, 52 * N_BPSC * N, CallDeint11n, T_CTX, T_NEXT>(ctx, next, d_out, CallDeint11n{ N_BPSC, SPATIAL_STREAM, N }, stream) {}
};
// TMimoChannelComp: NSTREAM = 2 ports in the reference (two RX chains in, two spatial streams out); here the two streams are the two halves of
// one burst: IPORT COMPLEX16 x 2 x 64N (chain 0 symbols, then chain 1) -> OPORT COMPLEX16 x 2 x 64N (stream 0, then stream 1).
struct CallMimoComp11n {
    const sora_complex16* d_hinv; size_t n;
    int operator()(const sora_complex16* in, sora_complex16* out, void* st) const { return sora_hip_mimo_comp11n(d_hinv, nullptr, in, in + 64 * n, out, out + 64 * n, n, st); }
};
template <size_t N, class T_CTX, class T_NEXT>
struct THip11nMimoComp : THipStage<sora_complex16, 128 * N, sora_complex16, 128 * N, CallMimoComp11n, T_CTX, T_NEXT> {
    THip11nMimoComp(T_CTX& ctx, T_NEXT* next, const sora_complex16* d_hinv, sora_complex16* d_out, void* stream = nullptr)
        : THipStage<sora_complex16, 128 * N, sora_complex16, 128 * N, CallMimoComp11n, T_CTX, T_NEXT>(ctx, next, d_out, CallMimoComp11n{ d_hinv, N }, stream) {}
};

// ISource over a batch of captures = the whole demod graph behind one handle (brick.h:343-353: Process/Seek/Reset/Flush).
class THipRx11aSource {
public:
    THipRx11aSource(CF_Error& ctx, const sora_rx_cfg& cfg) : ctx_(ctx), rx_(nullptr) { ctx_.error_code = (uint32_t)sora_rx_create(&cfg, &rx_); }
    ~THipRx11aSource() { if (rx_) sora_rx_destroy(rx_); }
    THipRx11aSource(const THipRx11aSource&) = delete;
    THipRx11aSource& operator=(const THipRx11aSource&) = delete;
    void Bind(const sora_complex16* d_iq, const sora_capture_desc* caps, size_t ncaps) { d_iq_ = d_iq; caps_ = caps; ncaps_ = ncaps; }
    bool Process()
    {
        if (!rx_) return false;
        const int rc = sora_rx_process_dev(rx_, d_iq_, caps_, ncaps_);
        if (rc != SORA_OK) { ctx_.error_code = (uint32_t)rc; return false; }
        return true;
    }
    void Reset() { if (rx_) sora_rx_reset(rx_); }
    void Flush() { if (rx_) sora_rx_flush(rx_); }
    sora_rx_t* handle() { return rx_; }
private:
    CF_Error& ctx_; sora_rx_t* rx_;
    const sora_complex16* d_iq_ = nullptr; const sora_capture_desc* caps_ = nullptr; size_t ncaps_ = 0;
};

// The 802.11b graph as one ISource: CreateDemodGraph (fb11bdemod_config.hpp:122-172) + MAC11b_Receive over a batch of 44 MHz captures.
class THipRx11bSource {
public:
    THipRx11bSource(CF_Error& ctx, const sora_rx_cfg& cfg) : ctx_(ctx), rx_(nullptr) { ctx_.error_code = (uint32_t)sora_rx11b_create(&cfg, &rx_); }
    ~THipRx11bSource() { if (rx_) sora_rx11b_destroy(rx_); }
    THipRx11bSource(const THipRx11bSource&) = delete;
    THipRx11bSource& operator=(const THipRx11bSource&) = delete;
    void Bind(const sora_complex16* d_iq, const sora_capture_desc* caps, size_t ncaps) { d_iq_ = d_iq; caps_ = caps; ncaps_ = ncaps; }
    bool Process()
    {
        if (!rx_) return false;
        const int rc = sora_rx11b_process_dev(rx_, d_iq_, caps_, ncaps_);
        if (rc != SORA_OK) { ctx_.error_code = (uint32_t)rc; return false; }
        return true;
    }
    void Reset() {}                                      // every call starts from the graph's initial state
    void Flush() { if (rx_) (void)sora_rx11b_synchronize(rx_); }   // the handle keeps two calls in flight: Flush() = all of them have finished
    sora_rx11b_t* handle() { return rx_; }
private:
    CF_Error& ctx_; sora_rx11b_t* rx_;
    const sora_complex16* d_iq_ = nullptr; const sora_capture_desc* caps_ = nullptr; size_t ncaps_ = 0;
};

// The 802.11n 2x2 graph as one ISource: CreateDemodGraph11n (fb11ndemod_config.hpp:166-257) + RxThread over a batch of two-chain
// 40 MHz captures (what TMemSamples2 is initialised with: MemSamplesDesc::Init(2, InputBuf, nCnt), fb11n_demod.cpp:113-117).
class THipRx11nSource {
public:
    THipRx11nSource(CF_Error& ctx, const sora_rx_cfg& cfg) : ctx_(ctx), rx_(nullptr) { ctx_.error_code = (uint32_t)sora_rx11n_create(&cfg, &rx_); }
    ~THipRx11nSource() { if (rx_) sora_rx11n_destroy(rx_); }
    THipRx11nSource(const THipRx11nSource&) = delete;
    THipRx11nSource& operator=(const THipRx11nSource&) = delete;
    void Bind(const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_capture_desc* caps, size_t ncaps) { d_iq_[0] = d_iq0; d_iq_[1] = d_iq1;
        caps_ = caps; ncaps_ = ncaps; }
    bool Process()
    {
        if (!rx_) return false;
        const int rc = sora_rx11n_process_dev(rx_, d_iq_[0], d_iq_[1], caps_, ncaps_);
        if (rc != SORA_OK) { ctx_.error_code = (uint32_t)rc; return false; }
        return true;
    }
    void Reset() {}                                      // every call starts from the graph's initial state
    void Flush() { if (rx_) (void)sora_rx11n_synchronize(rx_); }   // (sora_rx11n_set_depth calls in flight)
    sora_rx11n_t* handle() { return rx_; }
private:
    CF_Error& ctx_; sora_rx11n_t* rx_;
    const sora_complex16* d_iq_[2] = { nullptr, nullptr }; const sora_capture_desc* caps_ = nullptr; size_t ncaps_ = 0;
};

}  // namespace sora_brick
