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
    void Flush() {}
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
    void Bind(const sora_complex16* d_iq0, const sora_complex16* d_iq1, const sora_capture_desc* caps, size_t ncaps) { d_iq_[0] = d_iq0; d_iq_[1] = d_iq1; caps_ = caps; ncaps_ = ncaps; }
    bool Process()
    {
        if (!rx_) return false;
        const int rc = sora_rx11n_process_dev(rx_, d_iq_[0], d_iq_[1], caps_, ncaps_);
        if (rc != SORA_OK) { ctx_.error_code = (uint32_t)rc; return false; }
        return true;
    }
    void Reset() {}                                      // every call starts from the graph's initial state
    void Flush() {}
    sora_rx11n_t* handle() { return rx_; }
private:
    CF_Error& ctx_; sora_rx11n_t* rx_;
    const sora_complex16* d_iq_[2] = { nullptr, nullptr }; const sora_capture_desc* caps_ = nullptr; size_t ncaps_ = 0;
};

}  // namespace sora_brick
