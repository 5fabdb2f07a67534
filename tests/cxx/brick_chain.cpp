// brick_chain.cpp -- a BRICK-shaped graph built from the adapters of include/sora_brick.hpp and EXECUTED on the GPU
// (tests/test_gpu_hosts.py): source pin -> THipFFT64 -> THip11aDemap<N_BPSC> -> THip11aDeinterleave<N_BPSC> -> sink,
// sink-first construction and Process()/Flush()/Reset() exactly as a CREATE_BRICK_* chain of the reference is driven
// (kernel/brick/inc/brick.h:174-282).  Reads <in.bin> = n x 64 COMPLEX16, feeds them N symbols per burst, writes the
// de-interleaved soft values to <out.bin>.  usage: brick_chain <n_bpsc> <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sora_brick.hpp"

using namespace sora_brick;

// TSink: copies every burst it is handed into a host vector (the graph's TBB11aFrameSink stand-in)
template <size_t BURST>
class TCollect {
public:
    explicit TCollect(CF_Error&) {}
    void Reset() { resets++; }
    void Flush() { flushes++; }
    template <class T_IPIN> bool Process(T_IPIN& ipin)
    {
        while (ipin.check_read()) {
            const size_t at = host.size();
            host.resize(at + BURST);
            if (sora_hip_stream_synchronize(nullptr) != SORA_OK || sora_hip_memcpy_d2h(host.data() + at, ipin.peek(), BURST) != SORA_OK) return false;
            ipin.pop();
        }
        return true;
    }
    std::vector<uint8_t> host; int resets = 0, flushes = 0;
};

template <int NB>
static int run(const std::vector<sora_complex16>& in, const char* outp)
{
    constexpr size_t N = 16;                                           // symbols per burst
    const size_t nsym = in.size() / 64, nburst = nsym / N;
    sora_complex16* d_in = (sora_complex16*)sora_hip_malloc(64 * N * sizeof(sora_complex16));
    sora_complex16* d_fft = (sora_complex16*)sora_hip_malloc(64 * N * sizeof(sora_complex16));
    uint8_t* d_soft = (uint8_t*)sora_hip_malloc(48 * NB * N); uint8_t* d_de = (uint8_t*)sora_hip_malloc(48 * NB * N);
    if (!d_in || !d_fft || !d_soft || !d_de) { fprintf(stderr, "device memory: %s\n", sora_hip_last_error()); return 1; }
    CF_Error ctx;
    TCollect<48 * NB * N> sink(ctx);                                   // sink first, as CREATE_BRICK_SINK / _FILTER do
    THip11aDeinterleave<NB, N, CF_Error, decltype(sink)> deint(ctx, &sink, d_de);
    THip11aDemap<NB, N, CF_Error, decltype(deint)> demap(ctx, &deint, d_soft);
    THipFFT64<N, CF_Error, decltype(demap)> fft(ctx, &demap, d_fft);
    DevicePin<sora_complex16, 64 * N> src(d_in);
    fft.Reset();
    for (size_t b = 0; b < nburst; b++) {
        if (sora_hip_memcpy_h2d(src.append(), in.data() + b * 64 * N, 64 * N * sizeof(sora_complex16)) != SORA_OK) return 1;
        if (!fft.Process(src)) { fprintf(stderr, "Process failed: error_code %08x (%s)\n", ctx.error_code, sora_hip_last_error()); return 1; }
    }
    fft.Flush();
    if (sink.resets != 1 || sink.flushes != 1) { fprintf(stderr, "Reset/Flush did not reach the sink\n"); return 1; }
    FILE* fo = fopen(outp, "wb");
    if (!fo || fwrite(sink.host.data(), 1, sink.host.size(), fo) != sink.host.size()) return 1;
    fclose(fo);
    printf("brick chain: %zu symbols in %zu bursts, %zu soft values out\n", nburst * N, nburst, sink.host.size());
    return 0;
}

int main(int argc, char** argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s <n_bpsc> <in.bin> <out.bin>\n", argv[0]); return 2; }
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<sora_complex16> in((size_t)bytes / sizeof(sora_complex16));
    if (fread(in.data(), sizeof(sora_complex16), in.size(), f) != in.size()) return 1;
    fclose(f);
    switch (atoi(argv[1])) {
    case 1: return run<1>(in, argv[3]);
    case 2: return run<2>(in, argv[3]);
    case 4: return run<4>(in, argv[3]);
    case 6: return run<6>(in, argv[3]);
    }
    return 2;
}
