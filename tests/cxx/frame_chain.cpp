// frame_chain.cpp -- ONE 802.11a FRAME decoded by a graph of BRICK-shaped adapters (include/sora_brick.hpp), executed on the GPU
// (tests/test_gpu_hosts.py).  The graph is the data path of CreateDemodGraph11a (kernel/bb/demod11/fb11ademod_config.hpp:188-226):
//   T11aLTS                                                                              (144 samples -> the frame's context record)
//   T11aDataSymbol..TChannelEqualization -> TPhaseCompensate+TPilotTrack -> drop         (the SIGNAL symbol: tracker state only)
//   T11aDataSymbol..TChannelEqualization -> TPhaseCompensate+TPilotTrack -> T11aDemap<N> -> T11aDeinterleave<N> -> T11aViterbi
// one symbol per Process() -- the reference's own burst -- sink-first construction, errors through CF_Error.  Timing and the RX vector
// (start sample, symbols, rate, length: what TCCA11a and T11aPLCPParser put into the context) come from the command line.
// usage: frame_chain <capture20.bin> <start> <nsym> <n_bpsc> <code_rate 0|1|2> <frame_length> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sora_brick.hpp"

using namespace sora_brick;

template <int NB>
static int run(const std::vector<sora_complex16>& cap, size_t start, uint32_t nsym, int code_rate, uint16_t length, const char* outp)
{
    sora_complex16* d_cap = (sora_complex16*)sora_hip_malloc(cap.size() * sizeof(sora_complex16));
    sora_lts11a_ctx* d_ctx = (sora_lts11a_ctx*)sora_hip_malloc(sizeof(sora_lts11a_ctx));
    sora_track11a_state* d_state = (sora_track11a_state*)sora_hip_malloc(sizeof(sora_track11a_state));
    uint32_t* d_one = (uint32_t*)sora_hip_malloc(8);
    sora_complex16* d_eq = (sora_complex16*)sora_hip_malloc(64 * 4); sora_complex16* d_trk = (sora_complex16*)sora_hip_malloc(64 * 4);
    uint8_t* d_soft = (uint8_t*)sora_hip_malloc(48 * NB); uint8_t* d_de = (uint8_t*)sora_hip_malloc(48 * NB);
    uint8_t* d_frame = (uint8_t*)sora_hip_malloc(16 + (size_t)nsym * 48 * NB + 64); uint8_t* d_out = (uint8_t*)sora_hip_malloc(4096);
    if (!d_cap || !d_ctx || !d_state || !d_one || !d_eq || !d_trk || !d_soft || !d_de || !d_frame || !d_out) { fprintf(stderr, "device memory: %s\n", sora_hip_last_error()); return 1; }
    sora_track11a_state st; memset(&st, 0, sizeof(st));                // the reset state of CF_PhaseCompensate / CF_PilotTrack (pilot.hpp:143-152)
    st.symbol_count = 127; for (int k = 0; k < 64; k++) st.comp[k].re = 0x7fff;
    const uint32_t one[2] = { 0u, 1u };                                 // d_first = 0, d_nsym = 1: one symbol per burst
    if (sora_hip_memcpy_h2d(d_cap, cap.data(), cap.size() * sizeof(sora_complex16)) != SORA_OK || sora_hip_memcpy_h2d(d_state, &st, sizeof(st)) != SORA_OK ||
        sora_hip_memcpy_h2d(d_one, one, sizeof(one)) != SORA_OK) return 1;

    CF_Error ctx;
    THip11aLTS<CF_Error> lts(ctx, d_ctx);
    // SIGNAL branch (sink first)
    TDrop<CF_Error> drop(ctx);
    THip11aPilotTrack<1, CF_Error, decltype(drop)> sig_track(ctx, &drop, d_one, d_one + 1, d_state, d_trk);
    THip11aSymFront<1, CF_Error, decltype(sig_track)> sig_front(ctx, &sig_track, d_ctx, d_eq);
    // data branch
    THip11aViterbi<NB, CF_Error, TDrop<CF_Error>> vit(ctx, nullptr, d_frame, d_out);
    THip11aDeinterleave<NB, 1, CF_Error, decltype(vit)> deint(ctx, &vit, d_de);
    THip11aDemap<NB, 1, CF_Error, decltype(deint)> demap(ctx, &deint, d_soft);
    THip11aPilotTrack<1, CF_Error, decltype(demap)> track(ctx, &demap, d_one, d_one + 1, d_state, d_trk);
    THip11aSymFront<1, CF_Error, decltype(track)> front(ctx, &track, d_ctx, d_eq);
    vit.SetFrame(length, code_rate, nsym);

    DevicePin<sora_complex16, 144> lts_pin(d_cap + start); lts_pin.append();
    if (!lts.Process(lts_pin)) { fprintf(stderr, "T11aLTS failed: %08x\n", ctx.error_code); return 1; }
    DevicePin<sora_complex16, 80> sym_pin;
    sym_pin.bind(d_cap + start + 144); sym_pin.append();
    if (!sig_front.Process(sym_pin)) { fprintf(stderr, "SIGNAL branch failed: %08x (%s)\n", ctx.error_code, sora_hip_last_error()); return 1; }
    for (uint32_t k = 1; k <= nsym; k++) {
        sym_pin.bind(d_cap + start + 144 + 80 * (size_t)k); sym_pin.append();
        if (!front.Process(sym_pin)) { fprintf(stderr, "data branch failed at symbol %u: %08x (%s)\n", k, ctx.error_code, sora_hip_last_error()); return 1; }
    }
    front.Flush();
    if (!vit.decoded()) { fprintf(stderr, "the Viterbi brick did not fire\n"); return 1; }
    std::vector<uint8_t> out((size_t)length + 2);
    if (sora_hip_stream_synchronize(nullptr) != SORA_OK || sora_hip_memcpy_d2h(out.data(), vit.output(), out.size()) != SORA_OK) return 1;
    FILE* fo = fopen(outp, "wb");
    if (!fo || fwrite(out.data(), 1, out.size(), fo) != out.size()) return 1;
    fclose(fo);
    printf("frame chain: %u symbols, %zu decoded bytes\n", nsym, out.size());
    return 0;
}

int main(int argc, char** argv)
{
    if (argc != 8) { fprintf(stderr, "usage: %s <capture20.bin> <start> <nsym> <n_bpsc> <code_rate> <frame_length> <out.bin>\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<sora_complex16> cap((size_t)bytes / sizeof(sora_complex16));
    if (fread(cap.data(), sizeof(sora_complex16), cap.size(), f) != cap.size()) return 1;
    fclose(f);
    const size_t start = (size_t)atol(argv[2]); const uint32_t nsym = (uint32_t)atol(argv[3]); const int cr = atoi(argv[5]); const uint16_t len = (uint16_t)atoi(argv[6]);
    if (start + 144 + 80 * ((size_t)nsym + 1) > cap.size()) { fprintf(stderr, "frame reaches past the capture\n"); return 2; }
    switch (atoi(argv[4])) {
    case 1: return run<1>(cap, start, nsym, cr, len, argv[7]);
    case 2: return run<2>(cap, start, nsym, cr, len, argv[7]);
    case 4: return run<4>(cap, start, nsym, cr, len, argv[7]);
    case 6: return run<6>(cap, start, nsym, cr, len, argv[7]);
    }
    return 2;
}
