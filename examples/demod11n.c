/* demod11n.c -- a plain-C host for the 802.11n 2x2 receive graph of libsora_hip.so, shaped like the reference's offline harness
 *   demod11 --802.11n.brick -d -f <prefix>     (kernel/bb/demod11/main.cpp:59-229, fb11n_demod.cpp:92-140: loads <prefix>_0.dmp
 *   and <prefix>_1.dmp, one Sora RX_BLOCK dump per RX chain, 40 MHz)
 * De-frames both dumps on the GPU (sora_hip_ingest) and hands them to the 11n graph as ONE two-chain capture; prints what RxThread
 * would have reported.
 * Build: gcc -std=c11 -Iinclude examples/demod11n.c -Lsora_amd/lib -lsora_hip -Wl,-rpath,$PWD/sora_amd/lib -o demod11n
 * Usage: demod11n <prefix> [--raw14] [--out mpdu.bin] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sora_hip.h"

static sora_complex16* load_chain(const char* prefix, int chain, int raw14, size_t* n_out)
{
    char path[1024]; snprintf(path, sizeof(path), "%s_%d.dmp", prefix, chain);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "Failed to load input file.\n"); return NULL; }
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* file = (unsigned char*)malloc((size_t)bytes + 16);
    if (!file || fread(file, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "Failed to load input file.\n"); fclose(f); free(file); return NULL; }
    fclose(f);
    const unsigned flags = SORA_INGEST_RXBLOCK | (raw14 ? SORA_INGEST_RAW14 : 0u);      /* LoadSoraDumpFile on the device */
    size_t n = sora_hip_ingest_count((size_t)bytes, flags);
    void* d_file = sora_hip_malloc((size_t)bytes + 16);
    sora_complex16* d_iq = (sora_complex16*)sora_hip_malloc((n + 64) * sizeof(sora_complex16));
    size_t got = 0;
    int rc = (d_file && d_iq) ? sora_hip_memcpy_h2d(d_file, file, (size_t)bytes) : SORA_ERR_FAILED;
    if (rc == SORA_OK) rc = sora_hip_ingest(d_file, (size_t)bytes, flags, d_iq, n + 64, &got, NULL);
    if (rc == SORA_OK) rc = sora_hip_stream_synchronize(NULL);
    sora_hip_free(d_file); free(file);
    if (rc != SORA_OK) { fprintf(stderr, "ingest: %d (%s)\n", rc, sora_hip_last_error()); sora_hip_free(d_iq); return NULL; }
    *n_out = n;
    return d_iq;
}

int main(int argc, char** argv)
{
    const char* prefix = NULL; const char* outp = NULL; int raw14 = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--raw14")) raw14 = 1;
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outp = argv[++i];
        else prefix = argv[i];
    }
    if (!prefix) { fprintf(stderr, "usage: %s <prefix> [--raw14] [--out mpdu.bin]   (reads <prefix>_0.dmp and <prefix>_1.dmp)\n", argv[0]); return 2; }
    size_t n0 = 0, n1 = 0;
    sora_complex16* d_iq0 = load_chain(prefix, 0, raw14, &n0);
    sora_complex16* d_iq1 = d_iq0 ? load_chain(prefix, 1, raw14, &n1) : NULL;
    if (!d_iq0 || !d_iq1) return 1;
    size_t n = n0 < n1 ? n0 : n1;
    n -= n % 28;                                                                         /* whole source bursts */
    if (n == 0) { fprintf(stderr, "empty capture\n"); return 1; }
    printf("Demodulate 11n with brick demod graph on MI355X: %zu samples @40 MHz per chain\n", n);

    sora_rx_cfg cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device = 0; cfg.sample_rate_mhz = 40; cfg.max_captures = 1;
    cfg.max_total_samples = (uint64_t)n; cfg.max_frames_per_capture = 256;
    sora_rx11n_t* rx = NULL;
    int rc = sora_rx11n_create(&cfg, &rx);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx11n_create: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    sora_capture_desc cap; cap.offset = 0; cap.nsamples = (uint32_t)n; cap.capture_id = 0;
    rc = sora_rx11n_process_dev(rx, d_iq0, d_iq1, &cap, 1);
    sora_frame_result* res = (sora_frame_result*)malloc(256 * sizeof(*res));
    uint8_t* mpdu = (uint8_t*)malloc(256 * 4096);
    size_t nres = 0;
    if (rc == SORA_OK) rc = sora_rx11n_results(rx, res, 256, &nres, mpdu, 256 * 4096);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx11n: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    size_t good = 0, frames = 0; int wrote = 0;
    for (size_t i = 0; i < nres; i++) {
        const sora_frame_result* r = &res[i];
        if (r->error_code == SORA_E_FRAME_OK || r->error_code == (uint32_t)SORA_E_CRC32_FAIL) {
            frames++;
            printf("[frame %zu] seen at sample %u  MCS %u  length %u  FCS %08x  %s\n", frames, r->end_sample, r->rate_kbps, r->length,
                   r->crc32, r->error_code == SORA_E_FRAME_OK ? "FRAME_OK" : "CRC32_FAIL");
            if (r->error_code == SORA_E_FRAME_OK) {
                good++;
                if (outp && !wrote) { FILE* fo = fopen(outp, "wb"); if (fo) { fwrite(mpdu + r->mpdu_offset, 1, r->length, fo); fclose(fo); wrote = 1; } }
            }
        } else printf("err = %08X at sample %u\n", r->error_code, r->end_sample);      /* RxThread's printf (fb11n_demod.cpp:58) */
    }
    printf("good %zu / bad %zu\n", good, frames - good);
    sora_rx11n_destroy(rx);
    sora_hip_free(d_iq0); sora_hip_free(d_iq1);
    free(res); free(mpdu);
    return 0;
}
