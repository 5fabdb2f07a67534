/* demod11b.c -- a plain-C host for the 802.11b receive graph of libsora_hip.so, shaped like the reference's offline harness
 *   demod11 --802.11b.brick -d -f <dump>      (kernel/bb/demod11/main.cpp:59-229, fb11b_demod.cpp:79-117)
 * Loads a Sora RX_BLOCK dump recorded at 44 MHz, de-frames it on the GPU (sora_hip_ingest) and hands it to the 11b graph
 * as ONE capture; prints what MAC11b_Receive would have reported.
 * Build: gcc -std=c11 -Iinclude examples/demod11b.c -Lsora_amd/lib -lsora_hip -Wl,-rpath,$PWD/sora_amd/lib -o demod11b
 * Usage: demod11b <file.dmp> [--raw14] [--out mpdu.bin] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sora_hip.h"

int main(int argc, char** argv)
{
    const char* path = NULL; const char* outp = NULL; int raw14 = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--raw14")) raw14 = 1;
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outp = argv[++i];
        else path = argv[i];
    }
    if (!path) { fprintf(stderr, "usage: %s <file.dmp> [--raw14] [--out mpdu.bin]\n", argv[0]); return 2; }
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "Failed to load input file.\n"); return 1; }
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* file = (unsigned char*)malloc((size_t)bytes + 16);
    if (!file || fread(file, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "Failed to load input file.\n"); return 1; }
    fclose(f);

    const unsigned flags = SORA_INGEST_RXBLOCK | (raw14 ? SORA_INGEST_RAW14 : 0u);      /* LoadSoraDumpFile on the device */
    size_t n = sora_hip_ingest_count((size_t)bytes, flags);
    n -= n % 28;                                                                         /* whole source bursts */
    void* d_file = sora_hip_malloc((size_t)bytes + 16);
    sora_complex16* d_iq = (sora_complex16*)sora_hip_malloc((n + 64) * sizeof(sora_complex16));
    size_t got = 0;
    int rc = (d_file && d_iq) ? sora_hip_memcpy_h2d(d_file, file, (size_t)bytes) : SORA_ERR_FAILED;
    if (rc == SORA_OK) rc = sora_hip_ingest(d_file, (size_t)bytes, flags, d_iq, n + 64, &got, NULL);
    if (rc == SORA_OK) rc = sora_hip_stream_synchronize(NULL);
    if (rc != SORA_OK || n == 0) { fprintf(stderr, "ingest: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    printf("Demodulate 11b on MI355X: %zu samples @44 MHz\n", n);

    sora_rx_cfg cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device = 0; cfg.sample_rate_mhz = 44; cfg.max_captures = 1;
    cfg.max_total_samples = (uint64_t)n; cfg.max_frames_per_capture = 256;
    sora_rx11b_t* rx = NULL;
    rc = sora_rx11b_create(&cfg, &rx);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx11b_create: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    sora_capture_desc cap; cap.offset = 0; cap.nsamples = (uint32_t)n; cap.capture_id = 0;
    rc = sora_rx11b_process_dev(rx, d_iq, &cap, 1);
    sora_frame_result* res = (sora_frame_result*)malloc(256 * sizeof(*res));
    uint8_t* mpdu = (uint8_t*)malloc(256 * 4096);
    size_t nres = 0;
    if (rc == SORA_OK) rc = sora_rx11b_results(rx, res, 256, &nres, mpdu, 256 * 4096);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx11b: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    size_t good = 0, frames = 0; int wrote = 0;
    for (size_t i = 0; i < nres; i++) {
        const sora_frame_result* r = &res[i];
        if (r->error_code == SORA_E_FRAME_OK || r->error_code == (uint32_t)SORA_E_CRC32_FAIL) {
            frames++;
            printf("[frame %zu] ends at sample %u  %u kbps  length %u  FCS %06x..  %s\n", frames, r->end_sample, r->rate_kbps, r->length,
                   r->crc32 & 0xFFFFFFu, r->error_code == SORA_E_FRAME_OK ? "FRAME_OK" : "CRC32_FAIL");
            if (r->error_code == SORA_E_FRAME_OK) {
                good++;
                if (outp && !wrote) { FILE* fo = fopen(outp, "wb"); if (fo) { fwrite(mpdu + r->mpdu_offset, 1, r->length, fo); fclose(fo); wrote = 1; } }
            }
        } else printf("err = %08X at sample %u\n", r->error_code, r->end_sample);      /* what MAC11b_Receive prints */
    }
    printf("good %zu / bad %zu\n", good, frames - good);
    sora_rx11b_destroy(rx);
    sora_hip_free(d_file); sora_hip_free(d_iq);
    free(res); free(mpdu); free(file);
    return 0;
}
