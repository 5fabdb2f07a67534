/* demod11a.c -- a plain-C host for libsora_hip.so, shaped like the reference's offline harness
 *   demod11 --802.11a.brick -d -f <dump> -p 40      (kernel/bb/demod11/main.cpp:59-229, fb11a_demod.cpp:88-120)
 * Loads a Sora RX_BLOCK dump (kernel/brick/inc/brickutil.h:20-58: 16-byte descriptor + 28 COMPLEX16 per 128-byte
 * block), hands it to the GPU receive path as ONE capture and prints what RxThread would have reported.
 * Build: gcc -std=c11 -Iinclude examples/demod11a.c -Lsora_amd/lib -lsora_hip -Wl,-rpath,$PWD/sora_amd/lib -o demod11a
 * Usage: demod11a <file.dmp> [--raw14] [--rate 44|40|20] [--out mpdu.bin]     (--rate = sampling rate of the dump)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sora_hip.h"

/* The dump goes to the GPU as it is on disk; de-framing, the sign fix, the optional 44 -> 40 MHz resampler and the
 * optional /2 decimation run there (sora_hip_ingest).  Returns the raw bytes (malloc'd). */
static long load_file(const char* path, unsigned char** out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -2;                                   /* BK_ERROR_FILE_NOT_FOUND */
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* buf = (unsigned char*)malloc((size_t)bytes + 16);
    if (!buf || fread(buf, 1, (size_t)bytes, f) != (size_t)bytes) { fclose(f); free(buf); return -1; }
    fclose(f);
    *out = buf;
    return bytes;
}

int main(int argc, char** argv)
{
    const char* path = NULL; const char* outp = NULL; int raw14 = 0; unsigned rate = 40;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--raw14")) raw14 = 1;
        else if (!strcmp(argv[i], "--rate") && i + 1 < argc) rate = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outp = argv[++i];
        else path = argv[i];
    }
    if (!path) { fprintf(stderr, "usage: %s <file.dmp> [--raw14] [--rate 44|40|20] [--out mpdu.bin]\n", argv[0]); return 2; }
    unsigned char* file = NULL;
    long bytes = load_file(path, &file);
    if (bytes <= 0) { fprintf(stderr, "Failed to load input file.\n"); return 1; }
    /* LoadSoraDumpFile + (TDownSample44_40) on the device */
    unsigned flags = SORA_INGEST_RXBLOCK | (raw14 ? SORA_INGEST_RAW14 : 0u) | (rate == 44 ? SORA_INGEST_44TO40 : 0u);
    if (rate == 44) rate = 40;
    size_t n = sora_hip_ingest_count((size_t)bytes, flags);
    n -= n % (rate == 40 ? 28 : 14);                     /* whole source bursts (memsource.hpp:87-114) */
    void* d_file = sora_hip_malloc((size_t)bytes + 16);
    sora_complex16* d_iq = (sora_complex16*)sora_hip_malloc((n + 64) * sizeof(sora_complex16));
    if (!d_file || !d_iq || sora_hip_memcpy_h2d(d_file, file, (size_t)bytes) != SORA_OK) { fprintf(stderr, "device memory: %s\n", sora_hip_last_error()); return 1; }
    size_t got = 0;
    int rc = sora_hip_ingest(d_file, (size_t)bytes, flags, d_iq, n + 64, &got, NULL);
    if (rc == SORA_OK) rc = sora_hip_stream_synchronize(NULL);   /* the handle's streams do not follow the null stream */
    if (rc != SORA_OK || n == 0) { fprintf(stderr, "sora_hip_ingest: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    printf("Demodulate 11a on MI355X: %zu samples @%u MHz\n", n, rate);

    sora_rx_cfg cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device = 0; cfg.sample_rate_mhz = rate; cfg.max_captures = 1;
    cfg.max_total_samples = (uint64_t)n; cfg.max_frames_per_capture = 64;
    sora_rx_t* rx = NULL;
    rc = sora_rx_create(&cfg, &rx);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx_create: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    sora_capture_desc cap; cap.offset = 0; cap.nsamples = (uint32_t)n; cap.capture_id = 0;
    rc = sora_rx_process_dev(rx, d_iq, &cap, 1);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx_process_dev: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    sora_frame_result res[64]; size_t nres = 0;
    uint8_t* mpdu = (uint8_t*)malloc(64 * 2504);
    rc = sora_rx_results(rx, res, 64, &nres, mpdu, 64 * 2504);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx_results: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    size_t good = 0;
    for (size_t i = 0; i < nres; i++) {
        const sora_frame_result* r = &res[i];
        printf("[frame %zu] samples %u..%u  %u kbps  length %u  symbols %u  FCS %08x  %s\n", i, r->start_sample, r->end_sample,
               r->rate_kbps, r->length, r->nsym, r->crc32,
               r->error_code == SORA_E_FRAME_OK ? "FRAME_OK" : (r->error_code == (uint32_t)SORA_E_CRC32_FAIL ? "CRC32_FAIL" : "PLCP_HEADER_FAIL"));
        if (r->error_code == SORA_E_FRAME_OK) good++;
    }
    printf("good %zu / bad %zu\n", good, nres - good);
    if (outp && nres && res[0].error_code == SORA_E_FRAME_OK) {
        FILE* fo = fopen(outp, "wb");
        if (fo) { fwrite(mpdu + res[0].mpdu_offset, 1, res[0].length, fo); fclose(fo); }
    }
    sora_rx_destroy(rx);
    sora_hip_free(d_file); sora_hip_free(d_iq);
    free(mpdu); free(file);
    return 0;
}
