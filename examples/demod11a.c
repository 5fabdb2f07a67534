/* demod11a.c -- a plain-C host for libsora_hip.so, shaped like the reference's offline harness
 *   demod11 --802.11a.brick -d -f <dump> -p 40      (kernel/bb/demod11/main.cpp:59-229, fb11a_demod.cpp:88-120)
 * Loads a Sora RX_BLOCK dump (kernel/brick/inc/brickutil.h:20-58: 16-byte descriptor + 28 COMPLEX16 per 128-byte
 * block), hands it to the GPU receive path as ONE capture and prints what RxThread would have reported.
 * Build: gcc -std=c11 -Iinclude examples/demod11a.c -Lsora_amd/lib -lsora_hip -Wl,-rpath,$PWD/sora_amd/lib -o demod11a
 * Usage: demod11a <file.dmp> [--raw14] [--rate 40|20] [--out mpdu.bin]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sora_hip.h"

static long load_dump(const char* path, sora_complex16** out, int raw14)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -2;                                   /* BK_ERROR_FILE_NOT_FOUND */
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    long nblk = bytes / 128, n = 0;
    sora_complex16* iq = (sora_complex16*)malloc((size_t)(nblk * 28 + 28) * sizeof(sora_complex16));
    unsigned char blk[128];
    while (fread(blk, 1, 128, f) == 128) {
        memcpy(iq + n, blk + 16, 28 * sizeof(sora_complex16));
        if (raw14)                                       /* 14-bit two's complement, zero-extended */
            for (int i = 0; i < 28; i++) {
                iq[n + i].re = (int16_t)(uint16_t)((uint16_t)iq[n + i].re << 2);
                iq[n + i].im = (int16_t)(uint16_t)((uint16_t)iq[n + i].im << 2);
            }
        n += 28;
    }
    fclose(f);
    *out = iq;
    return n;
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
    if (!path) { fprintf(stderr, "usage: %s <file.dmp> [--raw14] [--rate 40|20] [--out mpdu.bin]\n", argv[0]); return 2; }
    sora_complex16* iq = NULL;
    long n = load_dump(path, &iq, raw14);
    if (n <= 0) { fprintf(stderr, "Failed to load input file.\n"); return 1; }
    printf("Demodulate 11a on MI355X: %ld samples @%u MHz\n", n, rate);

    sora_rx_cfg cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device = 0; cfg.sample_rate_mhz = rate; cfg.max_captures = 1;
    cfg.max_total_samples = (uint64_t)n; cfg.max_frames_per_capture = 64;
    sora_rx_t* rx = NULL;
    int rc = sora_rx_create(&cfg, &rx);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx_create: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
    sora_capture_desc cap; cap.offset = 0; cap.nsamples = (uint32_t)n; cap.capture_id = 0;
    rc = sora_rx_process(rx, iq, (size_t)n, &cap, 1);
    if (rc != SORA_OK) { fprintf(stderr, "sora_rx_process: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
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
    free(mpdu); free(iq);
    return 0;
}
