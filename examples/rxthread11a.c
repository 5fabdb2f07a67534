/* rxthread11a.c -- a plain-C host that keeps several receive calls in flight and takes their results in the order they finish:
 * RxThread's loop (kernel/bb/demod11/fb11a_demod.cpp:37-71: wait for the graph, look at the frame, hand the MPDU on) per CALL,
 * with the decoder running beside reception as ViterbiThread does (fb11a_demod.cpp:117-120) -- here: calls on the handle's pipelines.
 * The dump is cut into `pieces` captures of equal length (whole source bursts); every call decodes all of them; `calls` calls are
 * made with `depth` in flight; each call's rows and MPDU bytes are delivered to page-locked host memory (sora_rx_deliver_async) and
 * collected with sora_rx_wait_any, and every delivered table is compared with the first call's.
 * Build: gcc -std=c11 -Iinclude examples/rxthread11a.c -Lsora_amd/lib -lsora_hip -Wl,-rpath,$PWD/sora_amd/lib -o rxthread11a
 * Usage: rxthread11a <file.dmp> [--raw14] [--calls N] [--depth D]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sora_hip.h"

enum { MAXF = 8, MAXD = 16 };
struct slot { sora_frame_result* rows; uint32_t* nrows; uint8_t* mpdu; int ticket; };

int main(int argc, char** argv)
{
    const char* path = NULL; int raw14 = 0, calls = 24, depth = 4;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--raw14")) raw14 = 1;
        else if (!strcmp(argv[i], "--calls") && i + 1 < argc) calls = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--depth") && i + 1 < argc) depth = atoi(argv[++i]);
        else path = argv[i];
    }
    if (!path || depth < 1 || depth > MAXD || calls < 1) { fprintf(stderr, "usage: %s <file.dmp> [--raw14] [--calls N] [--depth 1..16]\n", argv[0]); return 2; }
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "Failed to load input file.\n"); return 1; }
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* file = (unsigned char*)malloc((size_t)bytes + 16);
    if (!file || fread(file, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "Failed to load input file.\n"); return 1; }
    fclose(f);
    const unsigned flags = SORA_INGEST_RXBLOCK | (raw14 ? SORA_INGEST_RAW14 : 0u);
    size_t n = sora_hip_ingest_count((size_t)bytes, flags);
    n -= n % 28;
    void* d_file = sora_hip_malloc((size_t)bytes + 16);
    sora_complex16* d_iq = (sora_complex16*)sora_hip_malloc((n + 64) * sizeof(sora_complex16));
    size_t got = 0;
    if (!d_file || !d_iq || sora_hip_memcpy_h2d(d_file, file, (size_t)bytes) != SORA_OK || sora_hip_ingest(d_file, (size_t)bytes, flags, d_iq, n + 64, &got, NULL) != SORA_OK
        || sora_hip_stream_synchronize(NULL) != SORA_OK) { fprintf(stderr, "ingest: %s\n", sora_hip_last_error()); return 1; }

    sora_rx_cfg cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device = 0; cfg.sample_rate_mhz = 40; cfg.max_captures = 1; cfg.max_total_samples = (uint64_t)n; cfg.max_frames_per_capture = MAXF;
    sora_rx_t* rx = NULL;
    if (sora_rx_create(&cfg, &rx) != SORA_OK) { fprintf(stderr, "sora_rx_create: %s\n", sora_hip_last_error()); return 1; }
    if (sora_rx_set_depth(rx, depth) < 0) { fprintf(stderr, "sora_rx_set_depth: %s\n", sora_hip_last_error()); return 1; }
    sora_capture_desc cap; cap.offset = 0; cap.nsamples = (uint32_t)n; cap.capture_id = 0;

    struct slot s[MAXD]; size_t mpdu_bytes = 0;
    sora_frame_result first_rows[MAXF]; uint32_t first_n = 0; uint8_t* first_mpdu = NULL;
    int in_flight = 0, submitted = 0, collected = 0, differing = 0, out_of_order = 0, last_done = 0;
    memset(s, 0, sizeof(s));
    while (collected < calls) {
        if (submitted < calls && in_flight < depth) {
            const int rc = sora_rx_process_dev(rx, d_iq, &cap, 1);                /* returns at once: the call is enqueued */
            if (rc != SORA_OK) { fprintf(stderr, "sora_rx_process_dev: %d (%s)\n", rc, sora_hip_last_error()); return 1; }
            const int t = sora_rx_ticket(rx);                                     /* the call's ticket */
            if (!mpdu_bytes) {                                                    /* (the same geometry every call) */
                mpdu_bytes = sora_rx_mpdu_bytes(rx, t);
                for (int k = 0; k < depth; k++) {
                    s[k].rows = (sora_frame_result*)sora_hip_host_alloc(sizeof(sora_frame_result) * MAXF);
                    s[k].nrows = (uint32_t*)sora_hip_host_alloc(sizeof(uint32_t)); s[k].mpdu = (uint8_t*)sora_hip_host_alloc(mpdu_bytes);
                    if (!s[k].rows || !s[k].nrows || !s[k].mpdu) { fprintf(stderr, "sora_hip_host_alloc: %s\n", sora_hip_last_error()); return 1; }
                }
                first_mpdu = (uint8_t*)malloc(mpdu_bytes);
            }
            int k = 0; while (s[k].ticket) k++;                                   /* a free buffer: at most `depth` are taken */
            if (sora_rx_deliver_async(rx, t, s[k].rows, MAXF, s[k].nrows, s[k].mpdu, mpdu_bytes) != SORA_OK) { fprintf(stderr, "sora_rx_deliver_async: %s\n", sora_hip_last_error()); return 1; }
            s[k].ticket = t; in_flight++; submitted++;
            continue;
        }
        int done = 0;
        if (sora_rx_wait_any(rx, &done) != SORA_OK) { fprintf(stderr, "sora_rx_wait_any: %s\n", sora_hip_last_error()); return 1; }
        int k = 0; while (k < depth && s[k].ticket != done) k++;
        if (k == depth) { fprintf(stderr, "sora_rx_wait_any returned ticket %d, which is not in flight\n", done); return 1; }
        if (done < last_done) out_of_order++;
        last_done = done;
        const uint32_t nr = *s[k].nrows;
        if (collected == 0) {                                                     /* what the MAC would be handed: the first call's frames */
            first_n = nr; memcpy(first_rows, s[k].rows, sizeof(sora_frame_result) * (nr < MAXF ? nr : MAXF)); memcpy(first_mpdu, s[k].mpdu, mpdu_bytes);
            for (uint32_t i = 0; i < nr && i < MAXF; i++) {
                const sora_frame_result* r = &s[k].rows[i];
                printf("[frame %u] samples %u..%u  %u kbps  length %u  FCS %08x  %s\n", i, r->start_sample, r->end_sample, r->rate_kbps, r->length, r->crc32,
                       r->error_code == SORA_E_FRAME_OK ? "FRAME_OK" : (r->error_code == (uint32_t)SORA_E_CRC32_FAIL ? "CRC32_FAIL" : "PLCP_HEADER_FAIL"));
            }
        } else if (nr != first_n || memcmp(first_rows, s[k].rows, sizeof(sora_frame_result) * (nr < MAXF ? nr : MAXF)) != 0) differing++;
        else {
            for (uint32_t i = 0; i < nr && i < MAXF; i++)
                if (s[k].rows[i].error_code == SORA_E_FRAME_OK && memcmp(first_mpdu + s[k].rows[i].mpdu_offset, s[k].mpdu + s[k].rows[i].mpdu_offset, s[k].rows[i].length) != 0) { differing++; break; }
        }
        s[k].ticket = 0; in_flight--; collected++;
    }
    int none = 0;
    const int rc_idle = sora_rx_wait_any(rx, &none);                              /* nothing is in flight any more: refused, not blocked */
    printf("calls %d, in flight %d, collected %d, tables differing from the first call's %d, idle wait_any %s\n", calls, depth, collected, differing, rc_idle == SORA_OK ? "returned a ticket" : "refused");
    (void)out_of_order;
    for (int k = 0; k < depth; k++) { sora_hip_host_free(s[k].rows); sora_hip_host_free(s[k].nrows); sora_hip_host_free(s[k].mpdu); }
    free(first_mpdu); free(file);
    sora_rx_destroy(rx); sora_hip_free(d_iq); sora_hip_free(d_file);
    return differing ? 3 : 0;
}
