/* shard11a.c -- a plain-C multi-GPU host for libsora_hip.so: BASELINE config 5 ("N concurrent 20/40 MHz captures sharded across the
 * GPUs of one node") without Python.  One process per GPU; every process loads the same dump, takes its block of the `--captures N`
 * copies of it (sora_shard_partition: captures are the shard, fb11ademod_config.hpp:68-95 -- no state crosses captures), runs the
 * receive path on its own GPU and joins the others in ONE exchange, the gather of the result rows and MPDUs
 * (sora_shard_gather_results_mpdu = three ncclAllGather over RCCL/xGMI: counts, rows, dense MPDU blocks).  Rank 0 creates the RCCL id and leaves it in --id-file; the others wait for the file.
 * Build: gcc -std=c11 -Iinclude examples/shard11a.c -Lsora_amd/lib -lsora_hip -Wl,-rpath,$PWD/sora_amd/lib -o shard11a
 * Usage: for r in 0..W-1:  shard11a <file.dmp> [--raw14] [--rate 40|20] --captures N --world W --rank r --id-file /tmp/sora.id &
 *        (device = rank; one node)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "sora_hip.h"

static long load_file(const char* path, unsigned char** out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -2;
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* buf = (unsigned char*)malloc((size_t)bytes + 16);
    if (!buf || fread(buf, 1, (size_t)bytes, f) != (size_t)bytes) { fclose(f); free(buf); return -1; }
    fclose(f);
    *out = buf;
    return bytes;
}

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, " (%s)\n", sora_hip_last_error()); return 1; } while (0)

int main(int argc, char** argv)
{
    const char* path = NULL; const char* idfile = NULL; int raw14 = 0, world = 1, rank = 0; unsigned rate = 40; size_t ncap_total = 8;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--raw14")) raw14 = 1;
        else if (!strcmp(argv[i], "--rate") && i + 1 < argc) rate = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--captures") && i + 1 < argc) ncap_total = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "--world") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc) idfile = argv[++i];
        else path = argv[i];
    }
    if (!path || !idfile || world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "usage: %s <file.dmp> [--raw14] [--rate 40|20] --captures N --world W --rank r --id-file F\n", argv[0]); return 2; }

    /* ---- the RCCL id: made on rank 0, carried by a file */
    uint8_t id[SORA_SHARD_ID_BYTES];
    if (rank == 0) {
        if (sora_shard_unique_id(id) != SORA_OK) DIE("sora_shard_unique_id");
        char tmp[1024]; snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
        FILE* f = fopen(tmp, "wb"); if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) DIE("cannot write %s", tmp);
        fclose(f); rename(tmp, idfile);
    } else {
        FILE* f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); tries++) { struct timespec ts = { 0, 100 * 1000 * 1000 }; nanosleep(&ts, NULL); }
        if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) DIE("cannot read %s", idfile);
        fclose(f);
    }
    sora_shard_t* sh = NULL;
    if (sora_shard_create(id, world, rank, /*device*/ rank, &sh) != SORA_OK) DIE("sora_shard_create");

    /* ---- this rank's captures: block `rank` of ncap_total copies of the dump, numbered globally */
    size_t first = 0, mine = 0;
    sora_shard_partition(ncap_total, world, rank, &first, &mine);
    unsigned char* file = NULL;
    long bytes = load_file(path, &file);
    if (bytes <= 0) { fprintf(stderr, "Failed to load input file.\n"); return 1; }
    const unsigned flags = SORA_INGEST_RXBLOCK | (raw14 ? SORA_INGEST_RAW14 : 0u);
    size_t n = sora_hip_ingest_count((size_t)bytes, flags);
    n -= n % (rate == 40 ? 28 : 14);
    const size_t stride = (n + 63) / 64 * 64;
    void* d_file = sora_hip_malloc((size_t)bytes + 16);
    sora_complex16* d_iq = (sora_complex16*)sora_hip_malloc((stride * (mine ? mine : 1) + 64) * sizeof(sora_complex16));
    if (!d_file || !d_iq || sora_hip_memcpy_h2d(d_file, file, (size_t)bytes) != SORA_OK) DIE("device memory");
    sora_capture_desc* caps = (sora_capture_desc*)calloc(mine ? mine : 1, sizeof(*caps));
    for (size_t c = 0; c < mine; c++) {
        size_t got = 0;
        if (sora_hip_ingest(d_file, (size_t)bytes, flags, d_iq + c * stride, stride, &got, NULL) != SORA_OK) DIE("sora_hip_ingest");
        caps[c].offset = c * stride; caps[c].nsamples = (uint32_t)n; caps[c].capture_id = (uint32_t)(first + c);
    }
    if (sora_hip_stream_synchronize(NULL) != SORA_OK) DIE("synchronize");

    const uint32_t max_frames = 8;
    sora_rx_cfg cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device = rank; cfg.sample_rate_mhz = rate; cfg.max_captures = (uint32_t)(mine ? mine : 1);
    cfg.max_total_samples = (uint64_t)stride * (mine ? mine : 1); cfg.max_frames_per_capture = max_frames;
    sora_rx_t* rx = NULL;
    if (sora_rx_create(&cfg, &rx) != SORA_OK) DIE("sora_rx_create");
    if (sora_rx_process_dev(rx, d_iq, caps, mine) != SORA_OK) DIE("sora_rx_process_dev");

    /* ---- the one exchange: every rank ends up with every rank's rows */
    const size_t per_rank = ((ncap_total + (size_t)world - 1) / (size_t)world) * max_frames;
    sora_frame_result* all = (sora_frame_result*)calloc(per_rank * (size_t)world, sizeof(*all));
    uint32_t* counts = (uint32_t*)calloc((size_t)world, sizeof(uint32_t));
    size_t total = 0, mpdu_total = 0;
    const size_t mpdu_per_rank = per_rank * 2504;                                      /* rows x the longest MPDU the 802.11a graph accepts */
    uint8_t* mpdus = (uint8_t*)malloc(mpdu_per_rank * (size_t)world);
    if (!mpdus) DIE("host memory");
    if (sora_shard_gather_results_mpdu(sh, rx, 0, per_rank, all, counts, &total, mpdu_per_rank, mpdus, &mpdu_total) != SORA_OK) DIE("sora_shard_gather_results_mpdu");
    if (rank == 0) {
        size_t good = 0;
        uint32_t fnv = 2166136261u;                                                     /* over every gathered MPDU byte, in row order */
        for (size_t i = 0; i < total; i++) {
            good += all[i].error_code == SORA_E_FRAME_OK;
            if (all[i].error_code == SORA_E_FRAME_OK || all[i].error_code == SORA_E_CRC32_FAIL)
                for (uint32_t k = 0; k < all[i].length; k++) fnv = (fnv ^ mpdus[all[i].mpdu_offset + k]) * 16777619u;
        }
        printf("world %d: %zu captures, %zu frames gathered (", world, ncap_total, total);
        for (int r = 0; r < world; r++) printf("%s%u", r ? " + " : "", counts[r]);
        printf("), good %zu / bad %zu; first: capture %u %u kbps length %u FCS %08x, last: capture %u; %zu MPDU bytes gathered, fnv1a %08x\n", good, total - good,
               total ? all[0].capture_id : 0, total ? all[0].rate_kbps : 0, total ? all[0].length : 0, total ? all[0].crc32 : 0,
               total ? all[total - 1].capture_id : 0, mpdu_total, fnv);
    }
    sora_rx_destroy(rx); sora_shard_destroy(sh);
    sora_hip_free(d_file); sora_hip_free(d_iq);
    free(all); free(counts); free(caps); free(file); free(mpdus);
    if (rank == 0) remove(idfile);
    return 0;
}
