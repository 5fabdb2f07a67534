/* winmodel.c -- design model (a tool, not product, not oracle) of the WINDOW-PARALLEL decode of the K=7 trellis with exact verification.
 *
 * The reference decoder (T11aViterbi<..,256,24> over TViterbiCore, viterbi.hpp:148-235, viterbicore.h:293-555) is one serial chain per frame:
 * 8-bit wrapping metrics, decision in the LSB, normalisation whenever (step & 7) == 0 after a puncture group, a trace-back every 256 output
 * bits.  The chain's state at a normalisation point is the 64 seven-bit metrics (the LSBs are overwritten by the next step).  A UNIT decodes
 * windows k0 .. k1-1 of a frame: it starts W steps before its VERIFY POINT b = floor24(256 k0) from all-zero metrics, records its vector at
 * b ("spec"), decodes its windows, and records its vector at the next unit's verify point ("end").  spec(unit u) == end(unit u - 1) for every
 * u >= 1 proves, by induction from the first unit (which starts at step 0 from the reference's initial metrics), that every unit was on the
 * reference's own trajectory from its verify point on: all decisions a window's walk reads are then the reference's.  A frame with any
 * mismatch is decoded again by the serial kernel.  This file measures how often that happens, by warm-up length and noise level.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int parity7(int v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1; }
static uint8_t cA[64][2], cB[64][2];
static int inited;
static void init(void)
{
    if (inited) return;
    for (int n = 0; n < 64; n++) for (int br = 0; br < 2; br++) { int r = (br << 6) | n; cA[n][br] = (uint8_t)parity7(r & 0155); cB[n][br] = (uint8_t)parity7(r & 0117); }
    inited = 1;
}
static inline uint8_t bm(uint8_t s, int bit) { return (uint8_t)(bit ? 2 * (7 - s) : 2 * s); }
static uint64_t acs(uint8_t m[64], int which, uint8_t sa, uint8_t sb)
{
    uint8_t nm[64]; uint64_t dec = 0;
    for (int n = 0; n < 64; n++) {
        uint8_t c0 = m[n >> 1], c1 = m[32 + (n >> 1)];
        if (which != 2) { c0 = (uint8_t)(c0 + bm(sa, cA[n][0])); c1 = (uint8_t)(c1 + bm(sa, cA[n][1])); }
        if (which != 1) { c0 = (uint8_t)(c0 + bm(sb, cB[n][0])); c1 = (uint8_t)(c1 + bm(sb, cB[n][1])); }
        c0 &= 0xFE; c1 |= 0x01;
        nm[n] = c0 < c1 ? c0 : c1;
        dec |= (uint64_t)(nm[n] & 1) << n;
    }
    memcpy(m, nm, 64);
    return dec;
}
static void normalize(uint8_t m[64])
{
    uint8_t mn = 255;
    for (int n = 0; n < 64; n++) if (m[n] < mn) mn = m[n];
    mn &= 0xFE;
    for (int n = 0; n < 64; n++) m[n] = (uint8_t)(m[n] - mn);
}
static int argmin_state(const uint8_t m[64])
{
    int best = 0; uint32_t bk = 0xFFFFFFFFu;
    for (int n = 0; n < 64; n++) { uint32_t k = ((uint32_t)m[n] << 8) | ((uint32_t)n << 2); if (k < bk) { bk = k; best = n; } }
    return best;
}
/* dec[] is indexed by ABSOLUTE column; columns below `lowest` do not exist in this unit (the walk must not reach them) */
static int traceback(const uint64_t* dec, uint32_t cur, const uint8_t m[64], uint8_t* out, uint32_t bits, uint32_t lookahead, uint32_t lowest)
{
    int st = argmin_state(m);
    int pos = st | ((m[st] & 1) << 6);
    uint32_t col = cur;
    for (uint32_t i = 0; i < lookahead; i++) { col--; if (col < lowest) return -1; pos = (pos >> 1) & 0x3F; pos |= (int)((dec[col] >> pos) & 1) << 6; }
    uint8_t* po = out + (bits >> 3);
    for (uint32_t i = 0; i < bits >> 3; i++) {
        uint8_t oc = 0;
        for (int j = 0; j < 8; j++) { oc = (uint8_t)((oc << 1) | ((pos >> 6) & 1)); col--; if (col < lowest) return -1; pos = (pos >> 1) & 0x3F; pos |= (int)((dec[col] >> pos) & 1) << 6; }
        *--po = oc;
    }
    return 0;
}

/* One unit.  Steps s0 .. until window k1 - 1 has been traced (or the frame's end); metrics start from `init` (64 bytes) at step s0.
 * snap_at[i] (absolute steps, multiples of 24, ascending, 0xFFFFFFFF = none): the vector (m & 0xFE) at that step goes to snaps[i].
 * Decoded bytes go to out (absolute byte positions).  Returns the number of bytes written, < 0 on a model error. */
static int run_unit(const uint8_t* soft, uint32_t nsoft, int cr, uint32_t frame_length, uint32_t s0, const uint8_t* initm, uint32_t k0, uint32_t k1,
                    const uint32_t snap_at[2], uint8_t snaps[2][64], uint8_t* out, uint64_t* dec /* [nsoft + 8] scratch, absolute columns */)
{
    const uint32_t DEPTH = 256, LOOK = 24, PREFIX = 6;
    const int GB = cr == 0 ? 2 : cr == 2 ? 4 : 3, GS = cr == 0 ? 1 : cr == 2 ? 3 : 2;
    uint8_t m[64]; memcpy(m, initm, 64);
    uint32_t tr = s0, ob = 256 * k0; int nout = 0;
    const uint8_t* p = soft + (size_t)(s0 / GS) * GB; const uint8_t* end = soft + nsoft;
    const uint32_t tr_end = frame_length * 8 + 16 + PREFIX;
    dec[s0] = 0; for (int n = 0; n < 64; n++) dec[s0] |= (uint64_t)(m[n] & 1) << n;
    uint8_t buf[512];
    while (p < end) {
        if (cr == 0)      { dec[tr + 1] = acs(m, 0, p[0], p[1]); tr += 1; p += 2; }
        else if (cr == 2) { dec[tr + 1] = acs(m, 0, p[0], p[1]); dec[tr + 2] = acs(m, 1, p[2], 0); dec[tr + 3] = acs(m, 2, 0, p[3]); tr += 3; p += 4; }
        else              { dec[tr + 1] = acs(m, 0, p[0], p[1]); dec[tr + 2] = acs(m, 1, p[2], 0); tr += 2; p += 3; }
        if ((tr & 7) == 0) normalize(m);
        for (int i = 0; i < 2; i++) if (tr == snap_at[i]) for (int n = 0; n < 64; n++) snaps[i][n] = m[n] & 0xFE;
        uint32_t cnt = 0, look = 0;
        if (tr >= tr_end) { cnt = tr_end - ob - PREFIX; look = tr - tr_end; }
        else if (tr >= ob + DEPTH + LOOK + PREFIX) { uint32_t remain = (tr - (ob + DEPTH + LOOK + PREFIX)) % 8; cnt = DEPTH; look = LOOK + remain; }
        if (cnt) {
            if (traceback(dec, tr, m, buf, cnt, look, s0 + 1) < 0) return -1000;
            memcpy(out + (ob >> 3), buf, cnt >> 3);
            ob += cnt; nout += (int)(cnt >> 3);
            if (tr >= tr_end) break;
            if (ob >= 256 * k1) break;
        }
    }
    return nout;
}

int wm_sequential(const uint8_t* soft, uint32_t nsoft, int cr, uint32_t frame_length, uint8_t* out)
{
    init();
    uint8_t m0[64]; for (int n = 0; n < 64; n++) m0[n] = 0x30; m0[0] = 0;
    uint64_t* dec = (uint64_t*)malloc(((size_t)nsoft + 16) * 8);
    const uint32_t none[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu }; uint8_t sn[2][64];
    const int r = run_unit(soft, nsoft, cr, frame_length, 0, m0, 0, 0x7FFFFFu, none, sn, out, dec);
    free(dec);
    return r;
}

/* The windowed decode: units of mwin windows, warm-up W steps (a multiple of 24).  Returns the number of unit boundaries whose verification
 * FAILED (0: the output is proven equal to the serial decode's), *nunits = units of the frame; out = the windowed output as it stands. */
int wm_windowed(const uint8_t* soft, uint32_t nsoft, int cr, uint32_t frame_length, uint32_t W, uint32_t mwin, uint8_t* out, uint32_t* nunits, uint32_t* first_fail)
{
    init();
    const uint32_t tr_end = frame_length * 8 + 16 + 6;
    /* windows: ob = 0, 256, ...; window k is a partial one iff a later event exists; the last event is the frame's end.  Number of
     * events = number of k with 256 k < obits (the last one takes what is left: up to 256 + 24 + ... bits) -- but the final trace takes over
     * as soon as tr >= tr_end, which can swallow a window whose threshold lies past tr_end: windows with 256 k + 286 > tr_end do not fire. */
    const uint32_t GSv = cr == 0 ? 1 : cr == 2 ? 3 : 2;
    uint32_t nwin = 0;
    while ((256 * nwin + 286 + GSv - 1) / GSv * GSv < tr_end) nwin++;              /* partial windows: the first check at or past 256 k + 286 comes before the frame's end */
    const uint32_t nev = nwin + 1;                                                 /* + the final trace */
    uint32_t nu = (nev + mwin - 1) / mwin;
    uint64_t* dec = (uint64_t*)malloc(((size_t)nsoft + 16) * 8);
    uint8_t (*spec)[64] = malloc((size_t)nu * 64), (*endv)[64] = malloc((size_t)nu * 64);
    uint8_t m0[64]; for (int n = 0; n < 64; n++) m0[n] = 0x30; m0[0] = 0;
    uint8_t z[64]; memset(z, 0, 64);
    int fails = 0; *first_fail = 0xFFFFFFFFu;
    for (uint32_t u = 0; u < nu; u++) {
        const uint32_t k0 = u * mwin, k1 = (u + 1 == nu) ? 0x7FFFFFu : (u + 1) * mwin;
        const uint32_t b = 256 * k0 / 24 * 24, bn = (u + 1 == nu) ? 0xFFFFFFFFu : 256 * k1 / 24 * 24;
        uint32_t s0 = 0; const uint8_t* im = m0;
        if (u > 0) { s0 = b >= W ? b - W : 0; im = s0 == 0 ? m0 : z; }
        const uint32_t snap_at[2] = { u > 0 ? b : 0xFFFFFFFFu, bn };
        uint8_t sn[2][64]; memset(sn, 0xEE, sizeof sn);
        const int r = run_unit(soft, nsoft, cr, frame_length, s0, im, k0, k1, snap_at, sn, out, dec);
        if (r < 0) { free(dec); free(spec); free(endv); return r; }
        memcpy(spec[u], sn[0], 64); memcpy(endv[u], sn[1], 64);
    }
    for (uint32_t u = 1; u < nu; u++) if (memcmp(spec[u], endv[u - 1], 64) != 0) { fails++; if (*first_fail == 0xFFFFFFFFu) *first_fail = u; }
    *nunits = nu;
    free(dec); free(spec); free(endv);
    return fails;
}
