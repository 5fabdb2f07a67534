/* so_ingest.c -- CPU restatement of the capture-ingest stages that sit in front of the 802.11a receive graph.
 *
 * TEST INFRASTRUCTURE (see so_oracle.h).  Follows, in the reference tree:
 *   TDownSample44_40            kernel/bb/Brick11/src/sampling.hpp:35-66   (28 samples in, 28 out when available)
 *   Down44to40::Resample        kernel/bb/Brick11/src/44MTo40M.hpp:62-123  (linear interpolation, 11 -> 10)
 *   TDownSample2                kernel/bb/Brick11/src/samples.hpp:9-47     (8 in, 4 out: even samples, no filter)
 * The de-framing of RX_BLOCK dumps and the 14 -> 16-bit sign fix are in so_rx11a.c (so_load_dump).
 */
#include "so_oracle.h"

static const int kLinR[11] = { 1, 115, 102, 90, 77, 64, 51, 38, 26, 13, 0 };   /* 44MTo40M.hpp:35-39 */
static const int kLinL[11] = { 0, 0, 13, 26, 38, 51, 64, 77, 90, 102, 115 };

/* Whole 28-sample RX blocks of `in` through the resampler; output forwarded in 28-sample blocks as the brick does
 * (a tail of fewer than 28 produced samples stays inside the brick).  Returns the samples written. */
int so_down44to40(const so_c16* in, uint32_t n_in, so_c16* out, uint32_t max_out)
{
    so_c16 buf[84];
    int length = 0;
    int last_re = 0, last_im = 0, last_index = -1;                              /* LastIQRight, LastIQIndex */
    uint32_t n_out = 0;
    for (uint32_t b = 0; b + 28 <= n_in; b += 28) {
        int index11 = 0;
        for (int i = 0; i < 28; i++) {
            const so_c16 x = in[b + (uint32_t)i];
            index11 = (last_index + i + 1) % 11;
            if (index11 > 1) {
                buf[length].re = (int16_t)((last_re + x.re * kLinL[index11]) >> 7);
                buf[length].im = (int16_t)((last_im + x.im * kLinL[index11]) >> 7);
                length++;
                last_re = x.re * kLinR[index11]; last_im = x.im * kLinR[index11];
            } else if (index11 == 0) {
                buf[length++] = x;
            } else {
                last_re = x.re * kLinR[1]; last_im = x.im * kLinR[1];
            }
        }
        last_index = index11;
        if (length >= 28) {                                                     /* GetOutStream(28) + Move() */
            if (n_out + 28 > max_out) break;
            for (int i = 0; i < 28; i++) out[n_out + (uint32_t)i] = buf[i];
            n_out += 28;
            length -= 28;
            for (int i = 0; i < length; i++) buf[i] = buf[i + 28];
        }
    }
    return (int)n_out;
}

/* TDownSample2: bursts of 8 -> 4, output j = input 2j (samples.hpp:36-39) */
int so_downsample2(const so_c16* in, uint32_t n_in, so_c16* out, uint32_t max_out)
{
    uint32_t n = 0;
    for (uint32_t b = 0; b + 8 <= n_in && n + 4 <= max_out; b += 8)
        for (int j = 0; j < 4; j++) out[n++] = in[b + 2u * (uint32_t)j];
    return (int)n;
}
