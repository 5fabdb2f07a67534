"""benchlib.stages -- the per-stage rows: streaming kernels against the HBM roofline (stages, ingest, tx)."""
import json
import os
import sys
import time

import numpy as np

from benchlib.common import *  # noqa: F401,F403
from benchlib.common import _traffic_profile  # noqa: F401


def bench_stages(torch, sora_amd, dev, nsym=1 << 20, reps=12, nsets=3):
    """The per-stage entry points (what the BRICK adapters call), each over `nsym` OFDM symbols resident in HBM, against the
    HBM roofline with SURVEY.md section 8(d)'s algorithmic bytes per symbol: FFT 256 in + 256 out; symbol front end
    (T11aDataSymbol..TChannelEqualization) 320 in + 256 out; demap (64-QAM) 256 in + 288 out; de-interleave 288 + 288;
    Viterbi (54 Mbps frames of 56 symbols) 288 soft bytes in + 27 decoded bytes out; FFT<128> 512 + 512.
    Round 4: every stage cycles through `nsets` DISTINCT input / output buffer sets, so consecutive launches share no
    line and the bytes in play (1.6-1.8 GB) are far past the 256 MiB Infinity Cache -- round 3's single 537-604 MB set only just exceeded it."""
    from sora_amd import capi
    L = capi.load()
    out = {}
    g = torch.Generator(device=dev); g.manual_seed(7)

    def timed(fns, nbytes, label, n=nsym):
        for f in fns:
            f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(reps):
            fns[i % len(fns)]()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[label] = {"symbols": n, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "achieved": round(nbytes / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
                      "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4), "gsymbols_per_s": round(n / ms / 1e6, 3),
                      "buffer_sets": len(fns), "bytes_in_play": int(nbytes) * len(fns)}

    st = capi._stream_ptr(None)
    P = capi._dev_ptr
    xs = [torch.randint(-6000, 6000, (nsym, 64, 2), dtype=torch.int16, device=dev, generator=g) for _ in range(nsets)]
    ys = [torch.empty_like(xs[0]) for _ in range(nsets)]
    timed([(lambda x=x, y=y: L.sora_hip_fft64(P(x), P(y), nsym, st)) for x, y in zip(xs, ys)], nsym * 512, "fft64")
    del ys
    softs = [torch.empty((nsym, 288), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    timed([(lambda x=x, o=o: L.sora_hip_demap11a(P(x), P(o), 6, nsym, st)) for x, o in zip(xs, softs)], nsym * (256 + 288), "demap11a_qam64")
    des = [torch.empty_like(softs[0]) for _ in range(nsets)]
    timed([(lambda i=i, o=o: L.sora_hip_deinterleave11a(P(i), P(o), 6, nsym, st)) for i, o in zip(softs, des)], nsym * 576, "deinterleave11a_qam64")
    del des, softs
    n128 = nsym // 2
    x128 = [x.view(n128, 128, 2) for x in xs]; y128 = [torch.empty_like(x128[0]) for _ in range(nsets)]
    timed([(lambda x=x, y=y: L.sora_hip_fft128(P(x), P(y), n128, st)) for x, y in zip(x128, y128)], n128 * 1024, "fft128", n128)
    del y128, x128, xs
    x80 = [torch.randint(-6000, 6000, (nsym, 80, 2), dtype=torch.int16, device=dev, generator=g) for _ in range(nsets)]
    nctx = 4096
    lts_in = torch.randint(-6000, 6000, (nctx, 144, 2), dtype=torch.int16, device=dev, generator=g)
    ctx = sora_amd.lts11a(lts_in)
    idx = (torch.arange(nsym, device=dev, dtype=torch.int32) // 256) % nctx
    eqs = [torch.empty((nsym, 64, 2), dtype=torch.int16, device=dev) for _ in range(nsets)]
    timed([(lambda x=x, e=e: L.sora_hip_symfront11a(P(x), P(ctx), P(idx), P(e), nsym, st)) for x, e in zip(x80, eqs)], nsym * 576, "symfront11a")
    # the three one-multiply bricks alone: 256 in + 256 out per symbol, a frame's 256 bytes of coefficients shared by its 256 symbols
    # (SURVEY section 8d counts a private coefficient read per symbol for the equaliser: 768)
    del x80
    x64 = [torch.randint(-6000, 6000, (nsym, 64, 2), dtype=torch.int16, device=dev, generator=g) for _ in range(nsets)]
    stt = torch.randint(-32768, 32767, (nctx, 134), dtype=torch.int16, device=dev, generator=g)
    timed([(lambda x=x, e=e: L.sora_hip_freq_comp11a(P(x), P(ctx), P(idx), P(e), nsym, st)) for x, e in zip(x64, eqs)], nsym * 512, "freq_comp11a")
    timed([(lambda x=x, e=e: L.sora_hip_equalize11a(P(x), P(ctx), P(idx), P(e), nsym, st)) for x, e in zip(x64, eqs)], nsym * 512, "equalize11a")
    timed([(lambda x=x, e=e: L.sora_hip_phase_comp11a(P(x), P(stt), P(idx), P(e), nsym, st)) for x, e in zip(x64, eqs)], nsym * 512, "phase_comp11a")
    del x64, eqs, idx, stt
    # Viterbi: frames of 56 symbols x 216 soft values (the bench frame), random soft values 0..7
    nfr = 8192; nso = 56 * 288
    sv = torch.randint(0, 8, (nfr * nso,), dtype=torch.uint8, device=dev, generator=g)
    so = (torch.arange(nfr, device=dev, dtype=torch.int32) * nso).contiguous(); ns = torch.full((nfr,), nso, dtype=torch.int32, device=dev)
    fl = torch.full((nfr,), MPDU_LEN, dtype=torch.int16, device=dev)
    vo = torch.zeros((nfr, 1536), dtype=torch.uint8, device=dev); oo = (torch.arange(nfr, device=dev, dtype=torch.int32) * 1536).contiguous()
    timed([lambda: L.sora_hip_viterbi11a(P(sv), P(so), P(ns), P(fl), 2, P(vo), P(oo), nfr, st)],
          nfr * 56 * (288 + 27), "viterbi11a_r34", nfr * 56)
    return out


def bench_ingest(torch, sora_amd, dev, nbytes=256 << 20, reps=20, nsets=3):
    """Row f3 (capture ingest): a 44 MHz RX_BLOCK dump resident in HBM -> de-framed, sign-fixed, resampled 40 MHz stream.
    A pure streaming kernel: algorithmic bytes = dump bytes read + samples written, against the HBM roofline.  `nsets` distinct dumps in turn
    (round 4: 0.8 GB of input in play instead of one 256 MiB buffer that is exactly the size of the Infinity Cache)."""
    flags = sora_amd.INGEST_RXBLOCK | sora_amd.INGEST_RAW14 | sora_amd.INGEST_44TO40
    raws = [torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    n_out = sora_amd.ingest_count(nbytes, flags)
    for r in raws:
        out = sora_amd.ingest(r, flags, sync=False)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps):
        out = sora_amd.ingest(raws[i % nsets], flags, sync=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = nbytes + 4 * n_out
    del raws, out
    return {"workload": "%d MiB Sora RX_BLOCK dump @44 MHz -> de-frame + 14->16 bit + 44->40 MHz (%d samples out), %d distinct dumps in turn" % (nbytes >> 20, n_out, nsets),
            "bound": "hbm", "ms": round(ms, 4), "algorithmic_bytes": alg, "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / HBM_PEAK, 4), "msamples_per_s_in": round(nbytes / 128 * 28 / ms / 1e3, 1)}


def bench_tx(torch, sora_amd, nframes=4096, reps=10):
    """Row f2 (transmitter): the same 4096 x 1500-byte 54 Mbps frames modulated on the GPU (COMPLEX8 @40 MHz out)."""
    rng = np.random.default_rng(0x5EED)
    mpdus = [bytes(rng.integers(0, 256, MPDU_LEN - 4).astype(np.uint8)) for _ in range(64)] * (nframes // 64)
    out, off = sora_amd.tx11a(mpdus, [RATE_KBPS] * nframes)                  # builds the device arrays; also the warm-up
    import ctypes
    from sora_amd import capi
    lens = torch.full((nframes,), MPDU_LEN - 4, dtype=torch.int32, device=out.device)
    rate = torch.full((nframes,), RATE_KBPS, dtype=torch.int32, device=out.device)
    seed = torch.full((nframes,), 0xFF, dtype=torch.uint8, device=out.device)
    moff = torch.arange(nframes, dtype=torch.int32, device=out.device) * (MPDU_LEN - 4)
    blob = torch.randint(0, 256, (nframes * (MPDU_LEN - 4),), dtype=torch.uint8, device=out.device)
    ooff = torch.arange(nframes, dtype=torch.int64, device=out.device) * (off[1] - off[0])
    L = capi.load()
    call = lambda: L.sora_hip_tx11a(capi._dev_ptr(blob), capi._dev_ptr(moff), capi._dev_ptr(lens), capi._dev_ptr(rate), capi._dev_ptr(seed),
                                    nframes, capi._dev_ptr(out), capi._dev_ptr(ooff), capi._stream_ptr(None))
    call()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nsamp = int(off[1] - off[0]) * nframes
    alg = nframes * (MPDU_LEN - 4) + 2 * nsamp
    return {"workload": "%d frames x %d-byte MPDU at 54 Mbps -> COMPLEX8 @40 MHz (%d samples)" % (nframes, MPDU_LEN, nsamp),
            "bound": "hbm", "ms": round(ms, 4), "msamples_per_s_out": round(nsamp / ms / 1e3, 1), "algorithmic_bytes": alg,
            "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / HBM_PEAK, 4)}
