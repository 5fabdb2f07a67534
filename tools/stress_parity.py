#!/usr/bin/env python3
"""stress_parity.py -- randomised parity hunt: many synthetic captures (random rates, lengths, noise, CFO, DC offset,
gaps, several frames per capture, truncated frames, pure noise) through the GPU receive path and through the oracle;
every result row must be identical.  Run on the GPU box:  python tools/stress_parity.py [--captures N] [--seed S]
Exit code 1 and a dump of the offending capture (gpurun_out/stress_fail_*.npy) on the first mismatch."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import awgn, batch, pad_capture, same_results            # noqa: E402
from oracle.pyoracle import Oracle, RATES                              # noqa: E402


def random_capture(o, rng, mhz):
    kind = rng.integers(0, 10)
    parts = []
    if kind == 0:                                                        # noise only, sometimes loud enough to trip carrier sense
        n = int(rng.integers(2, 200)) * 28
        return pad_capture(np.rint(rng.normal(0, rng.choice([30, 300, 3000]), (n, 2))).astype(np.int16), mhz)
    nfr = int(rng.choice([1, 1, 1, 2, 3]))
    for _ in range(nfr):
        rate = int(rng.choice(RATES)); L = int(rng.choice([1, 5, 20, 60, 150, 400, 900, 1500]))
        mp = rng.integers(0, 256, L).astype(np.uint8).tobytes()
        cap = o.tx_capture(mp, rate, seed=int(rng.integers(1, 128)), lead=int(rng.integers(0, 120)), tail=int(rng.choice([40, 160, 200, 400, 900])))
        parts.append(cap)
    x = np.concatenate(parts)
    if kind == 1:                                                        # truncated: the last frame runs past the capture
        x = x[:int(len(x) * rng.uniform(0.3, 0.95))]
    if rng.random() < 0.3:                                               # carrier frequency offset
        f = rng.uniform(-80e3, 80e3)
        z = (x[:, 0].astype(np.float64) + 1j * x[:, 1]) * np.exp(2j * np.pi * f * np.arange(len(x)) / 40e6)
        x = np.stack([np.rint(z.real), np.rint(z.imag)], 1)
    x = x.astype(np.float64)
    if rng.random() < 0.3:                                               # DC offset (TDCRemoveEx / TDCEstimator path)
        x += rng.uniform(-600, 600, size=(1, 2))
    if rng.random() < 0.2:                                               # gain
        x *= rng.uniform(0.25, 1.6)
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    sigma = float(rng.choice([0, 0, 60, 150, 400, 900, 2000]))
    if sigma:
        x = awgn(x, sigma, int(rng.integers(1 << 30)))
    if mhz == 20:
        x = x[::2].copy()
    return pad_capture(x, mhz)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--captures", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=100)
    args = ap.parse_args()
    import torch
    import sora_amd
    o = Oracle()
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); nfr = 0; nok = 0
    for b0 in range(0, args.captures, args.batch):
        mhz = int(rng.choice([20, 40]))
        caps = [random_capture(o, rng, mhz) for _ in range(min(args.batch, args.captures - b0))]
        iq, descs = batch(caps)
        rx = sora_amd.Rx(len(caps), len(iq), sample_rate_mhz=mhz, max_frames_per_capture=8)
        rx.process_dev(torch.from_numpy(iq).cuda(), descs)
        got = rx.results(); rx.close()
        want = []
        for i, c in enumerate(caps):
            for r in o.rx_capture(c, mhz):
                r = dict(r); r["capture_id"] = i; want.append(r)
        ok, why = same_results(got, want)
        if not ok:
            # find the first offending capture and save it
            for i, c in enumerate(caps):
                g = [r for r in got if r["capture_id"] == i]; w = [r for r in want if r["capture_id"] == i]
                if not same_results(g, w)[0]:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.save(os.path.join(ROOT, "gpurun_out", "stress_fail_%dmhz_seed%d_cap%d.npy" % (mhz, args.seed, b0 + i)), c)
                    print("MISMATCH batch %d capture %d (%d MHz): %s" % (b0, i, mhz, same_results(g, w)[1]))
                    break
            print("FAILED:", why)
            return 1
        nfr += len(want); nok += sum(r["error_code"] == 1 for r in want)
    print("stress parity OK: %d captures, %d frames (%d FRAME_OK) identical, %.1f s" % (args.captures, nfr, nok, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
