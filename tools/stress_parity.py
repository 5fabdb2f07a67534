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
from gpu_util import batch, random_capture, same_results                # noqa: E402
from oracle.pyoracle import Oracle, RATES                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--captures", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--trellis", type=int, default=0, help="0: the library's choice (the window-parallel kernel for batches this small), 1: k_viterbi16w, 16: k_viterbi16, 64: k_viterbi")
    ap.add_argument("--front", type=int, default=0, help="0: the library's choice, 1: k_frame, 3: k_sym_front -> k_track_lds -> k_sym_back, 4: k_pipe (needs few rows in flight: --batch 1 or 2, --depth 1)")
    ap.add_argument("--depth", type=int, default=0, help="calls in flight the handle is sized for (0: its default)")
    ap.add_argument("--max-frames", type=int, default=8)
    ap.add_argument("--pipe-wait-us", type=int, default=-1, help="bound of the waits inside a k_pipe launch (0: every wait that is not satisfied at once gives up -> the finishing kernel makes the call again)")
    ap.add_argument("--noise-frames", type=float, default=0.0, help="share of captures whose data field is replaced by noise behind an intact SIGNAL symbol (the window-parallel trellis's proof fails: the serial path)")
    args = ap.parse_args()
    import torch
    import sora_amd
    o = Oracle()
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); nfr = 0; nok = 0
    rec = {"boundaries": 0, "boundaries_failed": 0, "frames_decoded_again": 0, "units": 0}
    pipe = {"calls_made_again": 0, "backoffs": 0, "calls": 0}
    for b0 in range(0, args.captures, args.batch):
        mhz = int(rng.choice([20, 40]))
        caps = [random_capture(o, rng, mhz) for _ in range(min(args.batch, args.captures - b0))]
        for i in range(len(caps)):
            if rng.random() < args.noise_frames and len(caps[i]) > 3000:
                c = caps[i].astype(np.int32); a = (700 if mhz == 20 else 1400); c[a:len(c) - 200] = np.rint(rng.normal(0, 2500, (len(c) - 200 - a, 2)))
                caps[i] = np.clip(c, -32768, 32767).astype(np.int16)
        iq, descs = batch(caps)
        rx = sora_amd.Rx(len(caps), len(iq), sample_rate_mhz=mhz, max_frames_per_capture=args.max_frames)
        if args.depth:
            rx.set_depth(args.depth)
        if args.trellis:
            rx.set_trellis(args.trellis)
        if args.front:
            rx.set_front(args.front)
            assert rx.front() == args.front, "the handle does not run front %d in this shape (it would run %d)" % (args.front, rx.front())
        if args.pipe_wait_us >= 0:
            rx.set_pipe_wait_us(args.pipe_wait_us)
        rx.process_dev(torch.from_numpy(iq).cuda(), descs)
        got = rx.results()
        for k, v in rx.window_stats().items():
            rec[k] += v
        if args.front == 4:
            for k, v in rx.pipe_stats().items():
                pipe[k] += v
            pipe["calls"] += 1
        rx.close()
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
    print("stress parity OK: %d captures, %d frames (%d FRAME_OK) identical, %.1f s; window-parallel trellis: %s%s" % (args.captures, nfr, nok, time.time() - t0, rec,
          "; k_pipe: %s" % pipe if args.front == 4 else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
