#!/usr/bin/env python3
"""sweep_lengths_vs_reference.py -- CPU-only: every MPDU length 1..2496 at every 802.11a rate, clean and noisy, through
oracle/so_rx11a.c and through the reference's own receive graph (oracle/_ref/libsora_refgraph.so); all events identical.
Exercises every alignment of the frame end against OFDM symbols, Viterbi windows (256 + 24) and source bursts."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import awgn, pad_capture, same_as_reference_graph        # noqa: E402
from oracle.pyoracle import Oracle, ReferenceGraph, RATES               # noqa: E402


def main():
    o = Oracle(); g = ReferenceGraph()
    assert g.available()
    rng = np.random.default_rng(7); t0 = time.time(); n = bad = 0
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for rate in RATES:
        for ln in range(1, 2497, step):
            mp = rng.integers(0, 256, ln).astype(np.uint8).tobytes()
            cap = o.tx_capture(mp, rate, seed=1 + ln % 127, lead=int(rng.integers(0, 60)), tail=200)
            if ln % 2:
                cap = awgn(cap, float(rng.choice([60, 300, 900])), ln)
            cap = pad_capture(cap, 40)
            ok, why = same_as_reference_graph(o.rx_capture(cap, 40), g.rx11a(cap))
            n += 1
            if not ok:
                bad += 1; print("MISMATCH rate %d length %d: %s" % (rate, ln, why), flush=True)
        print("rate %d done, %d frames, %d mismatches, %.0f s" % (rate, n, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
