#!/usr/bin/env python3
"""sweep_lengths_gpu.py -- on the GPU box: every MPDU length 1..2496 at every 802.11a rate (clean / noisy alternating)
through the GPU receive path and through the reference's own receive graph (oracle/_ref/libsora_refgraph.so, or the
oracle where that is absent); every event identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import awgn, batch, pad_capture, same_as_reference_graph  # noqa: E402
from oracle.pyoracle import Oracle, ReferenceGraph, RATES                # noqa: E402


def main():
    import torch
    import sora_amd
    o = Oracle(); g = ReferenceGraph()
    rng = np.random.default_rng(7); t0 = time.time(); n = 0
    for rate in RATES:
        caps = []
        for ln in range(1, 2497):
            mp = rng.integers(0, 256, ln).astype(np.uint8).tobytes()
            cap = o.tx_capture(mp, rate, seed=1 + ln % 127, lead=int(rng.integers(0, 60)), tail=200)
            if ln % 2:
                cap = awgn(cap, float(rng.choice([60, 300, 900])), ln)
            caps.append(pad_capture(cap, 40))
        iq, descs = batch(caps)
        rx = sora_amd.Rx(len(caps), len(iq), sample_rate_mhz=40, max_frames_per_capture=4)
        rx.process_dev(torch.from_numpy(iq).cuda(), descs)
        got = rx.results(); rx.close()
        by = {}
        for r in got:
            by.setdefault(r["capture_id"], []).append(r)
        for i, c in enumerate(caps):
            if g.available():
                want = g.rx11a(c)
            else:
                want = [dict(r, sample_index=-(-r["end_sample"] * 2 // 28) * 28) for r in o.rx_capture(c, 40)]
            ok, why = same_as_reference_graph(by.get(i, []), want)
            if not ok:
                print("MISMATCH rate %d length %d: %s" % (rate, i + 1, why)); return 1
            n += 1
        print("rate %d: %d lengths identical (%.0f s)" % (rate, len(caps), time.time() - t0), flush=True)
    print("length sweep OK: %d frames identical on the GPU and the %s" % (n, "reference graph" if g.available() else "oracle"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
