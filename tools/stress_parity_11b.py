#!/usr/bin/env python3
"""stress_parity_11b.py -- randomised parity hunt for the 802.11b path: captures built from the reference's own modulator
(oracle/_ref/libsora_refgraph.so must be present) through the GPU (sora_rx11b_*), the reference's receive graph and
oracle/so_rx11b.c; every event must be identical.  Run on the GPU box:  python tools/stress_parity_11b.py [--captures N] [--seed S]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import random_capture_11b, same_as_reference_11b           # noqa: E402
from oracle.pyoracle import Oracle, ReferenceGraph                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--captures", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=500)
    args = ap.parse_args()
    import torch
    import sora_amd
    o = Oracle(); g = ReferenceGraph()
    assert g.available(), "needs oracle/_ref/libsora_refgraph.so"
    rng = np.random.default_rng(args.seed)
    t0 = time.time(); nev = nok = 0
    for b0 in range(0, args.captures, args.batch):
        caps = [random_capture_11b(g, rng) for _ in range(min(args.batch, args.captures - b0))]
        descs, pos = [], 0
        for i, c in enumerate(caps):
            descs.append((pos, len(c), i)); pos += len(c)
        rx = sora_amd.Rx11b(len(caps), pos, max_frames_per_capture=64)
        rx.process_dev(torch.from_numpy(np.concatenate(caps)).cuda(), descs)
        got = rx.results(); rx.close()
        for i, c in enumerate(caps):
            rows = [r for r in got if r["capture_id"] == i]
            ev = g.rx11b(c, max_frames=64)
            ok, why = same_as_reference_11b(rows, ev)
            if ok:
                ok, why = same_as_reference_11b(rows, [dict(r, sample_index=r["end_sample"]) for r in o.rx11b_capture(c, max_frames=64)])
            if not ok:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.save(os.path.join(ROOT, "gpurun_out", "stress11b_fail_seed%d_cap%d.npy" % (args.seed, b0 + i)), c)
                print("MISMATCH capture %d: %s" % (b0 + i, why)); return 1
            nev += len(ev); nok += sum(e["error_code"] == 1 for e in ev)
    print("11b stress parity OK: %d captures, %d events (%d FRAME_OK) identical on GPU, reference graph and oracle, %.1f s" % (args.captures, nev, nok, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
