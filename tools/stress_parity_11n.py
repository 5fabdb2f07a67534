"""GPU (sora_rx11n_*) against the CPU oracle on many random two-chain 802.11n captures built from the recorded waveforms of the
reference modulator (tests/golden/refgraph_11n.npz): decoded frames, FCS failures, header failures (MCS 12), frames cut by the end
of the capture, several frames per capture.  usage: python tools/stress_parity_11n.py [captures] [seed] [trellis: 64 | 16 | 1 | 0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import capture_11n, same_events_11n  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402


def main():
    import torch
    import sora_amd
    ncap = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    z = np.load(os.path.join(ROOT, "tests", "golden", "refgraph_11n.npz"))
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(seed); o = Oracle()
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if g.available():                                                       # the compiled reference modulator: fresh frames of any length
        for _ in range(36):
            mcs = int(rng.choice([8, 9, 10, 10, 12])); ln = int(rng.integers(1, 1497)) if rng.integers(0, 3) else int(rng.integers(1, 60))
            frames.append(g.tx11n(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), mcs))
    caps = []
    for t in range(ncap):
        fr = [frames[int(i)] for i in rng.integers(0, len(frames), size=int(rng.integers(1, 5)))]
        caps.append(capture_11n(rng, fr, sigma=float(rng.choice([3, 20, 60, 200, 600, 1500])), cut=float(rng.uniform(0.05, 1.0)) if t % 3 == 2 else None))
    iq0 = np.concatenate([a for a, _ in caps]); iq1 = np.concatenate([b for _, b in caps])
    descs = []; off = 0
    for i, (a, _) in enumerate(caps):
        descs.append((off, len(a), i)); off += len(a)
    rx = sora_amd.Rx11n(ncap, len(iq0), max_frames_per_capture=8)
    if len(sys.argv) > 3:                                                  # trellis kernel: 64, 16, 1 (the window-parallel form, round 6), 0 = automatic
        rx.set_trellis(int(sys.argv[3]))
    t0 = time.perf_counter()
    rx.process_dev(torch.from_numpy(iq0).cuda(), torch.from_numpy(iq1).cuda(), descs)
    per = [[] for _ in caps]
    for r in rx.results():
        per[r["capture_id"]].append(r)
    tg = time.perf_counter() - t0
    bad = 0; nev = 0; kinds = {}
    for i, (a, b) in enumerate(caps):
        want = o.rx11n_capture(a, b)
        ok, why = same_events_11n(per[i], want, position="end_sample")
        nev += len(want)
        for e in want:
            kinds[hex(e["error_code"])] = kinds.get(hex(e["error_code"]), 0) + 1
        if not ok:
            bad += 1
            if bad < 6:
                print("capture", i, why, [(hex(e["error_code"]), e["rate_kbps"], e["length"], e["end_sample"]) for e in per[i]],
                      [(hex(e["error_code"]), e["rate_kbps"], e["length"], e["end_sample"]) for e in want])
    print("captures %d  events %d %s  mismatching captures %d  (gpu incl. copies %.2f s)" % (ncap, nev, kinds, bad, tg))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
