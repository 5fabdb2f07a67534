"""legacy_vs_brick.py [frames] -- CPU: the reference's legacy dot11a receiver against its brick graph (both compiled from the reference sources,
oracle/_ref) on frames of every rate at four noise levels: how often each decodes, and whether the MPDUs agree where both do."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import awgn                                             # noqa: E402
from oracle.pyoracle import Oracle, ReferenceGraph, ReferenceLegacy, RATES   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    o = Oracle(); g = ReferenceGraph(); lg = ReferenceLegacy()
    rng = np.random.default_rng(7)
    stat = {}
    for i in range(n):
        rate = RATES[i % 8]; ln = int(rng.integers(20, 1500)); sigma = [0, 150, 500, 900][(i // 8) % 4]
        mp = rng.integers(0, 256, ln).astype(np.uint8).tobytes()
        cap = o.tx_capture(mp, rate, lead=int(rng.integers(300, 900)) // 28 * 28, tail=1400)
        if sigma:
            cap = awgn(cap, sigma, i)
        cap = cap[:len(cap) // 28 * 28]
        eb = [e for e in g.rx11a(cap) if e["error_code"] == 1]; el = [e for e in lg.rx11a(cap) if e["hr"] == 0x202]
        k = stat.setdefault(sigma, {"frames": 0, "both": 0, "same_bytes": 0, "only_brick": 0, "only_legacy": 0, "neither": 0})
        k["frames"] += 1
        if eb and el:
            k["both"] += 1; k["same_bytes"] += eb[0]["mpdu"] == el[0]["mpdu"]
        elif eb: k["only_brick"] += 1
        elif el: k["only_legacy"] += 1
        else: k["neither"] += 1
    for s in sorted(stat): print("sigma", s, stat[s])


if __name__ == "__main__":
    main()
