#!/usr/bin/env python3
"""fuzz_oracle_vs_reference.py <first seed> <seeds> <captures per seed> -- CPU-only: oracle/so_rx11a.c against the reference's own
802.11a receive graph compiled from its sources (oracle/_ref/libsora_refgraph.so) on random captures (tests/gpu_util.random_capture);
every event (error code, source position, rate, length, FCS, MPDU bytes) must be identical.  Offending captures are saved as .npy."""
import sys, os, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_util import random_capture, same_as_reference_graph, source_position_44, upsample_40_to_44
from oracle.pyoracle import Oracle, ReferenceGraph
o = Oracle(); g = ReferenceGraph()
seed0 = int(sys.argv[1]); nseeds = int(sys.argv[2]); per = int(sys.argv[3])
bad = 0; nev = 0; t0 = time.time()
for s in range(seed0, seed0 + nseeds):
    rng = np.random.default_rng(s)
    for i in range(per):
        c = random_capture(o, rng, 40)
        ev = g.rx11a(c)
        ok, why = same_as_reference_graph(o.rx_capture(c, 40), ev)
        nev += len(ev)
        if not ok:
            bad += 1; np.save('fuzz_fail_%d_%d.npy' % (s, i), c); print("MISMATCH seed", s, "capture", i, why, flush=True)
print("fuzz done: seeds %d..%d x %d captures, %d events, %d mismatches, %.0f s" % (seed0, seed0 + nseeds - 1, per, nev, bad, time.time() - t0), flush=True)
