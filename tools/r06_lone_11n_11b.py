"""A lone 802.11n capture (one MCS 10 frame of 1000 bytes, two chains) and a lone 802.11b capture (one 1 Mbps frame of 500 bytes) `reps` times, one call in flight, for
a kernel trace: rocprofv3 --kernel-trace --output-format rocpd -d <dir> -o p -- python tools/r06_lone_11n_11b.py 30; tools/rocprof_kernels.py <db> --last N shows a call's chain."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                                                # noqa: E402
import sora_amd                                                             # noqa: E402
from oracle.pyoracle import ReferenceGraph                                  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
which = sys.argv[2] if len(sys.argv) > 2 else "11n"
g = ReferenceGraph()
if which == "11n":
    s0, s1 = g.tx11n(np.random.default_rng(12).integers(0, 256, 1000).astype(np.uint8).tobytes(), 10)
    n = (len(s0) + 800 + 1200 + 27) // 28 * 28
    c = np.zeros((2, n, 2), np.float64)
    c[0, 800:800 + len(s0)] = s0 + 0.1 * s1; c[1, 800:800 + len(s0)] = s1 + 0.1 * s0
    c = np.clip(np.rint(c + np.random.default_rng(6).normal(0, 20, c.shape)), -32768, 32767).astype(np.int16)
    rx = sora_amd.Rx11n(1, 2 * n + 4096, max_frames_per_capture=4); rx.set_depth(1); rx.set_trellis(64)
    d0 = torch.from_numpy(c[0]).cuda(); d1 = torch.from_numpy(c[1]).cuda(); one = [(0, n, 0)]
    for _ in range(reps):
        rx.wait(rx.process_dev(d0, d1, one))
else:
    s8 = g.tx11b(np.random.default_rng(11).integers(0, 256, 500).astype(np.uint8).tobytes(), 1000)
    n = (len(s8) + 1200 + 2800 + 27) // 28 * 28
    cap = np.zeros((n, 2), np.int16); cap[1200:1200 + len(s8)] = s8.astype(np.int16) << 8
    cap = np.clip(cap + np.rint(np.random.default_rng(5).normal(0, 40, cap.shape)), -32768, 32767).astype(np.int16)
    rx = sora_amd.Rx11b(1, 2 * n + 4096, max_frames_per_capture=4)
    d = torch.from_numpy(cap).cuda(); one = [(0, n, 0)]
    for _ in range(reps):
        rx.wait(rx.process_dev(d, one))
rx.close()
