#!/usr/bin/env python3
"""

# (round 5) the probe hooks live in the TOOLS variant of the library only: build it once and load it
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if not _os.environ.get("SORA_HIP_LIB"):
    from sora_amd import build as _b
    _v = _os.path.join(_os.path.dirname(_b.LIB), "variants", "tools.so")
    _os.environ["SORA_HIP_LIB"] = _v if _os.path.exists(_v) else _b.build_variant("tools", ["SORA_TOOLS"])
r04_trellis_only.py -- on the GPU box: the trellis kernel's own throughput by launch size and by launches in flight.  One handle, calls that launch ONLY
k_viterbi16 (tool hook sora_internal_rx_only; the earlier stages' arrays are those of one full call), `depth` calls in flight on the handle's pipelines.
Prints ms per 4096 captures.  (profiles/r04_w_corun.txt: two launches of 1024 waves side by side finish sooner than one of 2048.)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import sora_amd
from sora_amd import capi
import bench
from oracle.pyoracle import Oracle
L = capi.load()
L.sora_internal_rx_only.argtypes = [ctypes.c_void_p, ctypes.c_uint]
MASK = int(os.environ.get("ONLY_MASK", "4"))
o = Oracle()
print("# kernels launched per call: mask %d (1 k_scan, 2 k_frame, 4 k_viterbi16, 8 k_finish); ms per 4096 captures" % MASK)
print("%-20s" % "captures per launch" + "".join("%10s" % ("x%d" % d) for d in (1, 2, 3, 4, 8)))
for nf in (2048, 4096, 8192, 16384, 32768):
    iq, descs, _ = bench.make_workload(o, nf, seed0=0)
    d_iq = torch.from_numpy(iq).cuda(); descs = sora_amd.Rx.captures(descs)
    row = "%-20d" % nf
    for depth in (1, 2, 3, 4, 8):
        rx = sora_amd.Rx(max_captures=nf, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
        rx.set_depth(depth); rx.set_trellis(16); rx.wait_for_producer = False
        for _ in range(depth):
            rx.process_dev(d_iq, descs)                                   # a full call on every pipeline: all stages' arrays exist
        rx.flush()
        assert L.sora_internal_rx_only(rx._h, MASK) == 0
        for _ in range(2 * depth):
            rx.process_dev(d_iq, descs)
        rx.flush()
        n = max(8 * depth, int(40 * 4096 / nf) * depth)
        t0 = time.perf_counter()
        for _ in range(n):
            rx.process_dev(d_iq, descs)                                   # blocks until the pipeline's previous call (depth calls ago) has finished
        rx.flush()
        ms = (time.perf_counter() - t0) / n * 1e3
        row += "%10.4f" % (ms * 4096 / nf)
        rx.close()
    print(row)
    del d_iq
