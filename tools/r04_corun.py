#!/usr/bin/env python3
"""

# (round 5) the probe hooks live in the TOOLS variant of the library only: build it once and load it
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if not _os.environ.get("SORA_HIP_LIB"):
    from sora_amd import build as _b
    _v = _os.path.join(_os.path.dirname(_b.LIB), "variants", "tools.so")
    _os.environ["SORA_HIP_LIB"] = _v if _os.path.exists(_v) else _b.build_variant("tools", ["SORA_TOOLS"])
r04_corun.py -- on the GPU box: how much does each front-end kernel slow the trellis kernel down when both are on the chip?  Two handles of 16384 captures
each, one call in flight each, two host threads: handle T launches ONLY the trellis kernel (the chip's trellis slots full: 2048 waves), handle F launches only
k_scan, only k_frame or only k_finish, back to back (tool hook sora_internal_rx_only; the skipped stages' arrays stay as a full call left them).
Prints the mean duration of T's launches alone and beside each F, and F's beside T."""
import ctypes, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import sora_amd
from sora_amd import capi
import bench
from oracle.pyoracle import Oracle
NF = int(os.environ.get("CORUN_FRAMES", "16384"))
iq, descs, _ = bench.make_workload(Oracle(), NF, seed0=0)
d_iq = torch.from_numpy(iq).cuda(); descs = sora_amd.Rx.captures(descs)
L = capi.load()
L.sora_internal_rx_only.argtypes = [ctypes.c_void_p, ctypes.c_uint]

def handle(trellis):
    rx = sora_amd.Rx(max_captures=NF, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(1); rx.set_trellis(trellis); rx.wait_for_producer = False
    rx.wait(rx.process_dev(d_iq, descs))                                   # a full call: every stage's arrays exist
    return rx
T = handle(16); F = handle(16)
torch.cuda.synchronize()

def loop(rx, n, out, stop=None):
    t0 = time.perf_counter(); k = 0
    while k < n and not (stop and stop.is_set()):
        rx.wait(rx.process_dev(d_iq, descs)); k += 1
    out.append((time.perf_counter() - t0) / max(1, k) * 1e3)

def alone(rx, mask, n=60):
    assert L.sora_internal_rx_only(rx._h, mask) == 0
    o = []; loop(rx, 5, []); loop(rx, n, o); return o[0]
names = {1: "k_scan", 2: "k_frame", 8: "k_finish", 4: "k_viterbi16"}
print("# %d captures per launch; ms per launch (host loop: launch + wait)" % NF)
base = {m: alone(F, m) for m in (1, 2, 8)}
t_alone = alone(T, 4)
print("%-12s alone %.4f" % (names[4], t_alone))
for m in (1, 2, 8):
    print("%-12s alone %.4f" % (names[m], base[m]))
if os.environ.get("CORUN_ALONE"):
    sys.exit(0)
for m in (1, 2, 8, 4):
    L.sora_internal_rx_only(F._h, m); L.sora_internal_rx_only(T._h, 4)
    loop(F, 3, []); loop(T, 3, [])
    stop = threading.Event(); fo = []; to = []
    th = threading.Thread(target=loop, args=(F, 10 ** 9, fo, stop)); th.start()
    time.sleep(0.05)
    loop(T, 60, to)
    stop.set(); th.join()
    fa = base.get(m, t_alone)
    print("k_viterbi16 beside %-12s: %.4f ms (x %.2f)   %-12s beside it: %.4f ms (x %.2f)" % (names[m], to[0], to[0] / t_alone, names[m], fo[0], fo[0] / fa))
