#!/usr/bin/env python3
"""

# (round 5) the probe hooks live in the TOOLS variant of the library only: build it once and load it
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if not _os.environ.get("SORA_HIP_LIB"):
    from sora_amd import build as _b
    _v = _os.path.join(_os.path.dirname(_b.LIB), "variants", "tools.so")
    _os.environ["SORA_HIP_LIB"] = _v if _os.path.exists(_v) else _b.build_variant("tools", ["SORA_TOOLS"])
r04_own_timeline.py [--frames F] [--depth D] [--calls N] -- on the GPU box: the headline loop's kernel timeline WITHOUT a profiler attached
(rocprofv3's kernel trace makes the host the bottleneck at eight calls in flight: 0.62 instead of 0.40 ms per step).  The library's own HIP events
(sora_rx_set_profiling) are read against one time base (tool hook sora_internal_rx_timeline).  Prints, for the steady state: the step period, how many
launches of each kernel run at once (time-weighted), the share of time no trellis / no front-end kernel is running, and the gaps inside a pipeline."""
import argparse, collections, ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4096); ap.add_argument("--depth", type=int, default=8); ap.add_argument("--calls", type=int, default=600)
ap.add_argument("--trellis", type=int, default=16); ap.add_argument("--no-profile-events", action="store_true")
a = ap.parse_args()
import torch
import sora_amd
from sora_amd import capi
import bench
from oracle.pyoracle import Oracle
oracle = Oracle()
iq, descs, _ = bench.make_workload(oracle, a.frames, seed0=0)
d_iqs = [torch.from_numpy(iq).cuda() for _ in range(4 if a.frames <= 16384 else 2)]
descs = sora_amd.Rx.captures(descs)
rx = sora_amd.Rx(max_captures=a.frames, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
rx.set_depth(a.depth); rx.set_trellis(a.trellis)
torch.cuda.synchronize(); rx.wait_for_producer = False
t = rx.process_dev(d_iqs[0], descs)
nb = a.depth + 2
bufs = [sora_amd.HostResults(a.frames * 2, rx.mpdu_bytes(t)) for _ in range(nb)]
rx.wait(t)
L = capi.load()
L.sora_internal_rx_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]

host = []                                                               # per call: ticket, submit begin / end, wait begin / end (host clock, ms since the time base)
host_t0 = [0.0]

def loop(n):
    first = None
    for _ in range(n):
        h0 = time.perf_counter()
        tk = rx.process_dev(d_iqs[(rx.ticket() + 1) % len(d_iqs)], descs)
        rx.deliver_async(tk, bufs[tk % nb])
        h1 = time.perf_counter()
        if first is None:
            first = tk
        if tk - first >= a.depth - 1:
            rx.wait(tk - (a.depth - 1))
        h2 = time.perf_counter()
        host.append((tk, 1e3 * (h0 - host_t0[0]), 1e3 * (h1 - host_t0[0]), 1e3 * (h2 - host_t0[0])))
    rx.flush()
loop(3 * a.depth)
t0 = time.perf_counter(); loop(a.calls); plain = (time.perf_counter() - t0) / a.calls * 1e3
rx.set_profiling(True)
assert L.sora_internal_rx_timeline(rx._h, 1, None, 0, None) == 0
host_t0[0] = time.perf_counter(); del host[:]; tk0 = rx.ticket()
t0 = time.perf_counter(); loop(a.calls); prof = (time.perf_counter() - t0) / a.calls * 1e3
out = np.zeros(8 * (a.calls + 64), np.float32); n = ctypes.c_size_t(0)
assert L.sora_internal_rx_timeline(rx._h, 0, out.ctypes.data, out.size, ctypes.byref(n)) == 0
rec = out[:n.value].reshape(-1, 7)
if os.environ.get("TL_RAW"):
    np.savez(os.environ["TL_RAW"], rec=rec, host=np.array(host), depth=a.depth, frames=a.frames, first_ticket=tk0 + 1)
print("captures per call %d, calls in flight %d: %.4f ms per call without events, %.4f with; %d calls logged" % (a.frames, a.depth, plain, prof, len(rec)))
names = ["memset+caps", "k_scan", "k_frame", "trellis", "k_finish"]
rec = rec[np.argsort(rec[:, 1])]
lo, hi = rec[len(rec) // 5, 1], rec[-len(rec) // 5, 1]                      # steady state: the middle three fifths
ev = []
for r in rec:
    for k, nm in enumerate(names):
        s, e = float(r[1 + k]), float(r[2 + k])
        if s >= lo and e <= hi:
            ev.append((s, e, nm, int(r[0])))
print("steady window %.2f ms; period %.4f ms per call" % (hi - lo, (hi - lo) / max(1, sum(1 for r in rec if lo <= r[1] < hi))))
for nm in names:
    d = [e - s for s, e, n2, _ in ev if n2 == nm]
    pts = sorted([(s, 1) for s, e, n2, _ in ev if n2 == nm] + [(e, -1) for s, e, n2, _ in ev if n2 == nm])
    cur = 0; last = lo; h = collections.Counter()
    for tt, dd in pts:
        h[cur] += tt - last; last = tt; cur += dd
    h[cur] += hi - last
    tot = sum(h.values())
    print("  %-12s mean %.4f ms; at once: mean %.2f  %s" % (nm, np.mean(d), sum(k * v for k, v in h.items()) / tot, {k: "%.0f %%" % (100 * v / tot) for k, v in sorted(h.items())}))
# trellis and front end together
def share(pred):
    pts = sorted([(s, 1) for s, e, n2, _ in ev if pred(n2)] + [(e, -1) for s, e, n2, _ in ev if pred(n2)])
    cur = 0; last = lo; h = collections.Counter()
    for tt, dd in pts:
        h[cur] += tt - last; last = tt; cur += dd
    h[cur] += hi - last
    tot = sum(h.values()); return {k: "%.0f %%" % (100 * v / tot) for k, v in sorted(h.items())}
print("  front-end kernels (scan, frame, finish) at once:", share(lambda n2: n2 in ("k_scan", "k_frame", "k_finish")))
print("  any kernel at once:", share(lambda n2: True))
# per pipeline: time from the end of a call's k_finish to the start of the same pipeline's next call (delivery, host, queueing)
byp = collections.defaultdict(list)
for r in rec:
    if lo <= r[1] <= hi:
        byp[int(r[0])].append(r)
gaps = []
for p, L2 in byp.items():
    for x, y in zip(L2, L2[1:]):
        gaps.append(float(y[1] - x[6]))
print("  a pipeline idles %.4f ms between k_finish and its next call's first kernel (mean; delivery + host turn-around); call latency first kernel -> k_finish end %.4f ms" % (np.mean(gaps), float(np.mean(rec[:, 6] - rec[:, 1]))))
