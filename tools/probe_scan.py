"""Developer probe: where does k_scan spend its cycles?  Builds the library with -DSORA_SCAN_PROBE (clock64 around the regions of k_scan,
block 0 only) and runs it on the bench workload (one 54 Mbps 1500-byte frame per capture at sample 0, 160 samples of silence behind it)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sora_amd import build as b
so = os.path.join(ROOT, "sora_amd", "lib", "variants", "scan_probe.so")
stale = os.path.exists(so) and any(os.path.getmtime(os.path.join(b.CSRC, f)) > os.path.getmtime(so) for f in b.SOURCES + b.HEADERS)
if not os.path.exists(so) or stale or "--rebuild" in sys.argv:                     # (build it where there is no GPU to pay for: python tools/probe_scan.py --build-only)
    so = b.build_variant("scan_probe", ["SORA_SCAN_PROBE"])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["SORA_HIP_LIB"] = so
import torch
import sora_amd
from oracle.pyoracle import Oracle
import bench
o = Oracle()
if "--fsample6" in sys.argv:                                               # the single capture of BASELINE configs[1] instead (40 MHz, one 6 Mbps frame of 465 symbols)
    g = np.load(os.path.join(ROOT, "tests", "golden", "fsample6_40mhz_i8.npz"))
    iq = g["iq_i8"].astype(np.int16) << 8
    iq = np.ascontiguousarray(iq[:len(iq) // 28 * 28]); n = 1; descs = [(0, len(iq), 0)]
    rx = sora_amd.Rx(1, len(iq), sample_rate_mhz=40, max_frames_per_capture=2)
elif "--shard" in sys.argv:                                                # BASELINE configs[4]'s per-GPU share: 32 captures x 16 frames back to back (the probe sums capture 0's sixteen frames)
    from benchlib.common import CAPTURE_SAMPLES
    iq, _, _ = bench.make_workload(o, 32 * 16, seed0=5151)
    n = 32; descs = [(i * 16 * CAPTURE_SAMPLES, 16 * CAPTURE_SAMPLES, i) for i in range(32)]
    rx = sora_amd.Rx(32, len(iq), sample_rate_mhz=20, max_frames_per_capture=18)
else:
    n = 4096
    iq, descs, _ = bench.make_workload(o, n, 0, distinct=64)
    rx = sora_amd.Rx(n, len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
rx.set_depth(1)
d = torch.from_numpy(iq).cuda()
for _ in range(3):
    rx.process_dev(d, descs); rx.flush()
rx.set_profiling(True); rx.process_dev(d, descs); rx.flush(); print(rx.kernel_times()); rx.set_profiling(False)
L = ctypes.CDLL(so)
buf = (ctypes.c_ulonglong * 16)()
L.sora_debug_scan_probe(buf, 1)
rx.process_dev(d, descs); rx.flush()
L.sora_debug_scan_probe(buf, 0)
names = ["establish_sync", "loop top .. start of a carrier-sense pass", "header section (LTS + SIGNAL)", "carrier-sense passes (<= 16 bursts)", "check_sync passes (<= 16 bursts)", "whole kernel (capture 0)", "ring fill for the header", "frame row + reset"]
for i, nme in enumerate(names):
    print("%-32s %9d ticks  %5d times" % (nme, buf[i], buf[8 + i]))
