"""Developer probe: where does k_scan spend its cycles? (build with -DSORA_SCAN_PROBE)"""
import ctypes, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
so = "/tmp/libsora_probe.so"
src = [os.path.join(ROOT, "sora_amd", "csrc", f) for f in ("k_scan.hip", "k_rx.hip", "k_stage.hip", "sora_hip.cpp")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-DSORA_SCAN_PROBE"] + src + ["-o", so])
import torch
from sora_amd import build as b
b.LIB = so
import sora_amd
from oracle.pyoracle import Oracle
from gpu_util import make_capture, batch
o = Oracle()
caps = [make_capture(o, 54000, 1496, seed=i, rate_mhz=20, sigma=300, tail=320)[0] for i in range(64)]
iq, descs = batch(caps)
rx = sora_amd.Rx(len(caps), len(iq), sample_rate_mhz=20)
d = torch.from_numpy(iq).cuda()
for _ in range(3):
    rx.process_dev(d, descs); rx.flush()
rx.set_profiling(True); rx.process_dev(d, descs); rx.flush(); print(rx.kernel_times())
L = ctypes.CDLL(so)
buf = (ctypes.c_ulonglong * 16)()
L.sora_debug_scan_probe(buf, 1)
rx.process_dev(d, descs); rx.flush()
L.sora_debug_scan_probe(buf, 0)
names = ["carrier sense (all bursts)", "T11aLTS", "SIGNAL chain"]
for i, n in enumerate(names):
    print("%-28s %9d cycles" % (n, buf[i]))
