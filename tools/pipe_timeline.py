"""Developer probe: where does k_pipe (sora_rx_set_front(4), k_rx.hip) spend a single capture's time?  Builds the tools variant with -DSORA_DBG_PIPE_TIMELINE (10 ns
stamps of the launch's hand-offs, written behind its hand-off words) and prints them for fsample-6 as one capture, relative to front workgroup 0's start.
Run on the GPU box: python tools/pipe_timeline.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sora_amd import build as b                                             # noqa: E402
so = os.path.join(ROOT, "sora_amd", "lib", "variants", "pipe_timeline.so")
if not os.path.exists(so) or "--rebuild" in sys.argv:
    so = b.build_variant("pipe_timeline", ["SORA_TOOLS", "SORA_DBG_PIPE_TIMELINE"])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["SORA_HIP_LIB"] = so
import torch                                                                # noqa: E402
import sora_amd                                                             # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "fsample6_40mhz_i8.npz"))
iq = g["iq_i8"].astype(np.int16) << 8
iq = np.ascontiguousarray(iq[:len(iq) // 28 * 28])
rx = sora_amd.Rx(1, len(iq), sample_rate_mhz=40, max_frames_per_capture=2)
rx.set_depth(1)
assert rx.front() == 4, rx.front()
d = torch.from_numpy(iq).cuda()
L = ctypes.CDLL(so)
L.sora_internal_rx_arrays.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
L.sora_hip_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
for rep in range(4):
    rx.wait(rx.process_dev(d, [(0, len(iq), 0)]))
    ptrs = (ctypes.c_void_p * 9)(); slots = ctypes.c_uint32(); nrows = ctypes.c_uint32()
    assert L.sora_internal_rx_arrays(rx._h, ptrs, ctypes.byref(slots), ctypes.byref(nrows)) == 0
    st = np.zeros(1024, np.uint32)
    assert L.sora_hip_memcpy_d2h(st.ctypes.data, ctypes.c_void_p(ptrs[8]), 4096) == 0
    t0 = int(st[0])
    us = lambda i: ((int(st[i]) - t0) & 0xFFFFFFFF) / 100.0                 # noqa: E731
    names = {8: "front workgroup 0 published", 1: "tracker: front flags seen + acquire", 2: "chain starts", 3: "chain ends", 4: "helper 1 ends", 5: "helper 2 ends", 6: "helper 3 ends"}
    if rep == 3:
        for i in (8, 1, 2, 3, 4, 5, 6):
            print("%-40s %8.2f us" % (names[i], us(i)))
        for w in range(96):
            if st[16 + 2 * w]:                                           # (a wave without units leaves no start stamp)
                print("trellis wave %2d: starts %8.2f us, ends %8.2f us" % (w, us(16 + 2 * w), us(17 + 2 * w)))
    st[:] = 0
rx.close()
