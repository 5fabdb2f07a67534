"""dbg_arrays.py [out.npz] -- on the GPU box: run fsample-6 through the library named by SORA_HIP_LIB (or the default) and dump the arrays
between the kernels of the call (sora_internal_rx_arrays).  Run once per build variant and compare the files (tools/dbg_compare.py)."""

# (round 5) the probe hooks live in the TOOLS variant of the library only: build it once and load it
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if not _os.environ.get("SORA_HIP_LIB"):
    from sora_amd import build as _b
    _v = _os.path.join(_os.path.dirname(_b.LIB), "variants", "tools.so")
    _os.environ["SORA_HIP_LIB"] = _v if _os.path.exists(_v) else _b.build_variant("tools", ["SORA_TOOLS"])

import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sora_amd
from sora_amd import capi

g = np.load(os.path.join(ROOT, "tests", "golden", "fsample6_40mhz_i8.npz"))
iq = g["iq_i8"].astype(np.int16) << 8
n = len(iq) // 28 * 28
rx = sora_amd.Rx(1, n, sample_rate_mhz=40, max_frames_per_capture=2)
d = torch.from_numpy(np.ascontiguousarray(iq[:n])).cuda()
rx.process_dev(d, [(0, n, 0)])
res = rx.results()
print([(r["error_code"], r["rate_kbps"], r["length"], r["nsym"]) for r in res])
L = capi.load()
ptrs = (ctypes.c_void_p * 9)(); slots = ctypes.c_uint32(0); nrows = ctypes.c_uint32(0)
L.sora_internal_rx_arrays.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
assert L.sora_internal_rx_arrays(rx._h, ptrs, ctypes.byref(slots), ctypes.byref(nrows)) == 0
S = slots.value


def grab(ptr, nbytes):
    if not ptr:
        return np.zeros(0, np.uint8)
    h = np.zeros(nbytes, np.uint8)
    assert L.sora_hip_memcpy_d2h(h.ctypes.data, ctypes.c_void_p(ptr), nbytes) == 0
    return h
out = {"frames": grab(ptrs[0], 64 * nrows.value).view(np.uint32).reshape(-1, 16), "slot_row": grab(ptrs[1], 4 * S).view(np.uint32),
       "eq": grab(ptrs[2], 256 * S).view(np.int16).reshape(S, 64, 2), "track": grab(ptrs[3], 8 * S).view(np.int16).reshape(S, 4),
       "soft": grab(ptrs[4], 108 * S), "slots": np.array([S])}
np.savez(sys.argv[1] if len(sys.argv) > 1 else "/tmp/dbg.npz", **out)
fr = out["frames"][0]
print("row0: capture", fr[0], "start", fr[1], "end", fr[2], "err", hex(fr[3]), "rate", fr[4], "len/nsym", fr[5] & 0xFFFF, fr[5] >> 16, "cr/nb", fr[6] & 0xFFFF, fr[6] >> 16, "slot0", fr[7])
sr = out["slot_row"]; own = np.nonzero(sr != 0xFFFFFFFF)[0]
print("slots", S, "owned", len(own), own[:5], own[-5:] if len(own) else "")
print("eq[slot0+1][:8]", out["eq"][fr[7] + 1][:8].tolist() if len(out["eq"]) else None)
print("track[slot0+1..+4]", out["track"][fr[7] + 1:fr[7] + 5].tolist() if len(out["track"]) else None)
print("soft first 24 bytes of the frame", out["soft"][fr[7] * 108:fr[7] * 108 + 24].tolist())
