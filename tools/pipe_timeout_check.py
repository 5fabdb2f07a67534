"""Developer check of k_pipe's safety net (k_rx.hip): a build in which the front workgroups never publish their flags (-DSORA_DBG_PIPE_LOSE_FLAGS, tools variant) must
end every wait after its bound (20 ms; sora_rx_set_pipe_wait_us) and STILL deliver the reference's rows: the finishing kernel behind the launch makes the call's data
field again (k_rx.hip: k_win_redo_finish_pipe) -- no hang, no SORA_E_INTERNAL_TIMEOUT row, no wrong MPDU; the handle then keeps to the three-kernel chain.  Run on the
GPU box.  (The product build's own test of the same path: tests/test_gpu_pipe.py, with the bound at zero.)"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sora_amd import build as b                                             # noqa: E402
so = os.path.join(ROOT, "sora_amd", "lib", "variants", "pipe_lose_flags.so")
if not os.path.exists(so) or "--rebuild" in sys.argv:
    so = b.build_variant("pipe_lose_flags", ["SORA_TOOLS", "SORA_DBG_PIPE_LOSE_FLAGS"])
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
t0 = time.time()
res = rx.results(ticket=rx.process_dev(d, [(0, len(iq), 0)]))
dt = time.time() - t0
print("call took %.3f s; rows: %s" % (dt, [(hex(r["error_code"] & 0xFFFFFFFF), r["length"]) for r in res]))
assert 0.015 < dt < 2.0, dt
assert len(res) == 1 and res[0]["error_code"] == 1 and hashlib.sha256(res[0]["mpdu"]).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62", res
st = rx.pipe_stats()
assert st == {"calls_made_again": 1, "backoffs": 1}, st
assert rx.front() == 3                                                      # the handle leaves k_pipe alone for its next calls
res = rx.results(ticket=rx.process_dev(d, [(0, len(iq), 0)]))
assert len(res) == 1 and res[0]["error_code"] == 1, res
rx.close()
print("k_pipe gives up after its bound and the call still delivers the reference's frame: OK", st)
