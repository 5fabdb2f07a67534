"""Developer check: k_pipe (a lone capture, hand-offs between the workgroups of one launch) while ANOTHER handle keeps the chip full -- eight 4096-capture calls in flight,
submitted by a second thread.  Every lone-capture call's MPDU is compared; reported: calls, wrong results, SORA_E_INTERNAL_TIMEOUT rows, the latency distribution.
Run on the GPU box: python tools/pipe_under_load.py [--calls 3000]"""
import argparse
import hashlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=3000)
    ap.add_argument("--front", type=int, default=0)
    a = ap.parse_args()
    import torch
    import sora_amd
    import bench
    from oracle.pyoracle import Oracle
    o = Oracle()
    iq, descs, _ = bench.make_workload(o, 4096, seed0=0)
    d_big = torch.from_numpy(iq).cuda(); dd = sora_amd.Rx.captures(descs)
    big = sora_amd.Rx(max_captures=4096, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    big.set_depth(8)
    g = np.load(os.path.join(ROOT, "tests", "golden", "fsample6_40mhz_i8.npz"))
    cap = g["iq_i8"].astype(np.int16) << 8
    cap = np.ascontiguousarray(cap[:len(cap) // 28 * 28])
    d_cap = torch.from_numpy(cap).cuda()
    small = sora_amd.Rx(1, len(cap), sample_rate_mhz=40, max_frames_per_capture=2)
    small.set_depth(1)
    if a.front:
        small.set_front(a.front)
    want = "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    stop = threading.Event(); nbig = [0]

    def load():
        tickets = []
        while not stop.is_set():
            tickets.append(big.process_dev(d_big, dd))
            if len(tickets) >= 8:
                big.wait(tickets.pop(0)); nbig[0] += 1
        for t in tickets:
            big.wait(t)

    th = threading.Thread(target=load); th.start()
    time.sleep(0.2)
    bad = timeouts = 0; ts = []
    for _ in range(a.calls):
        t0 = time.perf_counter()
        res = small.results(ticket=small.process_dev(d_cap, [(0, len(cap), 0)]))
        ts.append(time.perf_counter() - t0)
        if len(res) != 1 or (res[0]["error_code"] & 0xFFFFFFFF) == 0x8000F001:
            timeouts += 1
        elif res[0]["error_code"] != 1 or hashlib.sha256(res[0]["mpdu"]).hexdigest() != want:
            bad += 1
    stop.set(); th.join()
    ts = np.array(ts) * 1e3
    print("front %d: %d lone-capture calls beside %d batch calls: %d wrong, %d timed out; ms per call (incl. results): median %.3f, p99 %.3f, max %.3f; proof record %s"
          % (small.front(), a.calls, nbig[0], bad, timeouts, np.median(ts), np.percentile(ts, 99), ts.max(), small.window_stats()))
    small.close(); big.close()
    return 1 if bad or timeouts else 0


if __name__ == "__main__":
    sys.exit(main())
