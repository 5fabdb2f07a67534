"""Round 5: what the window-parallel trellis (sora_rx_set_trellis(1), k_vitwin.hip) buys where few frames are in flight.
  (a) fsample-6 as one capture: process -> wait by trellis kernel, with the library's per-kernel times;
  (b) the 4096-frame batch of BASELINE configs[2]: ms per call with 1, 2, 3, 4, 8 calls in flight, by trellis kernel (no delivery);
      the window-parallel rows are checked against the serial kernel's rows and MPDU bytes.
Usage (GPU box): python tools/r05_window_latency.py [--frames 4096] > gpurun_out/r05_window_latency.json"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--profile-single", type=int, default=0, help="only N calls of fsample-6 as one capture with the library's automatic kernel choice (for rocprofv3 --kernel-trace --stats)")
    ap.add_argument("--profile-batch", type=int, default=0, help="only N lone window-parallel calls of the batch (for rocprofv3 --kernel-trace --stats)")
    a = ap.parse_args()
    import torch
    import sora_amd
    import bench
    from oracle.pyoracle import Oracle
    o = Oracle()
    dev = "cuda:0"
    out = {}
    names = {64: "k_viterbi", 16: "k_viterbi16", 1: "k_viterbi16w"}
    if a.profile_single:
        g = np.load(os.path.join(ROOT, "tests", "golden", "fsample6_40mhz_i8.npz"))
        iq40 = g["iq_i8"].astype(np.int16) << 8
        n = len(iq40) // 28 * 28
        d = torch.from_numpy(np.ascontiguousarray(iq40[:n])).to(dev)
        rx = sora_amd.Rx(1, n, sample_rate_mhz=40, max_frames_per_capture=2)
        rx.set_depth(1)
        for _ in range(a.profile_single):
            rx.wait(rx.process_dev(d, [(0, n, 0)]))
        print(json.dumps({"front": rx.front(), "trellis": rx.trellis()})); rx.close()
        return
    if a.profile_batch:
        iq, descs, _ = bench.make_workload(o, a.frames, seed0=0)
        d_iq = torch.from_numpy(iq).to(dev); dd = sora_amd.Rx.captures(descs)
        rx = sora_amd.Rx(max_captures=a.frames, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
        rx.set_depth(1); rx.set_trellis(1)
        for _ in range(a.profile_batch):
            rx.wait(rx.process_dev(d_iq, dd))
        print(json.dumps(rx.window_stats())); rx.close()
        return
    # (a) one capture
    g = np.load(os.path.join(ROOT, "tests", "golden", "fsample6_40mhz_i8.npz"))
    iq40 = g["iq_i8"].astype(np.int16) << 8
    n = len(iq40) // 28 * 28
    d = torch.from_numpy(np.ascontiguousarray(iq40[:n])).to(dev)
    rx = sora_amd.Rx(1, n, sample_rate_mhz=40, max_frames_per_capture=2)
    rx.set_depth(1)
    one = [(0, n, 0)]
    single = {}
    for front, lanes in ((1, 64), (1, 16), (1, 1), (3, 64), (3, 1), (4, 1)):
        rx.set_front(front); rx.set_trellis(lanes); rx.flush()
        t = rx.process_dev(d, one); res = rx.results(ticket=t)
        ok = len(res) == 1 and res[0]["error_code"] == 1 and hashlib.sha256(res[0]["mpdu"]).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
        for _ in range(5):
            rx.wait(rx.process_dev(d, one))
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); rx.wait(rx.process_dev(d, one)); ts.append(time.perf_counter() - t0)
        rx.set_graph(1)
        for _ in range(5):
            rx.wait(rx.process_dev(d, one))
        tg = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); rx.wait(rx.process_dev(d, one)); tg.append(time.perf_counter() - t0)
        rx.set_graph(0)
        rx.set_profiling(True)
        for _ in range(10):
            rx.wait(rx.process_dev(d, one))
        rx.flush(); kt = rx.kernel_times(); rx.set_profiling(False)
        assert rx.front() == front, (rx.front(), front)
        single[("k_pipe (k_sym_front, k_track_lds, k_sym_back, k_viterbi16w in one launch)" if front == 4 else ("k_frame" if front == 1 else "k_sym_front+k_track_lds+k_sym_back") + " | " + names[lanes])] = {
            "decode_ms": round(float(np.median(ts)) * 1e3, 4), "min_ms": round(float(np.min(ts)) * 1e3, 4), "decode_ms_as_one_graph_launch": round(float(np.median(tg)) * 1e3, 4),
            "mpdu_sha256_ok": bool(ok), "kernel_ms": {k: round(v, 4) for k, v in kt.items()}}
    single["window_stats"] = rx.window_stats()
    rx.close()
    out["fsample6_single_capture"] = single
    # (b) the batch
    nfr = a.frames
    iq, descs, _ = bench.make_workload(o, nfr, seed0=0)
    d_iq = torch.from_numpy(iq).to(dev); dd = sora_amd.Rx.captures(descs)
    rx = sora_amd.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(1); rx.set_trellis(64)
    base = rx.results(ticket=rx.process_dev(d_iq, dd))
    key = lambda r: (r["capture_id"], r["error_code"], r["end_sample"], r["rate_kbps"], r["length"], r["crc32"], r["mpdu"])  # noqa: E731
    rx.set_trellis(1)
    got = rx.results(ticket=rx.process_dev(d_iq, dd))
    out["batch_windowed_equals_serial"] = bool([key(r) for r in got] == [key(r) for r in base]) and len(base) == nfr
    table = {}
    for lanes in (64, 16, 1):
        row = {}
        for depth in (1, 2, 3, 4, 8):
            rx.set_depth(depth); rx.set_trellis(lanes); rx.flush()
            for _ in range(depth + 2):
                rx.process_dev(d_iq, dd)
            rx.flush()
            ncalls = max(16, 6 * depth)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tickets = []
            for i in range(ncalls):
                tickets.append(rx.process_dev(d_iq, dd))
                if len(tickets) >= depth:
                    rx.wait(tickets.pop(0))
            for t in tickets:
                rx.wait(t)
            row[str(depth)] = round((time.perf_counter() - t0) / ncalls * 1e3, 4)
        if lanes == 1 or lanes == 64:
            rx.set_depth(1); rx.set_trellis(lanes); rx.flush(); rx.set_profiling(True)
            for _ in range(8):
                rx.wait(rx.process_dev(d_iq, dd))
            rx.flush(); row["kernel_ms_one_call_in_flight"] = {k: round(v, 4) for k, v in rx.kernel_times().items()}; rx.set_profiling(False)
        table[names[lanes]] = row
    out["batch_ms_per_call_by_calls_in_flight"] = table
    out["window_stats"] = rx.window_stats()
    rx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
