"""benchlib.rows -- the widened rows (802.11b, 802.11n, the 40 MHz HT receiver) and the multi-frame shard shape, each delivered, compared and gated against the reference."""
import json
import os
import sys
import time

import numpy as np

from benchlib.common import *  # noqa: F401,F403
from benchlib.common import _traffic_profile  # noqa: F401
from benchlib.cpu import _cpu_worker_11b, _cpu_worker_11n, host_cores  # noqa: F401


def bench_11b(torch, sora_amd, dev, ncaps=8192, reps=5, cpu=True, rate_kbps=1000):
    """Row f4 (802.11b receive graph): `ncaps` 44 MHz captures of one 1 Mbps DBPSK frame each (the modulator output recorded
    in tests/golden/refgraph_11b.npz, or a 500-byte frame from the compiled reference modulator when that library is
    here), noise added on the device.  A streaming integer path: 4 B per sample against the HBM roofline; the reference's
    own 11b graph is timed on one host core beside it when oracle/_ref is present."""
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if g.available():
        s8 = g.tx11b(np.random.default_rng(11).integers(0, 256, 500 if rate_kbps == 1000 else 1500).astype(np.uint8).tobytes(), rate_kbps); what = "500-byte MPDU" if rate_kbps == 1000 else "1500-byte MPDU"
    elif rate_kbps != 1000:
        return {"skipped": "needs oracle/_ref/libsora_refgraph.so (the capture comes from the reference's modulator)"}
    else:
        s8 = np.load(os.path.join(ROOT, "tests", "golden", "refgraph_11b.npz"))["tx_2"]; what = "40-byte MPDU (recorded modulator output)"
    n = (len(s8) + 1200 + 2800 + 27) // 28 * 28
    base = np.zeros((n, 2), np.int16); base[1200:1200 + len(s8)] = s8.astype(np.int16) << 8
    b = torch.from_numpy(base).to(dev).to(torch.float32)
    gen = torch.Generator(device=dev); gen.manual_seed(1102)
    iq = torch.empty((ncaps, n, 2), dtype=torch.int16, device=dev)
    for i in range(0, ncaps, 64):
        k = min(64, ncaps - i)
        iq[i:i + k] = (b[None] + 40.0 * torch.randn((k, n, 2), generator=gen, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    descs = sora_amd.Rx.captures([(i * n, n, i) for i in range(ncaps)])
    rx = sora_amd.Rx11b(ncaps, ncaps * n, max_frames_per_capture=4)
    flat = iq.view(-1, 2)
    torch.cuda.synchronize()                                            # the handle's stream does not follow torch's
    rx.wait_for_producer = False
    depth = rx.calls_in_flight()
    mlen = 500 if rate_kbps == 1000 else 1500
    # the handle's default pass plan is automatic (sora_rx11b_set_single_pass = 2): it measures, on the device, how many captures a call's first pass
    # handed to the CCK instantiation and plans the following calls accordingly -- no hint from the host.  That is the row's number; the two fixed
    # plans are timed beside it.
    ms, delivery, first = timed_with_delivery(sora_amd, rx, lambda: rx.process_dev(flat, descs), depth, max(reps, 20), ncaps * 4, ncaps * (mlen + 4) + 4096)
    passes = {"automatic": round(ms, 3)}
    if rate_kbps != 1000:
        key = lambda rows: [(r["capture_id"], r["error_code"], r["end_sample"], r["length"], r["crc32"], r["mpdu"]) for r in rows]
        same = True
        for plan, name in ((0, "two_passes"), (1, "single_pass")):
            rx.synchronize(); rx.set_single_pass(plan)
            ms_p, _, first_p = timed_with_delivery(sora_amd, rx, lambda: rx.process_dev(flat, descs), depth, max(reps, 20), ncaps * 4, ncaps * (mlen + 4) + 4096)
            passes[name] = round(ms_p, 3); same = same and key(first_p) == key(first)
        passes["same_table"] = same
        rx.synchronize(); rx.set_single_pass(2)
    ok = sum(r["error_code"] == 1 for r in first)
    rx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):                                                 # one call at a time, for the record
        rx.wait(rx.process_dev(flat, descs))
    ms1 = (time.perf_counter() - t0) / 10 * 1e3
    out = {"workload": "%d captures x one %s frame, %s, long preamble (%d samples @44 MHz each), AWGN" % (ncaps, "1 Mbps DBPSK" if rate_kbps == 1000 else "%g Mbps CCK" % (rate_kbps / 1000.0), what, n),
           "ms": round(ms, 3), "ms_one_call_in_flight": round(ms1, 3), "calls_in_flight": depth, "msamples_per_s": round(ncaps * n / ms / 1e3, 1), "frames_ok": ok, "frames": ncaps,
           "bound": "hbm", "algorithmic_bytes": 4 * ncaps * n, "achieved": round(4.0 * ncaps * n / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
           "unit": "GB/s", "frac": round(4.0 * ncaps * n / (ms * 1e-3) / HBM_PEAK, 4), "delivery": delivery, "ms_by_kernel_plan": passes}
    if g.available():                                                   # the whole batch against the compiled reference graph, capture by capture
        from gpu_util import same_as_reference_11b
        host = iq.cpu().numpy()
        out["parity"] = reference_gate(first, ncaps, lambda i: g.rx11b(host[i], max_frames=4), same_as_reference_11b)
        del host
    else:
        out["parity"] = {"against": None, "captures_checked": 0, "ok": None, "note": "oracle/_ref/libsora_refgraph.so is not here"}
    if cpu and g.available():                                           # the reference's 11b graph on this box's host cores, side by side
        import multiprocessing as mp
        import tempfile
        cores = host_cores()
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "iq11b.npy"); np.save(path, iq[:8].cpu().numpy())
            with mp.get_context("spawn").Pool(cores) as pool:
                one = pool.apply(_cpu_worker_11b, ((path, 2.0),))
                allc = pool.map(_cpu_worker_11b, [(path, 4.0)] * cores)
        out["cpu_reference_msamples_per_s_one_core"] = round(one, 2)
        out["cpu_reference_msamples_per_s"] = round(sum(allc), 1); out["cpu_reference_cores"] = cores
    rx.close(); del iq, flat
    return out


def bench_11n(torch, sora_amd, dev, ncaps=8192, reps=5):
    """Row f1 (802.11n 2x2 receive graph): `ncaps` two-chain 40 MHz captures of one MCS 10 frame each (a 1000-byte MPDU from the
    compiled reference modulator when that library is here, else the recorded 150-byte one of tests/golden/refgraph_11n.npz)
    through a 2x2 channel with cross-talk, noise added on the device.  8 B per sample pair against the HBM roofline; the
    reference's own graph is timed on the host cores beside it when oracle/_ref is present."""
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if g.available():
        s0, s1 = g.tx11n(np.random.default_rng(12).integers(0, 256, 1000).astype(np.uint8).tobytes(), 10); what = "1000-byte MPDU"
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "refgraph_11n.npz")); s0, s1 = z["tx2_0"], z["tx2_1"]; what = "150-byte MPDU (recorded modulator output)"
    n = (len(s0) + 800 + 1200 + 27) // 28 * 28
    base = np.zeros((2, n, 2), np.float32)
    base[0, 800:800 + len(s0)] = s0 + 0.1 * s1; base[1, 800:800 + len(s0)] = s1 + 0.1 * s0
    b = torch.from_numpy(base).to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1103)
    iq = torch.empty((2, ncaps, n, 2), dtype=torch.int16, device=dev)
    for i in range(0, ncaps, 64):
        k = min(64, ncaps - i)
        for c in range(2):
            iq[c, i:i + k] = (b[c][None] + 20.0 * torch.randn((k, n, 2), generator=gen, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    descs = sora_amd.Rx.captures([(i * n, n, i) for i in range(ncaps)])
    rx = sora_amd.Rx11n(ncaps, ncaps * n, max_frames_per_capture=4)
    f0 = iq[0].view(-1, 2); f1 = iq[1].view(-1, 2)
    torch.cuda.synchronize()
    rx.wait_for_producer = False
    mlen = 1000 if g.available() else 150
    res = {}
    D11N = 8
    for lanes in (64, 16):                                                   # both trellis kernels (sora_rx11n_set_trellis), eight calls in flight
        rx.set_trellis(lanes); rx.set_depth(D11N)
        ms_, delivery_, first_ = timed_with_delivery(sora_amd, rx, lambda: rx.process_dev(f0, f1, descs), D11N, max(reps, 36), ncaps * 4, ncaps * (mlen + 4) + 4096)
        res[lanes] = (ms_, delivery_, first_)
    best = min(res, key=lambda l: res[l][0])
    ms, delivery, first = res[best]
    ok = sum(r["error_code"] == 1 for r in first)
    one_by = {}; win_rows = None
    for lanes in (64, 16, 1):                                                # a lone call: each trellis kernel alone on the chip (1 = the window-parallel form, round 6)
        rx.set_trellis(lanes); rx.set_depth(1)
        if lanes == 1:
            win_rows = rx.results(ticket=rx.process_dev(f0, f1, descs))
        for _ in range(3):
            rx.wait(rx.process_dev(f0, f1, descs))
        t0 = time.perf_counter()
        for _ in range(10):
            rx.wait(rx.process_dev(f0, f1, descs))
        one_by[{64: "k_viterbi11n", 16: "k_viterbi16_11n", 1: "k_viterbi16w_11n"}[lanes]] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
    ms1 = min(one_by.values())
    rx.set_trellis(best)
    out = {"workload": "%d two-chain captures x one MCS 10 frame, %s (%d samples @40 MHz per chain each), 2x2 cross-talk, AWGN" % (ncaps, what, n),
           "ms": round(ms, 3), "ms_one_call_in_flight": round(ms1, 3), "ms_one_call_in_flight_by_trellis_kernel": one_by, "calls_in_flight": D11N, "trellis_kernel": {64: "k_viterbi11n", 16: "k_viterbi16_11n"}[best],
           "ms_by_trellis_kernel": {"k_viterbi11n": round(res[64][0], 3), "k_viterbi16_11n": round(res[16][0], 3)},
           "msamples_per_s": round(ncaps * n / ms / 1e3, 1), "frames_ok": ok, "frames": ncaps,
           "bound": "hbm", "algorithmic_bytes": 8 * ncaps * n, "achieved": round(8.0 * ncaps * n / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
           "unit": "GB/s", "frac": round(8.0 * ncaps * n / (ms * 1e-3) / HBM_PEAK, 4), "delivery": delivery}
    if g.available():                                                        # the whole batch against the compiled reference graph, capture by capture
        from gpu_util import same_events_11n
        h0 = iq[0].cpu().numpy(); h1 = iq[1].cpu().numpy()
        out["parity"] = reference_gate(first, ncaps, lambda i: g.rx11n(h0[i], h1[i]), lambda got, want: same_events_11n(got, want, position="sample_index"))
        out["parity"]["both_trellis_kernels_same_table"] = [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in res[64][2]] == [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in res[16][2]]
        out["parity"]["window_parallel_trellis_same_table"] = [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in res[64][2]] == [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in win_rows]
        del h0, h1
    else:
        out["parity"] = {"against": None, "captures_checked": 0, "ok": None, "note": "oracle/_ref/libsora_refgraph.so is not here"}
    if g.available():
        import multiprocessing as mp
        import tempfile
        cores = host_cores()
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "iq11n.npz"); np.savez(path, a=iq[0, :8].cpu().numpy(), b=iq[1, :8].cpu().numpy())
            with mp.get_context("spawn").Pool(cores) as pool:
                one = pool.apply(_cpu_worker_11n, ((path, 2.0),))
                allc = pool.map(_cpu_worker_11n, [(path, 4.0)] * cores)
        out["cpu_reference_msamples_per_s_one_core"] = round(one, 2)
        out["cpu_reference_msamples_per_s"] = round(sum(allc), 1); out["cpu_reference_cores"] = cores
    rx.close(); del iq, f0, f1
    return out


def bench_ht40(torch, sora_amd, dev, nframes=4096, trellis=(64, 16)):
    """BASELINE configs[3] (802.11n 2x2 40 MHz HT: 128-point FFT, MMSE detection, one decoder per spatial stream) on RAW CAPTURES -- parity
    unpinned for the 40 MHz extension, the reference has no such receiver; its own 20 MHz front-end bricks find
    and parse the frames.  `nframes` two-chain 40 MHz captures of one HT-mixed frame each: legacy preamble + HT-SIG + HT-STF + 2 HT-LTF +
    data, MCS 14 (64-QAM 3/4 on both streams), a 1500-byte PSDU per stream, from the numpy model of the format (oracle/py_ht40.py tx_frame;
    its preamble is pinned through the restated reference receiver) through a 2x2 channel with cross-talk; noise added on the device.
    sora_ht40_process_captures_dev: carrier sense, L-LTF, L-SIG / HT-SIG, CFO and noise variance, then the data field."""
    from oracle import py_ht40 as m
    rng = np.random.default_rng(40)
    ps = [m.add_fcs(rng.integers(0, 256, 1496, dtype=np.uint8).tobytes()) for _ in range(2)]
    x, nsym, pre = m.tx_frame(ps, 14)
    H = np.array([[1.0, 0.3j], [0.25, 0.9 * np.exp(0.7j)]])
    y = (H @ x) * 250.0
    lead = 400
    n = (lead + y.shape[1] + 600 + 27) // 28 * 28
    base = np.zeros((2, n, 2), np.float32); base[:, lead:lead + y.shape[1], 0] = y.real; base[:, lead:lead + y.shape[1], 1] = y.imag
    b = torch.from_numpy(base).to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(4040)
    iq = torch.empty((2, nframes, n, 2), dtype=torch.int16, device=dev)
    sigma = 12.0
    for i in range(0, nframes, 64):
        k = min(64, nframes - i)
        for c in range(2):
            iq[c, i:i + k] = (b[c][None] + sigma * torch.randn((k, n, 2), generator=gen, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    caps = sora_amd.Rx.captures([(i * n, n, i) for i in range(nframes)])
    rx = sora_amd.RxHt40(nframes, nframes * 2 * (nsym * 648 + 64))
    f0 = iq[0].view(-1, 2); f1 = iq[1].view(-1, 2)
    torch.cuda.synchronize()
    rx.wait_for_producer = False
    depth = rx.calls_in_flight()
    res = {}
    for lanes in trellis:
        rx.set_trellis(lanes)
        res[lanes] = timed_with_delivery(sora_amd, rx, lambda: rx.process_captures_dev(f0, f1, caps, max_frames_per_capture=2), depth, 20, 4 * nframes, 2 * nframes * 1500 + 4096)   # (room for two rows per event the captures could hold)
    best = min(res, key=lambda l: res[l][0])
    ms, delivery, first = res[best]
    ok = sum(r["error_code"] == 1 and r["mpdu"] == ps[r["stream"]] and r["rate_kbps"] == 14 for r in first)
    # the data field alone (the caller supplies what the front end would find): sora_ht40_process_dev
    descs = sora_amd.RxHt40.frames([(i * n + lead + pre, 6, 2, 1500, 1500, 0, 2 * sigma * sigma / 128.0, i) for i in range(nframes)])
    rx.set_trellis(best)
    ms_df, _, first_df = timed_with_delivery(sora_amd, rx, lambda: rx.process_dev(f0, f1, descs), depth, 20, 2 * nframes, 2 * nframes * 1500 + 4096)
    ok_df = sum(r["error_code"] == 1 and r["mpdu"] == ps[r["stream"]] for r in first_df)
    samples = nframes * n                                                    # per chain, 40 MHz: the whole capture is input now
    alg = 8.0 * samples + 2.0 * 1500 * nframes                               # both chains read once + the decoded PSDUs
    return {"workload": "%d two-chain 40 MHz captures x one HT-mixed frame, MCS 14 (64-QAM 3/4 on both streams), 1500-byte PSDU per stream (%d data symbols; %d samples per chain and capture), 2x2 cross-talk, AWGN; front end + unbiased MMSE on the estimated noise variance" % (nframes, nsym, n),
            "parity": "unpinned for the 40 MHz extension (the reference has no 40 MHz / MMSE / per-stream-decoder receiver): loop-back against oracle/py_ht40.py; the front end is the reference's 20 MHz bricks (pinned), the model's preamble is pinned through the restated reference receiver (tests/test_ht40_preamble_model.py)",
            "ms": round(ms, 3), "calls_in_flight": depth, "trellis_kernel": {64: "k_viterbi11n", 16: "k_viterbi16_11n"}[best],
            "ms_by_trellis_kernel": {{64: "k_viterbi11n", 16: "k_viterbi16_11n"}[l]: round(res[l][0], 3) for l in res},
            "ms_data_field_only": round(ms_df, 3), "psdus_ok_data_field_only": ok_df,
            "msamples_per_s": round(samples / ms / 1e3, 1), "decoded_mbit_per_s": round(2 * 1500 * 8 * nframes / ms / 1e3, 1),
            "psdus_ok": ok, "psdus": 2 * nframes, "delivery": delivery, "bound": "hbm", "algorithmic_bytes": int(alg), "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(alg / (ms * 1e-3) / HBM_PEAK, 4)}


def bench_shard_shape(torch, sora_amd, dev, oracle, ncaps=32, nframes=16, reps=60):
    """SURVEY section 8(d) config 5's per-GPU share (BASELINE configs[4]): 32 captures of 16 frames each (1500 bytes at 54 Mbps, the headline's frames back to back,
    160 samples of silence between them) -- 512 frames per call, far too few to fill the chip with a frame per wave, and k_scan walks each capture's sixteen frames
    one after the other.  One call in flight and eight, every call's rows checked against the compiled reference graph over the WHOLE capture (event for event,
    MPDU bytes included); kernel times of a lone call; the HBM roofline with the headline's bytes per sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import same_as_reference_graph, same_results
    from oracle.pyoracle import ReferenceGraph
    iq, _, _ = make_workload(oracle, ncaps * nframes, seed0=5151)
    caps = iq.reshape(ncaps, nframes * CAPTURE_SAMPLES, 2)
    g = ReferenceGraph(); have_ref = g.available()
    d_iq = torch.from_numpy(iq).to(dev)
    descs = sora_amd.Rx.captures([(i * nframes * CAPTURE_SAMPLES, nframes * CAPTURE_SAMPLES, i) for i in range(ncaps)])
    rx = sora_amd.Rx(max_captures=ncaps, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=nframes + 2)
    out = {"workload": "%d captures x %d frames of %d bytes at 54 Mbps, %d samples @20 MHz per capture" % (ncaps, nframes, MPDU_LEN, nframes * CAPTURE_SAMPLES),
           "frames_per_call": ncaps * nframes}
    # parity gate: every capture, whole table
    rx.set_depth(1)
    res = rx.results(ticket=rx.process_dev(d_iq, descs))
    ok = len(res) == ncaps * nframes; why = ""
    for i in range(ncaps):
        rows = [r for r in res if r["capture_id"] == i]
        if have_ref:
            o_, w_ = same_as_reference_graph(rows, g.rx11a(np.repeat(caps[i], 2, axis=0), max_frames=nframes + 4))
        else:
            want = [dict(r, capture_id=i) for r in oracle.rx_capture(caps[i], 20)]
            o_, w_ = same_results(rows, want)
        if not o_:
            ok = False; why = why or "capture %d: %s" % (i, w_)
    out["parity"] = {"against": "reference" if have_ref else "port", "captures_checked": ncaps, "frames": len(res), "frames_ok": sum(r["error_code"] == 1 for r in res), "ok": bool(ok), "why": why}
    samples = ncaps * nframes * FRAME_SAMPLES
    by = {}
    for depth in (1, 8):
        rx.set_depth(depth); rx.flush()
        chains = "%s | %s" % ({1: "k_frame", 3: "k_sym_front+k_track_lds+k_sym_back"}[rx.front()], TRELLIS_NAMES[rx.trellis()])
        for _ in range(depth + 2):
            rx.process_dev(d_iq, descs)
        rx.flush()
        n = reps * depth
        torch.cuda.synchronize(); t0 = time.perf_counter(); tickets = []
        for _ in range(n):
            tickets.append(rx.process_dev(d_iq, descs))
            if len(tickets) >= depth:
                rx.wait(tickets.pop(0))
        for t in tickets:
            rx.wait(t)
        ms = (time.perf_counter() - t0) / n * 1e3
        by["calls_in_flight_%d" % depth] = {"ms_per_call": round(ms, 4), "msamples_per_s": round(samples / ms / 1e3, 1), "kernels": chains + " (the library's choice)",
                                            "hbm_frac": round(samples * ALG_BYTES_PER_SAMPLE / (ms * 1e-3) / HBM_PEAK, 5)}
    # the same eight calls in flight with every pipeline's chain replayed as ONE hipGraph launch (sora_rx_set_graph: the calls are identical -- same buffer, same capture
    # set): what is left of the host's share when a call is one enqueue instead of five
    rx.set_depth(8); rx.set_graph(1); rx.flush()
    for _ in range(24):
        rx.process_dev(d_iq, descs)
    rx.flush()
    n = reps * 8
    torch.cuda.synchronize(); t0 = time.perf_counter(); tickets = []
    for _ in range(n):
        tickets.append(rx.process_dev(d_iq, descs))
        if len(tickets) >= 8:
            rx.wait(tickets.pop(0))
    for t in tickets:
        rx.wait(t)
    ms = (time.perf_counter() - t0) / n * 1e3
    g_res = rx.results(ticket=rx.process_dev(d_iq, descs))
    by["calls_in_flight_8_graph_replay"] = {"ms_per_call": round(ms, 4), "msamples_per_s": round(samples / ms / 1e3, 1), "hbm_frac": round(samples * ALG_BYTES_PER_SAMPLE / (ms * 1e-3) / HBM_PEAK, 5),
                                             "same_rows_as_the_checked_call": [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in g_res] == [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in res]}
    rx.set_graph(0)
    rx.set_depth(1); rx.flush(); rx.set_profiling(True)
    for _ in range(10):
        rx.wait(rx.process_dev(d_iq, descs))
    rx.flush(); out["kernel_ms_one_call_in_flight"] = {k: round(v, 4) for k, v in rx.kernel_times().items()}; rx.set_profiling(False)
    # the round-4 kernels on the same shape, one call in flight, for the record
    rx.set_front(1); rx.set_trellis(64); rx.flush()
    for _ in range(3):
        rx.wait(rx.process_dev(d_iq, descs))
    t0 = time.perf_counter()
    for _ in range(reps):
        rx.wait(rx.process_dev(d_iq, descs))
    out["one_call_in_flight_round4_kernels_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    out["window_trellis_record"] = rx.window_stats()
    rx.close()
    out.update(by)
    out["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_call": int(samples * ALG_BYTES_PER_SAMPLE), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                       "achieved": round(samples * ALG_BYTES_PER_SAMPLE / (by["calls_in_flight_8"]["ms_per_call"] * 1e-3) / 1e9, 2), "frac": by["calls_in_flight_8"]["hbm_frac"]}
    return out
