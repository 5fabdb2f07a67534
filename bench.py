#!/usr/bin/env python3
"""bench.py -- headline benchmark: IQ Msamples/s through the 802.11a 54 Mbps RX PHY on MI355X.

Workload (BASELINE.json configs[2]): 4096 independent 20 MHz captures per GPU, each holding one 54 Mbps
frame with a 1500-byte MPDU (PLCP LENGTH 1500 -> 56 data symbols, 4880 samples) followed by 160 samples of
silence (capture = 5040 samples = 360 source bursts).  Synthetic IQ: fixed-seed payloads through the
restated reference transmitter (oracle/so_tx11a.c), AWGN at ~30/27 dB SNR on 3 of every 4 captures.
A "step" = one sora_rx_process_dev call over the whole batch, inputs resident in HBM.  `value` counts the
4880 frame samples per capture (19.99 Msamples per 4096 frames).

Multi-GPU (--gpus N, launched by torch.distributed.run): captures are the natural shard -- each rank runs
its own 4096-capture batch end to end (weak scaling), no data-path collective; one all-reduce of the
frame counters after the timed region is the only exchange.

Prints ONE JSON line on rank 0 (see the contract in the task description), with `roofline` for the
dominant kernel (HIP events on the library's own stream) and `cpu_baseline` (the oracle timed on one host core).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FRAMES_PER_GPU = 4096
MPDU_LEN = 1500            # incl. FCS
RATE_KBPS = 54000
FRAME_SAMPLES = 4880       # 160 STS + 160 LTS + 80 SIGNAL + 56*80 data @20 MHz
CAPTURE_SAMPLES = 5040     # + 160 silence; 360 source bursts of 14
ALG_BYTES_PER_SAMPLE = 4.0 + 216 / 8.0 / 80.0     # 4.3375
HBM_PEAK = 8.0e12
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r01_traffic.json")     # rocprofv3 --pmc summary (tools/collect_profiles.sh)


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (FETCH_SIZE x2 + WRITE_SIZE, see the file), or None."""
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        return t["kernels"][kernel]["hbm_bytes"] if t.get("frames_per_launch") == FRAMES_PER_GPU else None
    except (OSError, KeyError, ValueError):
        return None


def measured_valu(kernel=None):
    """Wave-level VALU instructions per launch (SQ_INSTS_VALU, same PMC summary): of `kernel`, or of the whole call."""
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        if t.get("frames_per_launch") != FRAMES_PER_GPU:
            return None
        return t["kernels"][kernel]["valu_insts"] if kernel else t["total_valu_insts_per_call"]
    except (OSError, KeyError, ValueError):
        return None


def make_workload(oracle, nframes, seed0, distinct=512):
    """-> (iq int16 [nframes*CAPTURE_SAMPLES, 2], descs, payloads)"""
    from gpu_util import pad_capture
    base, payloads = [], []
    for i in range(min(distinct, nframes)):
        rng = np.random.default_rng(0x5EED0000 + seed0 + i)
        mp = rng.integers(0, 256, MPDU_LEN - 4).astype(np.uint8).tobytes()
        cap = oracle.tx_capture(mp, RATE_KBPS, seed=1 + (seed0 + i) % 127, lead=0, tail=320, rate_mhz=20)
        cap = pad_capture(cap, 20)
        assert len(cap) == CAPTURE_SAMPLES, len(cap)
        base.append(cap); payloads.append(mp)
    iq = np.empty((nframes, CAPTURE_SAMPLES, 2), np.int16)
    rng = np.random.default_rng(seed0 + 77)
    for i in range(nframes):
        c = base[i % len(base)].astype(np.int32)
        k = i % 4
        if k:                                   # clean / ~30 dB / ~27 dB / ~30 dB
            sigma = (0, 300, 420, 300)[k]
            c = c + np.rint(rng.normal(0.0, sigma, c.shape)).astype(np.int32)
        iq[i] = np.clip(c, -32768, 32767)
    descs = [(i * CAPTURE_SAMPLES, CAPTURE_SAMPLES, i) for i in range(nframes)]
    return iq.reshape(-1, 2), descs, [payloads[i % len(base)] for i in range(nframes)]


def _cpu_worker(args):
    """One host process of the CPU baseline over its share of the captures (cycled) for `seconds`.  kind "reference":
    the reference's own brick graph compiled from its sources (oracle/_ref/libsora_refgraph.so; it takes the 40 MHz
    stream its harness reads, so every 20 MHz sample is doubled -- TDownSample2 drops the copies); kind "port": the
    scalar C restatement."""
    path, nframes, first, stride, seconds, kind = args
    x = np.load(path, mmap_mode="r").reshape(nframes, CAPTURE_SAMPLES, 2)
    caps = np.stack([np.array(x[(first + k * stride) % nframes]) for k in range(max(1, min(64, nframes // max(1, stride))))])
    if kind == "reference":
        from oracle.pyoracle import ReferenceGraph
        g = ReferenceGraph()
        caps = np.repeat(caps, 2, axis=1)                                 # input preparation, not timed
        run = lambda: g.rx11a_bench(caps)                                 # noqa: E731  (the loop over captures is inside the library)
    else:
        from oracle.pyoracle import Oracle
        o = Oracle()
        run = lambda: sum(int(len(r) == 1 and r[0]["error_code"] == 1) for r in (o.rx_capture(c, 20) for c in caps))  # noqa: E731
    run()                                                                # tables + page-in, untimed
    t0 = time.perf_counter(); n = 0; ok = 0
    while time.perf_counter() - t0 < seconds:
        ok += run()
        n += len(caps)
    return n, ok, time.perf_counter() - t0


def _cpu_worker_11b(args):
    """One host process of the 802.11b CPU baseline: the reference's own 11b graph over the sample captures for `seconds`."""
    path, seconds = args
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    sample = np.load(path)
    g.rx11b_bench(sample[:1])
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        g.rx11b_bench(sample); k += len(sample)
    return k * sample.shape[1] / (time.perf_counter() - t0) / 1e6


def _cpu_worker_11n(args):
    """One host process of the 802.11n CPU baseline: the reference's own 2x2 graph over the sample captures for `seconds`."""
    path, seconds = args
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    z = np.load(path); a = z["a"]; b = z["b"]
    g.rx11n_bench(a[:1], b[:1])
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        g.rx11n_bench(a, b); k += len(a)
    return k * a.shape[1] / (time.perf_counter() - t0) / 1e6


def host_cores():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 256))


def cpu_baseline(iq, nframes, budget_s=10.0):
    """The reference receive path on this box's host cores over a bounded sample of the same captures: one process per
    usable core (affinity and cgroup quota), each cycling through its share of the captures for about budget_s seconds.
    kind "reference" = the reference's own SSE brick graph (CreateDemodGraph11a_40M + the RxThread loop) compiled from
    its sources into oracle/_ref; where that library is absent, kind "port" = the scalar C restatement.  The other one
    and the single-process rates are reported beside it."""
    import multiprocessing as mp
    import tempfile
    from oracle.pyoracle import ReferenceGraph
    cores = host_cores()
    have_ref = ReferenceGraph().available()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "iq.npy")
        np.save(path, iq)
        with mp.get_context("spawn").Pool(cores) as pool:
            for kind, secs in ((("reference", budget_s),) if have_ref else ()) + (("port", budget_s if not have_ref else 4.0),):
                one = pool.apply(_cpu_worker, ((path, nframes, 0, 1, 2.0, kind),))
                res = pool.map(_cpu_worker, [(path, nframes, k, cores, secs, kind) for k in range(cores)])
                out[kind] = {"value": round(sum(r[0] * FRAME_SAMPLES / r[2] for r in res) / 1e6, 3),   # side by side: rates add
                             "single": round(one[0] * FRAME_SAMPLES / one[2] / 1e6, 4),
                             "n": sum(r[0] for r in res), "ok": sum(r[1] for r in res), "secs": secs}
    kind = "reference" if have_ref else "port"
    m = out[kind]
    what = ("the reference's brick graph compiled from its sources (oracle/_ref/libsora_refgraph.so, SSE)" if have_ref
            else "oracle/so_rx11a.c (scalar C restatement)")
    r = {"value": m["value"], "unit": "Msamples/s", "cores": cores, "kind": kind, "single_core_value": m["single"],
         "sample": "%d captures of this workload (cycled), %d processes x %.0f s, %s" % (m["n"], cores, m["secs"], what),
         "frames_ok": m["ok"], "frames_run": m["n"]}
    if have_ref:
        r["port_value"] = out["port"]["value"]; r["port_single_core_value"] = out["port"]["single"]
    return r


def valu_roofline(nframes, ms_step):
    """What actually bounds this path: vector-ALU issue.  Wave-level VALU instructions of one receive call (rocprofv3
    SQ_INSTS_VALU, profiles/r01_traffic.json) over the measured step time, against 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles
    per wave64 instruction (MI355X_MICROARCH.md: a wave64 VALU op issues over 2 cycles)."""
    n = measured_valu() if nframes == FRAMES_PER_GPU else None
    if not n:
        return None
    peak = 256 * 4 * 2.4e9 / 2
    ach = n / (ms_step * 1e-3)
    return {"insts_per_call": n, "achieved": round(ach / 1e9, 1), "peak": round(peak / 1e9, 1), "unit": "G wave-instr/s",
            "frac": round(ach / peak, 4), "dominant_kernel_insts": measured_valu("k_viterbi")}


def bench_ingest(torch, sora_amd, dev, nbytes=256 << 20, reps=20):
    """Row f3 (capture ingest): a 44 MHz RX_BLOCK dump resident in HBM -> de-framed, sign-fixed, resampled 40 MHz stream.
    A pure streaming kernel: algorithmic bytes = dump bytes read + samples written, against the HBM roofline."""
    flags = sora_amd.INGEST_RXBLOCK | sora_amd.INGEST_RAW14 | sora_amd.INGEST_44TO40
    raw = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
    n_out = sora_amd.ingest_count(nbytes, flags)
    for _ in range(3):
        out = sora_amd.ingest(raw, flags, sync=False)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        out = sora_amd.ingest(raw, flags, sync=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = nbytes + 4 * n_out
    del raw, out
    return {"workload": "%d MiB Sora RX_BLOCK dump @44 MHz -> de-frame + 14->16 bit + 44->40 MHz (%d samples out)" % (nbytes >> 20, n_out),
            "bound": "hbm", "ms": round(ms, 4), "algorithmic_bytes": alg, "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / HBM_PEAK, 4), "msamples_per_s_in": round(nbytes / 128 * 28 / ms / 1e3, 1)}


def bench_11b(torch, sora_amd, dev, ncaps=8192, reps=5):
    """Row f4 (802.11b receive graph): `ncaps` 44 MHz captures of one 1 Mbps DBPSK frame each (the modulator output recorded
    in tests/golden/refgraph_11b.npz, or a 500-byte frame from the compiled reference modulator when that library is
    here), noise added on the device.  A streaming integer path: 4 B per sample against the HBM roofline; the reference's
    own 11b graph is timed on one host core beside it when oracle/_ref is present."""
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if g.available():
        s8 = g.tx11b(np.random.default_rng(11).integers(0, 256, 500).astype(np.uint8).tobytes(), 1000); what = "500-byte MPDU"
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
    rx.process_dev(flat, descs); res = rx.results()
    ok = sum(r["error_code"] == 1 for r in res)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        rx.process_dev(flat, descs)
    rx.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    out = {"workload": "%d captures x one 1 Mbps DBPSK frame, %s, long preamble (%d samples @44 MHz each), AWGN" % (ncaps, what, n),
           "ms": round(ms, 3), "msamples_per_s": round(ncaps * n / ms / 1e3, 1), "frames_ok": ok, "frames": ncaps,
           "bound": "hbm", "algorithmic_bytes": 4 * ncaps * n, "achieved": round(4.0 * ncaps * n / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
           "unit": "GB/s", "frac": round(4.0 * ncaps * n / (ms * 1e-3) / HBM_PEAK, 4)}
    if g.available():                                                   # the reference's 11b graph on this box's host cores, side by side
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
    rx.process_dev(f0, f1, descs); res = rx.results()
    ok = sum(r["error_code"] == 1 for r in res)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        rx.process_dev(f0, f1, descs)
    rx.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    out = {"workload": "%d two-chain captures x one MCS 10 frame, %s (%d samples @40 MHz per chain each), 2x2 cross-talk, AWGN" % (ncaps, what, n),
           "ms": round(ms, 3), "msamples_per_s": round(ncaps * n / ms / 1e3, 1), "frames_ok": ok, "frames": ncaps,
           "bound": "hbm", "algorithmic_bytes": 8 * ncaps * n, "achieved": round(8.0 * ncaps * n / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
           "unit": "GB/s", "frac": round(8.0 * ncaps * n / (ms * 1e-3) / HBM_PEAK, 4)}
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


def bench_tx(torch, sora_amd, nframes=4096, reps=10):
    """Row f2 (transmitter): the same 4096 x 1500-byte 54 Mbps frames modulated on the GPU (COMPLEX8 @40 MHz out)."""
    rng = np.random.default_rng(0x5EED)
    mpdus = [bytes(rng.integers(0, 256, MPDU_LEN - 4).astype(np.uint8)) for _ in range(64)] * (nframes // 64)
    out, off = sora_amd.tx11a(mpdus, [RATE_KBPS] * nframes)                  # builds the device arrays; also the warm-up
    import ctypes
    from sora_amd import capi
    lens = torch.full((nframes,), MPDU_LEN - 4, dtype=torch.int32, device=out.device)
    rate = torch.full((nframes,), RATE_KBPS, dtype=torch.int32, device=out.device)
    seed = torch.full((nframes,), 0xFF, dtype=torch.uint8, device=out.device)
    moff = torch.arange(nframes, dtype=torch.int32, device=out.device) * (MPDU_LEN - 4)
    blob = torch.randint(0, 256, (nframes * (MPDU_LEN - 4),), dtype=torch.uint8, device=out.device)
    ooff = torch.arange(nframes, dtype=torch.int64, device=out.device) * (off[1] - off[0])
    L = capi.load()
    call = lambda: L.sora_hip_tx11a(capi._dev_ptr(blob), capi._dev_ptr(moff), capi._dev_ptr(lens), capi._dev_ptr(rate), capi._dev_ptr(seed),
                                    nframes, capi._dev_ptr(out), capi._dev_ptr(ooff), capi._stream_ptr(None))
    call()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nsamp = int(off[1] - off[0]) * nframes
    alg = nframes * (MPDU_LEN - 4) + 2 * nsamp
    return {"workload": "%d frames x %d-byte MPDU at 54 Mbps -> COMPLEX8 @40 MHz (%d samples)" % (nframes, MPDU_LEN, nsamp),
            "bound": "hbm", "ms": round(ms, 4), "msamples_per_s_out": round(nsamp / ms / 1e3, 1), "algorithmic_bytes": alg,
            "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / HBM_PEAK, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="captures per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=0, help="process calls in flight on the handle's internal pipelines (0 = library default)")
    ap.add_argument("--check", type=int, default=32, help="captures compared with the oracle after the timed region")
    args = ap.parse_args()

    import torch
    import sora_amd
    from oracle.pyoracle import Oracle

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    oracle = Oracle()
    nfr = args.frames
    iq, descs, payloads = make_workload(oracle, nfr, seed0=rank * 100003)
    d_iq = torch.from_numpy(iq).to(dev)
    descs = sora_amd.Rx.captures(descs)           # packed sora_capture_desc[]: built once, submitted every step
    rx = sora_amd.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, device=local_rank, max_frames_per_capture=2)

    if args.depth:
        rx.set_depth(args.depth)
    depth = rx.set_depth(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rx.process_dev(d_iq, descs)
    rx.flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rx.process_dev(d_iq, descs)
    barrier()
    t1 = time.perf_counter()
    # the same K steps once more with HIP events around every kernel launch (on the streams the kernels run on): the
    # roofline's launch durations are means over this region; `value` comes from the un-instrumented region above
    # (recording 6 events per call costs a few percent, reported as ms_per_step_profiled)
    rx.set_profiling(True)
    barrier()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        rx.process_dev(d_iq, descs)
    barrier()
    t3 = time.perf_counter()
    ktimes = rx.kernel_times()
    rx.set_profiling(False)

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was timed
    res = rx.results()
    n_ok = sum(1 for r in res if r["error_code"] == sora_amd.E_FRAME_OK)
    n_payload_ok = sum(1 for r in res if r["error_code"] == sora_amd.E_FRAME_OK and r["mpdu"][:-4] == payloads[r["capture_id"]])
    parity_ok = None
    if args.check:
        from gpu_util import oracle_results, same_results
        x = iq.reshape(nfr, CAPTURE_SAMPLES, 2)
        idx = list(range(0, nfr, max(1, nfr // args.check)))[:args.check]
        want = []
        for i in idx:
            for r in oracle.rx_capture(x[i], 20):
                r = dict(r); r["capture_id"] = i; want.append(r)
        got = [r for r in res if r["capture_id"] in set(idx)]
        parity_ok, why = same_results(got, want)
        if not parity_ok:
            print("PARITY MISMATCH vs oracle:", why, file=sys.stderr)
    counters = torch.tensor([len(res), n_ok, n_payload_ok], device=dev, dtype=torch.int64)
    gathered_rows = None
    if world > 1:
        # the path's one exchange step: RCCL all-gather of the device-packed result rows + all-reduce of counters
        from sora_amd.shard import gather_rows
        rows, nrows, _ = rx.results_dev()
        rx.flush()
        allrows, per_rank = gather_rows(rows, int(nrows.item()), max_rows_per_rank=nfr * 2)
        gathered_rows = int(allrows.shape[0])
        dist.all_reduce(counters)
    tot_frames, tot_ok, tot_payload_ok = [int(v) for v in counters.tolist()]

    total_samples = float(nfr) * FRAME_SAMPLES * world * args.steps
    msps = total_samples / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    if rank == 0:
        dom = max((k for k in ktimes if k.startswith("k_")), key=lambda k: ktimes[k])
        launch_bytes = nfr * FRAME_SAMPLES * ALG_BYTES_PER_SAMPLE
        ach = launch_bytes / (ktimes[dom] * 1e-3)
        out = {
            "metric": "IQ Msamples/s through 802.11a 54 Mbps RX PHY",
            "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_step_profiled": round((t3 - t2) / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 IQ / u8 path metrics", "data": "synthetic",
            "config": {"workload": "802.11a 54 Mbps (64-QAM r=3/4) RX, %d captures/GPU x one 1500-byte frame (4880 samples @20 MHz, +160 silence), AWGN 30/27 dB on 3 of 4" % nfr,
                       "frames_per_gpu": nfr, "samples_per_frame": FRAME_SAMPLES, "capture_samples": CAPTURE_SAMPLES, "calls_in_flight": depth,
                       "sharding": "captures per rank, no data-path collective"},
            "decoded_mbit_per_s": round(msps * (MPDU_LEN * 8.0 / FRAME_SAMPLES), 2),
            "frames": tot_frames, "gathered_rows": gathered_rows, "frames_crc_ok": tot_ok, "frames_payload_ok": tot_payload_ok, "oracle_parity_sample_ok": parity_ok,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK, 5), "traffic": measured_traffic(dom) if nfr == FRAMES_PER_GPU else None,
                         "algorithmic_bytes_per_launch": launch_bytes, "kernel_ms": round(ktimes[dom], 4),
                         "whole_path_frac": round(msps * 1e6 / world * ALG_BYTES_PER_SAMPLE / HBM_PEAK, 5),
                         "valu": valu_roofline(nfr, ms_per_step)},
            "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
        }
        if world == 1:
            out["ingest"] = bench_ingest(torch, sora_amd, dev)
            out["tx"] = bench_tx(torch, sora_amd)
            out["rx11b"] = bench_11b(torch, sora_amd, dev)
            out["rx11n"] = bench_11n(torch, sora_amd, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(iq, nfr)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
