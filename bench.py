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

The timed region: the K-step block (--steps) is repeated until at least --min-seconds have passed; every step is a
process call whose dense result rows AND MPDU array are delivered to page-locked host memory behind the kernels, and the
oldest call in flight is waited for and its rows compared with the verified ones.  Before anything is timed, the first
call's rows are compared, capture by capture, with what the compiled reference graph reports (--check 0 = all captures).

Prints ONE JSON line on rank 0 (see the contract in the task description), with `roofline` for the dominant kernel (HIP
events on the library's own streams, one call in flight) and `cpu_baseline` (the reference's SSE graph compiled from its
sources, on all usable host cores; the scalar restatement where that library is absent).
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per stream-priority level (default 4).  Round 3's bench
# set GPU_MAX_HW_QUEUES=16 before HIP started, because eight pipelines of one priority ran three at a time; since round 4 the library spreads a
# handle's pipelines over the three priority levels (sora_internal_stream_create) and gets a hardware queue per pipeline by itself: the default
# run sets NO environment variable (config.hw_queues = null; profiles/r04_m_stream_priorities.txt).  --hw-queues N still sets it, for A/B runs.
TRELLIS_NAMES = {64: "k_viterbi", 16: "k_viterbi16", 1: "k_viterbi16w"}      # sora_rx_set_trellis: two frames per wave / eight per wave / window-parallel (round 5)


def _early_hw_queues(argv):
    """--hw-queues N, read before the HIP runtime starts: N > 0 sets GPU_MAX_HW_QUEUES (unless the environment already does),
    0 leaves the runtime's default alone (`config.hw_queues` is then null)."""
    for i, a in enumerate(argv):
        if a == "--hw-queues" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--hw-queues="):
            return int(a.split("=", 1)[1])
    return None


_HWQ = _early_hw_queues(sys.argv)
if _HWQ is None:
    _HWQ = 0
if _HWQ > 0:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(_HWQ))

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
PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
TRAFFIC_JSON = os.path.join(PROFILES, "r05_final_traffic.json")          # rocprofv3 --pmc passes of this command (tools/collect_profiles.sh): replayed, not measured by this run
VALU_PEAK_JSON = os.path.join(PROFILES, "r04_valu_peak.json")          # tools/calib/valu_peak.hip on one MI355X: what the chip sustains per instruction kind


def _traffic_profile():
    """The committed PMC summary, or (None, why).  It is replayed into the bench line only while it belongs to THIS tree: tools/summarize_pmc.py stamps it with the
    hash of every source and header the library is built from (sora_amd.build.sources_sha256), and a summary whose stamp is missing or differs is refused."""
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
    except (OSError, ValueError) as e:
        return None, "no PMC summary (%s)" % e.__class__.__name__
    try:
        from sora_amd import build as _b
        now = _b.sources_sha256()
    except Exception as e:
        return None, "sources hash unavailable (%r)" % e
    if t.get("sources_sha256") != now:
        return None, "stale: %s was collected for sources %s, this tree is %s" % (os.path.basename(TRAFFIC_JSON), str(t.get("sources_sha256"))[:16], now[:16])
    if t.get("frames_per_launch") != FRAMES_PER_GPU:
        return None, "collected for %s frames per launch" % t.get("frames_per_launch")
    return t, "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS in separate passes, corrected as tools/summarize_pmc.py documents; replayed, not measured by this run; sources stamp matches)" % os.path.basename(TRAFFIC_JSON)


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (FETCH_SIZE x2 + WRITE_SIZE, see the file), or None."""
    t, _ = _traffic_profile()
    try:
        return t["kernels"][kernel]["hbm_bytes"] if t else None
    except KeyError:
        return None


def measured_valu(kernel=None):
    """Wave-level VALU instructions per launch (SQ_INSTS_VALU, same PMC summary): of `kernel`, or of the whole call."""
    t, _ = _traffic_profile()
    try:
        return None if not t else t["kernels"][kernel]["valu_insts"] if kernel else t["total_valu_insts_per_call"]
    except KeyError:
        return None


def workload_payload(seed0, i, nframes, distinct=512):
    """MPDU (without FCS) of capture i of make_workload(.., nframes, seed0): what any rank can recompute about any other rank's batch"""
    rng = np.random.default_rng(0x5EED0000 + seed0 + i % min(distinct, nframes))
    return rng.integers(0, 256, MPDU_LEN - 4).astype(np.uint8).tobytes()


def exchange_results(torch, rx, d_iq, descs, dev, nfr, maxf):
    """The multi-GPU path's one exchange step, on an initialised process group: one more call, then RCCL all-gathers of
    {rows, MPDU bytes} per rank, the device-packed result rows and the dense MPDU blocks (sora_amd.shard.gather_mpdus) -- every MPDU of
    every rank reaches every host (fb11a_demod.cpp:64-70 for a sharded batch) -- and a check of every gathered MPDU against the payload
    its rank transmitted (rank r's batch comes from seed0 = r * 100003, so any rank can recompute it)."""
    from sora_amd.shard import gather_mpdus
    rx.process_dev(d_iq, descs)
    rows, nrows, mpdu_ptr = rx.results_dev()
    rx.flush()

    class _Arr:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    mpdu_dev = torch.as_tensor(_Arr(mpdu_ptr, rx.mpdu_bytes(rx.ticket())), device=dev)
    tg0 = time.perf_counter()
    allrows, allmp, per_rank = gather_mpdus(rows, int(nrows.item()), mpdu_dev, max_rows_per_rank=nfr * maxf, max_bytes_per_rank=nfr * maxf * MPDU_LEN)
    torch.cuda.synchronize()
    tg1 = time.perf_counter()
    ar = allrows.cpu().numpy().view(np.uint32); am = allmp.cpu().numpy()
    okm = 0; k = 0
    for rr, cnt in enumerate(per_rank):                                  # rank rr's rows: its captures were made from seed0 = rr * 100003
        for w in ar[k:k + cnt]:
            if int(w[3]) == 1:
                o_, ln = int(w[8]), int(w[5] & 0xFFFF)
                okm += bytes(am[o_:o_ + ln - 4]) == workload_payload(rr * 100003, int(w[0]), nfr)
        k += cnt
    return {"rows": int(allrows.shape[0]), "rows_per_rank": per_rank, "mpdu_bytes": int(am.size), "mpdus_equal_to_the_transmitted_payloads": int(okm),
            "exchange_ms": round((tg1 - tg0) * 1e3, 3),
            "bytes_per_rank": {"counts": 8, "rows": 36 * nfr * maxf, "mpdu_block": nfr * maxf * MPDU_LEN}}


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
    elif kind == "reference_mt":                                          # the reference's native split: RxThread here, ViterbiThread behind TThreadSeparator on a second core (fb11a_demod.cpp:83-120)
        import ctypes
        mt = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsora_refgraph_mt.so")); mt.ref_rx11a_bench_mt.restype = ctypes.c_uint32
        caps = np.ascontiguousarray(np.repeat(caps, 2, axis=1))
        run = lambda: mt.ref_rx11a_bench_mt(caps.ctypes.data_as(ctypes.c_void_p), caps.shape[0], caps.shape[1], 1)  # noqa: E731
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
        if have_ref and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsora_refgraph_mt.so")) and cores >= 2:
            # "single-thread and the native demod || Viterbi split" -- one instance of the two-thread harness (two cores), then cores // 2 of them side by
            # side.  Each instance is a process of its own that leaves through os._exit: its ViterbiThread spins on the separator's queue for good, as the reference's does.
            def mt_run(n_inst, secs):
                import subprocess
                ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-mt-worker", "%s,%d,%d,%d,%g" % (path, nframes, k, n_inst, secs)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                      for k in range(n_inst)]
                res = []
                for q in ps:
                    try:
                        o_, _ = q.communicate(timeout=secs + 60)
                        res.append(json.loads(o_.decode().strip().splitlines()[-1]))
                    except Exception:
                        q.kill()
                return res
            one = mt_run(1, 3.0); pairs = max(1, cores // 2); many = mt_run(pairs, 4.0)
            if one and len(many) == pairs:
                out["reference_mt"] = {"single": round(one[0][0] * FRAME_SAMPLES / one[0][2] / 1e6, 4), "value": round(sum(r[0] * FRAME_SAMPLES / r[2] for r in many) / 1e6, 3), "instances": pairs,
                                       "ok": one[0][1], "n": one[0][0]}
    kind = "reference" if have_ref else "port"
    m = out[kind]
    what = ("the reference's brick graph compiled from its sources (oracle/_ref/libsora_refgraph.so, SSE)" if have_ref
            else "oracle/so_rx11a.c (scalar C restatement)")
    r = {"value": m["value"], "unit": "Msamples/s", "cores": cores, "kind": kind, "single_core_value": m["single"],
         "sample": "%d captures of this workload (cycled), %d processes x %.0f s, %s" % (m["n"], cores, m["secs"], what),
         "frames_ok": m["ok"], "frames_run": m["n"]}
    if have_ref:
        r["port_value"] = out["port"]["value"]; r["port_single_core_value"] = out["port"]["single"]
    if "reference_mt" in out:
        mt = out["reference_mt"]
        r["two_thread_value"] = mt["single"]
        r["two_thread"] = {"value_one_instance_two_cores": mt["single"], "value_all_cores": mt["value"], "instances": mt["instances"], "frames_ok": mt["ok"], "frames_run": mt["n"],
                           "what": "the reference's own harness shape: RxThread on one core, ViterbiThread behind TThreadSeparator on a second (fb11a_demod.cpp:83-120, stdbrick.hpp:89-248; "
                                   "oracle/_ref/libsora_refgraph_mt.so, graph and thread kept across captures); the single-thread build above replaces the separator by TNoInline"}
    return r


def valu_roofline(nframes, ms_step):
    """What actually bounds this path: vector-ALU issue.  Wave-level VALU instructions of one receive call (rocprofv3 SQ_INSTS_VALU in separate
    --pmc passes -- replayed from the committed summary, counters need rocprof) over the measured step time, against what the chip SUSTAINS for
    this path's instruction mix: tools/calib/valu_peak.hip runs long unrolled streams of one instruction kind on every CU and reports wave-
    instructions per second of wall time (so the clock the chip holds under that load is in the number): v_add_u32 1166 G/s (2 cycles per
    wave64 instruction at ~2.28 GHz), v_pk_min_u16 / v_add_u32_dpp / any VOP3 ~580 G/s (half rate), and the trellis step's own mix -- four
    add, four add_dpp, four pk_min, xor, sub, each minimum depending on the two sums before it -- 607 G/s at ANY occupancy from one to eight
    waves per SIMD.  That last figure is `peak`: the ceiling for code made of add-compare-select steps."""
    n = measured_valu() if nframes == FRAMES_PER_GPU else None
    if not n:
        return None
    try:
        with open(VALU_PEAK_JSON) as f:
            pk = json.load(f)
        peak = pk["trellis_step_mix"]["g_per_s"]["4"] * 1e9; vop2 = pk["v_add_u32"]["g_per_s"]["4"]; half = pk["v_pk_min_u16"]["g_per_s"]["4"]
        src = "profiles/r04_valu_peak.json (tools/calib/valu_peak.hip, measured on one MI355X; not re-measured by this run)"
    except (OSError, KeyError, ValueError):
        peak = 256 * 4 * 2.4e9 / 2; vop2 = peak / 1e9; half = vop2 / 2; src = "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (no probe file)"
    ach = n / (ms_step * 1e-3)
    return {"insts_per_call": n, "insts_source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU; replayed)" % os.path.basename(TRAFFIC_JSON),
            "achieved": round(ach / 1e9, 1), "peak": round(peak / 1e9, 1), "peak_source": src, "unit": "G wave-instr/s", "frac": round(ach / peak, 4),
            "vop2_only_rate": vop2, "half_rate_instruction_rate": half,
            "dominant_kernel_insts": measured_valu("k_viterbi16") or measured_valu("k_viterbi")}


def bench_stages(torch, sora_amd, dev, nsym=1 << 20, reps=12, nsets=3):
    """The per-stage entry points (what the BRICK adapters call), each over `nsym` OFDM symbols resident in HBM, against the
    HBM roofline with SURVEY.md section 8(d)'s algorithmic bytes per symbol: FFT 256 in + 256 out; symbol front end
    (T11aDataSymbol..TChannelEqualization) 320 in + 256 out; demap (64-QAM) 256 in + 288 out; de-interleave 288 + 288;
    Viterbi (54 Mbps frames of 56 symbols) 288 soft bytes in + 27 decoded bytes out; FFT<128> 512 + 512.
    Round 4: every stage cycles through `nsets` DISTINCT input / output buffer sets, so consecutive launches share no
    line and the bytes in play (1.6-1.8 GB) are far past the 256 MiB Infinity Cache -- round 3's single 537-604 MB set only just exceeded it."""
    from sora_amd import capi
    L = capi.load()
    out = {}
    g = torch.Generator(device=dev); g.manual_seed(7)

    def timed(fns, nbytes, label, n=nsym):
        for f in fns:
            f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(reps):
            fns[i % len(fns)]()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[label] = {"symbols": n, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "achieved": round(nbytes / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
                      "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4), "gsymbols_per_s": round(n / ms / 1e6, 3),
                      "buffer_sets": len(fns), "bytes_in_play": int(nbytes) * len(fns)}

    st = capi._stream_ptr(None)
    P = capi._dev_ptr
    xs = [torch.randint(-6000, 6000, (nsym, 64, 2), dtype=torch.int16, device=dev, generator=g) for _ in range(nsets)]
    ys = [torch.empty_like(xs[0]) for _ in range(nsets)]
    timed([(lambda x=x, y=y: L.sora_hip_fft64(P(x), P(y), nsym, st)) for x, y in zip(xs, ys)], nsym * 512, "fft64")
    del ys
    softs = [torch.empty((nsym, 288), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    timed([(lambda x=x, o=o: L.sora_hip_demap11a(P(x), P(o), 6, nsym, st)) for x, o in zip(xs, softs)], nsym * (256 + 288), "demap11a_qam64")
    des = [torch.empty_like(softs[0]) for _ in range(nsets)]
    timed([(lambda i=i, o=o: L.sora_hip_deinterleave11a(P(i), P(o), 6, nsym, st)) for i, o in zip(softs, des)], nsym * 576, "deinterleave11a_qam64")
    del des, softs
    n128 = nsym // 2
    x128 = [x.view(n128, 128, 2) for x in xs]; y128 = [torch.empty_like(x128[0]) for _ in range(nsets)]
    timed([(lambda x=x, y=y: L.sora_hip_fft128(P(x), P(y), n128, st)) for x, y in zip(x128, y128)], n128 * 1024, "fft128", n128)
    del y128, x128, xs
    x80 = [torch.randint(-6000, 6000, (nsym, 80, 2), dtype=torch.int16, device=dev, generator=g) for _ in range(nsets)]
    nctx = 4096
    lts_in = torch.randint(-6000, 6000, (nctx, 144, 2), dtype=torch.int16, device=dev, generator=g)
    ctx = sora_amd.lts11a(lts_in)
    idx = (torch.arange(nsym, device=dev, dtype=torch.int32) // 256) % nctx
    eqs = [torch.empty((nsym, 64, 2), dtype=torch.int16, device=dev) for _ in range(nsets)]
    timed([(lambda x=x, e=e: L.sora_hip_symfront11a(P(x), P(ctx), P(idx), P(e), nsym, st)) for x, e in zip(x80, eqs)], nsym * 576, "symfront11a")
    # the three one-multiply bricks alone: 256 in + 256 out per symbol, a frame's 256 bytes of coefficients shared by its 256 symbols
    # (SURVEY section 8d counts a private coefficient read per symbol for the equaliser: 768)
    del x80
    x64 = [torch.randint(-6000, 6000, (nsym, 64, 2), dtype=torch.int16, device=dev, generator=g) for _ in range(nsets)]
    stt = torch.randint(-32768, 32767, (nctx, 134), dtype=torch.int16, device=dev, generator=g)
    timed([(lambda x=x, e=e: L.sora_hip_freq_comp11a(P(x), P(ctx), P(idx), P(e), nsym, st)) for x, e in zip(x64, eqs)], nsym * 512, "freq_comp11a")
    timed([(lambda x=x, e=e: L.sora_hip_equalize11a(P(x), P(ctx), P(idx), P(e), nsym, st)) for x, e in zip(x64, eqs)], nsym * 512, "equalize11a")
    timed([(lambda x=x, e=e: L.sora_hip_phase_comp11a(P(x), P(stt), P(idx), P(e), nsym, st)) for x, e in zip(x64, eqs)], nsym * 512, "phase_comp11a")
    del x64, eqs, idx, stt
    # Viterbi: frames of 56 symbols x 216 soft values (the bench frame), random soft values 0..7
    nfr = 8192; nso = 56 * 288
    sv = torch.randint(0, 8, (nfr * nso,), dtype=torch.uint8, device=dev, generator=g)
    so = (torch.arange(nfr, device=dev, dtype=torch.int32) * nso).contiguous(); ns = torch.full((nfr,), nso, dtype=torch.int32, device=dev)
    fl = torch.full((nfr,), MPDU_LEN, dtype=torch.int16, device=dev)
    vo = torch.zeros((nfr, 1536), dtype=torch.uint8, device=dev); oo = (torch.arange(nfr, device=dev, dtype=torch.int32) * 1536).contiguous()
    timed([lambda: L.sora_hip_viterbi11a(P(sv), P(so), P(ns), P(fl), 2, P(vo), P(oo), nfr, st)],
          nfr * 56 * (288 + 27), "viterbi11a_r34", nfr * 56)
    return out


def bench_ingest(torch, sora_amd, dev, nbytes=256 << 20, reps=20, nsets=3):
    """Row f3 (capture ingest): a 44 MHz RX_BLOCK dump resident in HBM -> de-framed, sign-fixed, resampled 40 MHz stream.
    A pure streaming kernel: algorithmic bytes = dump bytes read + samples written, against the HBM roofline.  `nsets` distinct dumps in turn
    (round 4: 0.8 GB of input in play instead of one 256 MiB buffer that is exactly the size of the Infinity Cache)."""
    flags = sora_amd.INGEST_RXBLOCK | sora_amd.INGEST_RAW14 | sora_amd.INGEST_44TO40
    raws = [torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    n_out = sora_amd.ingest_count(nbytes, flags)
    for r in raws:
        out = sora_amd.ingest(r, flags, sync=False)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps):
        out = sora_amd.ingest(raws[i % nsets], flags, sync=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = nbytes + 4 * n_out
    del raws, out
    return {"workload": "%d MiB Sora RX_BLOCK dump @44 MHz -> de-frame + 14->16 bit + 44->40 MHz (%d samples out), %d distinct dumps in turn" % (nbytes >> 20, n_out, nsets),
            "bound": "hbm", "ms": round(ms, 4), "algorithmic_bytes": alg, "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / HBM_PEAK, 4), "msamples_per_s_in": round(nbytes / 128 * 28 / ms / 1e3, 1)}



def bench_latency(torch, sora_amd, dev, rx_batch, d_iq, descs, nfr, reps=40):
    """The reference harness's own figure of merit (MACStopwatch.h:84-128,130-164): per frame, cost (time spent demodulating it) over required
    time (its samples / 40 MHz), with mean / max / std and the shares >= 0.8 and >= 1.0.
    (a) BASELINE configs[1]: kernel/test-data/fsample-6.dmp as ONE capture (tests/golden/fsample6_40mhz_i8.npz: the dump after the 14 -> 16 bit
        fix, 75,320 samples @40 MHz = 1.883 ms of air time, one 6 Mbps frame of 465 symbols): wall time of process -> wait with one call in
        flight, and the compiled reference graph on one host core beside it.
    (b) the 4096-frame batch, one call in flight (process -> deliver -> wait): every frame of a call costs that call's latency (they complete
        together), required = 9760 samples / 40 MHz = 244 us; the distribution is over the frames of `reps` calls.  The amortised cost
        (step time / frames) is what `realtime.factor` reports."""
    import hashlib
    from oracle.pyoracle import ReferenceGraph
    out = {}
    g = np.load(os.path.join(ROOT, "tests", "golden", "fsample6_40mhz_i8.npz"))
    iq40 = g["iq_i8"].astype(np.int16) << 8
    n = len(iq40) // 28 * 28
    d = torch.from_numpy(np.ascontiguousarray(iq40[:n])).to(dev)
    rx = sora_amd.Rx(1, n, sample_rate_mhz=40, max_frames_per_capture=2)
    rx.set_depth(1)
    one = [(0, n, 0)]
    t = rx.process_dev(d, one); res = rx.results(ticket=t)
    rx.wait_for_producer = False
    ok = len(res) == 1 and res[0]["error_code"] == sora_amd.E_FRAME_OK and hashlib.sha256(res[0]["mpdu"]).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    per = {}
    chains = {1: "k_frame", 3: "k_sym_front+k_track_lds+k_sym_back"}
    for front, lanes in ((1, 64), (1, 16), (1, 1), (3, 64), (3, 1)):
        rx.set_front(front); rx.set_trellis(lanes); rx.flush()
        ok = ok and [r["mpdu"] for r in rx.results(ticket=rx.process_dev(d, one))] == [res[0]["mpdu"]]
        for _ in range(5):
            rx.wait(rx.process_dev(d, one))
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); rx.wait(rx.process_dev(d, one)); ts.append(time.perf_counter() - t0)
        per[chains[front] + " | " + TRELLIS_NAMES[lanes]] = float(np.median(ts)) * 1e3
    rx.set_front(0); rx.set_trellis(0); rx.flush()                             # the library's own choice for a lone capture: the chains that spread ONE frame over the chip
    auto = chains[rx.front()] + " | " + TRELLIS_NAMES[rx.trellis()]
    rx.set_profiling(True)
    for _ in range(10):
        rx.wait(rx.process_dev(d, one))
    rx.flush(); kt = rx.kernel_times(); rx.set_profiling(False); wstats = rx.window_stats(); rx.close()
    air_ms = n / 40e3
    best = min(per.values())
    out["fsample6_single_capture"] = {
        "workload": "kernel/test-data/fsample-6.dmp after the 14->16 bit fix: one 6 Mbps frame, 1392 bytes, 465 data symbols, %d samples @40 MHz" % n,
        "air_time_ms": round(air_ms, 4), "decode_ms": round(per[auto], 4), "kernels": auto + " (the library's automatic choice)", "decode_ms_by_kernels": {k: round(v, 4) for k, v in per.items()},
        "decode_ms_best": round(best, 4),
        "realtime_factor": round(per[auto] / air_ms, 4), "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
        "kernel_ms_note": "the library's five timed intervals of the automatic chain: 'k_frame' = k_sym_front + k_track_lds + k_sym_back, 'k_viterbi' = k_viterbi16w + k_win_redo",
        "window_trellis_record": wstats, "mpdu_sha256_ok": bool(ok),
        "protocol": "sora_rx_process_dev + sora_rx_wait, one call in flight, samples resident in HBM; median of %d calls (host wall clock)" % reps}
    ref = ReferenceGraph()
    if ref.available():
        caps = np.ascontiguousarray(iq40[None, :n])
        ref.rx11a_bench(caps)
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 1.0:
            ref.rx11a_bench(caps, 4); k += 4
        cpu_ms = (time.perf_counter() - t0) / k * 1e3
        out["fsample6_single_capture"]["cpu_reference_decode_ms_one_core"] = round(cpu_ms, 4)
        out["fsample6_single_capture"]["cpu_reference_realtime_factor_one_core"] = round(cpu_ms / air_ms, 4)
    if rx_batch is None:
        return out
    # (b) the batch, one call in flight
    old_depth = rx_batch.set_depth(1); old_tr = rx_batch.set_trellis(-1); rx_batch.flush()
    req_us = 2 * FRAME_SAMPLES / 40.0
    dist = {}
    for lanes in (64, 16, 1):
        rx_batch.set_trellis(lanes); rx_batch.flush()
        buf = sora_amd.HostResults(nfr * 2, rx_batch.mpdu_bytes(rx_batch.process_dev(d_iq, descs))); rx_batch.flush()
        lat = []
        for i in range(reps + 3):
            t0 = time.perf_counter()
            tk = rx_batch.process_dev(d_iq, descs); rx_batch.deliver_async(tk, buf); rx_batch.wait(tk)
            if i >= 3:
                lat.append((time.perf_counter() - t0) * 1e6)
        buf.close()
        r = np.asarray(lat) / req_us                                         # every frame of call i has ratio r[i]
        dist[TRELLIS_NAMES[lanes]] = {
            "call_latency_ms": round(float(np.mean(lat)) / 1e3, 4), "frames": int(nfr * len(lat)), "required_us_per_frame": req_us,
            "ratio_mean": round(float(r.mean()), 3), "ratio_max": round(float(r.max()), 3), "ratio_std": round(float(r.std()), 3),
            "share_ge_0.8": round(float((r >= 0.8).mean()), 3), "share_ge_1.0": round(float((r >= 1.0).mean()), 3)}
    rx_batch.set_trellis(old_tr); rx_batch.set_depth(old_depth); rx_batch.flush()
    out["batch_per_frame"] = {"definition": "MACStopwatch's per-frame ratio cost / required with cost = the latency of the call the frame is in (process -> deliver -> wait, one call in flight: "
                                            "all %d frames of a call complete together) and required = %d samples / 40 MHz; >= 1.0 means a frame's result arrives later than its own air time, although "
                                            "the batch as a whole is decoded far faster than real time (realtime.factor, the amortised cost)" % (nfr, 2 * FRAME_SAMPLES),
                              "by_trellis_kernel": dist}
    return out


def bench_large_call(torch, sora_amd, local_rank, d_iqs, nfr, maxf, exp_rows, exp_mpdu, seconds, cores):
    """What a plain host gets when it hands over MORE PER CALL instead of keeping more calls in flight: the rotated device copies of the batch
    as ONE call of 8 x nfr captures, at most TWO such calls in flight, every call delivered (rows + MPDU bytes) and compared.  The first
    call is verified against the already verified nfr-capture table, quarter by quarter (capture_id and mpdu_offset shifted, everything else and
    every MPDU byte equal).  Reported per nfr captures, so that it reads beside ms_per_step."""
    copies = list(d_iqs) * max(1, 8 // len(d_iqs))                   # 8 x nfr captures per call: its trellis launch is two rounds of the chip's trellis slots
    g_n = len(copies); n_iq = d_iqs[0].shape[0]                     # (at 4 x nfr it is exactly ONE round, and the step is bimodal, 0.39-0.53 ms: profiles/r04_r_call_size.txt)
    big = torch.cat(copies)
    descs = sora_amd.Rx.captures([(g * n_iq + i * CAPTURE_SAMPLES, CAPTURE_SAMPLES, g * nfr + i) for g in range(g_n) for i in range(nfr)])
    rx = sora_amd.Rx(max_captures=g_n * nfr, max_total_samples=g_n * n_iq, sample_rate_mhz=20, device=local_rank, max_frames_per_capture=maxf)
    dep = 2
    rx.set_depth(dep)
    torch.cuda.synchronize()
    rx.wait_for_producer = False
    t = rx.process_dev(big, descs)
    nb = dep + TableChecker.EXTRA
    bufs = [sora_amd.HostResults(g_n * nfr * maxf, rx.mpdu_bytes(t)) for _ in range(nb)]
    rx.deliver_async(t, bufs[0]); rx.wait(t)
    n = int(bufs[0].nrows[0]); rows = bufs[0].rows[:n].copy(); mp = bufs[0].mpdu.copy()
    exp_n = len(exp_rows)
    ok = n == g_n * exp_n
    if ok:
        for g in range(g_n):
            q = rows[g * exp_n:(g + 1) * exp_n]
            ok = ok and all((q[f] == exp_rows[f]).all() for f in q.dtype.names if f not in ("capture_id", "mpdu_offset")) \
                and bool((q["capture_id"] == exp_rows["capture_id"] + g * nfr).all())
            good = np.nonzero(exp_rows["error_code"] == 1)[0]
            for k in good:
                a, b, ln = int(q["mpdu_offset"][k]), int(exp_rows["mpdu_offset"][k]), int(exp_rows["length"][k])
                if mp[a:a + ln].tobytes() != exp_mpdu[b:b + ln].tobytes():
                    ok = False
                    break
    chk = TableChecker(rows.tobytes(), mp, cores=cores)

    def block(k):
        first = None
        for _ in range(k):
            chk.release((rx.ticket() + 1) % nb)
            tk = rx.process_dev(big, descs)
            rx.deliver_async(tk, bufs[tk % nb])
            if first is None:
                first = tk
            if tk - first >= dep - 1:
                rx.wait(tk - (dep - 1)); b = bufs[(tk - (dep - 1)) % nb]
                chk.check((tk - (dep - 1)) % nb, int(b.nrows[0]) == n, b.rows[:n], b.mpdu)
        for old in range(max(first, tk - (dep - 1) + 1), tk + 1):
            rx.wait(old); b = bufs[old % nb]
            chk.check(old % nb, int(b.nrows[0]) == n, b.rows[:n], b.mpdu)
        rx.flush(); chk.drain()
    block(4)
    t0 = time.perf_counter(); block(8); probe = (time.perf_counter() - t0) / 8
    ncalls = max(16, int(seconds / max(probe, 1e-6)))
    chk.compared = chk.bad = 0
    t0 = time.perf_counter(); block(ncalls); dt = time.perf_counter() - t0
    out = {"captures_per_call": g_n * nfr, "calls_in_flight": dep, "trellis": TRELLIS_NAMES[rx.trellis()], "calls_timed": ncalls,
           "ms_per_call": round(1e3 * dt / ncalls, 4), "ms_per_%d_captures" % nfr: round(1e3 * dt / ncalls / g_n, 4),
           "msamples_per_s": round(g_n * nfr * FRAME_SAMPLES * ncalls / dt / 1e6, 1), "first_call_equals_the_verified_table": bool(ok),
           "calls_compared": chk.compared, "calls_with_wrong_rows": chk.bad, "input_bytes_per_call": int(big.numel() * 2),
           "note": "the same step protocol (process_dev -> deliver_async -> wait -> compare, no environment variable) with %d captures per call and two calls in flight" % (g_n * nfr)}
    chk.finish(); rx.close()
    del big
    return out


def bench_e2e(torch, sora_amd, dev, rx, iq, nfr, exp_rows, exp_mpdu, steps=24, nbatches=4):
    """Dump bytes in page-locked host memory -> sora_rx_process_dump (H2D copy + sora_hip_ingest + the receive chain on one stream, no host wait) ->
    rows and MPDUs delivered to the host: LoadSoraDumpFile -> graph -> MPDU buffer (brickutil.h:20-58, fb11a_demod.cpp:88-120) as one path.
    The workload's captures as a 40 MHz RX_BLOCK dump (every 20 MHz sample doubled -- TDownSample2, done by the ingest, drops the copies --
    128-byte blocks of a 16-byte descriptor + 28 samples): `nbatches` DIFFERENT dumps (the captures rotated by a quarter of the batch each)
    are submitted in turn, about 190 MB each, so the inputs of consecutive calls share nothing and their total is past the 256 MiB Infinity Cache."""
    from test_oracle_ingest import make_dump
    flags = sora_amd.INGEST_RXBLOCK | sora_amd.INGEST_DECIMATE2
    caps20 = iq.reshape(nfr, CAPTURE_SAMPLES, 2)
    blocks_per_cap = 2 * CAPTURE_SAMPLES // 28
    base = np.empty((nfr, blocks_per_cap * 128), np.uint8)
    for i in range(0, nfr, 256):
        c40 = np.repeat(caps20[i:i + 256], 2, axis=1).reshape(-1, 2)
        base[i:i + 256] = make_dump(c40, raw14=False, seed=i).reshape(-1, blocks_per_cap * 128)
    dumps, ids = [], []
    for k in range(nbatches):
        sh = (k * nfr) // nbatches
        perm = (np.arange(nfr) + sh) % nfr                                    # position i of dump k holds capture perm[i]
        t = torch.empty(base.size, dtype=torch.uint8).pin_memory()
        t.numpy().reshape(base.shape)[:] = base[perm]
        dumps.append(t); ids.append(perm)
    del base
    descs = [sora_amd.Rx.captures([(i * CAPTURE_SAMPLES, CAPTURE_SAMPLES, int(ids[k][i])) for i in range(nfr)]) for k in range(nbatches)]
    # what every call must deliver: per capture id {error_code, length, crc32} of the verified table (rows come in position order)
    order = np.argsort(exp_rows["capture_id"], kind="stable")
    want = {f: exp_rows[f][order] for f in ("capture_id", "error_code", "length", "crc32")}
    depth = 4
    old_depth = rx.set_depth(depth); rx.flush()
    nb = depth + 2
    bufs = [sora_amd.HostResults(nfr * 2, rx.mpdu_bytes(rx.ticket())) for _ in range(nb)]
    bad = [0]; checked = [0]; mp_checked = [0]

    def consume(tk, k):
        rx.wait(tk)
        b = bufs[tk % nb]
        n = int(b.nrows[0]); rows = b.rows[:n]
        o = np.argsort(rows["capture_id"], kind="stable")
        same = n == len(want["capture_id"]) and all(np.array_equal(rows[f][o], want[f]) for f in want)
        if same and k == 0:                                                   # the unrotated dump: the MPDU array byte for byte as well
            same = bool((b.mpdu == exp_mpdu).all()); mp_checked[0] += 1
        checked[0] += 1; bad[0] += 0 if same else 1

    def block(nsteps):
        pend = []
        for i in range(nsteps):
            k = i % nbatches
            tk = rx.process_dump(dumps[k], flags, descs[k])
            rx.deliver_async(tk, bufs[tk % nb]); pend.append((tk, k))
            if len(pend) >= depth:
                consume(*pend.pop(0))
        for tk, k in pend:
            consume(tk, k)
    block(nbatches + depth)                                                   # warm-up: every pipeline's staging buffers exist
    bad[0] = checked[0] = mp_checked[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block(steps)
    ms = (time.perf_counter() - t0) / steps * 1e3
    dump_bytes = int(dumps[0].numel())
    dump_row = {"workload": "%d dumps of %.1f MB (the batch's captures as a 40 MHz RX_BLOCK dump, rotated) in page-locked host memory, submitted in turn: sora_rx_process_dump = H2D copy + "
                            "sora_hip_ingest (de-frame, TDownSample2) + receive chain on the call's stream, then deliver_async of rows + MPDUs; %d calls in flight" % (nbatches, dump_bytes / 1e6, depth),
                "ms_per_step": round(ms, 4), "bytes_per_step": dump_bytes, "distinct_input_bytes": dump_bytes * nbatches,
                "pcie_gb_per_s_host_to_device": round(dump_bytes / ms / 1e6, 2), "msamples_per_s": round(nfr * FRAME_SAMPLES / ms / 1e3, 1),
                "decoded_mbit_per_s": round(nfr * MPDU_LEN * 8 / ms / 1e3, 1),
                "calls_delivered_and_checked": checked[0], "calls_with_wrong_rows": bad[0], "calls_with_mpdu_bytes_compared": mp_checked[0],
                "note": "bound by the host link: the 40 MHz dump is 9.4 bytes of PCIe traffic per 20 MHz sample decoded (RX_BLOCK framing, both 40 MHz samples of a pair)"}
    # the same host-fed loop with the stream the graph actually consumes -- descriptors stripped, TDownSample2 already applied (what brickutil.h:20-58 +
    # samples.hpp:36-39 leave: 4 bytes per 20 MHz sample), handed to the 20 MHz handle's sora_rx_process: H2D copy + receive chain + delivery, same comparison.
    del dumps
    streams = []
    for k in range(nbatches):
        t = torch.empty((nfr * CAPTURE_SAMPLES, 2), dtype=torch.int16).pin_memory()
        t.numpy().reshape(nfr, CAPTURE_SAMPLES, 2)[:] = caps20[ids[k]]
        streams.append(t)

    def block2(nsteps):
        pend = []
        for i in range(nsteps):
            k = i % nbatches
            tk = rx.process(streams[k].numpy(), descs[k])
            rx.deliver_async(tk, bufs[tk % nb]); pend.append((tk, k))
            if len(pend) >= depth:
                consume(*pend.pop(0))
        for tk, k in pend:
            consume(tk, k)
    block2(nbatches + depth)
    bad[0] = checked[0] = mp_checked[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block2(steps)
    ms2 = (time.perf_counter() - t0) / steps * 1e3
    sbytes = int(streams[0].numel()) * 2
    stream_row = {"workload": "%d streams of %.1f MB (the same captures as the 20 MHz COMPLEX16 stream the graph consumes: descriptors stripped, even samples only) in page-locked host memory: "
                              "sora_rx_process = H2D copy + receive chain, then deliver_async of rows + MPDUs; %d calls in flight" % (nbatches, sbytes / 1e6, depth),
                  "ms_per_step": round(ms2, 4), "bytes_per_step": sbytes, "pcie_gb_per_s_host_to_device": round(sbytes / ms2 / 1e6, 2), "msamples_per_s": round(nfr * FRAME_SAMPLES / ms2 / 1e3, 1),
                  "decoded_mbit_per_s": round(nfr * MPDU_LEN * 8 / ms2 / 1e3, 1),
                  "calls_delivered_and_checked": checked[0], "calls_with_wrong_rows": bad[0], "calls_with_mpdu_bytes_compared": mp_checked[0],
                  "note": "4.1 bytes of PCIe traffic per 20 MHz sample decoded: who strips the RX_BLOCK framing and drops the odd samples before the link (the capture front end, or a host pass) halves the link's load"}
    rx.set_depth(old_depth); rx.flush()
    for b in bufs:
        b.close()
    best = max((dump_row, stream_row), key=lambda r: r["msamples_per_s"])
    return {"msamples_per_s": best["msamples_per_s"], "ms_per_step": best["ms_per_step"], "path": "stripped_stream_20mhz" if best is stream_row else "rx_block_dump_40mhz",
            "rx_block_dump_40mhz": dump_row, "stripped_stream_20mhz": stream_row}


class TableChecker:
    """Byte-for-byte comparison of delivered tables with the verified first call's, off the submitting thread: comparing 6-12 MB of MPDUs
    takes a host core 1-2 ms -- longer than the GPU takes to decode them -- so a small pool of threads does it (numpy and memcmp release the
    GIL) while the main thread submits the next call.  A buffer is handed out again only after its comparison has finished."""
    EXTRA = 4                                                               # buffers beyond the calls in flight: the ones being compared

    def __init__(self, exp_rows_bytes, exp_mpdu, cores=None):
        from concurrent.futures import ThreadPoolExecutor

        def pin():                                                          # a checker thread never runs on the submit thread's core
            if cores and hasattr(os, "sched_setaffinity"):
                try:
                    os.sched_setaffinity(0, set(cores))
                except OSError:
                    pass
        self.pool = ThreadPoolExecutor(self.EXTRA, initializer=pin)
        self.rows = exp_rows_bytes
        m8 = exp_mpdu.size // 8 * 8
        self.m8 = m8; self.m = exp_mpdu.size
        self.head = np.frombuffer(exp_mpdu[:m8].tobytes(), np.uint64); self.tail = exp_mpdu[m8:].copy()
        self.pending = {}; self.compared = 0; self.bad = 0

    def _same(self, rows_view, mpdu_view):
        if rows_view.tobytes() != self.rows:
            return False
        if mpdu_view is None:
            return True
        return bool((mpdu_view[:self.m8].view(np.uint64) == self.head).all()) and bool((mpdu_view[self.m8:self.m] == self.tail).all())

    def check(self, key, counts_ok, rows_view, mpdu_view):
        """Queue buffer `key`'s comparison (its call has completed).  mpdu_view None: the row table only."""
        if mpdu_view is not None:
            self.mpdu_compared = getattr(self, "mpdu_compared", 0) + 1
        self.pending[key] = self.pool.submit(self._same, rows_view, mpdu_view) if counts_ok else None

    def release(self, key):
        """Before buffer `key` is written again: its comparison must be over."""
        if key in self.pending:
            f = self.pending.pop(key)
            self.compared += 1
            if f is None or not f.result():
                self.bad += 1

    def drain(self):
        for key in list(self.pending):
            self.release(key)

    def finish(self):
        self.drain()
        self.pool.shutdown()


def pin_rank_threads(local_rank, world):
    """Several ranks share one host: rank r takes the r-th slice of the usable cores, pins the calling (submit) thread to the slice's first core
    and returns the slice (the checker threads take the rest).  With one rank, or fewer than two cores per rank, nothing is pinned."""
    if not hasattr(os, "sched_getaffinity"):
        return []
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // max(1, world)
    if world <= 1 or per < 2:
        return []
    mine = cores[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, {mine[0]})
    except OSError:
        return []
    return mine


def timed_with_delivery(sora_amd, rx, submit, depth, reps, rows_cap, mpdu_cap):
    """The timed region of the widened rows, the headline's protocol: every step = one process call (submit() -> ticket) + deliver_async of
    its dense rows and MPDUs into page-locked host memory behind its kernels + wait for a call in flight (see anyorder) and comparison of the
    table it delivered (row bytes, MPDU bytes) with the first call's.  -> (ms per step, delivery object, the first call's result dicts)"""
    nb = depth + TableChecker.EXTRA
    bufs = [sora_amd.HostResults(rows_cap, mpdu_cap) for _ in range(nb)]
    t = submit(); rx.deliver_async(t, bufs[0]); rx.wait(t)
    first = bufs[0].results()
    n, m = int(bufs[0].counts[0]), int(bufs[0].counts[1])
    chk = TableChecker(bufs[0].rows[:n].tobytes(), bufs[0].mpdu[:m].copy())
    seq = [0]

    # handles with sora_*_wait_any take completions as they happen (and their next call reuses that pipeline); the two-slot handles wait for the older call
    anyorder = hasattr(rx, "wait_any")
    import collections
    free = collections.deque(range(nb)); pend = {}

    def consume_one():
        if anyorder:
            tk = rx.wait_any()
        else:
            tk = min(pend); rx.wait(tk)
        i = pend.pop(tk); b = bufs[i]
        chk.check(i, int(b.counts[0]) == n and int(b.counts[1]) == m, b.rows[:n], b.mpdu[:m])
        free.append(i)

    def block(k):
        for _ in range(k):
            i = free.popleft()
            chk.release(i)
            tk = submit()
            rx.deliver_async(tk, bufs[i]); pend[tk] = i
            if len(pend) >= depth:
                consume_one()
        while pend:
            consume_one()
    block(depth + 2)                                                        # warm-up
    t0 = time.perf_counter()
    block(reps)
    chk.finish()
    ms = (time.perf_counter() - t0) / reps * 1e3
    out = {"enabled": True, "calls_delivered_and_compared": chk.compared, "calls_with_wrong_tables": chk.bad, "rows_per_call": n, "mpdu_bytes_per_call": m,
           "protocol": "every step = process call + deliver_async (dense rows + MPDUs to pinned host memory) + wait for %s of %d calls in flight, whose rows and MPDU bytes "
                       "are compared with the verified first call's by a pool of %d host threads (inside the timed region)" % ("whichever finishes first" if anyorder else "the oldest", depth, TableChecker.EXTRA)}
    for b in bufs:
        b.close()
    return ms, out, first


def reference_gate(first, ncaps, ref_events, same):
    """Every capture of the batch against the compiled reference graph (ref_events(i) -> its events for capture i).  -> parity object"""
    per = [[] for _ in range(ncaps)]
    for r in first:
        per[r["capture_id"]].append(r)
    bad = 0; why0 = ""
    for i in range(ncaps):
        ok, why = same(per[i], ref_events(i))
        if not ok:
            bad += 1; why0 = why0 or "capture %d: %s" % (i, why)
    if bad:
        print("PARITY MISMATCH vs the reference graph: %d captures, first: %s" % (bad, why0), file=sys.stderr)
    return {"against": "reference", "captures_checked": ncaps, "ok": bad == 0, "captures_with_differences": bad}


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
    rx.set_trellis(best); rx.set_depth(1)
    t0 = time.perf_counter()
    for _ in range(10):
        rx.wait(rx.process_dev(f0, f1, descs))
    ms1 = (time.perf_counter() - t0) / 10 * 1e3
    out = {"workload": "%d two-chain captures x one MCS 10 frame, %s (%d samples @40 MHz per chain each), 2x2 cross-talk, AWGN" % (ncaps, what, n),
           "ms": round(ms, 3), "ms_one_call_in_flight": round(ms1, 3), "calls_in_flight": D11N, "trellis_kernel": {64: "k_viterbi11n", 16: "k_viterbi16_11n"}[best],
           "ms_by_trellis_kernel": {"k_viterbi11n": round(res[64][0], 3), "k_viterbi16_11n": round(res[16][0], 3)},
           "msamples_per_s": round(ncaps * n / ms / 1e3, 1), "frames_ok": ok, "frames": ncaps,
           "bound": "hbm", "algorithmic_bytes": 8 * ncaps * n, "achieved": round(8.0 * ncaps * n / ms / 1e6, 1), "peak": HBM_PEAK / 1e9,
           "unit": "GB/s", "frac": round(8.0 * ncaps * n / (ms * 1e-3) / HBM_PEAK, 4), "delivery": delivery}
    if g.available():                                                        # the whole batch against the compiled reference graph, capture by capture
        from gpu_util import same_events_11n
        h0 = iq[0].cpu().numpy(); h1 = iq[1].cpu().numpy()
        out["parity"] = reference_gate(first, ncaps, lambda i: g.rx11n(h0[i], h1[i]), lambda got, want: same_events_11n(got, want, position="sample_index"))
        out["parity"]["both_trellis_kernels_same_table"] = [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in res[64][2]] == [(r["capture_id"], r["error_code"], r["crc32"], r["mpdu"]) for r in res[16][2]]
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


def reference_rows(iq, nfr, oracle):
    """What the reference reports for every capture of the workload: the compiled reference graph (oracle/_ref, fresh
    graph state per capture is not needed: a capture ends in silence and the graph resets after every frame) where it is
    present, else the C restatement.  -> (kind, {capture: [events]})"""
    from oracle.pyoracle import ReferenceGraph
    x = iq.reshape(nfr, CAPTURE_SAMPLES, 2)
    g = ReferenceGraph()
    if g.available():
        return "reference", {i: g.rx11a(np.repeat(x[i], 2, axis=0)) for i in range(nfr)}     # the 40 MHz stream TDownSample2 halves
    return "port", {i: oracle.rx_capture(x[i], 20) for i in range(nfr)}


def check_against_reference(res, kind, want, idx):
    """GPU rows of the captures `idx` against the reference's events (every field the reference reports)."""
    from gpu_util import same_as_reference_graph, same_results
    by_cap = {}
    for r in res:
        by_cap.setdefault(r["capture_id"], []).append(r)
    for i in idx:
        got = by_cap.get(i, [])
        if kind == "reference":
            ok, why = same_as_reference_graph(got, want[i])
        else:
            w = []
            for r in want[i]:
                r = dict(r); r["capture_id"] = i; w.append(r)
            ok, why = same_results(got, w)
        if not ok:
            return False, "capture %d: %s" % (i, why)
    return True, ""



def bench_ht40(torch, sora_amd, dev, nframes=4096):
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
    for lanes in (64, 16):
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
            "ms_by_trellis_kernel": {"k_viterbi11n": round(res[64][0], 3), "k_viterbi16_11n": round(res[16][0], 3)},
            "ms_data_field_only": round(ms_df, 3), "psdus_ok_data_field_only": ok_df,
            "msamples_per_s": round(samples / ms / 1e3, 1), "decoded_mbit_per_s": round(2 * 1500 * 8 * nframes / ms / 1e3, 1),
            "psdus_ok": ok, "psdus": 2 * nframes, "delivery": delivery, "bound": "hbm", "algorithmic_bytes": int(alg), "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(alg / (ms * 1e-3) / HBM_PEAK, 4)}


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-mt-worker":                 # one instance of the reference's two-thread harness (cpu_baseline.two_thread): never returns normally
        p_, nf, first, stride, secs = sys.argv[2].split(",")
        print(json.dumps(list(_cpu_worker((p_, int(nf), int(first), int(stride), float(secs), "reference_mt")))), flush=True)
        os._exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--headline-only", action="store_true", help="stop after the timed region and print its step time only (kernel-trace runs: tools/trace_timeline.sh)")
    ap.add_argument("--wait-oldest", action="store_true", help="the host waits for its OLDEST ticket (round-3 loop) instead of taking completions as they come (sora_rx_wait_any)")
    ap.add_argument("--extra-launches", type=int, default=0, help="experiment: empty kernel launches appended to every call (tool hook sora_internal_rx_extra: needs the TOOLS variant of the library, SORA_HIP_LIB=sora_amd/lib/variants/tools.so from sora_amd.build.build_variant('tools', ['SORA_TOOLS']))")
    ap.add_argument("--no-plain", action="store_true", help="skip the plain_host section (experiments)")
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="captures per GPU (default: the BASELINE config; "
                    "--gpus 8 --frames 32 is BASELINE configs[4] literally: 256 captures over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage / ingest / tx / 11b / 11n sections")
    ap.add_argument("--depth", type=int, default=0, help="process calls in flight on the handle's internal pipelines (0 = library default)")
    ap.add_argument("--trellis", type=int, default=-1, help="trellis kernel: 64 = k_viterbi, 16 = k_viterbi16, 0 = the library's choice from the depth (default: leave the handle as created)")
    ap.add_argument("--check", type=int, default=0, help="captures compared with the reference after the timed region (0 = all)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="the timed region repeats the K-step block until it has lasted this long")
    ap.add_argument("--no-deliver", action="store_true", help="do not deliver rows + MPDUs to the host inside the timed region (round-1 behaviour)")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (read before HIP starts); 0 = leave the runtime default")
    ap.add_argument("--only", default="", help="run just one of the extra sections (stages, ingest, tx, rx11b, rx11b_cck, rx11n, rx11n_40, shard_32x16, latency) and print its object: for profiling that section alone")
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

    if args.only:
        sections = {"stages": lambda: bench_stages(torch, sora_amd, dev), "ingest": lambda: bench_ingest(torch, sora_amd, dev), "tx": lambda: bench_tx(torch, sora_amd),
                    "rx11b": lambda: bench_11b(torch, sora_amd, dev), "rx11b_cck": lambda: bench_11b(torch, sora_amd, dev, cpu=False, rate_kbps=11000),
                    "rx11n": lambda: bench_11n(torch, sora_amd, dev),
                    "rx11n_40": lambda: bench_ht40(torch, sora_amd, dev), "shard_32x16": lambda: bench_shard_shape(torch, sora_amd, dev, Oracle()),
                    "latency": lambda: bench_latency(torch, sora_amd, dev, None, None, None, 0)}
        print(json.dumps({args.only: sections[args.only]()}))
        return
    oracle = Oracle()
    nfr = args.frames
    MAXF = 2
    iq, descs, payloads = make_workload(oracle, nfr, seed0=rank * 100003)
    d_iq = torch.from_numpy(iq).to(dev)
    # one 80 MB input re-read every step sits in the 256 MiB Infinity Cache.  The timed steps rotate through NCOPIES device
    # copies of the batch at different addresses (same samples, so every call's table can be compared with the verified one): 320 MB of input in
    # play, consecutive calls share no line.
    NCOPIES = 4 if world == 1 else 2
    d_iqs = [d_iq] + [d_iq.clone() for _ in range(NCOPIES - 1)]
    descs = sora_amd.Rx.captures(descs)           # packed sora_capture_desc[]: built once, submitted every step
    rx = sora_amd.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, device=local_rank, max_frames_per_capture=MAXF)

    if args.depth:
        rx.set_depth(args.depth)
    depth = rx.set_depth(0)
    if args.trellis >= 0:
        rx.set_trellis(args.trellis)
    if args.extra_launches:
        import ctypes
        _L = sora_amd.capi.load(); _L.sora_internal_rx_extra.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        assert _L.sora_internal_rx_extra(rx._h, args.extra_launches) == 0
    torch.cuda.synchronize()
    rx.wait_for_producer = False                  # the inputs are resident in HBM from here on: no per-call wait for torch's stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the first call is checked before anything is timed: every capture against the reference (or --check of them)
    t = rx.process_dev(d_iq, descs)
    res = rx.results(ticket=t)
    kind, want = reference_rows(iq, nfr, oracle)
    idx = list(range(nfr)) if args.check <= 0 or args.check >= nfr else list(range(0, nfr, max(1, nfr // args.check)))[:args.check]
    parity_ok, why = check_against_reference(res, kind, want, idx)
    if not parity_ok:
        print("PARITY MISMATCH vs %s: %s" % (kind, why), file=sys.stderr)
    n_ok = sum(1 for r in res if r["error_code"] == sora_amd.E_FRAME_OK)
    n_payload_ok = sum(1 for r in res if r["error_code"] == sora_amd.E_FRAME_OK and r["mpdu"][:-4] == payloads[r["capture_id"]])

    # ---- result delivery inside the timed region: after every process call its dense rows and its MPDU array are copied
    # to page-locked host memory behind the kernels (sora_rx_deliver_async), and the oldest call in flight is waited for
    # and its rows compared with the verified ones -- what RxThread does per frame (fb11a_demod.cpp:37-71), per call.
    nb = depth + TableChecker.EXTRA
    bufs = [sora_amd.HostResults(nfr * MAXF, rx.mpdu_bytes(t)) for _ in range(nb)]
    rx.deliver_async(t, bufs[0]); rx.wait(t)
    exp_n = int(bufs[0].nrows[0]); exp_rows = bufs[0].rows[:exp_n].copy(); exp_mpdu = bufs[0].mpdu.copy()
    exp_bytes = exp_rows.tobytes()
    host_rows_ok = exp_n == len(res) and all(int(exp_rows[k]["crc32"]) == res[k]["crc32"] and int(exp_rows[k]["error_code"]) == res[k]["error_code"] for k in range(exp_n)) \
        and all(bytes(exp_mpdu[int(r["mpdu_offset"]):int(r["mpdu_offset"]) + int(r["length"])]) == res[k]["mpdu"] for k, r in enumerate(exp_rows) if int(r["error_code"]) == 1)
    stats = {"t_submit": 0.0, "t_wait": 0.0, "t_check": 0.0}
    share = pin_rank_threads(local_rank, world)                     # this rank's cores: the submit thread on the first, the checker's threads on the others
    chk = TableChecker(exp_bytes, exp_mpdu, cores=share[1:] if len(share) > 1 else None)   # rows AND MPDU bytes of every delivered call, compared by a few host threads

    # Completion order (round 4): calls in flight overtake one another (their streams sit on different dispatch priorities and share the chip), so
    # the loop takes whichever delivered call has finished (sora_rx_wait_any) and the next process call reuses THAT pipeline; --wait-oldest is the
    # round-3 loop (always wait for the oldest ticket), reported beside the headline as in_order_host.
    import collections
    inflight = {}                                                   # ticket -> index of the buffer it is delivered into
    free = collections.deque(range(nb))
    order = {"any": not args.wait_oldest}

    def consume_one():
        ta = time.perf_counter()
        if order["any"]:
            tk = rx.wait_any()
        else:
            tk = min(inflight); rx.wait(tk)
        tb = time.perf_counter()
        stats["t_wait"] += tb - ta
        bi = inflight.pop(tk); b = bufs[bi]
        # every call's rows AND MPDU bytes, whatever the number of ranks: the comparison runs on the checker's threads, which (like the submit
        # thread) are pinned to this rank's own share of the host's cores (pin_rank_threads) -- round 3 sampled every world-th call instead
        chk.check(bi, int(b.nrows[0]) == exp_n, b.rows[:exp_n], b.mpdu)
        free.append(bi)
        stats["t_check"] += time.perf_counter() - tb

    def run_block(k, deliver, dep=None):
        dep = dep or depth                                          # calls in flight on the handle right now (<= len(bufs) - EXTRA)
        for _ in range(k):
            ta = time.perf_counter()
            if deliver:
                bi = free.popleft()
                chk.release(bi)                                     # the buffer the next call will be delivered into: its comparison must be over
            tk = rx.process_dev(d_iqs[(rx.ticket() + 1) % NCOPIES], descs)
            if deliver:
                rx.deliver_async(tk, bufs[bi]); inflight[tk] = bi
            stats["t_submit"] += time.perf_counter() - ta
            if deliver and len(inflight) >= dep:
                consume_one()
        while inflight:
            consume_one()

    deliver = not args.no_deliver
    run_block(args.warmup, deliver)
    rx.flush()
    barrier()
    run_block(args.steps, deliver); rx.flush()                       # (the first block after the warm-up still pays first-use costs: size the region from the second)
    t0 = time.perf_counter()
    run_block(args.steps, deliver); rx.flush()
    probe = time.perf_counter() - t0
    repeats = max(1, int(np.ceil(args.min_seconds / max(probe, 1e-6))))
    if world > 1:                                  # every rank runs the same number of blocks
        r_t = torch.tensor([repeats], device=dev, dtype=torch.int64); dist.all_reduce(r_t, op=dist.ReduceOp.MAX); repeats = int(r_t.item())
    chk.drain(); chk.compared = chk.bad = 0; chk.mpdu_compared = 0
    barrier()
    for k_ in ("t_submit", "t_wait", "t_check"):
        stats[k_] = 0.0
    t0 = time.perf_counter()
    run_block(args.steps * repeats, deliver)                        # ONE continuous run of K x repeats steps: the calls in flight are collected at its end only
    rx.flush()                                                      # (round 3 drained the handle after every K steps: with K = 20 and eight calls in flight a fifth of the region was fill and drain)
    chk.drain()                                                     # every delivered table has been compared when the clock stops
    barrier()
    t1 = time.perf_counter()
    host_ms = {k_[2:]: round(1e3 * stats[k_] / (args.steps * repeats), 4) for k_ in ("t_submit", "t_wait", "t_check")}
    if args.headline_only:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(1e3 * (t1 - t0) / (args.steps * repeats), 4), "steps": args.steps * repeats, "calls_in_flight": depth, "captures_per_call": nfr,
                              "calls_with_wrong_rows": chk.bad, "host_ms_per_step": host_ms}))
        return
    timed_steps = args.steps * repeats
    stats["delivered"], stats["bad"] = chk.compared, chk.bad
    stats["mpdu_compared_timed"] = getattr(chk, "mpdu_compared", 0)
    mpdu_ok = bool(deliver) and all((b.mpdu == exp_mpdu).all() for b in bufs)     # the last calls' MPDU arrays once more, byte for byte
    # the same K steps once more with HIP events around every kernel launch (on the streams the kernels run on): the
    # roofline's launch durations are means over this region; `value` comes from the un-instrumented region above
    # (recording 6 events per call costs a few percent, reported as ms_per_step_profiled)
    rx.set_profiling(True)
    barrier()
    t2 = time.perf_counter()
    run_block(args.steps, deliver)
    rx.flush()
    barrier()
    t3 = time.perf_counter()
    ktimes = rx.kernel_times()
    rx.set_profiling(False)
    # and with ONE call in flight: each kernel alone on the chip (the per-kernel roofline without the CU sharing of overlapped calls).
    # The trellis kernel is pinned to the one the timed region used (left to itself the library picks k_viterbi for a single call in
    # flight and k_viterbi16 from depth 4); the other one is measured alone as well, for the record.
    lanes = rx.trellis(); trellis_setting = rx.set_trellis(-1)
    tname = TRELLIS_NAMES

    def alone(l):
        rx.set_trellis(l); rx.set_depth(1); rx.flush()
        rx.set_profiling(True)
        for _ in range(max(10, args.steps // 2)):
            rx.process_dev(d_iq, descs)
        rx.flush()
        kt = rx.kernel_times()
        rx.set_profiling(False)
        return {(tname[l] if k == "k_viterbi" else k): v for k, v in kt.items()}
    ktimes1 = alone(lanes)
    ktimes1_others = {l: alone(l) for l in TRELLIS_NAMES if l != lanes}
    ktimes = {(tname[lanes] if k == "k_viterbi" else k): v for k, v in ktimes.items()}

    # ---- what a plain host gets: process -> deliver -> wait with ONE or TWO calls in flight, i.e. at most two of the
    # handle's streams in use -- fewer than the runtime's default four hardware queues, so GPU_MAX_HW_QUEUES plays no part -- same timed-region
    # protocol (delivery + comparison inside), each trellis kernel pinned in turn.
    plain = {}
    if world == 1 and not args.no_plain:
        if order["any"] and deliver:                                 # the headline's calls in flight with the round-3 loop: always wait for the OLDEST ticket
            rx.set_trellis(lanes); rx.set_depth(depth); rx.flush(); order["any"] = False
            run_block(args.warmup, deliver); rx.flush(); chk.drain(); bad0 = chk.bad
            nblk = max(1, repeats // 4)
            tp0 = time.perf_counter()
            run_block(args.steps * nblk, deliver)
            rx.flush(); chk.drain()
            ms_p = (time.perf_counter() - tp0) / (nblk * args.steps) * 1e3
            plain["calls_in_flight_%d_waiting_for_the_oldest_ticket" % depth] = {"ms_per_step": round(ms_p, 4), "msamples_per_s": round(nfr * FRAME_SAMPLES / ms_p / 1e3, 1),
                                                                                 "calls_with_wrong_rows": chk.bad - bad0}
            order["any"] = True
        for dval in (1, 2):
            for l in (64, 16, 1):
                rx.set_trellis(l); rx.set_depth(dval); rx.flush()
                run_block(args.warmup, deliver, dval); rx.flush(); chk.drain(); bad0 = chk.bad
                nblk = max(1, repeats // 6)
                tp0 = time.perf_counter()
                run_block(args.steps * nblk, deliver, dval)
                rx.flush(); chk.drain()
                ms_p = (time.perf_counter() - tp0) / (nblk * args.steps) * 1e3
                plain["calls_in_flight_%d_%s" % (dval, tname[l])] = {"ms_per_step": round(ms_p, 4), "msamples_per_s": round(nfr * FRAME_SAMPLES / ms_p / 1e3, 1),
                                                                      "calls_with_wrong_rows": chk.bad - bad0}
        best1 = min((k for k in plain if k.startswith("calls_in_flight_1_")), key=lambda k: plain[k]["ms_per_step"])
        plain["calls_in_flight_1"] = dict(plain[best1], config=best1)           # (the library's automatic choice for one lone call of this size is the window-parallel trellis)
        best2 = min((k for k in plain if k.startswith("calls_in_flight_2")), key=lambda k: plain[k]["ms_per_step"])
        plain["best_with_at_most_two_calls_in_flight"] = dict(plain[best2], config=best2)
        try:
            plain["two_calls_in_flight_of_%d_captures" % (8 * nfr)] = bench_large_call(torch, sora_amd, local_rank, d_iqs, nfr, MAXF, exp_rows, exp_mpdu, args.min_seconds,
                                                                                           share[1:] if len(share) > 1 else None)
        except Exception as e:                                      # (a second 320 MB input and its workspace: report, do not lose the line)
            plain["two_calls_in_flight_of_%d_captures" % (8 * nfr)] = {"error": repr(e)}
        plain["note"] = ("the headline keeps %d calls in flight%s; with at most two, the chip holds at most 8192 frames = 1024 waves of k_viterbi16 (one per SIMD, each bound by its own "
                         "issue rate) or 4096 of k_viterbi (1.7x the instructions): DESIGN.md section 3.6, profiles/r04_a_depth_table.txt" % (depth, " on %s hardware queues" % os.environ["GPU_MAX_HW_QUEUES"] if os.environ.get("GPU_MAX_HW_QUEUES") else ""))
    rx.set_trellis(trellis_setting); rx.set_depth(depth); rx.flush()
    latency = e2e = None
    if world == 1 and not args.no_extras:
        latency = bench_latency(torch, sora_amd, dev, rx, d_iq, descs, nfr)
        e2e = bench_e2e(torch, sora_amd, dev, rx, iq, nfr, exp_rows, exp_mpdu)

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    counters = torch.tensor([len(res), n_ok, n_payload_ok, stats["delivered"], stats["bad"]], device=dev, dtype=torch.int64)
    gathered_rows = None; gathered = None
    if world > 1:
        # the path's one exchange step: RCCL all-gathers of the device-packed result rows AND the MPDUs (every MPDU
        # reaches the one host buffer, fb11a_demod.cpp:64-70) + all-reduce of counters.  Per rank and exchange: 8 bytes of counts,
        # 36 bytes x 2 x captures of rows, MPDU_LEN x captures of MPDU bytes.
        gathered = exchange_results(torch, rx, d_iq, descs, dev, nfr, MAXF)
        gathered_rows = gathered["rows"]
        dist.all_reduce(counters)
    tot_frames, tot_ok, tot_payload_ok, tot_delivered, tot_bad = [int(v) for v in counters.tolist()]

    total_samples = float(nfr) * FRAME_SAMPLES * world * timed_steps
    msps = total_samples / elapsed / 1e6
    ms_per_step = elapsed / timed_steps * 1e3
    if rank == 0:
        dom = max((k for k in ktimes if k.startswith("k_")), key=lambda k: ktimes[k])
        launch_bytes = nfr * FRAME_SAMPLES * ALG_BYTES_PER_SAMPLE
        ach = launch_bytes / (ktimes[dom] * 1e-3)
        ach1 = launch_bytes / (ktimes1[dom] * 1e-3)
        air_s = nfr * CAPTURE_SAMPLES / 20e6                                   # what one step's captures last on the air
        out = {
            "metric": "IQ Msamples/s through 802.11a 54 Mbps RX PHY",
            "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_steps": timed_steps, "timed_seconds": round(elapsed, 4),
            "ms_per_step": round(ms_per_step, 4), "ms_per_step_profiled": round((t3 - t2) / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 IQ / u8 path metrics", "data": "synthetic",
            "config": {"workload": "802.11a 54 Mbps (64-QAM r=3/4) RX, %d captures/GPU x one 1500-byte frame (4880 samples @20 MHz, +160 silence), AWGN 30/27 dB on 3 of 4" % nfr,
                       "frames_per_gpu": nfr, "samples_per_frame": FRAME_SAMPLES, "capture_samples": CAPTURE_SAMPLES, "input_copies_rotated": NCOPIES, "input_bytes_in_play": int(NCOPIES * iq.nbytes), "calls_in_flight": depth, "completions": "as they happen (sora_rx_wait_any)" if order["any"] else "oldest ticket first (sora_rx_wait)", "trellis_kernel": tname[lanes], "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "sharding": "captures per rank, no data-path collective",
                       "timed_region": "%d x %d steps in one continuous run; every step = process call + pack + async delivery of rows and MPDUs to pinned host memory + wait for %s, whose rows and MPDU bytes are compared with the verified ones by %d host threads%s" % (repeats, args.steps, "whichever call in flight finishes first (sora_rx_wait_any)" if order["any"] else "the oldest call in flight", TableChecker.EXTRA, "" if world == 1 else " (every rank pins its submit thread and its checker threads to its own slice of the host's cores)")
                                       if deliver else "%d x %d process calls, nothing delivered" % (repeats, args.steps)},
            "decoded_mbit_per_s": round(msps * (MPDU_LEN * 8.0 / FRAME_SAMPLES), 2),
            "frames": tot_frames, "gathered_rows": gathered_rows, "gathered": gathered, "frames_crc_ok": tot_ok, "frames_payload_ok": tot_payload_ok,
            "parity": {"against": kind, "captures_checked": len(idx), "ok": parity_ok, "host_rows_ok": host_rows_ok},
            "host_ms_per_step": host_ms,
            "delivery": {"enabled": deliver, "calls_delivered_and_compared": tot_delivered, "calls_with_wrong_rows": tot_bad, "rows_per_call": exp_n,
                         "row_bytes_per_call": 36 * nfr * MAXF, "mpdu_bytes_per_call": int(exp_mpdu.size), "last_calls_mpdu_ok": mpdu_ok, "calls_with_mpdu_bytes_compared": stats.get("mpdu_compared_timed", 0)},
            # the reference's own figure of merit (MACStopwatch.h:84-128): cost / required time, < 1 = faster than real time
            "realtime": {"factor": round(ms_per_step * 1e-3 / air_s, 7), "channels_20mhz_in_real_time": round(air_s / (ms_per_step * 1e-3), 1),
                         "call_latency_ms_one_in_flight": round(sum(v for k, v in ktimes1.items()), 4)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach1 / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(ach1 / HBM_PEAK, 5), "traffic": measured_traffic(dom) if nfr == FRAMES_PER_GPU else None,
                         "traffic_source": _traffic_profile()[1],
                         "algorithmic_bytes_per_launch": launch_bytes, "kernel_ms": round(ktimes1[dom], 4),
                         "kernel_ms_note": "mean launch duration with ONE call in flight (the kernel alone on the chip); with %d calls overlapped the same launch lasts %.4f ms (frac %.5f) because it shares the CUs" % (depth, ktimes[dom], ach / HBM_PEAK),
                         "whole_path_frac": round(msps * 1e6 / world * ALG_BYTES_PER_SAMPLE / HBM_PEAK, 5),
                         "other_trellis_kernels": {tname[l]: {"kernel_ms": round(kt[tname[l]], 4), "frac": round(launch_bytes / (kt[tname[l]] * 1e-3) / HBM_PEAK, 5)} for l, kt in ktimes1_others.items()},
                         "other_trellis_kernels_note": "sora_rx_set_trellis, each alone on the chip: k_viterbi = two frames per wave, k_viterbi16 = eight per wave (the one for 32768 and more captures in flight), k_viterbi16w = the frames' "
                                                       "trace-back windows decoded side by side and proven afterwards, with k_win_redo (the proof) in its time (the one below that; the automatic choice follows depth x max_captures)",
                         "valu": valu_roofline(nfr, ms_per_step)},
            "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
            "kernel_ms_one_call_in_flight": {k: round(v, 4) for k, v in ktimes1.items()},
        }
        try:                                                        # where the library this run loaded came from
            from sora_amd import build as _b
            out["build"] = dict(_b.build_info(), loaded=os.environ.get("SORA_HIP_LIB") or "sora_amd/lib/libsora_hip.so")
        except Exception as e:
            out["build"] = {"error": repr(e)}
        if plain:
            out["plain_host"] = plain
        if latency is not None:
            out["latency"] = latency
        if e2e is not None:
            out["e2e"] = e2e
        if world == 1 and not args.no_extras:
            out["stages"] = bench_stages(torch, sora_amd, dev)
            out["ingest"] = bench_ingest(torch, sora_amd, dev)
            out["tx"] = bench_tx(torch, sora_amd)
            out["rx11b"] = bench_11b(torch, sora_amd, dev)
            out["rx11b_cck"] = bench_11b(torch, sora_amd, dev, cpu=False, rate_kbps=11000)
            out["rx11n"] = bench_11n(torch, sora_amd, dev)
            out["rx11n_40"] = bench_ht40(torch, sora_amd, dev)
            out["shard_32x16"] = bench_shard_shape(torch, sora_amd, dev, oracle)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(iq, nfr)
            out["realtime"]["cpu_reference_factor_one_core"] = round(20.0 / out["cpu_baseline"]["single_core_value"], 4) if out["cpu_baseline"].get("single_core_value") else None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
