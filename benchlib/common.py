"""benchlib.common -- what every section of bench.py shares: the workload (BASELINE.json configs[2]), its constants, the reference gates, the delivered-and-compared
timed loop, and the replayed counter summaries.  (Round 5 split bench.py -- one 1500-line file -- into benchlib/: common, cpu, stages, latency, rows.)"""
import json
import os
import sys
import time

import numpy as np

TRELLIS_NAMES = {64: "k_viterbi", 16: "k_viterbi16", 1: "k_viterbi16w"}      # sora_rx_set_trellis: two frames per wave / eight per wave / window-parallel (round 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


FRAMES_PER_GPU = 4096
MPDU_LEN = 1500            # incl. FCS
RATE_KBPS = 54000
FRAME_SAMPLES = 4880       # 160 STS + 160 LTS + 80 SIGNAL + 56*80 data @20 MHz
CAPTURE_SAMPLES = 5040     # + 160 silence; 360 source bursts of 14
ALG_BYTES_PER_SAMPLE = 4.0 + 216 / 8.0 / 80.0     # 4.3375
HBM_PEAK = 8.0e12
PROFILES = os.path.join(ROOT, "profiles")
TRAFFIC_JSON = os.path.join(PROFILES, "r06_final_traffic.json")          # rocprofv3 --pmc passes of this command (tools/collect_profiles.sh): replayed, not measured by this run
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


def reference_rows(iq, nfr, oracle, fpc=1):
    """What the reference reports for every capture of the workload: the compiled reference graph (oracle/_ref, fresh
    graph state per capture is not needed: a capture ends in silence and the graph resets after every frame) where it is
    present, else the C restatement.  -> (kind, {capture: [events]})"""
    from oracle.pyoracle import ReferenceGraph
    x = iq.reshape(nfr, fpc * CAPTURE_SAMPLES, 2)                      # (fpc frames back to back per capture: the --shape shard workload)
    g = ReferenceGraph()
    if g.available():
        return "reference", {i: g.rx11a(np.repeat(x[i], 2, axis=0), max_frames=fpc + 4) for i in range(nfr)}     # the 40 MHz stream TDownSample2 halves
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


def exchange_results(torch, rx, d_iq, descs, dev, nfr, maxf, fpc=1):
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
        seen = {}                                                        # (capture, time order: the j-th row of a capture is its j-th frame = frame capture * fpc + j of the rank's workload)
        for w in ar[k:k + cnt]:
            j = seen.get(int(w[0]), 0); seen[int(w[0])] = j + 1
            if int(w[3]) == 1 and j < fpc:
                o_, ln = int(w[8]), int(w[5] & 0xFFFF)
                okm += bytes(am[o_:o_ + ln - 4]) == workload_payload(rr * 100003, int(w[0]) * fpc + j, nfr * fpc)
        k += cnt
    return {"rows": int(allrows.shape[0]), "rows_per_rank": per_rank, "mpdu_bytes": int(am.size), "mpdus_equal_to_the_transmitted_payloads": int(okm),
            "exchange_ms": round((tg1 - tg0) * 1e3, 3),
            "bytes_per_rank": {"counts": 8, "rows": 36 * nfr * maxf, "mpdu_block": nfr * maxf * MPDU_LEN}}
