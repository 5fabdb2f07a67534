"""benchlib.cpu -- the CPU baseline of bench.py: the reference's compiled graph (single-thread and its native two-thread harness) and the scalar restatement on the host's cores."""
import json
import os
import sys
import time

import numpy as np

from benchlib.common import *  # noqa: F401,F403
from benchlib.common import _traffic_profile  # noqa: F401


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
                ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-mt-worker", "%s,%d,%d,%d,%g" % (path, nframes, k, n_inst, secs)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
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
