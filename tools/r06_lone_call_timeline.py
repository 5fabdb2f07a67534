"""Developer probe: where a lone delivered 4096-capture call spends its time -- the library's timed intervals (set_profiling) for the plain and the bound delivery,
beside the host's wall clock for process -> deliver -> wait."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sora_amd, bench
from oracle.pyoracle import Oracle
nfr = 4096
iq, descs, _ = bench.make_workload(Oracle(), nfr, 0)
descs = sora_amd.Rx.captures(descs)
rx = sora_amd.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
d = torch.from_numpy(iq).cuda(); torch.cuda.synchronize(); rx.wait_for_producer = False
rx.set_depth(1); rx.set_trellis(1)
buf = sora_amd.HostResults(nfr * 2, rx.mpdu_bytes(rx.process_dev(d, descs))); rx.flush()
for bound in (False, True):
    for prof in (False, True):
        rx.set_profiling(prof)
        ts = []
        for i in range(40):
            t0 = time.perf_counter()
            if bound: rx.bind_mpdu(buf)
            tk = rx.process_dev(d, descs); t1 = time.perf_counter(); rx.deliver_async(tk, buf); t2 = time.perf_counter(); rx.wait(tk); t3 = time.perf_counter()
            if i >= 5: ts.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
        a = np.median(np.asarray(ts), axis=0) * 1e3
        print("bound" if bound else "plain", "profiling" if prof else "", "process %.3f deliver %.3f wait %.3f total %.3f ms" % tuple(a), rx.kernel_times() if prof else "")
