"""Developer run: bench_latency's batch_per_frame object alone (the 4096-frame batch as one call and as k calls taken as they complete)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sora_amd, bench
from oracle.pyoracle import Oracle
from benchlib.latency import bench_latency
nfr = 4096
iq, descs, _ = bench.make_workload(Oracle(), nfr, 0, distinct=64)
rx = sora_amd.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
d = torch.from_numpy(iq).cuda(); torch.cuda.synchronize(); rx.wait_for_producer = False
out = bench_latency(torch, sora_amd, torch.device("cuda", 0), rx, d, descs, nfr, reps=30)
print(json.dumps(out["batch_per_frame"], indent=1))
