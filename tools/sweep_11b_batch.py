"""How the 802.11b kernel's time grows with the number of captures in a batch (one wave per capture): tells a kernel that is bound by the
instruction issue of a SIMD (time ~ captures) from one bound by the latency of each wave's own dependent chain (time ~ rounds of resident waves).
    python tools/sweep_11b_batch.py 2048 4096 6144 8192 12288 16384"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, sora_amd, bench

if __name__ == "__main__":
  dev = torch.device("cuda:0")
  for n in [int(a) for a in sys.argv[1:]] or [4096, 8192]:
    r = bench.bench_11b(torch, sora_amd, dev, ncaps=n, reps=5, cpu=False)
    print(json.dumps({"captures": n, "ms": r["ms"], "us_per_capture": round(r["ms"] * 1e3 / n, 4), "frames_ok": r["frames_ok"]}), flush=True)
