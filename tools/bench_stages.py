"""The per-stage section of bench.py on its own (streaming stage kernels + ingest): python tools/bench_stages.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import sora_amd  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
out = {"stages": bench.bench_stages(torch, sora_amd, dev), "ingest": bench.bench_ingest(torch, sora_amd, dev)}
for k, v in list(out["stages"].items()) + [("ingest", out["ingest"])]:
    print("%-24s %8.4f ms  %8.1f GB/s  frac %.4f" % (k, v["ms"], v["achieved"], v["frac"]))
print(json.dumps(out))
