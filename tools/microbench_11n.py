"""Where k_rx11n's time goes: the same batch as noise only (carrier sense alone), and with one frame per capture at MCS 8 / 9 / 10
(symbols per frame vs trellis steps per frame differ).  Needs the compiled reference modulator (oracle/_ref) for the waveforms.
usage (on the GPU box): python tools/microbench_11n.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import sora_amd
from oracle.pyoracle import ReferenceGraph
g = ReferenceGraph()
def run(base0, base1, ncaps, label, sigma=20.0):
    n = base0.shape[0]; dev = "cuda:0"
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    iq = torch.empty((2, ncaps, n, 2), dtype=torch.int16, device=dev)
    b = [torch.from_numpy(base0).to(dev), torch.from_numpy(base1).to(dev)]
    for i in range(0, ncaps, 64):
        k = min(64, ncaps - i)
        for c in range(2):
            iq[c, i:i+k] = (b[c][None] + sigma * torch.randn((k, n, 2), generator=gen, device=dev)).round().clamp(-32768, 32767).to(torch.int16)
    descs = sora_amd.Rx.captures([(i * n, n, i) for i in range(ncaps)])
    rx = sora_amd.Rx11n(ncaps, ncaps * n, max_frames_per_capture=4)
    f0 = iq[0].view(-1, 2); f1 = iq[1].view(-1, 2)
    torch.cuda.synchronize(); rx.process_dev(f0, f1, descs); res = rx.results()
    ms = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(5): rx.process_dev(f0, f1, descs)
        rx.synchronize(); ms = min(ms, (time.perf_counter() - t0) / 5 * 1e3)
    print("%-28s ncaps %5d n %6d  %.3f ms  %.1f Msamples/s  events %d ok %d" % (label, ncaps, n, ms, ncaps * n / ms / 1e3, len(res), sum(r["error_code"] == 1 for r in res)))
    rx.close()
rng = np.random.default_rng(12)
for mcs, ln in ((10, 1000), (8, 1000), (9, 100)):
    s0, s1 = g.tx11n(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), mcs)
    n = (len(s0) + 2000 + 27) // 28 * 28
    base = np.zeros((2, n, 2), np.float32); base[0, 800:800+len(s0)] = s0 + 0.1 * s1; base[1, 800:800+len(s0)] = s1 + 0.1 * s0
    for nc in (4096, 8192) if mcs == 10 else (4096,):
        run(base[0], base[1], nc, "mcs %d len %d" % (mcs, ln))
    if mcs == 10: run(np.zeros_like(base[0]), np.zeros_like(base[1]), 4096, "noise only (same length)")
