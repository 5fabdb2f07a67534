"""Developer probe: where does the host spend a step of the rx11n_40 bench row?  Wraps the handle's calls with timers and runs the row."""
import os, sys, time, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sora_amd
from benchlib import rows, common
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); d = time.perf_counter() - t0
        acc[cls.__name__ + "." + name][0] += d; acc[cls.__name__ + "." + name][1] += 1
        return r
    setattr(cls, name, g)
for n in ("process_captures_dev", "process_dev", "deliver_async", "wait_any", "wait"):
    wrap(sora_amd.RxHt40, n)
wrap(common.TableChecker, "check"); wrap(common.TableChecker, "release")
out = rows.bench_ht40(torch, sora_amd, torch.device("cuda", 0), trellis=(16,))
print(json.dumps({k: out[k] for k in ("ms", "ms_by_trellis_kernel", "ms_data_field_only")}))
for k, (t, n) in sorted(acc.items()):
    print("%-36s %6d calls  %8.3f ms each" % (k, n, t / n * 1e3))
# pure host cost of a submit: the handle idle, one call, timed until the call returns (not until the GPU finishes)
from oracle import py_ht40 as m
import numpy as np
rng = np.random.default_rng(40)
ps = [m.add_fcs(rng.integers(0, 256, 1496, dtype=np.uint8).tobytes()) for _ in range(2)]
x, nsym, pre = m.tx_frame(ps, 14)
n = (400 + x.shape[1] + 600 + 27) // 28 * 28
nframes = 4096
iq = torch.zeros((2, nframes, n, 2), dtype=torch.int16, device="cuda")
y = (np.array([[1.0, 0.3j], [0.25, 0.9 * np.exp(0.7j)]]) @ x) * 250.0
b = np.zeros((2, n, 2), np.float32); b[:, 400:400 + y.shape[1], 0] = y.real; b[:, 400:400 + y.shape[1], 1] = y.imag
iq[:] = torch.from_numpy(b).to("cuda").round().to(torch.int16)[:, None]
caps = sora_amd.Rx.captures([(i * n, n, i) for i in range(nframes)])
rx = sora_amd.RxHt40(nframes, nframes * 2 * (nsym * 648 + 64))
f0 = iq[0].view(-1, 2); f1 = iq[1].view(-1, 2)
for _ in range(10):
    rx.process_captures_dev(f0, f1, caps, max_frames_per_capture=2)
rx.synchronize()
for k in range(4):
    t0 = time.perf_counter(); tk = rx.process_captures_dev(f0, f1, caps, max_frames_per_capture=2); t1 = time.perf_counter(); rx.wait(tk); t2 = time.perf_counter()
    print("idle handle: submit %.3f ms, then until done %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
