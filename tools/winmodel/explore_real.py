"""The window-parallel trellis's verification failure rate on the soft streams of the restated receive chain (oracle): the bench workload's
noise levels, fsample-6, and a noise sweep at every rate up to the point where frames stop decoding."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
from explore import run
from oracle.pyoracle import Oracle, rate_params
from gpu_util import make_capture

o = Oracle()
def soft_of(cap, mhz):
    res, tr = o.rx_capture(cap, mhz, trace=True)
    if not res or res[0]["error_code"] not in (1, 0x80000006): return None
    r = res[0]
    return tr["soft"].copy(), rate_params(r["rate_kbps"])[1], r["length"], r["error_code"] == 1

if os.path.exists("/root/reference/kernel/test-data/fsample-6.dmp"):
    iq = o.load_dump("/root/reference/kernel/test-data/fsample-6.dmp", raw14=True)
    s, cr, ln, ok = soft_of(iq, 40)
    for W in (48, 96, 144):
        f, nu, same, _ = run(s, cr, ln, W, 1)
        print(f"fsample-6: cr={cr} len={ln} crc_ok={ok} W={W}: failed boundaries {f}/{nu - 1} same={same}")
for rate in (54000, 48000, 36000, 24000, 12000, 6000):
    for sigma in (0, 300, 420, 600, 800, 1100, 1500, 2000):
        for W in (96, 144):
            nf = nb = nfr = ncrc = n = 0
            for i in range(12):
                cap, _ = make_capture(o, rate, 1500, seed=1000 + i, rate_mhz=20, sigma=sigma)
                g = soft_of(cap, 20)
                if g is None: continue
                s, cr, ln, ok = g
                f, nu, same, _ = run(s, cr, ln, W, 1)
                assert f > 0 or same
                n += 1; nf += f; nb += nu - 1; nfr += f > 0; ncrc += ok
            print(f"rate {rate} sigma {sigma:4d} W={W}: frames {n}, crc ok {ncrc}, failed boundaries {nf}/{nb}, frames with a failure {nfr}")
