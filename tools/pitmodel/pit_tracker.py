"""Design model (a tool): the pilot tracker's chain (TPhaseCompensate + TPilotTrack, freqoffset.hpp:28-30, pilot.hpp:166-233) solved PARALLEL IN TIME with an
exact fixed-point test.

The chain carries four 16-bit numbers from symbol to symbol (CFO_comp, SFO_comp and the two trackers) through two table look-ups per pilot: 465 symbols of
fsample-6 cost the GPU 0.30 ms as one serial chain.  Cut the frame's symbols into S segments; give every segment a GUESS of the state at its start; run all
segments side by side, each an exact copy of the serial chain from its guess; then compare: guess(segment j + 1) == end(segment j) for every j, with segment 0
started from the frame's true state, PROVES by induction that every segment ran from the true state, i.e. that the whole trajectory is the serial chain's.
Otherwise take the ends as the new guesses and go again.  After iteration k segments 0 .. k - 1 are certainly exact, so S iterations always suffice; the loop is
a phase-locked loop that forgets its starting state within a few symbols, so a handful do.  This file counts them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
from oracle.pyoracle import Oracle  # noqa: E402

o = Oracle()
USIN = o.usin_lut().astype(np.int64); UCOS = o.ucos_lut().astype(np.int64); UATAN = o.uatan2_lut().astype(np.int64)
SGN = [0,0,0,1,1,1,0,1, 1,1,1,0,0,1,0,1, 1,0,0,1,0,0,1,0, 0,0,0,0,0,1,0,0, 0,1,0,0,1,1,0,0, 0,1,0,1,1,1,0,1, 0,1,1,0,1,1,0,0, 0,0,0,1,1,0,0,1,
       1,0,1,0,1,0,0,1, 1,1,0,0,1,1,1,1, 0,1,1,0,1,0,0,0, 0,1,0,1,0,1,0,1, 1,1,1,1,0,1,0,0, 1,0,1,0,0,0,1,1, 0,1,1,1,0,0,0,1, 1,1,1,1,1,1,0,0]
PBIN = (43, 57, 7, 21); PC = (-21, -7, 7, 21)


def w16(v): return ((int(v) + 32768) & 0xFFFF) - 32768
def w32(v): return ((int(v) + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
def cdiv(a, b): return int(abs(a) // b) * (1 if a >= 0 else -1)      # C division: towards zero


def mul_q15(a, b):
    re = w32(a[0] * b[0] + a[1] * w16(-b[1])); im = w32(a[0] * b[1] + a[1] * b[0])
    return w16(re >> 15), w16(im >> 15)


def uatan2(y, x):
    def scope(v):
        a = abs(v); return a.bit_length() - 1 if a else 0
    sh = max(scope(x), scope(y)) - 6
    if sh > 0: y >>= sh; x >>= sh
    return int(UATAN[(y & 0xFF) * 256 + (x & 0xFF)])


def step(state, pil, count):
    """one symbol: state = (cfo, sfo, ctr, str) before it -> (state after it, (cfo, sfo, avg, del) = what the symbol's rotation uses)"""
    cfo, sfo, ctr, st = state
    th = []
    for k in range(4):
        arg = w16(cfo + PC[k] * sfo) & 0xFFFF
        p = mul_q15(pil[k], (int(UCOS[arg]), w16(-int(USIN[arg]))))
        t = uatan2(-p[1], -p[0]) if k == 3 else uatan2(p[1], p[0])
        if SGN[count]: t = w16(t + 0x8000)
        th.append(t)
    avg = w16(cdiv(th[0] + th[1] + th[2] + th[3], 4))
    dl = w16((cdiv(th[2] - th[0], 28) + cdiv(th[3] - th[1], 28)) >> 1)
    ctr = w16(ctr + (avg >> 2)); st = w16(st + (dl >> 2))
    return (w16(cfo + avg + ctr), w16(sfo + dl + st), ctr, st), (cfo, sfo, avg, dl)


def serial(state0, pilots):
    s = state0; traj = [s]
    for i, p in enumerate(pilots):
        s, _ = step(s, p, i % 127)
        traj.append(s)
    return traj


def pit(state0, pilots, nseg, guess="hold"):
    """-> (iterations until the fixed point, trajectory at segment starts)"""
    n = len(pilots); L = -(-n // nseg)
    starts = [min(j * L, n) for j in range(nseg + 1)]
    g = [state0] * nseg
    if guess == "ramp":                                           # CFO_comp grows by about the tracker per symbol
        g = [(w16(state0[0] + starts[j] * state0[2]), w16(state0[1] + starts[j] * state0[3]), state0[2], state0[3]) for j in range(nseg)]
    for it in range(1, nseg + 2):
        ends = []
        for j in range(nseg):
            s = g[j]
            for i in range(starts[j], starts[j + 1]):
                s, _ = step(s, pilots[i], i % 127)
            ends.append(s)
        ok = all(g[j + 1] == ends[j] for j in range(nseg - 1))
        if ok:
            return it, g, ends
        g = [state0] + ends[:-1]
    raise AssertionError("no fixed point")


def frame_pilots(cap, mhz):
    res, tr = o.rx_capture(cap, mhz, trace=True)
    if not res or res[0]["error_code"] not in (1, 0x80000006): return None
    eq = tr["eq"].astype(np.int64)                               # [nsym + 1][64][2]: symbol 0 is SIGNAL
    c = tr["ctx"]
    pil = [[(int(eq[s][b][0]), int(eq[s][b][1])) for b in PBIN] for s in range(len(eq))]
    return pil, res[0]


if __name__ == "__main__":
    from gpu_util import make_capture
    cases = []
    if os.path.exists("/root/reference/kernel/test-data/fsample-6.dmp"):
        cases.append(("fsample-6", o.load_dump("/root/reference/kernel/test-data/fsample-6.dmp", raw14=True), 40))
    for rate, ln, sigma, cfo in ((54000, 1500, 300, 0), (6000, 1500, 800, 30e3), (24000, 2000, 600, -60e3), (6000, 2500, 1500, 10e3), (54000, 1500, 0, 0)):
        cases.append(("%d/%d/s%d/cfo%g" % (rate, ln, sigma, cfo), make_capture(o, rate, ln, seed=rate // 1000 + ln, rate_mhz=20, sigma=sigma, cfo_hz=cfo)[0], 20))
    for name, cap, mhz in cases:
        g = frame_pilots(cap, mhz)
        if g is None: print(name, "no frame"); continue
        pil, r = g
        # the chain starts at the SIGNAL symbol from the all-zero tracking state (fb11ademod_config.hpp:68-95); the data symbols' chain starts from what SIGNAL leaves
        s0 = (0, 0, 0, 0)
        s1, _ = step(s0, pil[0], 127 % 127 if False else 0)       # SIGNAL is symbol_count 127 -> the reference resets the count: see k_frame ("127 -> 0 after the SIGNAL symbol")
        data = pil[1:]
        tr = serial(s1, data)
        for nseg in (16, 64, 128):
            for guess in ("hold", "ramp"):
                it, gs, ends = pit(s1, data, nseg, guess)
                L = -(-len(data) // nseg)
                assert all(gs[j] == tr[min(j * L, len(data))] for j in range(nseg)), "fixed point is not the serial trajectory!"
                print(f"{name}: {len(data)} symbols, {nseg} segments of {L}, guess {guess}: {it} iterations = {it * L} symbol-times (serial: {len(data)})")
