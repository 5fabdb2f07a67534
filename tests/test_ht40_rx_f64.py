"""The second, independently written receiver for the 40 MHz HT two-stream captures (oracle/ht40_rx_f64.py: float64, from IEEE 802.11n-2009
clause 20) on the CPU: it decodes what the capture generator (oracle/py_ht40.py) sends -- every MCS 8..14, lengths, carrier
offsets, cross-talk, noise, several frames per capture, a spoiled HT-SIG -- without sharing a line with it.  The GPU path is held to this
receiver in tests/test_gpu_ht40.py::test_gpu_psdus_equal_the_independent_float64_receiver."""
import os

import numpy as np

from oracle import ht40_rx_f64 as rxf
from oracle import py_ht40 as gen


def capture(rng, frames, sigma=8.0, cfo=0.0):
    segs, truth = [], []
    for mcs, ln, spoil in frames:
        ps = [gen.add_fcs(rng.integers(0, 256, ln - 4, dtype=np.uint8).tobytes()) for _ in range(2)]
        x, nsym, pre = gen.tx_frame(ps, mcs)
        if spoil:
            x[:, 800:1120] = x[:, 800:1120][:, ::-1] * 1j                        # garbage where HT-SIG should be
        ph = rng.uniform(0, 2 * np.pi, 4)
        H = np.array([[1.0 * np.exp(1j * ph[0]), 0.3 * np.exp(1j * ph[1])], [0.25 * np.exp(1j * ph[2]), 0.9 * np.exp(1j * ph[3])]])
        segs.append(gen.channel(x, H, 0.0, rng, cfo_step=cfo, lead=int(rng.integers(300, 900)))); truth.append((mcs, ln, ps, spoil))
    y = np.concatenate(segs + [np.zeros((2, 800, 2), np.int16)], axis=1).astype(np.float64)
    y += rng.normal(0, sigma, y.shape)
    return np.clip(np.rint(y), -32768, 32767).astype(np.int16), truth


def test_the_module_is_independent_of_the_generator():
    src = open(os.path.join(os.path.dirname(rxf.__file__), "ht40_rx_f64.py")).read()
    code = "\n".join(l for l in src.split("\n") if l.strip().startswith(("import ", "from ")))
    assert "py_ht40" not in code and "sora_amd" not in code and "pyoracle" not in code, code


def test_every_mcs_lengths_offsets_and_noise():
    rng = np.random.default_rng(99)
    n = 0
    for trial in range(42):
        mcs = 8 + trial % 7
        ln = int(rng.choice([5, 31, 64, 200, 333, 700, 1500][: 7 if trial % 9 == 0 else 6]))
        y, truth = capture(rng, [(mcs, ln, False)], sigma=float(rng.choice([3.0, 8.0, 12.0])), cfo=float(rng.choice([0.0, 21.0, -37.0])))
        fr = rxf.receive(y)
        assert len(fr) == 1 and fr[0].sig_ok and (fr[0].mcs, fr[0].length) == (mcs, ln), (trial, mcs, ln, [(f.mcs, f.length, f.sig_ok) for f in fr])
        assert fr[0].fcs_ok == [True, True] and fr[0].psdu == truth[0][2], (trial, mcs, ln)
        n += 1
    assert n == 42


def test_several_frames_per_capture_and_a_spoiled_ht_sig():
    rng = np.random.default_rng(5)
    y, truth = capture(rng, [(9, 120, False), (13, 400, False), (12, 64, True), (14, 300, False)], sigma=6.0)
    fr = rxf.receive(y)
    good = [f for f in fr if f.sig_ok]
    assert [(f.mcs, f.length) for f in good] == [(9, 120), (13, 400), (14, 300)]
    assert [f.psdu for f in good] == [t[2] for t in truth if not t[3]]
    assert all(f.fcs_ok == [True, True] for f in good)
    noise = np.clip(np.rint(rng.normal(0, 40, (2, 6000, 2))), -32768, 32767).astype(np.int16)
    assert rxf.receive(noise) == []
