"""802.11n 2x2 receive graph, whole path on the CPU: oracle/so_rx11n.c against the events the reference's own
CreateDemodGraph11n produced for the captures of tests/golden/refgraph_11n.npz (built from recorded waveforms of the reference
modulator), and live against oracle/_ref where the reference tree is present."""
import os

import numpy as np
import pytest

from gpu_util import capture_11n, same_events_11n
from oracle.pyoracle import Oracle, ReferenceGraph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refgraph_11n.npz")


@pytest.fixture(scope="module")
def o():
    return Oracle()


def golden_captures():
    z = np.load(GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(1145); k = 0
    for fr, cut, sg, cnt, det in zip(z["plan_frames"], z["plan_cut"], z["plan_sigma"], z["ev_count"], z["cca_detect"]):
        a, b = capture_11n(rng, [frames[int(i)] for i in str(fr).split(",")], sigma=float(sg), cut=None if cut < 0 else float(cut))
        want = [dict(sample_index=int(z["ev_pos"][j]), error_code=int(z["ev_err"][j]), rate_kbps=int(z["ev_mcs"][j]), length=int(z["ev_length"][j]), crc32=int(z["ev_crc"][j])) for j in range(k, k + cnt)]
        k += cnt
        yield a, b, want, [int(x) for x in str(det).split(",") if x], z


def test_oracle_equals_recorded_reference_events(o):
    nev = nok = 0
    for a, b, want, det, z in golden_captures():
        got = o.rx11n_capture(a, b)
        assert [(e["error_code"], e["rate_kbps"], e["length"], e["crc32"]) if e["error_code"] != 0x80000005 else (e["error_code"],) for e in got] == \
               [(e["error_code"], e["rate_kbps"], e["length"], e["crc32"]) if e["error_code"] != 0x80000005 else (e["error_code"],) for e in want]
        assert [e["end_sample"] for e in got] == [e["sample_index"] for e in want]
        for e in got:
            if e["error_code"] == 1:
                i = {8: 0, 9: 1, 10: 2}[e["rate_kbps"]]
                assert e["mpdu"][:-4] == z["mpdu%d" % i].tobytes(); nok += 1
        n4 = len(a) // 2 // 4 * 4
        assert o.cca11n(a[::2][:n4], b[::2][:n4]) == det
        nev += len(got)
    assert nev >= 12 and nok >= 6                                          # decoded frames, header failures (MCS 12) and cut frames all occur


def test_oracle_equals_reference_graph_live(o):
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not built (needs the reference tree)")
    rng = np.random.default_rng(2)
    txc = {}
    nev = 0
    for t in range(40):
        frames = []
        for _ in range(int(rng.integers(1, 4))):
            mcs = int(rng.choice([8, 9, 10, 10, 9, 8, 12])); ln = int(rng.integers(1, 300 if t % 5 else 1400))
            frames.append(g.tx11n(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), mcs))
        a, b = capture_11n(rng, frames, sigma=float(rng.choice([5, 20, 60, 200])), cut=float(rng.uniform(0.3, 1.0)) if t % 3 == 2 else None)
        want = g.rx11n(a, b); got = o.rx11n_capture(a, b)
        ok, why = same_events_11n(got, want, position="sample_index")
        assert ok, (t, why)
        nev += len(want)
        n4 = len(a) // 2 // 4 * 4; skip = int(rng.integers(0, 500))
        assert o.cca11n(a[::2][:n4], b[::2][:n4], skip) == g.cca11n(a[::2][:n4], b[::2][:n4], skip)
    assert nev > 40


def test_oracle_equals_reference_graph_under_multipath(o):
    """A 2x2 matrix of frequency-selective channels (every TX -> RX path its own 1-3 echoes, 1-8 samples @20 MHz behind its direct tap):
    the per-carrier 2x2 inverse of TMimoChannelEst (channel_11n.hpp:423-433: float, determinant / 65536 as the divisor) on unequal,
    partly ill-conditioned carriers.  Every event of the compiled reference graph against the restatement."""
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not built (needs the reference tree)")
    rng = np.random.default_rng(31)
    nev = nok = 0
    for t in range(60):
        frames = []
        for _ in range(int(rng.integers(1, 3))):
            mcs = int(rng.choice([8, 9, 10, 10, 9, 8])); ln = int(rng.integers(1, 400))
            frames.append(g.tx11n(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), mcs))
        a, b = capture_11n(rng, frames, sigma=float(rng.choice([5, 20, 60])), multipath_p=1.0)
        want = g.rx11n(a, b); got = o.rx11n_capture(a, b)
        ok, why = same_events_11n(got, want, position="sample_index")
        assert ok, (t, why)
        nev += len(want); nok += sum(e["error_code"] == 1 for e in want)
    assert nev > 60 and nok > 20, (nev, nok)
