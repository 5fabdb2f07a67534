"""802.11b: the literal C restatement of the reference's 11b receive graph (oracle/so_rx11b.c) against
the reference's own graph compiled from its sources (oracle/_ref/libsora_refgraph.so, ref_rx11b_capture) -- live where
that library exists, and against what it reported for recorded inputs everywhere (tests/golden/refgraph_11b.npz)."""
import os

import numpy as np
import pytest

from gpu_util import random_capture_11b, same_as_reference_11b
from oracle.pyoracle import Oracle, ReferenceGraph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def o():
    return Oracle()


def channel_11b(s8, seed):
    """The recorded modulator output -> a 44 MHz capture (lead, tail, DC, noise), deterministic in `seed`."""
    rng = np.random.default_rng(seed)
    x = np.zeros((int(rng.integers(0, 2000)) + len(s8) + 2800, 2))
    lead = len(x) - len(s8) - 2800
    x[lead:lead + len(s8)] = s8.astype(np.float64) * 256
    x += rng.uniform(-300, 300, size=(1, 2)) + rng.normal(0, float(rng.choice([0, 100, 600])), x.shape)
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    return x[:len(x) // 28 * 28]


@pytest.mark.parametrize("seed", [21, 22])
def test_oracle_11b_equals_reference_graph_on_random_captures(o, seed):
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not built (needs the reference tree)")
    rng = np.random.default_rng(seed)
    nev = nok = ncck = 0
    for i in range(400):
        c = random_capture_11b(g, rng)
        ev = g.rx11b(c)
        ok, why = same_as_reference_11b(o.rx11b_capture(c), ev)
        assert ok, "seed %d capture %d: %s" % (seed, i, why)
        nev += len(ev); nok += sum(e["error_code"] == 1 for e in ev); ncck += sum(e["error_code"] == 1 and e["rate_kbps"] > 2000 for e in ev)
    assert nev > 1000 and nok > 300 and ncck > 100


def test_oracle_11b_equals_reference_graph_under_multipath(o):
    """Echoes up to two chips behind the direct path (a quarter of them within 1 dB of it) on every capture: Barker despreading, the
    timing recurrence and the CCK correlators on smeared chips.  Every event against the compiled reference graph."""
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not built (needs the reference tree)")
    rng = np.random.default_rng(23)
    nev = nok = 0
    for i in range(300):
        c = random_capture_11b(g, rng, multipath_p=1.0)
        ev = g.rx11b(c)
        ok, why = same_as_reference_11b(o.rx11b_capture(c), ev)
        assert ok, "capture %d: %s" % (i, why)
        nev += len(ev); nok += sum(e["error_code"] == 1 for e in ev)
    assert nev > 500 and nok > 100


@pytest.mark.parametrize("fixture", ["refgraph_11b.npz", "refgraph_11b_cck.npz"])
def test_oracle_11b_equals_recorded_reference_events(o, fixture):
    z = np.load(os.path.join(GOLD, fixture))
    k = 0
    for f in range(int(z["frames"])):
        for rep in range(3):
            rows = o.rx11b_capture(channel_11b(z["tx_%d" % f], 100 * f + rep))
            n = int(z["ev_count"][3 * f + rep])
            assert len(rows) == n
            for r in rows:
                assert (r["error_code"], r["end_sample"]) == (z["ev_error"][k], z["ev_position"][k]), (f, rep)
                if r["error_code"] in (1, 0x80000006):
                    assert (r["rate_kbps"], r["length"], r["crc32"] & 0xFFFFFF) == (z["ev_rate"][k], z["ev_length"][k], z["ev_crc"][k] & 0xFFFFFF)
                    assert np.array_equal(np.frombuffer(r["mpdu"], np.uint8), z["mpdu_%d" % k])
                k += 1
    assert k == len(z["ev_error"]) and (z["ev_error"] == 1).sum() >= 12
