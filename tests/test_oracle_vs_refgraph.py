"""The C restatement (oracle/so_rx11a.c) against the REFERENCE'S OWN receive graph.

oracle/_ref/libsora_refgraph.so is every brick of CreateDemodGraph11a_40M (kernel/bb/demod11/fb11ademod_config.hpp:168-233)
compiled from the reference sources (oracle/build_ref.sh) and driven by the RxThread loop (fb11a_demod.cpp:29-81).
Where the library exists (build container, and the GPU box: it travels with the snapshot) the oracle is compared with it
live on random captures; everywhere, it is compared with the events that library produced for a committed list of
seeded captures (tests/golden/refgraph_events.npz, written by tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

from gpu_util import multipath_capture, random_capture, same_as_reference_graph, source_position, source_position_44, upsample_40_to_44
from oracle.pyoracle import Oracle, ReferenceGraph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def o():
    return Oracle()


@pytest.fixture(scope="module")
def graph():
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not built (needs the reference tree)")
    return g


def test_reference_graph_decodes_the_fixture(o, graph):
    z = np.load(os.path.join(GOLD, "fsample6_40mhz_i8.npz"))
    iq = z["iq_i8"].astype(np.int16) << 8
    ev = graph.rx11a(iq)
    assert len(ev) == 1 and ev[0]["error_code"] == 1 and ev[0]["rate_kbps"] == 6000 and ev[0]["length"] == 1392
    assert ev[0]["crc32"] == 0x80EF9B11
    assert hashlib.sha256(ev[0]["mpdu"]).hexdigest().startswith("5a13a477")
    ok, why = same_as_reference_graph(o.rx_capture(iq, 40), ev)
    assert ok, why


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_oracle_equals_reference_graph_on_random_captures(o, graph, seed):
    rng = np.random.default_rng(seed)
    nframes = 0
    for i in range(250):
        cap = random_capture(o, rng, 40, multipath_p=0.3)
        ev = graph.rx11a(cap)
        ok, why = same_as_reference_graph(o.rx_capture(cap, 40), ev)
        assert ok, "seed %d capture %d: %s" % (seed, i, why)
        nframes += len(ev)
    assert nframes > 150


def test_oracle_equals_reference_graph_under_multipath(o, graph):
    """frequency-selective channels -- 2-4 taps, echoes 1-8 samples behind the direct path, a deep-null case in
    two of five (an echo within 1 dB of the direct path).  T11aLTS::_channel_estimation divides per carrier by |Y_k|^2 >> 8 with C
    truncation and writes a ZERO coefficient where that divisor is zero (channel_11a.hpp:144-151: `if (e[j] != 0) ... else 0`); the
    restatement does the same (so_rx11a.c), and so does k_scan.  600 captures, every event compared."""
    rng = np.random.default_rng(20261005)
    nev = 0; kinds = {}
    for i in range(600):
        cap = multipath_capture(o, rng, 40)
        ev = graph.rx11a(cap)
        ok, why = same_as_reference_graph(o.rx_capture(cap, 40), ev)
        assert ok, "capture %d: %s" % (i, why)
        nev += len(ev)
        for e in ev:
            kinds[e["error_code"]] = kinds.get(e["error_code"], 0) + 1
    assert nev > 600 and kinds.get(0x1, 0) > 200 and kinds.get(0x80000006, 0) > 30, kinds     # decoded frames AND frames the channel broke


def test_the_two_thread_harness_reports_the_same_events(o, graph):
    """oracle/_ref/libsora_refgraph.so replaces the hop to the decoder thread (TThreadSeparator, stdbrick.hpp:89-248) by the
    reference's same-thread pass-through TNoInline -- the one non-syntactic patch of the compiled oracle.  This test runs
    the graph WITH the real separator, a second (joined) thread executing ViterbiThread (fb11a_demod.cpp:83-86) and the
    RxThread loop on the caller's thread (libsora_refgraph_mt.so), and shows the same events: error code, source position,
    rate, length, FCS, MPDU bytes.  (T11aDataSymbol flushes the decoder branch after the last symbol, PHY_11a.hpp:407-420,
    so the event is raised before the source call returns whatever the thread timing.)"""
    if graph.rx11a_two_threads(np.zeros((280, 2), np.int16)) is None:
        pytest.skip("oracle/_ref/libsora_refgraph_mt.so not built (needs the reference tree)")
    rng = np.random.default_rng(20260929)
    nev = 0; kinds = set()
    for i in range(1000):
        cap = random_capture(o, rng, 40)
        one = graph.rx11a(cap); two = graph.rx11a_two_threads(cap)
        assert one == two, "capture %d: %r vs %r" % (i, [(hex(e["error_code"]), e["sample_index"]) for e in one], [(hex(e["error_code"]), e["sample_index"]) for e in two])
        nev += len(one); kinds |= {e["error_code"] for e in one}
    assert nev > 1000 and {0x1, 0x80000005, 0x80000006} <= kinds


def test_negative_cfo_estimate_is_floored(o, graph):
    """FreqOffsetEstimate divides by a size_t: a negative angle is floored, not truncated (dspalg.hpp:242).  The
    recorded fixture has a non-negative offset, so only the reference graph itself shows this."""
    from gpu_util import make_capture
    seen = 0
    for seed in range(40):
        cap, mp = make_capture(o, 54000, 300, 100 + seed, cfo_hz=-30e3 - 700 * seed, sigma=40)
        rows = o.rx_capture(cap, 40)
        assert rows and rows[0]["cfo_est"] < 0
        ok, why = same_as_reference_graph(rows, graph.rx11a(cap))
        assert ok, why
        seen += rows[0]["error_code"] == 1
    assert seen >= 35


def test_oracle_equals_recorded_reference_events(o):
    """Runs everywhere: the events the compiled reference graph produced for seeded captures (make_golden.py)."""
    z = np.load(os.path.join(GOLD, "refgraph_events.npz"))
    seed = int(z["seed"]); n = int(z["captures"])
    rng = np.random.default_rng(seed)
    k = 0
    ev_cap, ev_err, ev_pos, ev_crc, ev_sha = z["ev_capture"], z["ev_error"], z["ev_position"], z["ev_crc32"], z["ev_mpdu_sha"]
    for i in range(n):
        cap = random_capture(o, rng, 40)
        rows = o.rx_capture(cap, 40)
        assert len(rows) == int((ev_cap == i).sum()), "capture %d: event count" % i
        for r in rows:
            assert ev_cap[k] == i and r["error_code"] == ev_err[k] and source_position(r["end_sample"]) == ev_pos[k], "capture %d" % i
            if r["error_code"] in (0x1, 0x80000006):
                assert r["crc32"] == ev_crc[k]
                assert hashlib.sha256(r["mpdu"]).digest()[:8] == ev_sha[k].tobytes(), "capture %d: MPDU bytes" % i
            k += 1
    assert k == len(ev_cap)


TX_CASES = [(rate, ln, seed) for rate in (6000, 9000, 12000, 18000, 24000, 36000, 48000, 54000)
            for ln, seed in ((1, 0xFF), (37, 0x5D), (400, 0), (1496, 0x7F))]


def _tx_payload(rate, ln):
    return np.random.default_rng(rate + ln).integers(0, 256, ln).astype(np.uint8).tobytes()


def test_oracle_transmitter_equals_reference_mod_graph(o, graph):
    """oracle/so_tx11a.c against CreateModGraph11a_40M + CreatePreamble11a_40M compiled from the reference sources:
    every COMPLEX8 sample, all eight rates, scrambler seeds including the all-zero register."""
    for rate, ln, seed in TX_CASES:
        mp = _tx_payload(rate, ln)
        assert np.array_equal(o.tx(mp, rate, seed=seed), graph.tx11a(mp, rate, seed=seed)), (rate, ln, seed)


def test_oracle_transmitter_equals_recorded_reference_samples(o):
    z = np.load(os.path.join(GOLD, "refgraph_events.npz"))
    for i, (rate, ln, seed) in enumerate(TX_CASES):
        x = o.tx(_tx_payload(rate, ln), rate, seed=seed)
        assert len(x) == z["tx_len"][i] and hashlib.sha256(x.tobytes()).digest()[:8] == z["tx_sha"][i].tobytes(), (rate, ln, seed)


def test_reference_11b_brick_path_runs_on_the_cpu(graph):
    """BASELINE configs[0] (plumbing, no GPU): the reference's own 802.11b BRICK receive path (CreateDemodGraph,
    fb11bdemod_config.hpp:122-172, driven as MAC11b_Receive does) decodes what the reference's own modulation graph
    (fb11bmod_config.hpp:28-50) emits: 1 Mbps DBPSK, 2 Mbps DQPSK, 5.5 and 11 Mbps CCK, long preamble, 44 MHz samples, with noise.
    (The CCK branches only loop back when the build has the reference's integer model: demap_dqpsk_bits, core/inc/soradsp.h:190-198,
    shifts an `unsigned long` by 31 -- one bit with the 32-bit long the code was written for, 0xFF.. with an LP64 long.  oracle/ref_flatten.py
    respells the keyword in the scratch copy; this test is what pins that.)"""
    rng = np.random.default_rng(802)
    for rate in (1000, 2000, 5500, 11000):
        for ln in (14, 300, 1500):
            mp = rng.integers(0, 256, ln).astype(np.uint8).tobytes()
            s8 = graph.tx11b(mp, rate)
            x = np.zeros((2000 + len(s8) + 2800, 2), np.int16)
            x[2000:2000 + len(s8)] = s8.astype(np.int16) << 8                 # `demod11 -c` (modulate11a.cpp:131-190)
            x = np.clip(x + np.rint(rng.normal(0, 150, x.shape)), -32768, 32767).astype(np.int16)[:len(x) // 28 * 28]
            ev = graph.rx11b(x)
            assert len(ev) == 1 and ev[0]["error_code"] == 1 and ev[0]["rate_kbps"] == rate and ev[0]["length"] == ln + 4, (rate, ln, ev)
            assert ev[0]["mpdu"][:ln] == mp


def test_oracle_44mhz_mode_equals_the_reference_44m_graph(o, graph):
    """CreateDemodGraph11a_44M = TDownSample44_40 in front of the same bricks.  The resampled stream is what
    so_down44to40 makes, but that brick has no Reset/Flush, so the samples queued behind it survive the reset that
    follows a frame (the 40 MHz graph flushes TMemSamples' queue): sample_rate_mhz = 44 selects that behaviour."""
    rng = np.random.default_rng(4440)
    nev = 0
    for i in range(250):
        c44 = upsample_40_to_44(random_capture(o, rng, 40))
        ev = graph.rx11a_44(c44)
        x40 = o.down44to40(c44)
        ok, why = same_as_reference_graph(o.rx_capture(x40[:len(x40) // 28 * 28], 44), ev, position=source_position_44)
        assert ok, "capture %d: %s" % (i, why)
        nev += len(ev)
    assert nev > 150


def test_reference_11n_2x2_brick_path_runs_on_the_cpu(graph):
    """SURVEY row f1 / BASELINE configs[3], plumbing: the reference's own 802.11n 2x2 graphs (fb11nmod_config.hpp,
    fb11ndemod_config.hpp:167-264 -- two 64-point FFTs, ZF detection, one Viterbi) compiled from the reference sources
    loop back through a 2x2 channel with cross-talk and noise at MCS 8, 9 and 10, the three the reference's receiver
    accepts (its SIG parser refuses MCS 11-14, which its modulator can emit: PLCP_HEADER_FAIL)."""
    rng = np.random.default_rng(1109)
    for mcs in (8, 9, 10, 12):
        for ln in (60, 1000):
            mp = rng.integers(0, 256, ln).astype(np.uint8).tobytes()
            s0, s1 = graph.tx11n(mp, mcs)
            n = (len(s0) + 1400) // 28 * 28
            a = np.zeros((n, 2)); b = np.zeros((n, 2))
            a[400:400 + len(s0)] = s0 + 0.1 * s1; b[400:400 + len(s0)] = s1 + 0.1 * s0
            a = np.clip(np.rint(a + rng.normal(0, 20, a.shape)), -32768, 32767).astype(np.int16)
            b = np.clip(np.rint(b + rng.normal(0, 20, b.shape)), -32768, 32767).astype(np.int16)
            ev = graph.rx11n(a, b)
            assert len(ev) == 1 and ev[0]["rate_kbps"] == mcs, (mcs, ln, ev)
            if mcs <= 10:
                assert ev[0]["error_code"] == 1 and ev[0]["length"] == ln + 4 and ev[0]["mpdu"][:ln] == mp, (mcs, ln)
            else:
                assert ev[0]["error_code"] == 0x80000005
