"""GPU 802.11b receive graph (sora_rx11b_*, row f4) against the reference's own graph compiled from its sources and against
the C restatement, event for event: error code, source position, rate, length, FCS word, MPDU bytes."""
import os

import numpy as np
import pytest

from gpu_util import random_capture_11b, same_as_reference_11b

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    return sora_amd


def run_11b(sora, caps, max_frames=16, single_pass=False):
    import torch
    parts, descs, pos = [], [], 0
    for i, c in enumerate(caps):
        descs.append((pos, len(c), i)); parts.append(c); pos += len(c)
    iq = np.concatenate(parts) if parts else np.zeros((0, 2), np.int16)
    rx = sora.Rx11b(max(1, len(caps)), max(28, len(iq)), max_frames_per_capture=max_frames)
    assert rx.set_single_pass(-1) == 2                                   # the default pass plan is automatic
    if single_pass:
        assert rx.set_single_pass(1) == 2 and rx.set_single_pass(-1) == 1
    rx.process_dev(torch.from_numpy(iq).cuda(), descs)
    res = rx.results(); rx.close()
    return res


def oracle_rows(oracle, c):
    rows = oracle.rx11b_capture(c, max_frames=64)
    return [dict(r, sample_index=r["end_sample"]) for r in rows]


@pytest.mark.parametrize("fixture", ["refgraph_11b.npz", "refgraph_11b_cck.npz"])
def test_gpu_11b_equals_recorded_reference_events(sora, oracle, fixture):
    from test_oracle_11b import channel_11b
    z = np.load(os.path.join(GOLD, fixture))
    caps = [channel_11b(z["tx_%d" % f], 100 * f + rep) for f in range(int(z["frames"])) for rep in range(3)]
    got = run_11b(sora, caps)
    k = 0
    for i in range(len(caps)):
        rows = [r for r in got if r["capture_id"] == i]
        assert len(rows) == int(z["ev_count"][i])
        for r in rows:
            assert (r["error_code"], r["end_sample"]) == (z["ev_error"][k], z["ev_position"][k]), i
            if r["error_code"] in (1, 0x80000006):
                assert (r["rate_kbps"], r["length"], r["crc32"] & 0xFFFFFF) == (z["ev_rate"][k], z["ev_length"][k], z["ev_crc"][k] & 0xFFFFFF)
                assert np.array_equal(np.frombuffer(r["mpdu"], np.uint8), z["mpdu_%d" % k])
            k += 1
    assert k == len(z["ev_error"])


def test_gpu_11b_equals_the_reference_graph_under_multipath(sora, oracle):
    """Echoes up to two chips behind the direct path on every capture (a quarter with an echo within 1 dB of it): 500 captures, every
    event of the compiled reference graph against the GPU graph (1 / 2 / 5.5 / 11 Mbps)."""
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present (the captures come from the reference's modulator)")
    rng = np.random.default_rng(1212)
    caps = [random_capture_11b(g, rng, multipath_p=1.0) for _ in range(500)]
    got = run_11b(sora, caps, max_frames=64)
    nev = nok = 0
    for i, c in enumerate(caps):
        ev = g.rx11b(c, max_frames=64)
        ok, why = same_as_reference_11b([r for r in got if r["capture_id"] == i], ev)
        assert ok, "capture %d vs the reference graph: %s" % (i, why)
        nev += len(ev); nok += sum(e["error_code"] == 1 for e in ev)
    assert nev > 800 and nok > 150, (nev, nok)


def test_gpu_11b_equals_reference_graph_and_oracle_on_random_captures(sora, oracle):
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present (the captures come from the reference's modulator)")
    rng = np.random.default_rng(1111)
    caps = [random_capture_11b(g, rng) for _ in range(400)]
    got = run_11b(sora, caps, max_frames=64)
    nev = nok = ncck = 0
    for i, c in enumerate(caps):
        rows = [r for r in got if r["capture_id"] == i]
        ev = g.rx11b(c, max_frames=64)
        ok, why = same_as_reference_11b(rows, ev)
        assert ok, "capture %d vs the reference graph: %s" % (i, why)
        ok, why = same_as_reference_11b(rows, oracle_rows(oracle, c))
        assert ok, "capture %d vs oracle/so_rx11b.c: %s" % (i, why)
        nev += len(ev); nok += sum(e["error_code"] == 1 for e in ev); ncck += sum(e["error_code"] == 1 and e["rate_kbps"] > 2000 for e in ev)
    assert nev > 1000 and nok > 300 and ncck > 100                      # all four rates decode, the CCK ones included


def test_gpu_11b_long_frames_of_every_rate(sora, oracle):
    """Frames long enough for many bulk passes (k_rx11b.hip: up to 64 source calls a pass; a CCK pass decodes ~57 code words), with a DC
    offset and two frames of different rates in one capture: every event as the reference graph and the C restatement report it."""
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present (the captures come from the reference's modulator)")
    rng = np.random.default_rng(1212)
    caps = []
    for rates, ln in (((1000, 11000), 700), ((2000, 5500), 1100), ((11000, 11000), 1500), ((5500, 2000), 1501), ((11000, 1000), 2000)):
        parts = [np.zeros((1400, 2), np.float64)]
        for r in rates:
            s8 = g.tx11b(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), r)
            parts += [s8.astype(np.float64) * 256.0, np.zeros((int(rng.integers(60, 120)) * 28, 2), np.float64)]
        x = np.concatenate(parts)
        x = x[:len(x) // 28 * 28] + rng.uniform(-400, 400, size=(1, 2)) + rng.normal(0, 60, (len(x) // 28 * 28, 2))
        caps.append(np.clip(np.rint(x), -32768, 32767).astype(np.int16))
    got = run_11b(sora, caps, max_frames=16)
    # sora_rx11b_set_single_pass: every capture straight through the CCK-capable kernel instead of two passes -- the same rows, MPDUs included
    assert run_11b(sora, caps, max_frames=16, single_pass=True) == got
    nok = 0
    for i, c in enumerate(caps):
        rows = [r for r in got if r["capture_id"] == i]
        ev = g.rx11b(c, max_frames=16)
        ok, why = same_as_reference_11b(rows, ev)
        assert ok, "capture %d vs the reference graph: %s" % (i, why)
        ok, why = same_as_reference_11b(rows, oracle_rows(oracle, c))
        assert ok, "capture %d vs oracle/so_rx11b.c: %s" % (i, why)
        nok += sum(e["error_code"] == 1 for e in ev)
    assert nok == 10                                                    # both frames of every capture decode


def test_11b_calls_in_flight_keep_their_results_apart(sora, oracle):
    """A handle keeps two calls in flight (own stream and result buffers each): four different batches issued back to back report the last
    batch's rows; issued one by one with a read-back after each, every batch reports its own (checked against the C restatement)."""
    import torch
    rng = np.random.default_rng(2323)
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present (the captures come from the reference's modulator)")
    batches = []
    for b in range(4):
        caps = [random_capture_11b(g, rng) for _ in range(24)]
        descs, pos = [], 0
        for i, c in enumerate(caps):
            descs.append((pos, len(c), i)); pos += len(c)
        batches.append((torch.from_numpy(np.concatenate(caps)).cuda(), descs, caps))
    rx = sora.Rx11b(24, max(int(d.shape[0]) for d, _, _ in batches), max_frames_per_capture=64)

    def check(res, caps):
        for i, c in enumerate(caps):
            ok, why = same_as_reference_11b([r for r in res if r["capture_id"] == i], oracle_rows(oracle, c))
            assert ok, "capture %d: %s" % (i, why)

    for d, descs, _ in batches:
        rx.process_dev(d, descs)
    check(rx.results(), batches[-1][2])
    for d, descs, caps in batches:
        rx.process_dev(d, descs)
        check(rx.results(), caps)
    # tickets: with calls issued back to back, every call in flight is collectable by its own ticket while the next ones run
    # (sora_rx11b_ticket / _wait / _results_of: the contract of sora_rx_*); a ticket whose slot has been reused is refused
    depth = rx.calls_in_flight()
    assert depth == 2
    tickets = []
    for k in range(7):
        d, descs, caps = batches[k % 4]
        tickets.append((rx.process_dev(d, descs), caps))
        if len(tickets) >= depth:
            t, c = tickets[-depth]
            rx.wait(t)
            check(rx.results(ticket=t), c)                               # the OLDER call, read while the newer one is in flight
    check(rx.results(ticket=tickets[-1][0]), tickets[-1][1])
    assert rx.ticket() == tickets[-1][0]
    # delivery without a host wait (sora_rx11b_deliver_async): rows + densely packed MPDUs into page-locked memory behind each call's
    # kernels, two calls in flight; the delivered table is the one results() reports
    key = lambda r: (r["capture_id"], r["end_sample"], r["error_code"], r["rate_kbps"], r["length"], r["crc32"], r["mpdu"])
    bufs = [sora.HostResults(24 * 64, 1 << 20) for _ in range(depth)]
    pend = []
    for k in range(5):
        d, descs, caps = batches[k % 4]
        t = rx.process_dev(d, descs)
        rx.deliver_async(t, bufs[k % depth])
        pend.append((t, bufs[k % depth]))
        if len(pend) >= depth:
            t0, b0 = pend.pop(0)
            rx.wait(t0)
            got = b0.results()
            assert [key(r) for r in got] == [key(r) for r in rx.results(ticket=t0)] and len(got) > 10
            assert int(b0.counts[1]) == sum(len(r["mpdu"]) for r in got)
    for b in bufs:
        b.close()
    with pytest.raises(sora.SoraError):
        rx.results(ticket=tickets[0][0])                                 # long reused
    with pytest.raises(sora.SoraError):
        rx.wait(10 ** 6)                                                 # never issued
    rx.synchronize(); rx.close()


def test_11b_capacity_and_argument_errors(sora):
    import torch
    with pytest.raises(Exception):
        sora.Rx11b(0, 28)
    rx = sora.Rx11b(1, 280)
    with pytest.raises(Exception):
        rx.process_dev(torch.zeros((30, 2), dtype=torch.int16).cuda(), [(0, 30, 0)])      # not whole source bursts
    with pytest.raises(Exception):
        rx.process_dev(torch.zeros((560, 2), dtype=torch.int16).cuda(), [(0, 560, 0)])    # more than max_total_samples
    rx.close()


def test_the_automatic_pass_plan_changes_nothing_but_the_time(sora, oracle):
    """sora_rx11b_set_single_pass(2), the default: after a call whose first pass handed most captures to the CCK instantiation the handle takes
    the single pass by itself (and a two-pass call every 16th, to measure again).  Thirty-six calls of CCK-heavy traffic followed by Barker-only
    traffic: every call's rows equal the fixed two-pass plan's."""
    import torch
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present (the captures come from the reference's modulator)")
    rng = np.random.default_rng(515)

    def batch(rates):
        caps = []
        for r in rates:
            s8 = g.tx11b(rng.integers(0, 256, int(rng.integers(20, 300))).astype(np.uint8).tobytes(), r)
            x = np.concatenate([np.zeros((28 * 40, 2)), s8.astype(np.float64) * 256.0, np.zeros((28 * 60, 2))])
            x = x[:len(x) // 28 * 28] + rng.normal(0, 50, (len(x) // 28 * 28, 2))
            caps.append(np.clip(np.rint(x), -32768, 32767).astype(np.int16))
        return caps
    cck = batch([11000, 5500, 11000, 2000, 11000, 5500, 11000, 11000])
    barker = batch([1000, 2000, 1000, 2000, 1000, 1000, 2000, 1000])
    want_cck = run_11b_fixed(sora, cck); want_barker = run_11b_fixed(sora, barker)
    n = max(sum(len(c) for c in cck), sum(len(c) for c in barker))
    rx = sora.Rx11b(8, n, max_frames_per_capture=16)

    def call(caps):
        descs, pos = [], 0
        for i, c in enumerate(caps):
            descs.append((pos, len(c), i)); pos += len(c)
        t = rx.process_dev(torch.from_numpy(np.concatenate(caps)).cuda(), descs)
        return rx.results(ticket=t)
    for k in range(20):
        assert call(cck) == want_cck, k
    for k in range(16):
        assert call(barker) == want_barker, k
    rx.close()


def run_11b_fixed(sora, caps):
    import torch
    descs, pos = [], 0
    for i, c in enumerate(caps):
        descs.append((pos, len(c), i)); pos += len(c)
    rx = sora.Rx11b(len(caps), pos, max_frames_per_capture=16)
    rx.set_single_pass(0)
    rx.process_dev(torch.from_numpy(np.concatenate(caps)).cuda(), descs)
    res = rx.results(); rx.close()
    return res


@pytest.mark.gpu
def test_11b_completions_are_taken_as_they_happen(sora, oracle):
    """sora_rx11b_wait_any: every delivered call comes back exactly once with its own table; nothing pending -> refused."""
    import torch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refgraph_11b.npz"))
    from test_oracle_11b import channel_11b
    batches = []
    for b in range(3):
        caps = [channel_11b(z["tx_%d" % ((i + b) % 4)], 300 + 17 * b + i) for i in range(3 + b)]
        n = max(len(c) for c in caps) // 28 * 28
        iq = np.concatenate([c[:n] if len(c) >= n else np.concatenate([c, np.zeros((n - len(c), 2), np.int16)]) for c in caps])
        descs = [(i * n, n, i) for i in range(len(caps))]
        batches.append((torch.from_numpy(np.ascontiguousarray(iq)).cuda(), descs))
    rx = sora.Rx11b(8, max(len(b[0]) for b in batches), max_frames_per_capture=8)
    key = lambda r: (r["capture_id"], r["end_sample"], r["error_code"], r["rate_kbps"], r["length"], r["crc32"], r["mpdu"])
    want = []
    for d, descs in batches:
        t = rx.process_dev(d, descs); want.append([key(r) for r in rx.results(ticket=t)])
    rx.synchronize()
    with pytest.raises(sora.SoraError):
        rx.wait_any()
    depth = rx.calls_in_flight()
    free = [sora.HostResults(8 * 8, 1 << 18) for _ in range(depth)]
    held = {}; seen = []; k = 0; first = t + 1

    def submit():
        nonlocal k
        d, descs = batches[k % 3]
        tk = rx.process_dev(d, descs); held[tk] = (free.pop(), k % 3); k += 1
        rx.deliver_async(tk, held[tk][0])

    def take():
        tk = rx.wait_any(); seen.append(tk)
        buf, which = held.pop(tk)
        assert [key(r) for r in buf.results()] == want[which] and len(want[which]) >= 1, tk
        free.append(buf)
    for _ in range(depth):
        submit()
    for _ in range(12):
        take(); submit()
    while held:
        take()
    assert sorted(seen) == list(range(first, first + k))
    for b in free:
        b.close()
    rx.synchronize(); rx.close()
