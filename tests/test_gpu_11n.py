"""802.11n 2x2 receive graph on the GPU (sora_rx11n_*, k_rx11n.hip) against the recorded events of the reference graph
(tests/golden/refgraph_11n.npz) and against the CPU oracle on random two-chain captures."""
import numpy as np
import pytest

from gpu_util import capture_11n, same_events_11n
from oracle.pyoracle import Oracle
from test_oracle_11n_graph import golden_captures

pytestmark = pytest.mark.gpu


def run_batch(caps, max_frames=8, trellis=None):
    import torch
    import sora_amd
    n = sum(len(a) for a, _ in caps)
    iq0 = np.concatenate([a for a, _ in caps]); iq1 = np.concatenate([b for _, b in caps])
    descs = []; off = 0
    for i, (a, _) in enumerate(caps):
        descs.append((off, len(a), i)); off += len(a)
    rx = sora_amd.Rx11n(len(caps), n, max_frames_per_capture=max_frames)
    if trellis is not None:
        rx.set_trellis(trellis)
    rx.process_dev(torch.from_numpy(iq0).cuda(), torch.from_numpy(iq1).cuda(), descs)
    per = [[] for _ in caps]
    for r in rx.results():
        per[r["capture_id"]].append(r)
    return per


def test_gpu_equals_recorded_reference_events():
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    gold = list(golden_captures())
    got = run_batch([(a, b) for a, b, _, _, _ in gold])
    for i, (a, b, want, _, z) in enumerate(gold):
        key = lambda e: (e["error_code"], e["rate_kbps"], e["length"], e["crc32"]) if e["error_code"] != 0x80000005 else (e["error_code"],)
        assert [key(e) for e in got[i]] == [key(e) for e in want], (i, [key(e) for e in got[i]], [key(e) for e in want])
        assert [e["end_sample"] for e in got[i]] == [e["sample_index"] for e in want], i
        for e in got[i]:
            if e["error_code"] == 1:
                assert e["mpdu"][:-4] == z["mpdu%d" % {8: 0, 9: 1, 10: 2}[e["rate_kbps"]]].tobytes()


def test_gpu_equals_oracle_on_random_captures():
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    o = Oracle()
    z = np.load(__import__("test_oracle_11n_graph").GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(77)
    caps = []
    for t in range(160):
        fr = [frames[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 4)))]
        caps.append(capture_11n(rng, fr, sigma=float(rng.choice([5, 20, 60, 200, 600])), cut=float(rng.uniform(0.2, 1.0)) if t % 3 == 2 else None))
    got = run_batch(caps)
    nev = 0
    for i, (a, b) in enumerate(caps):
        want = o.rx11n_capture(a, b)
        ok, why = same_events_11n(got[i], want)
        assert ok, (i, why, [(hex(e["error_code"]), e["rate_kbps"], e["length"]) for e in got[i]], [(hex(e["error_code"]), e["rate_kbps"], e["length"]) for e in want])
        assert [e["end_sample"] for e in got[i]] == [e["end_sample"] for e in want], i
        nev += len(want)
    assert nev > 150


def test_gpu_edge_cases():
    """Empty batch, silent and tiny captures, more frames than rows, captures at odd offsets inside one buffer."""
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    o = Oracle()
    z = np.load(__import__("test_oracle_11n_graph").GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rx = sora_amd.Rx11n(4, 1000)
    rx.process_dev(torch.zeros((28, 2), dtype=torch.int16).cuda(), torch.zeros((28, 2), dtype=torch.int16).cuda(), [])
    assert rx.results() == []
    rng = np.random.default_rng(5)
    a3, b3 = capture_11n(rng, [frames[0], frames[1], frames[0]], sigma=10.0)
    silent = np.zeros((28 * 40, 2), np.int16); tiny = rng.integers(-50, 51, size=(28, 2)).astype(np.int16)
    caps = [(silent, silent), (tiny, tiny), (a3, b3), (a3[:28 * 3], b3[:28 * 3])]
    got = run_batch(caps)
    for i, (a, b) in enumerate(caps):
        ok, why = same_events_11n(got[i], o.rx11n_capture(a, b), position="end_sample")
        assert ok, (i, why)
    assert got[0] == [] and got[1] == [] and len(got[2]) == 3
    # fewer rows than frames: the first rows are reported
    for mf in (1, 2):
        one = run_batch([(a3, b3)], max_frames=mf)
        ok, why = same_events_11n(one[0], o.rx11n_capture(a3, b3)[:mf], position="end_sample")
        assert ok, why
        # ... with their own MPDU bytes (no later frame's payload in the last row) and the last row flagged
        assert [r["mpdu"] for r in one[0]] == [r["mpdu"] for r in got[2][:mf]]
        assert [r["flags"] for r in one[0]] == [0] * (mf - 1) + [sora_amd.ROW_TRUNCATED]
    assert all(r["flags"] == 0 for r in got[2])
    # the same captures addressed at odd offsets of one buffer (descriptors need not be aligned)
    pad = 13
    iq0 = np.concatenate([np.zeros((pad, 2), np.int16), a3, np.zeros((7, 2), np.int16), a3[:2800]])
    iq1 = np.concatenate([np.zeros((pad, 2), np.int16), b3, np.zeros((7, 2), np.int16), b3[:2800]])
    rx = sora_amd.Rx11n(2, len(iq0))
    rx.process_dev(torch.from_numpy(iq0).cuda(), torch.from_numpy(iq1).cuda(), [(pad, len(a3), 7), (pad + len(a3) + 7, 2800, 9)])
    res = rx.results()
    for cid, (a, b) in ((7, (a3, b3)), (9, (a3[:2800], b3[:2800]))):
        ok, why = same_events_11n([r for r in res if r["capture_id"] == cid], o.rx11n_capture(a, b), position="end_sample")
        assert ok, (cid, why)
    with pytest.raises(sora_amd.SoraError):
        rx.process_dev(torch.from_numpy(iq0).cuda(), torch.from_numpy(iq1).cuda(), [(0, 27, 0)])       # not a whole source burst


def test_gpu_equals_the_live_reference_graph():
    """The GPU path against the reference ITSELF, directly: CreateDemodGraph11n compiled from the reference sources
    (oracle/_ref/libsora_refgraph.so, ref_rx11n_capture = the RxThread loop of fb11n_demod.cpp:30-85) on random two-chain
    captures -- frames of the compiled reference modulator at MCS 8, 9, 10 (decoded) and 11-14 (refused by the SIG parser),
    random lengths, 1-3 frames per capture, gain / phase / cross-talk / CFO / noise up to failure, frames cut by the end of
    the capture.  Error code, 40 MHz source position, MCS, length, FCS and MPDU bytes of every event."""
    import sora_amd
    from oracle.pyoracle import ReferenceGraph
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    rng = np.random.default_rng(20260926)
    frames = []
    for k in range(40):
        mcs = [8, 9, 10, 8, 9, 10, 10, 11, 12, 13, 14][k % 11]
        ln = int(rng.integers(1, 1497)) if k % 3 else int(rng.integers(1, 80))
        frames.append(g.tx11n(rng.integers(0, 256, ln).astype(np.uint8).tobytes(), mcs))
    caps = []
    for t in range(320):
        fr = [frames[int(i)] for i in rng.integers(0, len(frames), size=int(rng.integers(1, 4)))]
        caps.append(capture_11n(rng, fr, sigma=float(rng.choice([3, 20, 60, 200, 600, 1500])), cut=float(rng.uniform(0.05, 1.0)) if t % 3 == 2 else None))
    got = run_batch(caps)
    nev = 0; kinds = {}
    for i, (a, b) in enumerate(caps):
        want = g.rx11n(a, b)
        ok, why = same_events_11n(got[i], want, position="sample_index")
        assert ok, (i, why, [(hex(e["error_code"]), e["rate_kbps"], e["length"], e["end_sample"]) for e in got[i]],
                    [(hex(e["error_code"]), e["rate_kbps"], e["length"], e["sample_index"]) for e in want])
        nev += len(want)
        for e in want:
            kinds[e["error_code"]] = kinds.get(e["error_code"], 0) + 1
    assert nev > 320 and kinds.get(1, 0) > 100 and kinds.get(0x80000005, 0) > 50, kinds


def test_gpu_11n_equals_the_oracle_under_a_2x2_multipath_channel():
    """500 two-chain captures through a 2x2 matrix of frequency-selective channels (every TX -> RX path its own echoes): TMimoChannelEst's
    float per-carrier inverse on unequal and ill-conditioned carriers (channel_11n.hpp:423-433), the pilot tracker behind it.  Rows and
    MPDUs equal the oracle's (which tests/test_oracle_11n_graph.py pins to the compiled reference graph on the same kind of captures)."""
    import sora_amd
    from oracle.pyoracle import Oracle
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    o = Oracle()
    z = np.load(__import__("test_oracle_11n_graph").GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(79)
    caps = []
    for t in range(500):
        fr = [frames[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 3)))]
        caps.append(capture_11n(rng, fr, sigma=float(rng.choice([5, 20, 60, 200])), multipath_p=1.0))
    got = run_batch(caps)
    nev = nok = 0
    for i, (a, b) in enumerate(caps):
        want = o.rx11n_capture(a, b)
        ok, why = same_events_11n(got[i], want)
        assert ok, (i, why)
        assert [e["end_sample"] for e in got[i]] == [e["end_sample"] for e in want], i
        nev += len(want); nok += sum(e["error_code"] == 1 for e in want)
    assert nev > 500 and nok > 150, (nev, nok)


def test_both_trellis_kernels_decode_the_11n_graph_alike():
    """sora_rx11n_set_trellis: k_viterbi11n (64 lanes per frame pair) and k_viterbi16_11n (16 lanes per pair, eight frames per wave), both
    with the 192 / 36 window of T11aViterbi<5000*8, 312, 192, 36>, report the same rows and MPDUs -- frames cut by the end of the capture
    (zero-padded decoder input) included -- and those are the oracle's."""
    import sora_amd
    from oracle.pyoracle import Oracle
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    z = np.load(__import__("test_oracle_11n_graph").GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(78)
    caps = []
    for t in range(300):
        fr = [frames[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 5)))]
        caps.append(capture_11n(rng, fr, sigma=float(rng.choice([5, 60, 600, 1500])), cut=float(rng.uniform(0.05, 1.0)) if t % 2 else None))
    a = run_batch(caps, trellis=64)
    b = run_batch(caps, trellis=16)
    key = lambda e: (e["error_code"], e["end_sample"], e["rate_kbps"], e["length"], e["crc32"], e["mpdu"])
    assert sum(len(x) for x in a) > 300
    o = Oracle()
    for i in range(len(caps)):
        assert [key(e) for e in a[i]] == [key(e) for e in b[i]], i
        if i % 10 == 0:
            assert [key(e) for e in b[i]] == [key(e) for e in o.rx11n_capture(*caps[i])], i


def test_window_parallel_trellis_for_the_11n_decoder():
    """Round 6: T11aViterbi<5000*8, 312, 192, 36> decoded window-parallel with the units' proof (k_viterbi16w_11n + k_win_redo_11n, the 802.11a
    machinery instantiated for 192-bit windows and one byte per soft value).  Same rows and MPDUs as k_viterbi11n and the oracle on random captures (cut frames
    included); frames of up to 1500 bytes from the reference's own modulator against the compiled reference graph; a data field replaced by noise behind intact
    headers fails the proof and is decoded again by the serial kernel; it is the automatic choice for a handle with few frames in flight."""
    import torch
    import sora_amd
    from oracle.pyoracle import Oracle, ReferenceGraph
    z = np.load(__import__("test_oracle_11n_graph").GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(79)
    caps = []
    for t in range(120):
        fr = [frames[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 5)))]
        caps.append(capture_11n(rng, fr, sigma=float(rng.choice([5, 60, 600, 1500])), cut=float(rng.uniform(0.05, 1.0)) if t % 2 else None))
    a = run_batch(caps, trellis=64)
    b = run_batch(caps, trellis=sora_amd.TRELLIS_WINDOWED)
    key = lambda e: (e["error_code"], e["end_sample"], e["rate_kbps"], e["length"], e["crc32"], e["mpdu"])
    assert sum(len(x) for x in a) > 120
    for i in range(len(caps)):
        assert [key(e) for e in a[i]] == [key(e) for e in b[i]], i
    # long frames, the compiled reference as the judge; one capture at a time = the lone-capture layout (units of one window), and a batch of them
    g = ReferenceGraph()
    if g.available():
        big = []
        for k, (ln, mcs) in enumerate(((1000, 10), (1490, 9), (1496, 8), (700, 10), (64, 8))):
            s0, s1 = g.tx11n(np.random.default_rng(900 + k).integers(0, 256, ln).astype(np.uint8).tobytes(), mcs)
            n = (len(s0) + 800 + 1200 + 27) // 28 * 28
            c = np.zeros((2, n, 2), np.float64)
            c[0, 800:800 + len(s0)] = s0 + 0.1 * s1; c[1, 800:800 + len(s0)] = s1 + 0.1 * s0
            c = np.clip(np.rint(c + np.random.default_rng(k).normal(0, 20, c.shape)), -32768, 32767).astype(np.int16)
            big.append((c[0], c[1]))
        noisy = [(x.copy(), y.copy()) for x, y in big[:2]]
        for x, y in noisy:                                                   # the data field replaced by noise: the headers stay, the units' proof fails
            x[2400:len(x) - 1400] = np.rint(rng.normal(0, 2500, (len(x) - 3800, 2))); y[2400:len(y) - 1400] = np.rint(rng.normal(0, 2500, (len(y) - 3800, 2)))
        want = [key_ref(g.rx11n(x, y)) for x, y in big + noisy]
        for group in ([[c] for c in big + noisy] + [big + noisy]):
            rx = sora_amd.Rx11n(len(group), sum(len(x) for x, _ in group) + 4096, max_frames_per_capture=4)
            assert rx.trellis() == sora_amd.TRELLIS_WINDOWED                 # the automatic choice for so few frames
            iq0 = np.concatenate([x for x, _ in group]); iq1 = np.concatenate([y for _, y in group])
            descs = []; off = 0
            for i, (x, _) in enumerate(group):
                descs.append((off, len(x), i)); off += len(x)
            rx.process_dev(torch.from_numpy(iq0).cuda(), torch.from_numpy(iq1).cuda(), descs)
            per = [[] for _ in group]
            for r in rx.results():
                per[r["capture_id"]].append(r)
            st = rx.window_stats(); rx.close()
            assert st["units"] > 0
            for i, c in enumerate(group):
                j = next(k for k, d in enumerate(big + noisy) if d[0] is c[0])
                assert [(e["error_code"], e["rate_kbps"], e["length"], e["crc32"], e["mpdu"]) for e in per[i]] == want[j], (len(group), i)
            if len(group) > 1:
                assert st["boundaries_failed"] >= 1 and st["frames_decoded_again"] >= 1, st


def key_ref(events):
    return [(e["error_code"], e["rate_kbps"], e["length"], e["crc32"], e["mpdu"]) for e in events]


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_11n_calls_in_flight_are_collectable_by_ticket(depth):
    """sora_rx11n_set_depth: consecutive calls on different batches rotate over the handle's pipelines; each call's rows and MPDUs are read back by its
    ticket while later calls are in flight, and equal the oracle's; a ticket whose pipeline has been reused is refused."""
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    o = Oracle()
    z = np.load(__import__("test_oracle_11n_graph").GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(500 + depth)
    batches = []
    for b in range(depth + 2):
        caps = [capture_11n(rng, [frames[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 3)))], sigma=float(rng.choice([5, 60, 400]))) for _ in range(12)]
        iq0 = np.concatenate([a for a, _ in caps]); iq1 = np.concatenate([c for _, c in caps])
        descs = []; off = 0
        for i, (a, _) in enumerate(caps):
            descs.append((off, len(a), i)); off += len(a)
        batches.append((caps, torch.from_numpy(iq0).cuda(), torch.from_numpy(iq1).cuda(), descs))
    n_max = max(len(b[1]) for b in batches)
    rx = sora_amd.Rx11n(12, n_max, max_frames_per_capture=8)
    assert rx.set_depth(depth) == 1
    tickets = [rx.process_dev(b[1], b[2], b[3]) for b in batches]
    assert tickets == list(range(1, len(batches) + 1))
    for t, b in list(zip(tickets, batches))[-depth:]:                         # the calls still held by a pipeline
        per = [[] for _ in b[0]]
        for r in rx.results(ticket=t):
            per[r["capture_id"]].append(r)
        for i, (a, c) in enumerate(b[0]):
            ok, why = same_events_11n(per[i], o.rx11n_capture(a, c))
            assert ok, (t, i, why)
    with pytest.raises(Exception):
        rx.results(ticket=tickets[0])                                           # depth + 2 calls were made: the first call's pipeline has been reused
    # delivery without a host wait (sora_rx11n_deliver_async): rows + densely packed MPDUs behind each call's kernels, `depth` calls in flight
    key = lambda r: (r["capture_id"], r["end_sample"], r["error_code"], r["rate_kbps"], r["length"], r["crc32"], r["mpdu"])
    bufs = [sora_amd.HostResults(12 * 8, 1 << 18) for _ in range(depth)]
    pend = []
    for k in range(depth + 3):
        b = batches[k % len(batches)]
        t = rx.process_dev(b[1], b[2], b[3])
        rx.deliver_async(t, bufs[k % depth]); pend.append((t, bufs[k % depth]))
        if len(pend) >= depth:
            t0, b0 = pend.pop(0)
            rx.wait(t0)
            got = b0.results()
            assert [key(r) for r in got] == [key(r) for r in rx.results(ticket=t0)] and len(got) >= 12
    for bb in bufs:
        bb.close()
    rx.close()


def test_11n_completions_are_taken_as_they_happen():
    """sora_rx11n_wait_any: every delivered call comes back exactly once, with its own table; the released pipeline is the one reused."""
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    z = np.load(__import__("test_oracle_11n_graph").GOLD)
    frames = [(z["tx%d_0" % i], z["tx%d_1" % i]) for i in range(4)]
    rng = np.random.default_rng(777)
    batches = []
    for b in range(3):
        caps = [capture_11n(rng, [frames[int(i)] for i in rng.integers(0, 4, size=1 + b)], sigma=float(rng.choice([5, 60]))) for _ in range(4 + 4 * b)]
        iq0 = np.concatenate([a for a, _ in caps]); iq1 = np.concatenate([c for _, c in caps])
        descs = []; off = 0
        for i, (a, _) in enumerate(caps):
            descs.append((off, len(a), i)); off += len(a)
        batches.append((torch.from_numpy(iq0).cuda(), torch.from_numpy(iq1).cuda(), descs))
    rx = sora_amd.Rx11n(12, max(len(b[0]) for b in batches), max_frames_per_capture=8)
    depth = 4
    rx.set_depth(depth)
    key = lambda r: (r["capture_id"], r["end_sample"], r["error_code"], r["rate_kbps"], r["length"], r["crc32"], r["mpdu"])
    want = []
    for b in batches:                                                          # the tables to expect, one call at a time
        t = rx.process_dev(*b); want.append([key(r) for r in rx.results(ticket=t)])
    rx.synchronize()
    with pytest.raises(Exception):
        rx.wait_any()                                                         # no delivery is pending
    free = [sora_amd.HostResults(12 * 8, 1 << 18) for _ in range(depth)]
    held = {}; seen = []; k = 0

    def submit():
        nonlocal k
        t = rx.process_dev(*batches[k % 3]); held[t] = (free.pop(), k % 3); k += 1
        rx.deliver_async(t, held[t][0])

    def take():
        t = rx.wait_any(); seen.append(t)
        buf, which = held.pop(t)
        assert [key(r) for r in buf.results()] == want[which], t
        free.append(buf)
    first = t + 1
    for _ in range(depth):
        submit()
    for _ in range(24):
        take(); submit()
    while held:
        take()
    assert sorted(seen) == list(range(first, first + k))
    for bb in free:
        bb.close()
    rx.close()
