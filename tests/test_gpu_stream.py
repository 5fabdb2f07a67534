"""Stream continuation (sora_rx_set_stream_mode, include/sora_hip.h): a 40 MHz stream handed to the library in pieces cut at arbitrary
28-sample source bursts must yield exactly the events the reference's graph reports on the UNCUT stream -- the live-source case, where
TRxStream (kernel/brick/inc/rxstream.hpp:34-66) keeps feeding one graph whose DC estimate (dc.hpp:92-166) and carrier-sense state
(cca.hpp:126-158) carry over from read to read."""
import numpy as np
import pytest

from gpu_util import random_capture, same_as_reference_graph, source_position

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    sora_amd.load()
    assert sora_amd.device_count() > 0
    return sora_amd


def _stream(oracle, rng, ncaps):
    """several random captures back to back (frames of all rates and lengths, gaps, DC steps, gain steps, carrier offsets, noise, bare noise)"""
    x = np.concatenate([random_capture(oracle, rng, 40) for _ in range(ncaps)])
    return np.ascontiguousarray(x[:len(x) // 28 * 28])


def _reference_events(oracle, stream):
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if g.available():
        return g.rx11a(stream, max_frames=256), "reference"
    ev = []
    for r in oracle.rx_capture(stream, 40, max_frames=256):                      # the restatement (pinned to the compiled graph where that exists)
        e = dict(r); e["sample_index"] = source_position(r["end_sample"]); ev.append(e)
    return ev, "port"


def _run_in_pieces(sora, torch, streams, cuts, max_frames=32):
    """streams: list of int16 [n, 2]; cuts[k]: increasing stream positions (multiples of 28) at which stream k's pieces end.  Every call
    carries one capture per stream: what the call before left unconsumed plus the next piece."""
    ns = len(streams)
    rx = sora.Rx(ns, sum(len(s) for s in streams) + 64 * ns, sample_rate_mhz=40, max_frames_per_capture=max_frames)
    assert rx.set_stream_mode(1) == 0 and rx.set_stream_mode(-1) == 1
    base = [0] * ns
    events = [[] for _ in range(ns)]
    calls = 0
    for i in range(len(cuts[0])):
        segs, descs, off = [], [], 0
        for k in range(ns):
            seg = streams[k][base[k]:cuts[k][i]]
            pad = (-len(seg)) % 4
            segs.append(seg); descs.append((off, len(seg), k)); off += len(seg) + pad
            if pad:
                segs.append(np.zeros((pad, 2), np.int16))
        iq = np.ascontiguousarray(np.concatenate(segs)) if off else np.zeros((4, 2), np.int16)
        t = rx.process_dev(torch.from_numpy(iq).cuda(), descs)
        rows = rx.results(ticket=t)
        used = rx.stream_consumed(t, ns)
        calls += 1
        for r in rows:
            k = r["capture_id"]
            r = dict(r); r["start_sample"] += base[k] // 2; r["end_sample"] += base[k] // 2
            events[k].append(r)
        for k in range(ns):
            assert used[k] % 28 == 0 and used[k] <= descs[k][1], (used[k], descs[k])
            for r in rows:                                                       # a reported frame lies in front of the resume point
                if r["capture_id"] == k:
                    assert source_position(r["end_sample"]) <= used[k], (r["end_sample"], used[k])
            base[k] += int(used[k])
    rx.close()
    return events, calls


def test_pieces_cut_at_arbitrary_source_bursts_report_what_the_uncut_stream_reports(sora, oracle):
    import torch
    rng = np.random.default_rng(20261101)
    total_events = 0; kinds = set()
    for trial in range(12):
        ns = 1 + trial % 3
        streams = [_stream(oracle, rng, int(rng.integers(3, 9))) for _ in range(ns)]
        want = [_reference_events(oracle, s) for s in streams]
        npieces = int(rng.integers(2, 14))
        cuts = []
        for s in streams:
            inner = sorted(int(c) * 28 for c in rng.integers(1, len(s) // 28, size=npieces - 1))
            cuts.append(inner + [len(s)])
        got, calls = _run_in_pieces(sora, torch, streams, cuts)
        for k in range(ns):
            ok, why = same_as_reference_graph(got[k], want[k][0])
            assert ok, "trial %d stream %d (%d pieces, against the %s): %s" % (trial, k, npieces, want[k][1], why)
            total_events += len(want[k][0]); kinds.update(e["error_code"] for e in want[k][0])
    assert total_events > 60 and {0x1, 0x80000005} <= kinds, (total_events, kinds)


def test_a_frame_straddling_many_short_pieces_and_a_mode_switch(sora, oracle):
    """Pieces much shorter than a frame (a 1500-byte 6 Mbps frame lasts 80 k samples): nothing is consumed while the frame is under way, the
    host's tail grows, and the frame is reported once, when a piece finally holds its end.  sora_rx_reset starts the stream afresh."""
    import torch
    from gpu_util import make_capture
    rng = np.random.default_rng(7)
    cap = make_capture(oracle, 6000, 1500, seed=3, rate_mhz=40, sigma=40, lead=56 * 9, tail=560)[0]
    noise = np.rint(rng.normal(0, 30, (28 * 40, 2))).astype(np.int16)
    stream = np.ascontiguousarray(np.concatenate([noise, cap, noise, cap]))
    stream = stream[:len(stream) // 28 * 28]
    want, kind = _reference_events(oracle, stream)
    assert len(want) == 2 and all(e["error_code"] == 1 for e in want)
    cuts = [list(range(28 * 200, len(stream), 28 * 200)) + [len(stream)]]
    got, calls = _run_in_pieces(sora, torch, [stream], cuts, max_frames=4)
    ok, why = same_as_reference_graph(got[0], want)
    assert ok, why
    assert calls >= 25
    # without stream mode the same pieces are independent captures: the frames, cut by every piece, are lost
    rx = sora.Rx(1, len(stream), sample_rate_mhz=40, max_frames_per_capture=4)
    lost = 0
    for a, b in zip([0] + cuts[0][:-1], cuts[0]):
        rx.process_dev(torch.from_numpy(np.ascontiguousarray(stream[a:b])).cuda(), [(0, b - a, 0)])
        lost += len(rx.results())
    assert lost == 0
    rx.close()
