"""GPU transmitter (sora_hip_tx11a, row f2) against the oracle's restatement of the reference modulation graph, bit for
bit, and the loop back through the GPU receiver."""
import numpy as np
import pytest

from oracle.pyoracle import RATES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    return sora_amd


@pytest.mark.parametrize("rate", RATES)
def test_tx_matches_oracle(sora, oracle, rate):
    rng = np.random.default_rng(rate)
    lens = [1, 2, 3, 4, 5, 37, 100, 260, 1496, 2496 if rate >= 12000 else 700]
    seeds = [0xFF, 0x5B, 0x02, 0x01, 0x00, 0x7E, 0x81, 0x33, 0xFF, 0xA5]
    mpdus = [bytes(rng.integers(0, 256, L).astype(np.uint8)) for L in lens]
    out, off = sora.tx11a(mpdus, [rate] * len(lens), seeds)
    got = out.cpu().numpy()
    for f, (mp, sd) in enumerate(zip(mpdus, seeds)):
        want = oracle.tx(mp, rate, sd)
        assert off[f + 1] - off[f] == len(want) == sora.tx11a_samples(len(mp), rate)
        assert np.array_equal(got[off[f]:off[f + 1]], want), (rate, len(mp), hex(sd))


def test_tx_frames_at_sample_offsets_that_are_not_multiples_of_four(sora, oracle):
    """The kernel stores four samples at a time where a frame starts on an 8-byte boundary and two bytes at a time elsewhere: frames behind gaps of 1, 2, 3, 5 .. samples
    come out sample for sample as the oracle's, and the gaps stay untouched."""
    rng = np.random.default_rng(99)
    rates = [RATES[i % 8] for i in range(16)]
    gaps = [(1, 2, 3, 5, 0, 7, 4, 6)[i % 8] for i in range(16)]
    mpdus = [bytes(rng.integers(0, 256, 20 + 41 * i).astype(np.uint8)) for i in range(16)]
    out, off = sora.tx11a(mpdus, rates, [0x5B] * 16, gaps=gaps)
    got = out.cpu().numpy()
    for f in range(16):
        assert not got[off[f]:off[f] + gaps[f]].any()
        assert np.array_equal(got[off[f] + gaps[f]:off[f + 1]], oracle.tx(mpdus[f], rates[f], 0x5B)), (f, rates[f], gaps[f])


def test_tx_matches_the_reference_modulation_graph(sora):
    """Against the reference itself: CreateModGraph11a_40M + CreatePreamble11a_40M compiled from the reference sources
    (oracle/_ref/libsora_refgraph.so, oracle/build_ref.sh)."""
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    rng = np.random.default_rng(4242)
    rates = [RATES[i % 8] for i in range(40)]
    mpdus = [bytes(rng.integers(0, 256, 1 + 61 * i).astype(np.uint8)) for i in range(40)]
    seeds = [int(rng.integers(0, 256)) for _ in range(40)]
    out, off = sora.tx11a(mpdus, rates, seeds)
    got = out.cpu().numpy()
    for f in range(40):
        assert np.array_equal(got[off[f]:off[f + 1]], g.tx11a(mpdus[f], rates[f], seed=seeds[f])), (rates[f], len(mpdus[f]), seeds[f])


def test_tx_mixed_batch_loops_back_through_the_gpu_receiver(sora, oracle):
    import torch
    rng = np.random.default_rng(77)
    rates = [RATES[i % 8] for i in range(24)]
    mpdus = [bytes(rng.integers(0, 256, 60 + 53 * i).astype(np.uint8)) for i in range(24)]
    out, off = sora.tx11a(mpdus, rates, [1 + (i * 7) % 127 for i in range(24)])
    # `demod11 -c` (modulate11a.cpp:131-190): COMPLEX8 << 8 -> COMPLEX16 capture at 40 MHz; 200 samples of silence between frames
    caps, descs, pos = [], [], 0
    iq8 = out.cpu().numpy().astype(np.int16) << 8
    for f in range(24):
        seg = np.concatenate([iq8[off[f]:off[f + 1]], np.zeros((400, 2), np.int16)])
        seg = seg[:len(seg) // 28 * 28]
        caps.append(seg); descs.append((pos, len(seg), f)); pos += len(seg)
    iq = np.concatenate(caps)
    rx = sora.Rx(24, len(iq), sample_rate_mhz=40)
    rx.process_dev(torch.from_numpy(iq).cuda(), descs)
    res = rx.results()
    assert len(res) == 24
    for r in res:
        assert r["error_code"] == sora.E_FRAME_OK and r["mpdu"][:-4] == mpdus[r["capture_id"]] and r["rate_kbps"] == rates[r["capture_id"]]


def test_unsupported_rate_is_refused(sora):
    assert sora.tx11a_samples(100, 11000) == 0
    with pytest.raises(Exception):
        sora.tx11a([b"x" * 10], [11000])
