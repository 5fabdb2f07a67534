"""GPU capture ingest (sora_hip_ingest) against the oracle chain load_dump -> down44to40 -> downsample2, and the
fixture dump end to end: RX_BLOCK bytes in HBM -> samples -> decoded MPDU."""
import hashlib
import os

import numpy as np
import pytest

from test_oracle_ingest import make_dump

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    return sora_amd


@pytest.mark.parametrize("flags", [1, 3, 5, 7, 9, 11, 13, 15, 0, 4, 8, 12])
def test_ingest_matches_oracle(sora, oracle, flags):
    import torch
    rng = np.random.default_rng(100 + flags)
    if flags & 1:
        raw = rng.integers(0, 256, size=128 * 613 + (72 if flags & 2 else 0), dtype=np.uint8)
        s = oracle.load_dump(raw.tobytes(), raw14=bool(flags & 2))
        d = torch.from_numpy(raw).cuda()
    else:
        s = rng.integers(-32768, 32768, size=(28 * 300 + 11, 2)).astype(np.int16)
        d = torch.from_numpy(s).cuda()
        if flags & 2:
            s = (s.astype(np.int32) << 2).astype(np.int16)
    if flags & 4:
        s = oracle.down44to40(s)
    if flags & 8:
        s = oracle.downsample2(s)
    got = sora.ingest(d, flags).cpu().numpy()
    assert got.shape == s.shape and np.array_equal(got, s)


def test_ingest_large_against_closed_form(sora, oracle):
    """16 MB dump: every output sample against the vectorised closed form; a slice against the oracle."""
    import torch
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, size=128 * 131072, dtype=np.uint8)
    got = sora.ingest(torch.from_numpy(raw).cuda(), 1 | 4 | 8).cpu().numpy()
    x = raw.reshape(-1, 128)[:, 16:].copy().view(np.int16).reshape(-1, 2).astype(np.int64)
    R = np.array([1, 115, 102, 90, 77, 64, 51, 38, 26, 13, 0]); L = np.array([0, 0, 13, 26, 38, 51, 64, 77, 90, 102, 115])
    m = 2 * np.arange(len(got)); p, k = m // 10, m % 10
    interp = (x[11 * p + k] * R[k][:, None] + x[np.minimum(11 * p + k + 1, len(x) - 1)] * L[np.minimum(k + 1, 10)][:, None]) >> 7
    want = np.where((k == 0)[:, None], x[11 * p], interp).astype(np.int16)
    assert np.array_equal(got, want)
    head = oracle.downsample2(oracle.down44to40(oracle.load_dump(raw[:128 * 4096].tobytes())))
    assert np.array_equal(got[:len(head)], head)


def test_fixture_dump_from_hbm_to_mpdu(sora, oracle, golden_dir):
    """fsample-6 re-framed as a Sora dump (14-bit samples, descriptors): ingest on the GPU, then the receive path."""
    import torch
    g = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))
    iq = g["iq_i8"].astype(np.int16) << 8
    raw = make_dump(iq, raw14=True)
    d = sora.ingest(torch.from_numpy(raw).cuda(), sora.INGEST_RXBLOCK | sora.INGEST_RAW14)
    assert np.array_equal(d.cpu().numpy()[:len(iq)], iq)
    n = len(d) // 28 * 28
    rx = sora.Rx(1, n, sample_rate_mhz=40)
    rx.process_dev(d, [(0, n, 0)])
    res = rx.results()
    assert len(res) == 1 and res[0]["error_code"] == sora.E_FRAME_OK and res[0]["length"] == 1392
    assert hashlib.sha256(res[0]["mpdu"]).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    # and decimated on ingest, fed to the 20 MHz path
    d20 = sora.ingest(torch.from_numpy(raw).cuda(), sora.INGEST_RXBLOCK | sora.INGEST_RAW14 | sora.INGEST_DECIMATE2)
    n20 = len(d20) // 14 * 14
    rx20 = sora.Rx(1, n20, sample_rate_mhz=20)
    rx20.process_dev(d20, [(0, n20, 0)])
    r20 = rx20.results()
    assert len(r20) == 1 and r20[0]["mpdu"] == res[0]["mpdu"]


def test_44mhz_graph_on_the_gpu_equals_the_reference_44m_graph(sora, oracle):
    """Row f3 end to end against the reference itself: 44 MHz captures -> sora_hip_ingest(44->40) -> sora_rx with
    sample_rate_mhz = 44, compared event for event with CreateDemodGraph11a_44M compiled from the reference sources."""
    import torch
    from gpu_util import random_capture, same_as_reference_graph, source_position_44, upsample_40_to_44
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    rng = np.random.default_rng(4441)
    caps44 = [upsample_40_to_44(random_capture(oracle, rng, 40)) for _ in range(200)]
    parts, descs, pos = [], [], 0
    for i, c in enumerate(caps44):
        x = sora.ingest(torch.from_numpy(c).cuda(), sora.INGEST_44TO40)
        n = x.shape[0] // 28 * 28                                       # whole bursts of TDownSample44_40
        parts.append(x[:n]); descs.append((pos, n, i)); pos += n
    iq = torch.cat(parts)
    rx = sora.Rx(len(caps44), iq.shape[0], sample_rate_mhz=44, max_frames_per_capture=8)
    rx.process_dev(iq, descs)
    got = rx.results(); rx.close()
    nev = 0
    for i, c in enumerate(caps44):
        ev = g.rx11a_44(c)
        ok, why = same_as_reference_graph([r for r in got if r["capture_id"] == i], ev, position=source_position_44)
        assert ok, "capture %d: %s" % (i, why)
        nev += len(ev)
    assert nev > 120


def test_dump_bytes_in_host_memory_to_mpdus_in_one_call(sora, oracle, golden_dir):
    """sora_rx_process_dump: LoadSoraDumpFile -> graph -> MPDU buffer as one stream-ordered path (brickutil.h:20-58 in front of
    fb11a_demod.cpp:88-120).  The fixture dump from pinned host memory, three calls in flight on the handle's pipelines (their own
    staging buffers), 40 MHz graph and decimate-on-ingest 20 MHz graph; a dump that holds more samples than the handle is refused."""
    import torch
    g = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))
    iq = g["iq_i8"].astype(np.int16) << 8
    raw = make_dump(iq, raw14=True)
    pinned = torch.from_numpy(raw).pin_memory()
    want = "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    for mhz, flags in ((40, sora.INGEST_RXBLOCK | sora.INGEST_RAW14), (20, sora.INGEST_RXBLOCK | sora.INGEST_RAW14 | sora.INGEST_DECIMATE2)):
        n = sora.ingest_count(raw.size, flags)
        burst = 28 if mhz == 40 else 14
        rx = sora.Rx(1, n, sample_rate_mhz=mhz)
        rx.set_depth(3)
        tickets = [rx.process_dump(pinned, flags, [(0, n // burst * burst, 7)]) for _ in range(4)]
        for t in tickets[1:]:                                             # (the first ticket's pipeline has been reused by the fourth call)
            res = rx.results(ticket=t)
            assert len(res) == 1 and res[0]["error_code"] == sora.E_FRAME_OK and res[0]["capture_id"] == 7, (mhz, res)
            assert hashlib.sha256(res[0]["mpdu"]).hexdigest() == want
        t = rx.process_dump(raw, flags, [(0, n // burst * burst, 0)])     # pageable host memory works too (the copy is then synchronous)
        assert hashlib.sha256(rx.results(ticket=t)[0]["mpdu"]).hexdigest() == want
        rx.close()
    small = sora.Rx(1, 1024, sample_rate_mhz=40)
    with pytest.raises(sora.SoraError):
        small.process_dump(pinned, sora.INGEST_RXBLOCK | sora.INGEST_RAW14, [(0, 1008, 0)])
    small.close()
