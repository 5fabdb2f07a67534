"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Integer/byte work throughout => bit-exact (tolerance 0 LSB on every intermediate that is compared)."""
import hashlib
import os

import numpy as np
import pytest

from gpu_util import awgn, batch, make_capture, oracle_results, pad_capture, same_results
from oracle.pyoracle import CR_12, CR_23, CR_34, E_FRAME_OK, RATES, rate_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def sora():
    import sora_amd
    sora_amd.load()
    assert sora_amd.device_count() > 0
    return sora_amd


def run_rx(sora, torch, caps, rate_mhz, max_frames=4, fused=None, trellis=None):
    iq, descs = batch(caps)
    rx = sora.Rx(max_captures=max(1, len(caps)), max_total_samples=max(64, len(iq)), sample_rate_mhz=rate_mhz,
                 max_frames_per_capture=max_frames)
    if fused is not None:
        rx.set_fused(fused)
    if trellis is not None:
        rx.set_trellis(trellis)
    d = torch.from_numpy(iq).cuda()
    rx.process_dev(d, descs)
    res = rx.results()
    rx.close()
    return res


# ------------------------------------------------------------------ stage kernels
def test_fft64_bit_exact(sora, torch_cuda, oracle):
    torch = torch_cuda
    rng = np.random.default_rng(1)
    n = 4099
    amps = np.array([32767, 20000, 8000, 500, 30])[np.arange(n) % 5]
    x = (rng.integers(-32768, 32768, size=(n, 64, 2)) % (2 * amps[:, None, None] + 1) - amps[:, None, None]).astype(np.int16)
    x[7, 3] = (-32768, 32767); x[8] = 0; x[9] = 32767; x[10] = -32768
    got = sora.fft64(torch.from_numpy(x).cuda()).cpu().numpy()
    for i in range(n):                                                    # every symbol (the corner cases sit at 7..10)
        assert np.array_equal(got[i], oracle.fft(x[i], 64)), i


def test_demap_deinterleave_bit_exact(sora, torch_cuda, oracle):
    torch = torch_cuda
    rng = np.random.default_rng(2)
    x = rng.integers(-3000, 3000, size=(257, 64, 2)).astype(np.int16)
    x[0] = 32767; x[1] = -32768
    for nb in (1, 2, 4, 6):
        soft = sora.demap11a(torch.from_numpy(x).cuda(), nb)
        de = sora.deinterleave11a(soft, nb).cpu().numpy(); soft = soft.cpu().numpy()
        for i in range(len(x)):
            want = oracle.demap(nb, x[i])
            assert np.array_equal(soft[i], want)
            assert np.array_equal(de[i], oracle.deinterleave(nb, want))


@pytest.mark.parametrize("cr", [CR_12, CR_23, CR_34])
def test_viterbi_bit_exact_on_noise(sora, torch_cuda, oracle, cr):
    """Pure-noise and noisy-codeword soft streams: 8-bit wrapping metrics, LSB tie-breaks, window schedule."""
    torch = torch_cuda
    rng = np.random.default_rng(30 + cr)
    per = {CR_12: 2, CR_23: 3, CR_34: 4}[cr]
    lens = [1, 4, 30, 31, 32, 33, 36, 100, 1500, 2500, 257, 64]
    softs, offs, ns = [], [], []
    off = 0
    for L in lens:
        steps = L * 8 + 16 + 6 + 40
        nsoft = int(np.ceil(steps * per / {CR_12: 1, CR_23: 2, CR_34: 3}[cr] / 48.0)) * 48
        nsoft = (nsoft + per * 4 - 1) // (per * 4) * (per * 4)
        s = rng.integers(0, 8, size=nsoft).astype(np.uint8)
        h = nsoft // 2
        s[:h] = np.clip(rng.choice([0, 7], size=h) + rng.integers(-3, 4, size=h), 0, 7)
        softs.append(s); offs.append(off); ns.append(nsoft); off += nsoft + (-nsoft) % 4
    buf = np.zeros(off + 64, np.uint8)
    for s, o in zip(softs, offs):
        buf[o:o + len(s)] = s
    out = sora.viterbi11a(torch.from_numpy(buf).cuda(), torch.tensor(offs, dtype=torch.int32).cuda(),
                          torch.tensor(ns, dtype=torch.int32).cuda(), torch.tensor(lens, dtype=torch.int16).cuda(), cr).cpu().numpy()
    for i, L in enumerate(lens):
        want = oracle.viterbi_frame(softs[i], cr, L)
        assert len(want) == L + 2
        assert np.array_equal(out[i, :L + 2], want), (cr, L)
    # the same jobs out of a caller-owned workspace (no allocation, no host wait inside the call), twice in a row on one stream, and with an
    # odd job count (the last job has no pair mate)
    d_buf = torch.from_numpy(buf).cuda()
    ws = torch.empty(sora.viterbi11a_workspace_bytes(d_buf.numel(), len(lens)), dtype=torch.uint8, device="cuda")
    for n in (len(lens), len(lens) - 1):
        args = (torch.tensor(offs[:n], dtype=torch.int32).cuda(), torch.tensor(ns[:n], dtype=torch.int32).cuda(), torch.tensor(lens[:n], dtype=torch.int16).cuda())
        for _ in range(2):
            o2 = sora.viterbi11a_ws(d_buf, *args, cr, ws)
        torch.cuda.synchronize()
        o2 = o2.cpu().numpy()
        for i, L in enumerate(lens[:n]):
            assert np.array_equal(o2[i, :L + 2], out[i, :L + 2]), ("workspace", cr, L, n)
        # ... and through the other trellis kernel (k_viterbi16: a frame pair in 16 lanes x 4 registers, eight frames per wave): job counts
        # that leave rows of its last wave empty or half-filled
        o3 = sora.viterbi11a_ws(d_buf, *args, cr, ws, lanes_per_pair=16)
        torch.cuda.synchronize()
        o3 = o3.cpu().numpy()
        for i, L in enumerate(lens[:n]):
            assert np.array_equal(o3[i, :L + 2], out[i, :L + 2]), ("16 lanes per pair", cr, L, n)
    # the stage packs the caller's bytes to three bits per value (k_soft_pack3): jobs at ODD byte offsets, with junk in the upper five bits of
    # every byte (only the low three are soft values), decode to the same bytes
    offs2, o = [], 1
    for nsoft in ns:
        offs2.append(o); o += nsoft + 3
    buf2 = rng.integers(0, 32, size=o + 64).astype(np.uint8) << 3
    for s_, o_ in zip(softs, offs2):
        buf2[o_:o_ + len(s_)] |= s_
    d_buf2 = torch.from_numpy(buf2).cuda()
    ws2 = torch.empty(sora.viterbi11a_workspace_bytes(d_buf2.numel(), len(lens)), dtype=torch.uint8, device="cuda")
    args2 = (torch.tensor(offs2, dtype=torch.int32).cuda(), torch.tensor(ns, dtype=torch.int32).cuda(), torch.tensor(lens, dtype=torch.int16).cuda())
    for lanes in (64, 16):
        o4 = sora.viterbi11a_ws(d_buf2, *args2, cr, ws2, lanes_per_pair=lanes)
        torch.cuda.synchronize()
        o4 = o4.cpu().numpy()
        for i, L in enumerate(lens):
            assert np.array_equal(o4[i, :L + 2], out[i, :L + 2]), ("odd offsets", lanes, cr, L)
    # jobs whose nsoft is NOT a multiple of 8, laid back to back (no gap between a job's last value and the next job's first)
    # and listed in reverse order of their offsets -- round 3's placement (byte 3 ceil(off / 8), last group padded) let neighbours overwrite
    # each other's packed bytes here (off = 4, nsoft = 28 wrote bytes 3..14, the job at off = 32 starts at byte 12)
    ns3 = [n_ - 4 for n_ in ns]
    offs3, o = [], 4
    for nsoft in ns3:
        offs3.append(o); o += nsoft
    buf3 = rng.integers(0, 32, size=o + 64).astype(np.uint8) << 3
    for s_, o_, n_ in zip(softs, offs3, ns3):
        buf3[o_:o_ + n_] |= s_[:n_]
    order = list(range(len(lens)))[::-1]
    d_buf3 = torch.from_numpy(buf3).cuda()
    ws3 = torch.empty(sora.viterbi11a_workspace_bytes(d_buf3.numel(), len(lens)), dtype=torch.uint8, device="cuda")
    args3 = (torch.tensor([offs3[i] for i in order], dtype=torch.int32).cuda(), torch.tensor([ns3[i] for i in order], dtype=torch.int32).cuda(),
             torch.tensor([lens[i] for i in order], dtype=torch.int16).cuda())
    for lanes in (64, 16):
        o5 = sora.viterbi11a_ws(d_buf3, *args3, cr, ws3, lanes_per_pair=lanes)
        torch.cuda.synchronize()
        o5 = o5.cpu().numpy()
        for k, i in enumerate(order):
            assert np.array_equal(o5[k, :lens[i] + 2], out[i, :lens[i] + 2]), ("back to back, nsoft % 8 = 4", lanes, cr, lens[i])


# ------------------------------------------------------------------ whole path
def test_fsample6_golden(sora, torch_cuda, oracle, golden_dir):
    """config[1]: the reference's only IQ dump, 6 Mbps, on the GPU; MPDU sha256 + FCS as pinned in SURVEY.md 8c."""
    z = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))
    iq = z["iq_i8"].astype(np.int16) << 8
    assert len(iq) % 28 == 0
    for rate_mhz, cap in ((40, iq), (20, pad_capture(iq[::2].copy(), 20))):
        res = run_rx(sora, torch_cuda, [cap], rate_mhz)
        assert len(res) == 1
        r = res[0]
        assert r["error_code"] == E_FRAME_OK and r["rate_kbps"] == 6000 and r["length"] == 1392 and r["nsym"] == 465
        assert r["crc32"] == 0x80EF9B11
        assert hashlib.sha256(r["mpdu"]).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
        ok, why = same_results(res, oracle_results(oracle, [cap], rate_mhz))
        assert ok, why


@pytest.mark.parametrize("rate", RATES)
def test_all_rates_clean_and_noisy(sora, torch_cuda, oracle, rate):
    nb = rate_params(rate)[0]
    sigma = {1: 2200, 2: 1500, 4: 700, 6: 330}[nb]
    caps = []
    for i, (L, sg, mhz_lead) in enumerate([(40, 0, 0), (300, sigma, 12), (1500, sigma, 28), (1, 0, 4), (2496, sigma // 2, 0)]):
        c, _ = make_capture(oracle, rate, L, seed=rate + i, rate_mhz=40, sigma=sg, lead=mhz_lead)
        caps.append(c)
    got = run_rx(sora, torch_cuda, caps, 40)
    want = oracle_results(oracle, caps, 40)
    assert sum(r["error_code"] == E_FRAME_OK for r in want) >= 4
    ok, why = same_results(got, want)
    assert ok, why


def test_20mhz_input(sora, torch_cuda, oracle):
    caps = [make_capture(oracle, r, 200 + 10 * i, seed=i, rate_mhz=20, sigma=150, lead=8 * i)[0] for i, r in enumerate(RATES)]
    got = run_rx(sora, torch_cuda, caps, 20)
    ok, why = same_results(got, oracle_results(oracle, caps, 20))
    assert ok, why
    assert all(r["error_code"] == E_FRAME_OK for r in got) and len(got) == 8


def test_short_mpdus_crc_paths(sora, torch_cuda, oracle):
    """Payloads around the limits of the parallel CRC (serial below 4 bytes, 40-byte lane segments, ragged first segment)."""
    lens = [2, 3, 4, 5, 7, 8, 35, 36, 37, 39, 40, 41, 76, 80, 81, 119]
    caps = [make_capture(oracle, [12000, 54000][i % 2], L, seed=900 + i, rate_mhz=20, sigma=60, tail=160)[0] for i, L in enumerate(lens)]
    got = run_rx(sora, torch_cuda, caps, 20)
    want = oracle_results(oracle, caps, 20)
    ok, why = same_results(got, want)
    assert ok, why
    assert sum(r["error_code"] == E_FRAME_OK for r in got) == len(lens)


def test_cfo_and_heavy_noise(sora, torch_cuda, oracle):
    caps = [make_capture(oracle, 24000, 400, seed=5, sigma=300, cfo_hz=40e3)[0],
            make_capture(oracle, 54000, 800, seed=6, sigma=900)[0],          # CRC failure expected
            make_capture(oracle, 6000, 100, seed=7, sigma=6000)[0],          # may not even sync
            make_capture(oracle, 36000, 600, seed=8, sigma=500, cfo_hz=-60e3)[0]]
    got = run_rx(sora, torch_cuda, caps, 40)
    want = oracle_results(oracle, caps, 40)
    ok, why = same_results(got, want)
    assert ok, why
    assert any(r["error_code"] != E_FRAME_OK for r in want)


def test_multi_frame_and_edge_captures(sora, torch_cuda, oracle):
    rng = np.random.default_rng(11)
    parts = []
    for i, rate in enumerate((54000, 6000, 36000, 48000)):
        mp = rng.integers(0, 256, 200 + 100 * i).astype(np.uint8).tobytes()
        parts.append(oracle.tx_capture(mp, rate, lead=0, tail=400 + 52 * i))
    multi = pad_capture(awgn(np.concatenate(parts), 120, 3), 40)
    silent = np.zeros((2800, 2), np.int16)
    tiny = np.zeros((28, 2), np.int16)
    trunc = pad_capture(oracle.tx_capture(bytes(1000), 12000)[:9000], 40)
    overmtu = pad_capture(oracle.tx_capture(bytes(2497), 54000), 40)
    caps = [multi, silent, tiny, trunc, overmtu]
    got = run_rx(sora, torch_cuda, caps, 40, max_frames=8)
    want = oracle_results(oracle, caps, 40)
    ok, why = same_results(got, want)
    assert ok, why
    assert [r["capture_id"] for r in got].count(0) == 4


def test_batch_of_frames_54mbps(sora, torch_cuda, oracle):
    """A small version of config[2]: many independent 54 Mbps captures in one call."""
    caps = [make_capture(oracle, 54000, 1500, seed=100 + i, rate_mhz=20, sigma=120 + 10 * (i % 7), lead=0, tail=160)[0] for i in range(48)]
    got = run_rx(sora, torch_cuda, caps, 20, max_frames=2)
    want = oracle_results(oracle, caps, 20)
    ok, why = same_results(got, want)
    assert ok, why
    assert sum(r["error_code"] == E_FRAME_OK for r in got) >= 40


@pytest.mark.parametrize("graph", ["0", "1"])
def test_repeated_calls_replay_graph(sora, torch_cuda, oracle, graph, monkeypatch):
    """Identical consecutive calls (optionally replayed as one hipGraph launch, SORA_HIP_GRAPH=1); a different capture set drops the graph."""
    monkeypatch.setenv("SORA_HIP_GRAPH", graph)
    capsA = [make_capture(oracle, 36000, 400 + 11 * i, seed=500 + i, rate_mhz=20, sigma=100, tail=160)[0] for i in range(6)]
    capsB = [make_capture(oracle, 12000, 150 + 7 * i, seed=600 + i, rate_mhz=20, sigma=100, tail=160)[0] for i in range(5)]
    iqA, dA = batch(capsA); iqB, dB = batch(capsB)
    rx = sora.Rx(max_captures=8, max_total_samples=max(len(iqA), len(iqB)), sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(1)
    tA = torch_cuda.from_numpy(iqA).cuda(); tB = torch_cuda.from_numpy(iqB).cuda()
    wantA = oracle_results(oracle, capsA, 20); wantB = oracle_results(oracle, capsB, 20)
    for k in range(4):
        rx.process_dev(tA, dA)
        ok, why = same_results(rx.results(), wantA); assert ok, (k, why)
    for k in range(3):
        rx.process_dev(tB, dB)
        ok, why = same_results(rx.results(), wantB); assert ok, (k, why)
    tA.add_(0)                                     # same buffer, same descriptors, new contents are picked up by the replay
    tA.copy_(torch_cuda.from_numpy(iqA[::-1].copy()).cuda()); tA.copy_(torch_cuda.from_numpy(iqA).cuda())
    for k in range(3):
        rx.process_dev(tA, dA)
        ok, why = same_results(rx.results(), wantA); assert ok, (k, why)
    rx.close()


@pytest.mark.parametrize("depth", [1, 2, 4, 8, 16])
def test_calls_in_flight(sora, torch_cuda, oracle, depth):
    """Consecutive calls rotate over internal pipelines; results() always reports the most recent call."""
    sets = []
    for s in range(3):
        caps = [make_capture(oracle, [54000, 24000, 9000][s], 200 + 50 * s + 13 * i, seed=700 + 10 * s + i, rate_mhz=20, sigma=100, tail=160)[0] for i in range(4 + s)]
        iq, d = batch(caps)
        sets.append((torch_cuda.from_numpy(iq).cuda(), d, oracle_results(oracle, caps, 20)))
    rx = sora.Rx(max_captures=8, max_total_samples=max(len(t) for t, _, _ in sets), sample_rate_mhz=20, max_frames_per_capture=2)
    assert rx.set_depth(depth) == 8 and rx.set_depth(0) == depth
    assert rx.trellis() == sora.TRELLIS_WINDOWED                      # the automatic choice of the trellis kernel follows the capacity in flight (depth x max_captures): few frames -> cut into units
    big = sora.Rx(max_captures=4096, max_total_samples=4096 * 64, sample_rate_mhz=20, max_frames_per_capture=1)
    big.set_depth(depth); assert big.trellis() == (16 if depth * 4096 >= 32768 else sora.TRELLIS_WINDOWED)
    big.close()
    rx.set_trellis(16 if depth >= 4 else 64 if depth >= 2 else 0)     # (the deeper rotations run the eight-frames-per-wave kernel, as they do at full size; one call in flight: the library's choice)
    ncalls = max(9, 2 * depth + 1)                  # (every pipeline is used at least twice)
    tickets = []
    for k in range(ncalls):
        t, d, want = sets[k % 3]
        tickets.append(rx.process_dev(t, d))
        if k % 2:                                   # sometimes let several calls pile up before looking
            ok, why = same_results(rx.results(), want); assert ok, (k, why)
    for k in range(max(0, ncalls - depth), ncalls):  # every call still in flight is addressable by its ticket
        ok, why = same_results(rx.results(ticket=tickets[k]), sets[k % 3][2]); assert ok, ("ticket", k, why)
    rx.flush()
    ok, why = same_results(rx.results(), sets[(ncalls - 1) % 3][2]); assert ok, why
    rx.reset()
    with pytest.raises(Exception):
        rx.results()
    rx.close()


def test_host_buffer_entry_point(sora, torch_cuda, oracle):
    cap, mp = make_capture(oracle, 18000, 333, seed=2, rate_mhz=40, sigma=100)
    rx = sora.Rx(1, len(cap), sample_rate_mhz=40)
    rx.process(cap, [(0, len(cap), 77)])
    res = rx.results()
    assert len(res) == 1 and res[0]["capture_id"] == 77 and res[0]["mpdu"][:-4] == mp
    rx.reset()
    with pytest.raises(Exception):
        rx.results()


def test_capacity_errors(sora, torch_cuda):
    rx = sora.Rx(1, 2800, sample_rate_mhz=40)
    x = torch_cuda.zeros((5600, 2), dtype=torch_cuda.int16).cuda()
    with pytest.raises(sora.SoraError):
        rx.process_dev(x, [(0, 5600)])
    with pytest.raises(sora.SoraError):
        rx.process_dev(x, [(0, 1400), (1400, 1400)])
    with pytest.raises(sora.SoraError):
        rx.process_dev(x, [(2, 1400)])


def test_device_resident_results_and_row_codec(sora, torch_cuda, oracle):
    """sora_rx_results_dev packs dense rows on the device (what the RCCL all-gather ships); decode them back."""
    from sora_amd.shard import results_from_rows
    caps = [make_capture(oracle, r, 120 + i, seed=40 + i, rate_mhz=20, sigma=100)[0] for i, r in enumerate(RATES)]
    caps.insert(3, np.zeros((1400, 2), np.int16))                      # a silent capture: no row
    iq, descs = batch(caps)
    rx = sora.Rx(len(caps), len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    d = torch_cuda.from_numpy(iq).cuda()
    rx.process_dev(d, descs)
    rows, nrows, mpdu_ptr = rx.results_dev()
    rx.flush()
    n = int(nrows.item())
    got = results_from_rows(rows[:n].cpu().numpy())
    want = rx.results()
    assert n == len(want) == 8
    for g, w in zip(got, want):
        for k in ("capture_id", "start_sample", "end_sample", "error_code", "rate_kbps", "length", "nsym", "crc32", "cfo_est"):
            assert g[k] == w[k], k
    assert mpdu_ptr != 0


def test_randomised_captures_match_the_oracle(sora, torch_cuda, oracle):
    """A slice of tools/stress_parity.py: random rates, lengths, noise, CFO, DC, gaps, several frames per capture,
    truncated frames, pure noise -- every result row identical (the full hunt ran over 24,000 captures)."""
    from gpu_util import random_capture
    rng = np.random.default_rng(20260925)
    for mhz in (20, 40):
        caps = [random_capture(oracle, rng, mhz, multipath_p=0.3) for _ in range(150)]
        got = run_rx(sora, torch_cuda, caps, mhz, max_frames=8)
        ok, why = same_results(got, oracle_results(oracle, caps, mhz))
        assert ok, (mhz, why)


def test_gpu_equals_the_reference_graph(sora, torch_cuda, oracle):
    """The GPU path against the reference ITSELF: oracle/_ref/libsora_refgraph.so is CreateDemodGraph11a_40M compiled
    from the reference sources (oracle/build_ref.sh; it travels with the snapshot).  Every event the reference's
    RxThread loop reports -- error code, source position, rate, length, FCS, MPDU bytes -- must be what the GPU reports."""
    from gpu_util import random_capture, same_as_reference_graph
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    rng = np.random.default_rng(20260927)
    caps = [random_capture(oracle, rng, 40) for _ in range(300)]
    for k in range(40):                                                  # negative frequency offsets (floored CFO estimate)
        caps.append(make_capture(oracle, RATES[k % 8], 40 + 37 * k, 500 + k, cfo_hz=-20e3 - 1500 * k, sigma=30)[0])
    got = run_rx(sora, torch_cuda, caps, 40, max_frames=8)
    nev = 0
    for i, c in enumerate(caps):
        ev = g.rx11a(c)
        ok, why = same_as_reference_graph([r for r in got if r["capture_id"] == i], ev)
        assert ok, "capture %d: %s" % (i, why)
        nev += len(ev)
    assert nev > 300


@pytest.mark.parametrize("trellis", [64, 16])
def test_multipath_captures_equal_the_reference_graph(sora, torch_cuda, oracle, trellis):
    """frequency-selective channels on the headline path.  600 captures through 2-4 tap channels (echoes 1-8
    samples @20 MHz behind the direct path, 3-12 dB down, random phase), two in five with an echo within 1 dB of the direct path: deep
    nulls, where T11aLTS::_channel_estimation's truncating division by |Y_k|^2 >> 8 meets small divisors and, at zero, writes a zero
    coefficient (channel_11a.hpp:144-151) -- k_scan does the same.  Every event against the compiled reference graph; both trellis kernels."""
    from gpu_util import multipath_capture, same_as_reference_graph
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    rng = np.random.default_rng(20261005 + trellis)
    caps = [multipath_capture(oracle, rng, 40) for _ in range(600)]
    got = run_rx(sora, torch_cuda, caps, 40, max_frames=8, trellis=trellis)
    per = [[] for _ in caps]
    for r in got:
        per[r["capture_id"]].append(r)
    kinds = {}
    for i, c in enumerate(caps):
        ev = g.rx11a(c)
        ok, why = same_as_reference_graph(per[i], ev)
        assert ok, "capture %d: %s" % (i, why)
        for e in ev:
            kinds[e["error_code"]] = kinds.get(e["error_code"], 0) + 1
    assert kinds.get(0x1, 0) > 200 and kinds.get(0x80000006, 0) > 30, kinds    # decoded frames and frames the channel broke
    # the same captures as 20 MHz input (the even samples): what the oracle reports on them
    caps20 = [pad_capture(c[::2].copy(), 20) for c in caps[:200]]
    ok, why = same_results(run_rx(sora, torch_cuda, caps20, 20, max_frames=8, trellis=trellis), oracle_results(oracle, caps20, 20))
    assert ok, why


def test_20mhz_mode_equals_the_reference_graph_on_the_40mhz_stream(sora, torch_cuda, oracle):
    """sample_rate_mhz = 20 is DEFINED as the even samples of a 40 MHz stream (what TDownSample2 keeps): the GPU on x[::2]
    must report what the reference's 40 MHz graph reports on x."""
    from gpu_util import random_capture, same_as_reference_graph
    from oracle.pyoracle import ReferenceGraph
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    rng = np.random.default_rng(20260928)
    caps40 = [random_capture(oracle, rng, 40) for _ in range(200)]
    got = run_rx(sora, torch_cuda, [c[::2].copy() for c in caps40], 20, max_frames=8)
    for i, c in enumerate(caps40):
        ok, why = same_as_reference_graph([r for r in got if r["capture_id"] == i], g.rx11a(c))
        assert ok, "capture %d: %s" % (i, why)


# ------------------------------------------------------------------ tickets, asynchronous delivery, row limits
@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_every_call_in_flight_is_collectable_by_ticket(sora, torch_cuda, oracle, depth):
    """With `depth` calls in flight each process call has a ticket; the results of EVERY call -- not only the most recent --
    are addressable until the pipeline is reused, host rows and device rows alike; a reused ticket is refused."""
    sets = []
    for s in range(4):
        caps = [make_capture(oracle, [54000, 24000, 9000, 48000][s], 150 + 40 * s + 13 * i, seed=900 + 10 * s + i, rate_mhz=20, sigma=100, tail=160)[0] for i in range(3 + s)]
        iq, d = batch(caps)
        sets.append((torch_cuda.from_numpy(iq).cuda(), d, oracle_results(oracle, caps, 20)))
    rx = sora.Rx(max_captures=8, max_total_samples=max(len(t) for t, _, _ in sets), sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(depth)
    assert rx.ticket() == 0
    tickets = []
    for k in range(2 * depth + 1):                                   # fill the pipelines more than twice over
        t, d, _ = sets[k % 4]
        tickets.append((rx.process_dev(t, d), k % 4))
        assert rx.ticket() == tickets[-1][0]
    assert [t for t, _ in tickets] == list(range(1, 2 * depth + 2))
    for tk, s in tickets[-depth:]:                                    # the last `depth` calls are all collectable, in any order
        ok, why = same_results(rx.results(ticket=tk), sets[s][2]); assert ok, (tk, why)
        rows, nrows, _ = rx.results_dev(ticket=tk)
        rx.wait(tk)
        assert int(nrows.item()) == len(sets[s][2])
        r = rows[:len(sets[s][2])].cpu().numpy()
        assert [int(v) for v in r[:, 3].astype(np.uint32)] == [w["error_code"] for w in sets[s][2]]
    for tk, _ in tickets[:-depth]:                                    # older ones are gone
        with pytest.raises(sora.SoraError):
            rx.results(ticket=tk)
        with pytest.raises(sora.SoraError):
            rx.wait(tk)
    rx.close()


def test_results_are_delivered_to_pinned_host_memory_behind_the_kernels(sora, torch_cuda, oracle):
    """sora_rx_deliver_async: rows, row count and the MPDU array of a call arrive in page-locked host buffers without a
    host wait in between; three calls in flight, every call's delivery equals sora_rx_results of the same call."""
    sets = []
    for s in range(3):
        caps = [make_capture(oracle, [36000, 54000, 6000][s], 100 + 70 * s + 9 * i, seed=950 + 10 * s + i, rate_mhz=20, sigma=90, tail=160)[0] for i in range(5)]
        caps.append(np.zeros((1400, 2), np.int16))                   # a silent capture: no row
        iq, d = batch(caps)
        sets.append((torch_cuda.from_numpy(iq).cuda(), d, oracle_results(oracle, caps, 20)))
    n = max(len(t) for t, _, _ in sets)
    rx = sora.Rx(max_captures=8, max_total_samples=n, sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(3)
    bufs = [sora.HostResults(16, (n // 80 + 8 + 16) * 32) for _ in range(3)]
    for rnd in range(3):
        tks = []
        for s in range(3):
            t, d, _ = sets[(s + rnd) % 3]
            tk = rx.process_dev(t, d)
            assert rx.mpdu_bytes(tk) <= bufs[s].mpdu.size
            rx.deliver_async(tk, bufs[s]); tks.append(tk)
        for s, tk in enumerate(tks):
            want = sets[(s + rnd) % 3][2]
            rx.wait(tk)
            b = bufs[s]
            assert int(b.nrows[0]) == len(want)
            for row, w in zip(b.rows[:len(want)], want):
                for f in ("capture_id", "start_sample", "end_sample", "error_code", "rate_kbps", "length", "nsym", "crc32", "cfo_est"):
                    assert int(row[f]) == w[f], (f, int(row[f]), w[f])
                assert int(row["flags"]) == 0
                if w["error_code"] in (0x1, 0x80000006):
                    assert bytes(b.mpdu[int(row["mpdu_offset"]):int(row["mpdu_offset"]) + w["length"]]) == w["mpdu"]
    small = sora.HostResults(16, 64)
    with pytest.raises(sora.SoraError):                               # an MPDU buffer that is too small is refused, not overrun
        rx.deliver_async(rx.ticket(), small)
    rx.close()


def test_completions_are_taken_in_the_order_they_happen(sora, torch_cuda, oracle):
    """sora_rx_wait_any: the host takes whichever delivered call has finished (calls in flight overtake one another) and the next process call reuses
    THAT pipeline; every ticket comes back exactly once with its own results, older tickets stay addressable while a younger released one is recycled,
    and with nothing in flight the call fails instead of blocking."""
    sets = []
    for s in range(3):
        caps = [make_capture(oracle, [54000, 6000, 24000][s], 90 + 400 * (s == 1) + 11 * i, seed=1200 + 10 * s + i, rate_mhz=20, sigma=90, tail=160)[0] for i in range(2 + 2 * s)]
        iq, d = batch(caps)
        sets.append((torch_cuda.from_numpy(iq).cuda(), d, oracle_results(oracle, caps, 20)))
    n = max(len(t) for t, _, _ in sets)
    rx = sora.Rx(max_captures=8, max_total_samples=n, sample_rate_mhz=20, max_frames_per_capture=2)
    depth = 4
    rx.set_depth(depth)
    with pytest.raises(sora.SoraError):
        rx.wait_any()                                                # nothing in flight
    bufs = {}; free = [sora.HostResults(16, (n // 80 + 8 + 16) * 32) for _ in range(depth)]
    which = {}; seen = []

    def submit(k):
        t, d, _ = sets[k % 3]
        tk = rx.process_dev(t, d)
        bufs[tk] = free.pop(); which[tk] = k % 3
        rx.deliver_async(tk, bufs[tk])
        return tk

    def check(tk):
        want = sets[which[tk]][2]; b = bufs.pop(tk)
        assert int(b.nrows[0]) == len(want)
        for row, w in zip(b.rows[:len(want)], want):
            for f in ("capture_id", "start_sample", "end_sample", "error_code", "rate_kbps", "length", "nsym", "crc32", "cfo_est"):
                assert int(row[f]) == w[f], (tk, f, int(row[f]), w[f])
            if w["error_code"] == 0x1:
                assert bytes(b.mpdu[int(row["mpdu_offset"]):int(row["mpdu_offset"]) + w["length"]]) == w["mpdu"]
        free.append(b)
    k = 0
    for _ in range(depth):
        submit(k); k += 1
    for _ in range(40):
        tk = rx.wait_any(); seen.append(tk); check(tk)
        submit(k); k += 1
    while bufs:
        tk = rx.wait_any(); seen.append(tk); check(tk)
    assert sorted(seen) == list(range(1, k + 1))                     # every call exactly once
    with pytest.raises(sora.SoraError):
        rx.wait_any()
    # a released younger call is recycled ahead of older calls still held
    rx.flush()
    a = submit(k); b_ = submit(k + 1); c = submit(k + 2); d_ = submit(k + 3)
    rx.wait(c); check(c)                                              # c is released (delivered and waited for)
    e = submit(k + 4)                                                 # takes c's pipeline: a, b, d stay addressable
    for tk in (a, b_, d_, e):
        ok, why = same_results(rx.results(ticket=tk), sets[which[tk]][2]); assert ok, (tk, why)
    with pytest.raises(sora.SoraError):
        rx.results(ticket=c)
    rx.close()


def test_frames_beyond_the_row_limit_are_counted_and_flagged(sora, torch_cuda, oracle):
    """A capture with more frames than max_frames_per_capture: the rows that exist are the first frames, unchanged, and the
    last one carries SORA_ROW_TRUNCATED; nothing of a later frame leaks into a reported row (the reference reports every frame)."""
    rng = np.random.default_rng(77)
    parts = [oracle.tx_capture(rng.integers(0, 256, 120 + 60 * i).astype(np.uint8).tobytes(), (54000, 12000, 24000, 6000)[i], lead=0, tail=400) for i in range(4)]
    multi = pad_capture(awgn(np.concatenate(parts), 100, 5), 40)
    single = make_capture(oracle, 18000, 200, seed=5, rate_mhz=40, sigma=80, tail=600)[0]
    caps = [multi, single]
    want = oracle_results(oracle, caps, 40)
    assert [w["capture_id"] for w in want].count(0) == 4
    for mf in (1, 2, 3, 4, 5):
        got = run_rx(sora, torch_cuda, caps, 40, max_frames=mf)
        exp = [w for w in want if w["capture_id"] == 0][:mf] + [w for w in want if w["capture_id"] == 1]
        ok, why = same_results(got, exp); assert ok, (mf, why)
        flags = [r["flags"] for r in got]
        last0 = min(mf, 4) - 1
        assert flags == [1 if (i == last0 and mf < 4) else 0 for i in range(len(got))], (mf, flags)


def test_oversized_configuration_is_refused(sora):
    """A handle whose symbol-slot geometry would overflow the 32-bit offsets of the device tables is refused at creation."""
    with pytest.raises(sora.SoraError) as e:
        sora.Rx(max_captures=16, max_total_samples=1_300_000_000, sample_rate_mhz=20)
    assert e.value.code == -6


def test_host_descriptor_past_the_buffer_is_refused(sora, torch_cuda, oracle):
    cap, _ = make_capture(oracle, 18000, 100, seed=3, rate_mhz=40, sigma=50)
    rx = sora.Rx(2, 2 * len(cap), sample_rate_mhz=40)
    rx.process(cap, [(0, len(cap), 1)])
    good = rx.results()
    with pytest.raises(sora.SoraError):
        rx.process(cap, [(0, len(cap), 1), (len(cap) - 280, 560, 2)])
    ok, why = same_results(rx.results(), good); assert ok, why          # the refused call left the previous results intact
    rx.close()


def test_config3_full_batch_equals_the_reference_graph(sora, torch_cuda, oracle):
    """BASELINE configs[2] at its stated size: the bench workload (4096 captures x one 1500-byte 54 Mbps frame, same seeds)
    through one process call, every capture against the reference's own graph compiled from its sources -- event for event
    (error code, source position, rate, length, FCS, MPDU bytes).  bench.py gates its headline on the same comparison."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from oracle.pyoracle import ReferenceGraph
    if not ReferenceGraph().available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    nfr = bench.FRAMES_PER_GPU
    iq, descs, payloads = bench.make_workload(oracle, nfr, seed0=0)
    rx = sora.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    t = rx.process_dev(torch_cuda.from_numpy(iq).cuda(), sora.Rx.captures(descs))
    res = rx.results(ticket=t)
    rx.close()
    kind, want = bench.reference_rows(iq, nfr, oracle)
    assert kind == "reference" and sum(len(v) for v in want.values()) >= nfr
    ok, why = bench.check_against_reference(res, kind, want, range(nfr))
    assert ok, why
    good = sum(1 for r in res if r["error_code"] == E_FRAME_OK and r["mpdu"][:-4] == payloads[r["capture_id"]])
    assert good >= nfr * 0.95, good                                   # ~2 % of the 27 dB captures fail their FCS -- in the reference too (compared above)


# ------------------------------------------------------------------ the fused decode kernel (k_decode)
# A build variant since round 4: these three tests run only when the loaded library was built with SORA_WITH_K_DECODE
# (sora_amd.build.build_variant("fused", ["SORA_WITH_K_DECODE"]); SORA_HIP_LIB=sora_amd/lib/variants/fused.so); the default library
# answers sora_rx_set_fused(1) with SORA_E_NOT_SUPPORTED, which test_fused_mode_is_a_build_variant checks.
def _need_fused(sora):
    rx = sora.Rx(max_captures=1, max_total_samples=1024, sample_rate_mhz=20)
    try:
        rx.set_fused(1)
    except sora.SoraError:
        pytest.skip("k_decode is not part of this build of libsora_hip.so (build variant 'fused')")
    finally:
        rx.close()


def test_fused_mode_is_a_build_variant(sora, torch_cuda):
    rx = sora.Rx(max_captures=1, max_total_samples=1024, sample_rate_mhz=20)
    assert rx.set_fused(-1) == 0 and rx.set_fused(0) == 0
    try:
        assert rx.set_fused(1) == 0 and rx.set_fused(-1) == 1           # a variant build: the switch works
    except sora.SoraError as e:
        assert e.code & 0xFFFFFFFF == 0x80000003 and rx.set_fused(-1) == 0   # the default build: SORA_E_NOT_SUPPORTED, nothing changed
    rx.close()


def test_fused_decode_kernel_equals_the_oracle_on_random_captures(sora, torch_cuda, oracle):
    """sora_rx_set_fused(1): symbol waves feeding trellis waves through an LDS ring inside one kernel.  Same rows as the oracle,
    and as the split path, on random captures: all rates (so frames of different modulation share a trellis wave), lengths,
    noise up to failure, several frames per capture, truncation, pure noise."""
    _need_fused(sora)
    from gpu_util import random_capture
    rng = np.random.default_rng(20261001)
    for mhz in (20, 40):
        caps = [random_capture(oracle, rng, mhz) for _ in range(200)]
        want = oracle_results(oracle, caps, mhz)
        got = run_rx(sora, torch_cuda, caps, mhz, max_frames=8, fused=1)
        ok, why = same_results(got, want)
        assert ok, (mhz, why)
        ok, why = same_results(run_rx(sora, torch_cuda, caps, mhz, max_frames=8, fused=0), want)
        assert ok, (mhz, why)


def test_fused_decode_kernel_on_lengths_rates_and_odd_lists(sora, torch_cuda, oracle):
    """Every rate at lengths around the window schedule's corners, list sizes 1..5 per code rate (the last frame of an odd list
    runs alone in its trellis wave), frames of very different length sharing a wave."""
    _need_fused(sora)
    caps = []
    for i, rate in enumerate(RATES):
        for j, ln in enumerate((1, 5, 29, 30, 31, 33, 100, 257, 1024, 1500, 2304)[: 3 + (i % 5) * 2]):
            caps.append(make_capture(oracle, rate, ln, seed=3000 + 20 * i + j, rate_mhz=20, sigma=60 + 15 * j, tail=160)[0])
    want = oracle_results(oracle, caps, 20)
    ok, why = same_results(run_rx(sora, torch_cuda, caps, 20, max_frames=2, fused=1), want)
    assert ok, why
    for n in (1, 2, 3):                                               # a single frame, a single pair, an odd list
        ok, why = same_results(run_rx(sora, torch_cuda, caps[:n], 20, max_frames=2, fused=1), oracle_results(oracle, caps[:n], 20))
        assert ok, (n, why)


def test_fused_decode_kernel_full_batch_equals_the_reference_graph(sora, torch_cuda, oracle):
    """The bench workload (BASELINE configs[2], 4096 x 1500 B at 54 Mbps) through k_decode, every capture against the compiled reference graph."""
    _need_fused(sora)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from oracle.pyoracle import ReferenceGraph
    if not ReferenceGraph().available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    nfr = bench.FRAMES_PER_GPU
    iq, descs, _ = bench.make_workload(oracle, nfr, seed0=0)
    rx = sora.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    assert rx.set_fused(1) == 0 and rx.set_fused(-1) == 1
    d = torch_cuda.from_numpy(iq).cuda(); dd = sora.Rx.captures(descs)
    tickets = [rx.process_dev(d, dd) for _ in range(3)]              # three calls in flight on the fused path
    kind, want = bench.reference_rows(iq, nfr, oracle)
    for t in tickets:
        ok, why = bench.check_against_reference(rx.results(ticket=t), kind, want, range(nfr))
        assert ok, why
    rx.close()


# ------------------------------------------------------------------ the 16-lanes-per-pair trellis kernel (k_viterbi16)
def test_trellis16_equals_the_oracle_on_random_captures(sora, torch_cuda, oracle):
    """sora_rx_set_trellis(16): eight frames per wave, the coset layout of k_vit16.hip.  Same rows as the oracle on random captures:
    all rates (frames of different modulation and length share a wave), noise up to failure, several frames per capture, truncation."""
    from gpu_util import random_capture
    rng = np.random.default_rng(20261003)
    for mhz in (20, 40):
        caps = [random_capture(oracle, rng, mhz) for _ in range(200)]
        want = oracle_results(oracle, caps, mhz)
        ok, why = same_results(run_rx(sora, torch_cuda, caps, mhz, max_frames=8, trellis=16), want)
        assert ok, (mhz, why)


def test_trellis16_on_lengths_rates_and_partly_filled_waves(sora, torch_cuda, oracle):
    """Every rate at lengths around the window schedule's corners; list sizes 1..9 per code rate (rows of the last wave empty, a
    pair without its second frame), frames of very different length sharing a wave."""
    caps = []
    for i, rate in enumerate(RATES):
        for j, ln in enumerate((1, 5, 29, 30, 31, 33, 100, 257, 1024, 1500, 2304)[: 3 + (i % 5) * 2]):
            caps.append(make_capture(oracle, rate, ln, seed=3100 + 20 * i + j, rate_mhz=20, sigma=60 + 15 * j, tail=160)[0])
    want = oracle_results(oracle, caps, 20)
    ok, why = same_results(run_rx(sora, torch_cuda, caps, 20, max_frames=2, trellis=16), want)
    assert ok, why
    for n in (1, 2, 3, 7, 8, 9):
        ok, why = same_results(run_rx(sora, torch_cuda, caps[:n], 20, max_frames=2, trellis=16), oracle_results(oracle, caps[:n], 20))
        assert ok, (n, why)


def test_trellis16_full_batch_equals_the_reference_graph(sora, torch_cuda, oracle):
    """The bench workload (BASELINE configs[2], 4096 x 1500 B at 54 Mbps) through k_viterbi16 with four calls in flight, every capture of
    every call against the compiled reference graph."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from oracle.pyoracle import ReferenceGraph
    if not ReferenceGraph().available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not present")
    nfr = bench.FRAMES_PER_GPU
    iq, descs, _ = bench.make_workload(oracle, nfr, seed0=0)
    rx = sora.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
    rx.set_depth(4)
    rx.set_trellis(16)
    assert rx.set_trellis(-1) == 16 and rx.trellis() == 16
    d = torch_cuda.from_numpy(iq).cuda(); dd = sora.Rx.captures(descs)
    tickets = [rx.process_dev(d, dd) for _ in range(4)]
    kind, want = bench.reference_rows(iq, nfr, oracle)
    for t in tickets:
        ok, why = bench.check_against_reference(rx.results(ticket=t), kind, want, range(nfr))
        assert ok, why
    rx.close()


def test_gpu_mpdus_equal_the_legacy_receivers(sora, torch_cuda, oracle):
    """The second cross-check oracle: the reference's LEGACY dot11a C receiver compiled from its sources
    (oracle/_ref/libsora_reflegacy.so, tests/test_oracle_legacy.py).  An independent implementation: every frame it decodes with a good FCS the
    GPU path decodes too, to the same bytes."""
    from oracle.pyoracle import ReferenceLegacy
    lg = ReferenceLegacy()
    if not lg.available():
        pytest.skip("oracle/_ref/libsora_reflegacy.so not present")
    rng = np.random.default_rng(416)
    caps, want = [], []
    for i in range(48):
        rate = RATES[i % 8]; ln = int(rng.integers(20, 1400))
        mp = rng.integers(0, 256, ln).astype(np.uint8).tobytes()
        cap = oracle.tx_capture(mp, rate, lead=int(rng.integers(300, 900)) // 28 * 28, tail=1400)
        if i % 3:
            cap = awgn(cap, [0, 150, 450][i % 3], i)
        cap = cap[:len(cap) // 28 * 28]
        ev = [e for e in lg.rx11a(cap) if e["hr"] == 0x202]
        caps.append(cap); want.append(ev[0] if ev else None)
    got = run_rx(sora, torch_cuda, caps, 40)
    n = 0
    for i, w in enumerate(want):
        if w is None:
            continue
        rows = [r for r in got if r["capture_id"] == i and r["error_code"] == E_FRAME_OK]
        assert len(rows) == 1 and rows[0]["mpdu"] == w["mpdu"] and rows[0]["rate_kbps"] == w["rate_kbps"] and rows[0]["length"] == w["length"], i
        n += 1
    assert n >= 40, n
