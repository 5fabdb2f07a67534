"""The 40 MHz HT two-stream data field on the GPU (sora_ht40_*, k_ht40.hip; BASELINE configs[3]).  PARITY UNPINNED -- the reference has no such
receiver.  What is checked: (1) loop-back with the independent numpy model of the format (oracle/py_ht40.py, written from IEEE 802.11n-2009):
every modulation and code rate, 2x2 channels with cross-talk, CFO, noise -- both streams' PSDUs come back with a good FCS; (2) the pieces the
reference does have: with noise_var = 0 the detection weights are TMimoChannelEst's zero-forcing inverse bit for bit (sora_hip_mimo_est11n, itself
pinned to the reference brick, on the same channel matrices); (3) the MMSE weights against a float64 evaluation, tolerance +-1 LSB of the int16
weight (single-precision arithmetic on the GPU), and against float64 ALL the way from the raw samples, tolerance +-32 LSB (the fixed-point FFT<128>);
(4) the unchanged trellis kernel decodes the two streams of a frame in one wave."""
import numpy as np
import pytest

from oracle import py_ht40 as m

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    return torch, sora_amd


def make_frames(rng, specs, sigma=8.0, cfo_step=0.0):
    """specs: [(nbpsc, code_rate, len0, len1)] -> (iq [2, n, 2] int16, descriptors, psdus)"""
    parts, descs, psdus, pos = [], [], [], 0
    for i, (nb, cr, l0, l1) in enumerate(specs):
        ps = [m.add_fcs(rng.integers(0, 256, l0 - 4, dtype=np.uint8).tobytes()), m.add_fcs(rng.integers(0, 256, l1 - 4, dtype=np.uint8).tobytes())]
        x, nsym = m.tx(ps, nb, cr, seeds=(int(rng.integers(1, 128)), int(rng.integers(1, 128))))
        ph = rng.uniform(0, 2 * np.pi, 4)
        H = np.array([[1.0 * np.exp(1j * ph[0]), 0.35 * np.exp(1j * ph[1])], [0.3 * np.exp(1j * ph[2]), 0.9 * np.exp(1j * ph[3])]])
        lead = 64 * int(rng.integers(0, 4))
        y = m.channel(x, H, sigma, rng, cfo_step=cfo_step, lead=lead)
        pad = (-y.shape[1]) % 64
        y = np.concatenate([y, np.zeros((2, pad + 64, 2), np.int16)], axis=1)
        descs.append((pos + lead, nb, cr, l0, l1, int(round(-cfo_step)), float(2 * sigma * sigma / 128.0), i))
        parts.append(y); psdus.append(ps); pos += y.shape[1]
    return np.concatenate(parts, axis=1), descs, psdus


def comp0(x):
    """TFreqComp_11n at cfo = 0, theta = 0: sat((x * (32767 + 0j)) >> 15) -- not the identity (positive values lose one LSB)"""
    return ((x.astype(np.int64) * 32767) >> 15).astype(np.int16)


def run(env, iq, descs, want_w=False, trellis=None):
    torch, sora = env
    nsoft = sum(2 * (sora.ht40_symbols(d[3], d[4], d[1], d[2]) * 108 * d[1] + 64) for d in descs)
    rx = sora.RxHt40(len(descs), nsoft)
    if trellis is not None:
        rx.set_trellis(trellis)
    w = torch.zeros((len(descs), 4, 128, 2), dtype=torch.int16, device="cuda") if want_w else None
    rx.process_dev(torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda(), descs, w)
    res = rx.results(); rx.close()
    return res, (w.cpu().numpy() if want_w else None)


@pytest.mark.parametrize("nb,cr", [(1, 0), (2, 0), (2, 2), (4, 0), (4, 2), (6, 1), (6, 2)])
def test_loopback_every_modulation_and_rate(env, nb, cr):
    rng = np.random.default_rng(100 * nb + cr)
    specs = [(nb, cr, int(rng.integers(30, 700)), int(rng.integers(30, 700))) for _ in range(6)] + [(nb, cr, 40, 1500), (nb, cr, 1500, 1500)]
    iq, descs, psdus = make_frames(rng, specs, sigma=6.0)
    res, _ = run(env, iq, descs)
    assert len(res) == 2 * len(specs)
    for r in res:
        f, s = r["capture_id"], r["stream"]
        assert r["error_code"] == 1, (nb, cr, f, s, hex(r["error_code"]))
        assert r["mpdu"] == psdus[f][s], (nb, cr, f, s)


def test_loopback_with_carrier_offset_and_mixed_batch(env):
    rng = np.random.default_rng(7)
    specs = [(int(rng.choice([1, 2, 4, 6])), int(rng.choice([0, 1, 2])), int(rng.integers(20, 900)), int(rng.integers(20, 900))) for _ in range(40)]
    iq, descs, psdus = make_frames(rng, specs, sigma=5.0, cfo_step=37.0)      # 37 / 65536 of a turn per 40 MHz sample = 22.6 kHz
    res, _ = run(env, iq, descs)
    ok = sum(r["error_code"] == 1 and r["mpdu"] == psdus[r["capture_id"]][r["stream"]] for r in res)
    assert ok == 2 * len(specs), ok


def test_both_trellis_kernels_decode_the_streams_alike(env):
    """sora_ht40_set_trellis(16): the two streams of a frame as one pair in 16 lanes x 4 registers (k_viterbi16_11n), four frames per wave --
    same rows and PSDUs as the 64-lane kernel, on a mixed batch that includes frames too noisy to decode."""
    rng = np.random.default_rng(8)
    specs = [(int(rng.choice([1, 2, 4, 6])), int(rng.choice([0, 1, 2])), int(rng.integers(20, 900)), int(rng.integers(20, 900))) for _ in range(37)]
    iq, descs, psdus = make_frames(rng, specs, sigma=14.0, cfo_step=-21.0)
    a, _ = run(env, iq, descs, trellis=64)
    b, _ = run(env, iq, descs, trellis=16)
    key = lambda r: (r["capture_id"], r["stream"], r["error_code"], r["length"], r["crc32"], r["mpdu"])
    assert [key(r) for r in a] == [key(r) for r in b]
    assert sum(r["error_code"] == 1 for r in a) > len(specs)


def test_calls_in_flight_keep_their_results_apart(env):
    """A handle keeps several calls in flight (sora_ht40_calls_in_flight) (own stream and intermediates each): five different batches issued back to back, then the same
    five with the results read after every call -- the rows of the last call, and of every call read in turn, are the batch's own."""
    torch, sora = env
    rng = np.random.default_rng(77)
    batches = []
    for b in range(5):
        specs = [(int(rng.choice([2, 4, 6])), int(rng.choice([0, 2])), int(rng.integers(60, 900)), int(rng.integers(60, 900))) for _ in range(12)]
        iq, descs, psdus = make_frames(rng, specs, sigma=5.0)
        batches.append((torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda(), descs, psdus))
    nsoft = max(sum(2 * (sora.ht40_symbols(d[3], d[4], d[1], d[2]) * 108 * d[1] + 64) for d in descs) for _, _, descs, _ in batches)
    rx = sora.RxHt40(12, nsoft)

    def check(res, psdus):
        assert len(res) == 24
        for r in res:
            assert r["error_code"] == 1 and r["mpdu"] == psdus[r["capture_id"]][r["stream"]], (r["capture_id"], r["stream"], hex(r["error_code"]))

    for f0, f1, descs, _ in batches:                                    # no read-back in between
        rx.process_dev(f0, f1, descs)
    check(rx.results(), batches[-1][3])
    for f0, f1, descs, psdus in batches:                                # and one by one
        rx.process_dev(f0, f1, descs)
        check(rx.results(), psdus)
    # tickets: every call in flight is collectable by its own ticket while later ones run (sora_ht40_ticket / _wait / _results_of);
    # a ticket whose slot has been reused is refused
    depth = rx.calls_in_flight()
    assert depth >= 3
    tickets = []
    for k in range(9):
        f0, f1, descs, psdus = batches[k % 5]
        tickets.append((rx.process_dev(f0, f1, descs), psdus))
        if len(tickets) >= depth:
            t, ps = tickets[-depth]
            rx.wait(t)
            check(rx.results(ticket=t), ps)                              # the oldest call in flight, read while two newer ones run
    for t, ps in tickets[-depth:]:
        check(rx.results(ticket=t), ps)
    assert rx.ticket() == tickets[-1][0]
    with pytest.raises(sora.SoraError):
        rx.results(ticket=tickets[0][0])
    # delivery without a host wait (sora_ht40_deliver_async): two rows per frame and the PSDUs, densely packed, behind each call's kernels
    key = lambda r: (r["capture_id"], r["stream"], r["error_code"], r["rate_kbps"], r["length"], r["nsym"], r["crc32"], r["mpdu"])
    bufs = [sora.HostResults(24, 24 * 1024) for _ in range(depth)]
    pend = []
    for k in range(7):
        f0, f1, descs, psdus = batches[k % 5]
        t = rx.process_dev(f0, f1, descs)
        rx.deliver_async(t, bufs[k % depth]); pend.append((t, bufs[k % depth], psdus))
        if len(pend) >= depth:
            t0, b0, ps = pend.pop(0)
            rx.wait(t0)
            got = b0.results()
            check(got, ps)
            assert [key(r) for r in got] == [key(r) for r in rx.results(ticket=t0)]
    for b in bufs:
        b.close()
    rx.synchronize(); rx.close()


def test_noise_decides_and_mmse_is_not_worse_than_zf(env):
    """Over noise levels around the point where 64-QAM 3/4 begins to fail, the MMSE weights lose no more PSDUs than zero forcing (same captures)."""
    good = {"zf": 0, "mmse": 0}; total = 0
    for sigma in (90.0, 120.0, 150.0, 180.0):
        rng = np.random.default_rng(int(sigma))
        specs = [(6, 2, 300, 300)] * 32
        iq, descs, psdus = make_frames(rng, specs, sigma=sigma)
        for name, nv in (("zf", 0.0), ("mmse", None)):
            d2 = [d if nv is None else d[:6] + (nv,) + d[7:] for d in descs]
            res, _ = run(env, iq, d2)
            good[name] += sum(r["error_code"] == 1 and r["mpdu"] == psdus[r["capture_id"]][r["stream"]] for r in res)
        total += 2 * len(specs)
    assert 0 < good["zf"] < total, (good, total)                            # the operating points really are marginal
    assert good["mmse"] >= good["zf"], (good, total)


def test_zero_forcing_weights_are_the_reference_bricks(env):
    """noise_var = 0: the detection weights of carrier k equal TMimoChannelEst's inverse for the same channel matrix.  The 20 MHz brick
    (sora_hip_mimo_est11n, pinned to the reference in tests/test_11n_stages.py) is fed the HT-LTF bins of 57 carriers at a time, with the
    sign of the 40 MHz HT-LTF moved into the input so that both see the same H."""
    torch, sora = env
    rng = np.random.default_rng(5)
    iq, descs, _ = make_frames(rng, [(2, 0, 60, 60)] * 3, sigma=4.0)
    descs = [d[:6] + (0.0,) + d[7:] for d in descs]
    _, w = run(env, iq, descs, want_w=True)
    ltf20 = np.array([1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                      1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1])      # 20 MHz HT-LTF, -28..28
    for f, d in enumerate(descs):
        # the kernel's own FFT outputs are not exposed; rebuild them with the pinned FFT<128> stage from the frequency-compensated samples (cfo = 0 here)
        off = d[0]
        sym = comp0(np.stack([iq[r, off + 160 * s + 32: off + 160 * s + 160] for s in range(2) for r in range(2)]))   # [ltf sym, chain]
        Y = sora.fft128(torch.from_numpy(sym.copy()).cuda()).cpu().numpy().reshape(2, 2, 128, 2)                      # [sym][chain][bin]
        # a 40 MHz carrier is given to a 20 MHz bin whose HT-LTF sign is the same, so that the brick's sign rule is the kernel's
        pool = {1: [b for b in range(-28, 29) if ltf20[b + 28] == 1], -1: [b for b in range(-28, 29) if ltf20[b + 28] == -1]}
        todo = {1: [k for k in range(-58, 59) if m.HTLTF40[k + 58] == 1], -1: [k for k in range(-58, 59) if m.HTLTF40[k + 58] == -1]}
        checked = 0
        for sgn in (1, -1):
            for k0 in range(0, len(todo[sgn]), len(pool[sgn])):
                grp = todo[sgn][k0:k0 + len(pool[sgn])]
                l0 = np.zeros((1, 128, 2), np.int16); l1 = np.zeros((1, 128, 2), np.int16)
                for k, b20 in zip(grp, pool[sgn]):
                    for sy in range(2):
                        l0[0, 64 * sy + (b20 % 64)] = Y[sy, 0, k % 128]; l1[0, 64 * sy + (b20 % 64)] = Y[sy, 1, k % 128]
                _, hinv = sora.mimo_est11n(torch.from_numpy(l0).cuda(), torch.from_numpy(l1).cuda())
                hinv = hinv.cpu().numpy().reshape(2, 128, 2)
                for k, b20 in zip(grp, pool[sgn]):
                    got = w[f, :, k % 128]                                   # [4][2]: w00, w01, w10, w11
                    want = np.stack([hinv[0, b20 % 64], hinv[0, 64 + b20 % 64], hinv[1, b20 % 64], hinv[1, 64 + b20 % 64]])
                    assert np.array_equal(got, want), (f, k, got, want)
                    checked += 1
        assert checked == 114


def test_mmse_weights_against_float64(env):
    torch, sora = env
    rng = np.random.default_rng(6)
    iq, descs, _ = make_frames(rng, [(4, 0, 80, 80)] * 4, sigma=10.0)
    nv = 3000.0
    descs = [d[:6] + (nv,) + d[7:] for d in descs]
    _, w = run(env, iq, descs, want_w=True)
    worst = 0
    for f, d in enumerate(descs):
        off = d[0]
        sym = comp0(np.stack([iq[r, off + 160 * s + 32: off + 160 * s + 160] for s in range(2) for r in range(2)]))
        Y = sora.fft128(torch.from_numpy(sym.copy()).cuda()).cpu().numpy().reshape(2, 2, 128, 2).astype(np.int64)
        for k in range(-58, 59):
            v = int(m.HTLTF40[k + 58])
            if v == 0:
                continue
            b = k % 128
            def sat_half(a, c, sub):                                          # sra(csubs / cadds, 1) then the sign of the HT-LTF
                z = np.clip(a - c if sub else a + c, -32768, 32767) >> 1
                return z if v == 1 else -z
            H = np.zeros((2, 2), complex)
            for r in range(2):
                p, q = Y[0, r, b], Y[1, r, b]
                dd = sat_half(p, q, True); ss = sat_half(p, q, False)
                H[r, 0] = dd[0] + 1j * dd[1]; H[r, 1] = ss[0] + 1j * ss[1]
            W = np.linalg.solve(H.conj().T @ H + nv * np.eye(2), H.conj().T)
            W = W / np.real(np.diag(W @ H))[:, None] * 65536.0                # unbiased: each stream's own gain is 1
            want = np.array([[W[0, 0].real, W[0, 0].imag], [W[0, 1].real, W[0, 1].imag], [W[1, 0].real, W[1, 0].imag], [W[1, 1].real, W[1, 1].imag]])
            if np.abs(want).max() > 32000:
                continue
            worst = max(worst, float(np.abs(w[f, :, b].astype(float) - want).max()))
    assert worst <= 1.0, worst


# The stated tolerance of the FFT<128> path's intermediates: the detection weights the data field works with -- the first
# quantity behind the two HT-LTF transforms -- against an evaluation that is float64 ALL the way (numpy's FFT of the raw int16 samples, no fixed-point step
# anywhere), weights in the kernel's Q16 (1.0 = 65536).  Two sources of difference: the fixed-point transform (its stages shift and saturate; gain 1/128) and the
# single-precision solve.  Bound: |difference| <= kWeightTolLsb LSB on every data / pilot carrier whose weight fits the int16 (the others saturate by design).
kFft128Gain = 1.0 / 128.0
kWeightTolLsb = 32.0                                                   # (measured: 12.7 LSB worst over 684 carriers; 32 LSB = 2^-11 of a unit gain)


def test_detection_weights_against_an_all_float64_chain(env):
    torch, sora = env
    rng = np.random.default_rng(66)
    iq, descs, _ = make_frames(rng, [(4, 0, 80, 80)] * 6, sigma=10.0)
    nv = 3000.0
    descs = [d[:6] + (nv,) + d[7:] for d in descs]
    _, w = run(env, iq, descs, want_w=True)
    worst = 0.0; gains = []; checked = 0
    for f, d in enumerate(descs):
        off = d[0]
        Yf = np.zeros((2, 2, 128), complex)
        for s in range(2):
            for r in range(2):
                x = iq[r, off + 160 * s + 32: off + 160 * s + 160].astype(np.float64)
                Yf[s, r] = np.fft.fft(x[:, 0] + 1j * x[:, 1]) * kFft128Gain
        sym = comp0(np.stack([iq[r, off + 160 * s + 32: off + 160 * s + 160] for s in range(2) for r in range(2)]))
        Yg = sora.fft128(torch.from_numpy(sym.copy()).cuda()).cpu().numpy().reshape(2, 2, 128, 2).astype(np.float64)
        for k in range(-58, 59):
            v = int(m.HTLTF40[k + 58])
            if v == 0:
                continue
            b = k % 128
            gains.append(np.abs(Yg[0, 0, b, 0] + 1j * Yg[0, 0, b, 1]) / max(np.abs(Yf[0, 0, b]), 1e-9))
            H = np.zeros((2, 2), complex)
            for r in range(2):
                H[r, 0] = v * (Yf[0, r, b] - Yf[1, r, b]) / 2; H[r, 1] = v * (Yf[0, r, b] + Yf[1, r, b]) / 2
            W = np.linalg.solve(H.conj().T @ H + nv * np.eye(2), H.conj().T)
            W = W / np.real(np.diag(W @ H))[:, None] * 65536.0
            want = np.array([[W[0, 0].real, W[0, 0].imag], [W[0, 1].real, W[0, 1].imag], [W[1, 0].real, W[1, 0].imag], [W[1, 1].real, W[1, 1].imag]])
            if np.abs(want).max() > 32000:
                continue
            worst = max(worst, float(np.abs(w[f, :, b].astype(float) - want).max())); checked += 1
    print("FFT<128> path, detection weights against float64 all the way: worst |difference| %.2f LSB of Q16 over %d carriers; fixed-point / float transform gain %.4f .. %.4f"
          % (worst, checked, min(gains), max(gains)))
    assert checked >= 6 * 100, checked
    assert worst <= kWeightTolLsb, (worst, min(gains), max(gains))


# ------------------------------------------------------------------ raw captures: the front end (sora_ht40_process_captures_dev)
def _raw_captures(rng, specs, sigma=8.0, cfo=0.0):
    """specs: per capture a list of (mcs, length) frames (or a callable that spoils the waveform).  -> (iq [2, n, 2], descs, truth per capture)"""
    parts, descs, truth, pos = [], [], [], 0
    for ci, frames in enumerate(specs):
        segs = []; want = []
        for fi, (mcs, ln, spoil) in enumerate(frames):
            ps = [m.add_fcs(rng.integers(0, 256, ln - 4, dtype=np.uint8).tobytes()) for _ in range(2)]
            x, nsym, pre = m.tx_frame(ps, mcs)
            if spoil == "sig":                                           # garbage where L-SIG / HT-SIG should be: the header must fail
                x[:, 640:1120] = x[:, 640:1120][:, ::-1] * 1j
            ph = rng.uniform(0, 2 * np.pi, 4)
            H = np.array([[1.0 * np.exp(1j * ph[0]), 0.3 * np.exp(1j * ph[1])], [0.25 * np.exp(1j * ph[2]), 0.9 * np.exp(1j * ph[3])]])
            y = m.channel(x, H, 0.0, rng, cfo_step=cfo, lead=int(rng.integers(300, 900)))
            segs.append(y); want.append((mcs, ps, spoil))
        y = np.concatenate(segs + [np.zeros((2, 800, 2), np.int16)], axis=1).astype(np.float64)
        y += rng.normal(0, sigma, y.shape)
        y = np.clip(np.rint(y), -32768, 32767).astype(np.int16)
        n = y.shape[1] // 28 * 28
        parts.append(y[:, :n]); descs.append((pos, n, 100 + ci)); truth.append(want); pos += n
    return np.concatenate(parts, axis=1), descs, truth


def test_raw_captures_front_end_finds_parses_and_decodes(env):
    """BASELINE configs[3] on raw two-chain 40 MHz captures: carrier sense, L-LTF, L-SIG / HT-SIG (the reference's 20 MHz bricks on the
    duplicated legacy preamble), CFO and noise variance estimated, then the 40 MHz data field (FFT<128>, unbiased MMSE, a decoder per
    stream).  Loop-back against the numpy model (parity unpinned for the 40 MHz extension): every MCS 8..14, several frames per capture,
    a carrier offset, a frame whose SIG field is spoiled (one PLCP row), captures of pure noise."""
    torch, sora = env
    rng = np.random.default_rng(4040)
    specs = [[(8 + k % 7, int(rng.integers(40, 900)), None)] for k in range(14)]
    specs += [[(9, 120, None), (13, 700, None), (11, 64, None)], [(14, 1500, None), (10, 300, "sig"), (12, 333, None)], []]
    for cfo in (0.0, 21.0):
        iq, descs, truth = _raw_captures(rng, specs, sigma=8.0, cfo=cfo)
        nsoft = 2 * sum(2 * (m.nsym_for([ln, ln], *m.MCS2[mcs]) * 108 * m.MCS2[mcs][0] + 64) for fr in specs for mcs, ln, _ in fr)
        rx = sora.RxHt40(64, nsoft)
        t = rx.process_captures_dev(torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda(), descs, max_frames_per_capture=4)
        res = rx.results(ticket=t)
        per = {}
        for r in res:
            per.setdefault(r["capture_id"], []).append(r)
        for ci, want in enumerate(truth):
            got = per.get(100 + ci, [])
            exp = []
            for mcs, ps, spoil in want:
                exp += [("plcp",)] if spoil else [(mcs, 0, ps[0]), (mcs, 1, ps[1])]
            assert len(got) == len(exp), (cfo, ci, [(hex(r["error_code"]), r["rate_kbps"], r["stream"]) for r in got])
            ends = [r["end_sample"] for r in got]
            assert ends == sorted(ends)
            for r, e in zip(got, exp):
                if e[0] == "plcp":
                    assert r["error_code"] == 0x80000005, (cfo, ci, hex(r["error_code"]))
                else:
                    assert (r["error_code"], r["rate_kbps"], r["stream"], r["mpdu"]) == (1, e[0], e[1], e[2]), (cfo, ci, hex(r["error_code"]), r["rate_kbps"], r["stream"])
        rx.close()


def test_raw_captures_beyond_the_handles_capacity_are_reported(env):
    """The data field's tables are planned on the device (k_ht40_plan), so a batch that holds more frames than the handle was created
    for cannot be refused by the process call itself: sora_ht40_wait / _results_of report it (SORA_ERR_CAPACITY)."""
    torch, sora = env
    rng = np.random.default_rng(4242)
    specs = [[(9, 200, None)] for _ in range(6)]
    iq, descs, truth = _raw_captures(rng, specs, sigma=10.0)
    f0, f1 = torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda()
    rx = sora.RxHt40(3, 1 << 20)                                         # room for three frames; the captures hold six
    t = rx.process_captures_dev(f0, f1, descs, max_frames_per_capture=2)
    with pytest.raises(sora.SoraError):
        rx.wait(t)
    with pytest.raises(sora.SoraError):
        rx.results(ticket=t)
    rx.close()
    rx = sora.RxHt40(8, 1 << 20)                                         # ... and with room for them they are all decoded
    t = rx.process_captures_dev(f0, f1, descs, max_frames_per_capture=2)
    rx.wait(t)
    assert sum(r["error_code"] == 1 for r in rx.results(ticket=t)) >= 10
    rx.close()


def test_raw_capture_calls_in_flight_and_delivery(env):
    """tickets and sora_ht40_deliver_async with raw captures: three different batches in flight, every call collected by its ticket, the
    delivered tables equal to results_of row for row (decoded frames, headers that failed, the truncation flag)."""
    torch, sora = env
    rng = np.random.default_rng(4141)
    batches = []
    for b in range(4):
        specs = [[(8 + int(rng.integers(0, 7)), int(rng.integers(40, 600)), None)] for _ in range(10)]
        iq, descs, truth = _raw_captures(rng, specs, sigma=10.0)     # (below sigma ~ 6 the reference's carrier sense itself misfires on the frame's first samples: its integer energies are 0 / 1 there)
        batches.append((torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda(), descs, truth))
    rx = sora.RxHt40(16, 1 << 22)
    depth = rx.calls_in_flight()
    bufs = [sora.HostResults(16 * 2 * 2, 1 << 16) for _ in range(depth)]
    pend = []
    key = lambda r: (r["capture_id"], r["stream"], r["error_code"], r["rate_kbps"], r["end_sample"], r["length"], r["crc32"], r["mpdu"], r.get("flags", 0))
    checked = [0]

    def collect(t0, b0, tr):
        rx.wait(t0)
        got = b0.results(); ref = rx.results(ticket=t0)
        assert [key(r) for r in got] == [key(r) for r in ref]
        sent = {p for fr in tr for _, ps, _ in fr for p in ps}
        assert len(got) >= 18 and sum(r["error_code"] == 1 for r in got) >= 17 and all(r["mpdu"] in sent for r in got if r["error_code"] == 1)
        checked[0] += 1

    for k in range(depth + 5):
        f0, f1, descs, truth = batches[k % 4]
        t = rx.process_captures_dev(f0, f1, descs, max_frames_per_capture=2)
        rx.deliver_async(t, bufs[k % depth]); pend.append((t, bufs[k % depth], truth))
        if len(pend) >= depth:
            collect(*pend.pop(0))                                        # the oldest call in flight, while the newer ones run
    while pend:
        collect(*pend.pop(0))
    assert checked[0] == depth + 5
    for b in bufs:
        b.close()
    rx.synchronize(); rx.close()


def test_gpu_psdus_equal_the_independent_float64_receiver(env):
    """VERDICT r3 #9.  The 40 MHz extension has no reference implementation (parity stays UNPINNED); this is the cross-check that is not the
    author's own loop-back: oracle/ht40_rx_f64.py -- a float64 receiver written from IEEE 802.11n-2009 clause 20 with its own timing search,
    CFO estimate, channel estimates, MMSE detector, LLR demapper, de-interleaver and full-frame Viterbi; no import from the capture generator
    or the library -- decodes the SAME raw captures the HIP path decodes.  540 random frames over MCS 8..14, lengths 5..1500 bytes, 2x2
    cross-talk, carrier offsets, noise floors from 8 LSB to near the 64-QAM 3/4 threshold: every frame BOTH receivers record (the HIP front
    end is the reference's carrier sense, which loses a frame now and then -- DESIGN.md section 7 g1 -- the float receiver's is a matched
    filter) must carry the same HT-SIG, and wherever the float receiver's FCS is good the HIP path's PSDU must be byte-identical, both streams."""
    torch, sora = env
    from oracle import ht40_rx_f64 as rxf
    rng = np.random.default_rng(20261109)
    both = only_f64 = only_gpu = 0
    nframes = missed_by_gpu = missed_by_f64 = 0
    for batch in range(18):
        specs = []
        for c in range(10):
            specs.append([(8 + int(rng.integers(0, 7)), int(rng.choice([5, 20, 64, 150, 333, 700, 1100, 1500])) + 4 * c + 40 * k, None) for k in range(3)])
        sigma = float(rng.choice([8.0, 10.0, 12.0, 14.0])); cfo = float(rng.choice([0.0, 21.0, -33.0]))
        iq, descs, truth = _raw_captures(rng, specs, sigma=sigma, cfo=cfo)
        nsoft = 2 * sum(2 * (m.nsym_for([ln, ln], *m.MCS2[mcs]) * 108 * m.MCS2[mcs][0] + 64) for fr in specs for mcs, ln, _ in fr)
        rx = sora.RxHt40(len(specs) * 8, nsoft)
        t = rx.process_captures_dev(torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda(), descs, max_frames_per_capture=8)
        res = rx.results(ticket=t); rx.close()
        per = {}
        for r in res:
            if r["error_code"] in (0x1, 0x80000006):
                per.setdefault(r["capture_id"], {}).setdefault((r["rate_kbps"], r["length"]), {})[r["stream"]] = r
        for ci, (off, n, cid) in enumerate(descs):
            ref = {(f.mcs, f.length): f for f in rxf.receive(iq[:, off:off + n]) if f.sig_ok}
            got = per.get(cid, {})
            sent = {(mcs, ln) for mcs, ln, _ in specs[ci]}
            assert set(ref) <= sent and set(got) <= sent, (batch, ci, sorted(ref), sorted(got))      # neither receiver reads an HT-SIG that was not sent
            missed_by_gpu += len(set(ref) - set(got)); missed_by_f64 += len(set(got) - set(ref))
            for key in set(ref) & set(got):
                nframes += 1
                f = ref[key]
                for s in range(2):
                    r = got[key][s]
                    g_ok = r["error_code"] == 1
                    if f.fcs_ok[s] and g_ok:
                        assert r["mpdu"] == f.psdu[s], (batch, ci, key, s)
                        both += 1
                    elif f.fcs_ok[s]:
                        only_f64 += 1
                    elif g_ok:
                        only_gpu += 1
    print("frames decoded by both receivers: %d (x 2 streams); PSDUs byte-identical with both FCS good: %d; FCS good in the float64 receiver only: %d, on the GPU only: %d; "
          "frames only the float64 receiver found: %d, only the GPU: %d" % (nframes, both, only_f64, only_gpu, missed_by_gpu, missed_by_f64))
    assert nframes >= 500, (nframes, missed_by_gpu, missed_by_f64)
    assert missed_by_gpu <= 0.04 * 540 and missed_by_f64 <= 0.02 * 540, (missed_by_gpu, missed_by_f64)
    assert both >= 2 * nframes * 0.95 and only_f64 <= 2 * nframes * 0.04, (nframes, both, only_f64, only_gpu)


@pytest.mark.gpu
def test_raw_capture_completions_are_taken_as_they_happen(env):
    """sora_ht40_wait_any: eight raw-capture calls in flight, taken in the order they finish; every ticket exactly once, its delivered table equal to
    results_of, the released slot reused by the next call; nothing pending -> refused."""
    torch, sora = env
    rng = np.random.default_rng(5151)
    batches = []
    for b in range(3):
        specs = [[(8 + int(rng.integers(0, 7)), int(rng.integers(40, 400)), None)] for _ in range(6 + 2 * b)]
        iq, descs, truth = _raw_captures(rng, specs, sigma=10.0)
        batches.append((torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda(), descs))
    rx = sora.RxHt40(16, 1 << 22)
    with pytest.raises(sora.SoraError):
        rx.wait_any()
    depth = rx.calls_in_flight()
    key = lambda r: (r["capture_id"], r["stream"], r["error_code"], r["rate_kbps"], r["end_sample"], r["length"], r["crc32"], r["mpdu"])
    free = [sora.HostResults(16 * 2 * 2, 1 << 16) for _ in range(depth)]
    held = {}; seen = []; k = 0

    def submit():
        nonlocal k
        f0, f1, descs = batches[k % 3]
        t = rx.process_captures_dev(f0, f1, descs, max_frames_per_capture=2); held[t] = free.pop(); k += 1
        rx.deliver_async(t, held[t])

    def take():
        t = rx.wait_any(); seen.append(t)
        buf = held.pop(t)
        got = buf.results(); ref = rx.results(ticket=t)                  # (the slot is released, but nothing has reused it yet)
        assert [key(r) for r in got] == [key(r) for r in ref] and len(got) >= 10, t
        free.append(buf)
    for _ in range(depth):
        submit()
    for _ in range(20):
        take(); submit()
    while held:
        take()
    assert sorted(seen) == list(range(1, k + 1))
    with pytest.raises(sora.SoraError):
        rx.wait_any()
    for b in free:
        b.close()
    rx.synchronize(); rx.close()


@pytest.mark.gpu
def test_raw_capture_delivery_carries_failed_headers_and_truncation(env):
    """sora_ht40_deliver_async of a raw-capture call delivers the table sora_ht40_results_of reports: a frame whose SIG field is spoiled is ONE row with
    E_ERROR_PLCP_HEADER_FAIL between its neighbours' rows, and a capture with more frames than max_frames_per_capture ends in rows flagged SORA_ROW_TRUNCATED."""
    torch, sora = env
    rng = np.random.default_rng(6161)
    specs = [[(9, 150, None), (10, 300, "sig"), (12, 333, None)], [(11, 80, None)], [(13, 200, None), (8, 90, None), (14, 400, None)], []]
    iq, descs, truth = _raw_captures(rng, specs, sigma=8.0)
    f0, f1 = torch.from_numpy(iq[0].copy()).cuda(), torch.from_numpy(iq[1].copy()).cuda()
    key = lambda r: (r["capture_id"], r["stream"], r["error_code"], r["rate_kbps"], r["end_sample"], r["length"], r["crc32"], r["flags"], r["mpdu"])
    for mf in (4, 2):
        rx = sora.RxHt40(16, 1 << 21)
        buf = sora.HostResults(len(specs) * mf * 2, 1 << 16)
        t = rx.process_captures_dev(f0, f1, descs, max_frames_per_capture=mf)
        rx.deliver_async(t, buf); rx.wait(t)
        got = buf.results(); ref = rx.results(ticket=t)
        assert [key(r) for r in got] == [key(r) for r in ref], mf
        codes = [(r["capture_id"], r["error_code"]) for r in got]
        if mf == 4:
            assert codes.count((100, 0x80000005)) == 1 and sum(r["error_code"] == 1 for r in got) == 2 * 6 and all(r["flags"] == 0 for r in got)
            i = codes.index((100, 0x80000005))
            assert got[i - 1]["capture_id"] == 100 and got[i + 1]["capture_id"] == 100 and got[i]["mpdu"] == b""   # between its neighbours, in time order
        else:
            flagged = [r for r in got if r["flags"] & 1]
            assert {r["capture_id"] for r in flagged} == {100, 102} and all(r["flags"] == 0 for r in got if r["capture_id"] == 101)
        buf.close(); rx.close()
