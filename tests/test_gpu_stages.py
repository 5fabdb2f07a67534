"""GPU parity of the per-stage entry points (the brick-by-brick boundary) against the oracle's stage functions:
T11aLTS, the symbol front end, TPhaseCompensate+TPilotTrack, FFT<128>.  0 LSB everywhere."""
import ctypes

import numpy as np
import pytest

from gpu_util import awgn
from oracle.pyoracle import RATES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import sora_amd
    sora_amd.load()
    assert torch.cuda.is_available()
    return torch, sora_amd


def frames_for_stage_tests(oracle, n=6):
    """Noisy captures -> (capture@20MHz, frame start, nsym) using the oracle's own receiver for the timing."""
    out = []
    for i in range(n):
        rate = RATES[i % 8]
        rng = np.random.default_rng(500 + i)
        mp = rng.integers(0, 256, 150 + 40 * i).astype(np.uint8).tobytes()
        cap = awgn(oracle.tx_capture(mp, rate, lead=8 * i), 250, i)[::2].copy()
        z = (cap[:, 0].astype(np.float64) + 1j * cap[:, 1]) * np.exp(2j * np.pi * (i - 2) * 15e3 * np.arange(len(cap)) / 20e6)
        cap = np.stack([np.rint(z.real), np.rint(z.imag)], 1).astype(np.int16)
        res = oracle.rx_capture(cap, 20)
        if len(res) == 1 and res[0]["length"]:
            out.append((cap, res[0]["start_sample"], res[0]["nsym"]))
    assert len(out) >= 4
    return out


def test_lts_symfront_track_chain(env, oracle):
    torch, sora = env
    frames = frames_for_stage_tests(oracle)
    lts_in = np.stack([c[s:s + 144] for c, s, _ in frames])
    ctx_gpu = sora.lts11a(torch.from_numpy(lts_in).cuda())
    ctx_h = ctx_gpu.cpu().numpy()
    # ---- T11aLTS
    octx = []
    for i, (c, s, _) in enumerate(frames):
        k = oracle.new_ctx(); oracle.lts(k, c[s:s + 144]); octx.append(k)
        assert ctx_h[i, 0] == k.CFO_est
        assert np.array_equal(ctx_h[i, 2:130], np.array(k.FreqCoeffs[:], np.int16))
        assert np.array_equal(ctx_h[i, 130:258], np.array(k.ChannelCoeffs[:], np.int16))
    assert any(k.CFO_est != 0 for k in octx)
    # ---- symbol front end over every symbol (SIGNAL + data) of every frame
    syms, idx, first, nsym = [], [], [], []
    for i, (c, s, ns) in enumerate(frames):
        first.append(len(syms)); nsym.append(ns + 1)
        for k in range(ns + 1):
            p = s + 144 + 80 * k
            syms.append(c[p:p + 80]); idx.append(i)
    x = torch.from_numpy(np.stack(syms)).cuda()
    eq = sora.symfront11a(x, ctx_gpu, torch.tensor(idx, dtype=torch.int32).cuda())
    eq_h = eq.cpu().numpy()
    want_eq = np.stack([oracle.sym_front(octx[i], sy) for sy, i in zip(syms, idx)])
    assert np.array_equal(eq_h, want_eq)
    # ---- pilot tracking, frame by frame, from the reset state (CompCoeffs = 0x7fff, symbol_count = 127)
    st = np.zeros((len(frames), 134), np.int16)
    st[:, 4] = 127                                                     # symbol_count (uint32 at int16 index 4..5)
    st[:, 6::2] = 0x7fff                                               # comp[k].re
    st_d = torch.from_numpy(st).cuda()
    trk = sora.pilot_track11a(eq, torch.tensor(first, dtype=torch.int32).cuda(), torch.tensor(nsym, dtype=torch.int32).cuda(), st_d)
    trk_h = trk.cpu().numpy(); st_h = st_d.cpu().numpy()
    used = [b for b in range(64) if (1 <= b <= 26 or b >= 38)]
    for i in range(len(frames)):
        k = octx[i]
        for s_i in range(nsym[i]):
            w = oracle.sym_track(k, want_eq[first[i] + s_i])
            assert np.array_equal(trk_h[first[i] + s_i][used], w[used]), (i, s_i)
        assert (st_h[i, 0], st_h[i, 1], st_h[i, 2], st_h[i, 3]) == (k.CFO_comp, k.SFO_comp, k.CFO_tracker, k.SFO_tracker)
        assert int(st_h[i, 4:6].view(np.uint32)[0]) == k.symbol_count
        assert np.array_equal(st_h[i, 6:].reshape(64, 2)[used], np.array(k.CompCoeffs[:], np.int16).reshape(64, 2)[used])

    # ---- TPhaseCompensate and TPilotTrack as two bricks (sora_hip_phase_comp11a -> sora_hip_pilot11a), symbol by symbol as the graph runs them: the same outputs and
    # the same state as the fused entry point left
    st2 = torch.from_numpy(st).cuda()
    one_first = torch.zeros(1, dtype=torch.int32).cuda(); one_n = torch.ones(1, dtype=torch.int32).cuda()
    for i in range(min(2, len(frames))):
        for s_i in range(min(nsym[i], 12)):
            pcs = sora.phase_comp11a(eq[first[i] + s_i:first[i] + s_i + 1], st2[i:i + 1])
            o2 = sora.pilot11a(pcs, one_first, one_n, st2[i:i + 1]).cpu().numpy()[0]
            assert np.array_equal(o2[used], trk_h[first[i] + s_i][used]), (i, s_i)
        if nsym[i] <= 12:
            assert np.array_equal(st2[i].cpu().numpy(), st_h[i])


def test_fft128_bit_exact(env, oracle):
    torch, sora = env
    rng = np.random.default_rng(9)
    n = 301
    amps = np.array([32767, 20000, 8000, 500, 30])[np.arange(n) % 5]
    x = (rng.integers(-32768, 32768, size=(n, 128, 2)) % (2 * amps[:, None, None] + 1) - amps[:, None, None]).astype(np.int16)
    x[5, 9] = (-32768, 32767); x[6] = 32767; x[7] = -32768
    got = sora.fft128(torch.from_numpy(x).cuda()).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], oracle.fft(x[i], 128)), i


def test_fft128_matches_reference_vectors(env, golden_dir):
    """The committed vectors produced by the reference's own FFT<128> (tests/golden/ref_vectors.npz)."""
    import os
    torch, sora = env
    v = np.load(os.path.join(golden_dir, "ref_vectors.npz"))
    got = sora.fft128(torch.from_numpy(v["fft128_in"]).cuda()).cpu().numpy()
    assert np.array_equal(got, v["fft128_out"])
    got64 = sora.fft64(torch.from_numpy(v["fft64_in"]).cuda()).cpu().numpy()
    assert np.array_equal(got64, v["fft64_out"])


def _mul32(a, b):
    """vector128.h:1075-1081 on int64 arrays [...,2]: 32-bit wrapping parts of a x b with the im of b negated by _mm_sign_epi16 (wrapping)"""
    w16 = lambda v: ((v + 32768) & 0xFFFF) - 32768          # noqa: E731
    w32 = lambda v: ((v + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)  # noqa: E731
    re = w32(a[..., 0] * b[..., 0] + a[..., 1] * w16(-b[..., 1])); im = w32(a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0])
    return re, im, w16


def test_freq_comp_equalize_phase_comp_alone(env, oracle):
    """TFreqCompensation, TChannelEqualization and TPhaseCompensate as stand-alone stages (channel_11a.hpp:534-653, freqoffset.hpp:16-66): 4099 random
    symbols with the corner values, several coefficient sets picked per symbol, against the vector128 arithmetic restated in numpy AND -- where oracle/_ref is
    present -- against the reference's own primitives called in the bricks' order (oracle/ref_shim.cpp: ref_freq_comp64, ref_channel_equalize64, ref_phase_comp64)."""
    from oracle.pyoracle import Reference
    torch, sora = env
    rng = np.random.default_rng(44)
    n, m = 4099, 7
    amps = np.array([32767, 20000, 8000, 500, 30])[np.arange(n) % 5]
    x = (rng.integers(-32768, 32768, size=(n, 64, 2)) % (2 * amps[:, None, None] + 1) - amps[:, None, None]).astype(np.int16)
    x[7, 3] = (-32768, 32767); x[8] = 0; x[9] = 32767; x[10] = -32768
    ctx = rng.integers(-32768, 32768, size=(m, 258)).astype(np.int16)           # sora_lts11a_ctx: cfo_est, reserved, freq[64], chan[64]
    ctx[0, 2:] = 32767; ctx[1, 2:] = -32768
    st = rng.integers(-32768, 32768, size=(m, 134)).astype(np.int16)            # sora_track11a_state: 4 x int16, symbol_count, comp[64]
    st[0, 6:] = -32768
    idx = rng.integers(0, m, n).astype(np.int32); idx[:m] = np.arange(m)
    xd = torch.from_numpy(x).cuda(); id_ = torch.from_numpy(idx).cuda()
    got = {"freq_comp": sora.freq_comp11a(xd, torch.from_numpy(ctx).cuda(), id_).cpu().numpy(),
           "channel_equalize": sora.equalize11a(xd, torch.from_numpy(ctx).cuda(), id_).cpu().numpy(),
           "phase_comp": sora.phase_comp11a(xd, torch.from_numpy(st).cuda(), id_).cpu().numpy()}
    coef = {"freq_comp": ctx[:, 2:130].reshape(m, 64, 2), "channel_equalize": ctx[:, 130:258].reshape(m, 64, 2), "phase_comp": st[:, 6:].reshape(m, 64, 2)}
    X = x.astype(np.int64)
    for which in got:
        c = coef[which][idx].astype(np.int64)
        a = X >> 1 if which == "freq_comp" else X
        re, im, w16 = _mul32(a, c)
        sh = 8 if which == "channel_equalize" else 15
        want = np.stack([w16(re >> sh), w16(im >> sh)], -1).astype(np.int16)
        if which == "channel_equalize":
            want[:, 28:36] = 0
        assert np.array_equal(got[which], want), which
    # index NULL = set 0 for every symbol
    assert np.array_equal(sora.equalize11a(xd[:100], torch.from_numpy(ctx).cuda()).cpu().numpy(), sora.equalize11a(xd[:100], torch.from_numpy(ctx).cuda(), torch.zeros(100, dtype=torch.int32).cuda()).cpu().numpy())
    ref = Reference()
    if ref.available() and hasattr(ref.L, "ref_freq_comp64"):
        for i in list(range(12)) + list(range(n - 200, n)):
            for which in got:
                assert np.array_equal(got[which][i], ref.brick64(which, x[i], coef[which][idx[i]])), (which, i)
