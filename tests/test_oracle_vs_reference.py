"""Live comparison of the C restatement with the reference's own SSE kernels (oracle/_ref/libsora_ref.so).
Skipped where the reference build is unavailable; tests/test_oracle_golden.py holds the committed vectors."""
import numpy as np
import pytest

from oracle.pyoracle import CR_12, CR_23, CR_34, RATES, rate_params


def test_luts_equal_reference_headers(oracle, reference):
    assert np.array_equal(oracle.usin_lut(), reference.lut("usin"))      # core/inc/intalglut.h:4
    assert np.array_equal(oracle.ucos_lut(), reference.lut("ucos"))      # :3648
    assert np.array_equal(oracle.uatan2_lut(), reference.lut("uatan2"))  # :7332


@pytest.mark.parametrize("n", [64, 128])
@pytest.mark.parametrize("inverse", [False, True])
def test_fft_random(oracle, reference, n, inverse):
    rng = np.random.default_rng(n + inverse)
    for it in range(400):
        amp = [32767, 20000, 8000, 500, 30][it % 5]
        x = rng.integers(-amp, amp + 1, size=(n, 2)).astype(np.int16)
        if it % 7 == 0:
            x[rng.integers(0, n)] = (-32768, 32767)
        assert np.array_equal(oracle.fft(x, n, inverse), reference.fft(x, n, inverse))


def test_uatan2_random(oracle, reference):
    rng = np.random.default_rng(3)
    for it in range(20000):
        sc = [100, 3000, 100000, 2 ** 30][it % 4]
        y = int(rng.integers(-sc, sc)); x = int(rng.integers(-sc, sc))
        assert oracle.L.so_uatan2(y, x) == reference.L.ref_uatan2(y, x)


def test_demap_random(oracle, reference):
    rng = np.random.default_rng(4)
    for it in range(100):
        x = rng.integers(-3000, 3000, size=(64, 2)).astype(np.int16)
        for nb in (1, 2, 4, 6):
            assert np.array_equal(oracle.demap(nb, x), reference.demap(nb, x))


def test_viterbi_sig_random(oracle, reference):
    rng = np.random.default_rng(5)
    for it in range(300):
        s = rng.integers(0, 8, size=48).astype(np.uint8)
        assert oracle.viterbi_sig(s) == reference.viterbi_sig(s)


@pytest.mark.parametrize("rate", RATES)
def test_viterbi_on_real_noisy_frames(oracle, reference, rate):
    """Soft values of a real (noisy) frame, as the oracle's demap/deinterleave produced them, decoded by the
    reference's TViterbiCore with the T11aViterbi<5000*8,48,256,24> schedule (viterbi.hpp:148-235)."""
    rng = np.random.default_rng(rate)
    nb, cr, _ = rate_params(rate)
    mp = rng.integers(0, 256, 700).astype(np.uint8).tobytes()
    cap = oracle.tx_capture(mp, rate).astype(np.int32)
    sigma = {1: 2500, 2: 1800, 4: 900, 6: 450}[nb]
    cap += np.rint(rng.normal(0, sigma, cap.shape)).astype(np.int32)
    cap = np.clip(cap, -32768, 32767).astype(np.int16)
    res, tr = oracle.rx_capture(cap, 40, trace=True)
    assert len(res) == 1 and res[0]["rate_kbps"] == rate
    soft = tr["soft"]
    assert len(soft) == res[0]["nsym"] * 48 * nb
    assert (soft != 0).any() and (soft != 7).any()
    got = oracle.viterbi_frame(soft, cr, res[0]["length"])
    want = reference.viterbi_frame(soft, cr, res[0]["length"])
    assert np.array_equal(got, want)
    assert np.array_equal(got, tr["decoded"][:len(got)])


def test_crc32(oracle, reference):
    rng = np.random.default_rng(6)
    for n in (0, 1, 5, 100, 1500):
        b = rng.integers(0, 256, n).astype(np.uint8).tobytes()
        assert oracle.crc32(b) == reference.crc32(b)


def test_one_multiply_bricks_in_numpy_equal_the_reference_primitives(reference):
    """The numpy restatement tests/test_gpu_stages.py checks the stand-alone TFreqCompensation / TChannelEqualization / TPhaseCompensate stages with (on the GPU box)
    against the reference's own primitives in the bricks' order (oracle/ref_shim.cpp, channel_11a.hpp:548-574,642-644, freqoffset.hpp:28-30), here, where the
    reference library is freshly built: 300 random symbols and the corner values."""
    import os
    import importlib.util
    if not hasattr(reference.L, "ref_freq_comp64"):
        pytest.skip("oracle/_ref/libsora_ref.so predates the brick shims")
    spec = importlib.util.spec_from_file_location("tgs", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_stages.py"))
    tgs = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgs)
    rng = np.random.default_rng(1)
    for it in range(300):
        x = rng.integers(-32768, 32768, (64, 2)).astype(np.int16); c = rng.integers(-32768, 32768, (64, 2)).astype(np.int16)
        if it == 0: x[:] = -32768; c[:] = -32768
        if it == 1: x[:] = 32767; c[:] = -32768
        for which in ("freq_comp", "channel_equalize", "phase_comp"):
            a = x.astype(np.int64) >> 1 if which == "freq_comp" else x.astype(np.int64)
            re, im, w16 = tgs._mul32(a, c.astype(np.int64)); sh = 8 if which == "channel_equalize" else 15
            want = np.stack([w16(re >> sh), w16(im >> sh)], -1).astype(np.int16)
            if which == "channel_equalize":
                want[28:36] = 0
            assert np.array_equal(reference.brick64(which, x, c), want), (which, it)
