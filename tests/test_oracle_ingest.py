"""Capture ingest: the oracle's restatement against the reference's own resampler
(committed vectors + live when oracle/_ref exists), and the output-size arithmetic of the C ABI (no GPU needed)."""
import numpy as np
import pytest


def make_dump(iq40_i16, raw14=True, seed=0):
    """int16 [n,2] stream -> bytes of a Sora RX_BLOCK dump (16-byte descriptor + 28 samples per 128-byte block)"""
    rng = np.random.default_rng(seed)
    x = np.ascontiguousarray(iq40_i16, np.int16).reshape(-1, 2)
    nb = (len(x) + 27) // 28
    pad = np.zeros((nb * 28, 2), np.int16); pad[:len(x)] = x
    if raw14:
        pad = ((pad.astype(np.uint16) >> 2) & 0x3FFF).astype(np.uint16).view(np.int16)
    blocks = np.zeros((nb, 128), np.uint8)
    blocks[:, :16] = rng.integers(0, 256, size=(nb, 16), dtype=np.uint8)      # descriptors: content must not matter
    blocks[:, 16:] = pad.reshape(nb, 28 * 2).view(np.uint8).reshape(nb, 112)
    return blocks.reshape(-1)


def test_down44to40_matches_reference_vectors(oracle, golden_dir):
    import os
    v = np.load(os.path.join(golden_dir, "ref_vectors.npz"))
    got = oracle.down44to40(v["down44_in"])
    assert np.array_equal(got, v["down44_out"])


def test_down44to40_matches_live_reference(oracle, reference):
    if not reference.available():
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(44)
    for n in (28, 56, 28 * 11, 28 * 40 + 13, 28 * 131):
        x = rng.integers(-32768, 32768, size=(n, 2)).astype(np.int16)
        assert np.array_equal(oracle.down44to40(x), reference.down44to40(x)), n


def test_down44to40_closed_form(oracle):
    """out[10p] = x[11p]; out[10p+k] = (x[11p+k] R[k] + x[11p+k+1] L[k+1]) >> 7 -- what the GPU kernel computes per thread"""
    R = np.array([1, 115, 102, 90, 77, 64, 51, 38, 26, 13, 0]); L = np.array([0, 0, 13, 26, 38, 51, 64, 77, 90, 102, 115])
    rng = np.random.default_rng(3)
    x = rng.integers(-32768, 32768, size=(28 * 77, 2)).astype(np.int16)
    want = oracle.down44to40(x)
    m = np.arange(len(want)); p, k = m // 10, m % 10
    xi = x.astype(np.int64)
    interp = (xi[11 * p + k] * R[k][:, None] + xi[np.minimum(11 * p + k + 1, len(x) - 1)] * L[np.minimum(k + 1, 10)][:, None]) >> 7
    got = np.where((k == 0)[:, None], xi[11 * p], interp).astype(np.int16)
    assert np.array_equal(got, want)


def test_load_dump_and_downsample2(oracle):
    rng = np.random.default_rng(9)
    x = (rng.integers(-8192, 8192, size=(28 * 9 + 5, 2)) * 4).astype(np.int16)     # representable in 14 bits << 2
    raw = make_dump(x, raw14=True)
    got = oracle.load_dump(raw.tobytes(), raw14=True)
    assert np.array_equal(got[:len(x)], x) and len(got) == 28 * 10
    assert np.array_equal(oracle.downsample2(got[:24]), got[:24:2])


def test_ingest_count_matches_oracle(oracle):
    import sora_amd
    from sora_amd import INGEST_RXBLOCK as FB, INGEST_44TO40 as F4, INGEST_DECIMATE2 as F2
    rng = np.random.default_rng(1)
    for nbytes in (0, 16, 20, 127, 128, 128 * 3 + 16 + 8, 128 * 50, 128 * 977 + 60):
        raw = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        s = oracle.load_dump(raw.tobytes()) if nbytes else np.zeros((0, 2), np.int16)
        assert sora_amd.ingest_count(nbytes, FB) == len(s), nbytes
        r = oracle.down44to40(s)
        assert sora_amd.ingest_count(nbytes, FB | F4) == len(r), nbytes
        assert sora_amd.ingest_count(nbytes, FB | F4 | F2) == len(oracle.downsample2(r)), nbytes
        assert sora_amd.ingest_count(nbytes, FB | F2) == len(oracle.downsample2(s)), nbytes
