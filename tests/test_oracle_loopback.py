"""TX -> channel -> RX loop-back through the oracle (the reference's own test strategy: demod11 -m / -c / -d,
SURVEY.md section 4), every 802.11a rate, clean and impaired, single- and multi-frame captures."""
import numpy as np
import pytest

from oracle.pyoracle import E_CRC32_FAIL, E_FRAME_OK, RATES


def awgn(cap, sigma, seed):
    rng = np.random.default_rng(seed)
    x = cap.astype(np.int32) + np.rint(rng.normal(0, sigma, cap.shape)).astype(np.int32)
    return np.clip(x, -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("rate", RATES)
@pytest.mark.parametrize("length", [10, 100, 1500])
def test_clean_loopback(oracle, rate, length):
    rng = np.random.default_rng(rate + length)
    mp = rng.integers(0, 256, length).astype(np.uint8).tobytes()
    res = oracle.rx_capture(oracle.tx_capture(mp, rate), 40)
    assert len(res) == 1
    r = res[0]
    assert r["error_code"] == E_FRAME_OK and r["rate_kbps"] == rate and r["length"] == length + 4
    assert r["mpdu"][:-4] == mp


def test_max_length_frame(oracle):
    mp = bytes(range(256)) * 9 + bytes(192)          # 2496 + FCS = 2500 = MTU (PHY_11a.hpp:570-572)
    res = oracle.rx_capture(oracle.tx_capture(mp, 54000), 40)
    assert len(res) == 1 and res[0]["error_code"] == E_FRAME_OK and res[0]["mpdu"][:-4] == mp


def test_over_mtu_is_plcp_failure(oracle):
    mp = bytes(2497)
    res = oracle.rx_capture(oracle.tx_capture(mp, 54000), 40)
    assert res and res[0]["error_code"] == 0x80000005


@pytest.mark.parametrize("rate,sigma", [(6000, 2500), (24000, 700), (54000, 250)])
def test_noisy_loopback_decodes(oracle, rate, sigma):
    rng = np.random.default_rng(rate)
    mp = rng.integers(0, 256, 500).astype(np.uint8).tobytes()
    res = oracle.rx_capture(awgn(oracle.tx_capture(mp, rate), sigma, rate), 40)
    assert len(res) == 1 and res[0]["error_code"] == E_FRAME_OK and res[0]["mpdu"][:-4] == mp


def test_heavy_noise_gives_crc_failure(oracle):
    rng = np.random.default_rng(9)
    mp = rng.integers(0, 256, 800).astype(np.uint8).tobytes()
    res = oracle.rx_capture(awgn(oracle.tx_capture(mp, 54000), 900, 9), 40)
    assert len(res) == 1 and res[0]["error_code"] == E_CRC32_FAIL and res[0]["length"] == 804


def test_empty_and_silent_captures(oracle):
    assert oracle.rx_capture(np.zeros((0, 2), np.int16), 40) == []
    assert oracle.rx_capture(np.zeros((10000, 2), np.int16), 40) == []
    assert oracle.rx_capture(np.zeros((3, 2), np.int16), 20) == []


def test_truncated_frame_yields_nothing(oracle):
    mp = bytes(1000)
    cap = oracle.tx_capture(mp, 12000)
    assert oracle.rx_capture(cap[:len(cap) // 2], 40) == []


def test_multi_frame_capture(oracle):
    """Several frames of different rates back to back: per-frame Reset, carrier sense resumes after each."""
    rng = np.random.default_rng(11)
    parts, want = [], []
    for i, rate in enumerate((54000, 6000, 36000, 48000)):
        mp = rng.integers(0, 256, 200 + 100 * i).astype(np.uint8).tobytes()
        parts.append(oracle.tx_capture(mp, rate, lead=0, tail=400 + 52 * i)); want.append((rate, mp))
    res = oracle.rx_capture(np.concatenate(parts), 40)
    assert [(r["rate_kbps"], r["mpdu"][:-4]) for r in res] == want
    assert all(r["error_code"] == E_FRAME_OK for r in res)
    starts = [r["start_sample"] for r in res]
    assert starts == sorted(starts)


def test_20mhz_equals_even_samples_of_40mhz(oracle):
    rng = np.random.default_rng(12)
    mp = rng.integers(0, 256, 300).astype(np.uint8).tobytes()
    cap = awgn(oracle.tx_capture(mp, 48000, lead=36), 200, 3)
    a = oracle.rx_capture(cap, 40); b = oracle.rx_capture(cap[::2].copy(), 20)
    assert a == b and a[0]["error_code"] == E_FRAME_OK


def test_cfo_is_tracked(oracle):
    """+-40 kHz carrier offset on the 40 MHz stream: T11aLTS estimates it, pilot tracking follows the rest."""
    rng = np.random.default_rng(13)
    mp = rng.integers(0, 256, 400).astype(np.uint8).tobytes()
    cap = oracle.tx_capture(mp, 24000).astype(np.float64)
    n = np.arange(len(cap))
    for df in (40e3, -40e3):
        ph = np.exp(2j * np.pi * df * n / 40e6)
        z = (cap[:, 0] + 1j * cap[:, 1]) * ph
        x = np.stack([np.rint(z.real), np.rint(z.imag)], 1).astype(np.int16)
        res = oracle.rx_capture(x, 40)
        assert len(res) == 1 and res[0]["error_code"] == E_FRAME_OK and res[0]["mpdu"][:-4] == mp
        assert np.sign(res[0]["cfo_est"]) == np.sign(df)
