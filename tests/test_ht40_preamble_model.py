"""The HT-mixed 40 MHz preamble of oracle/py_ht40.py (tx_frame) is PINNED on the CPU: the legacy preamble and HT-SIG it generates, read the way
the GPU front end reads them (the even samples of x[n] j^n, py_ht40.front_end_view), go through the restated REFERENCE receiver of the
802.11n graph (oracle/so_rx11n.c, itself pinned to the compiled reference graph) -- TCCA11n finds the frame, T11nSigParser accepts L-SIG and
HT-SIG and reports the MCS and LENGTH that were sent.  (MCS 8..10 and lengths <= 1500 only: what the reference's parser lets through,
PHY_11n.hpp:497-505; the data field behind it is 40 MHz wide, so that receiver then reports a CRC failure -- expected.)"""
import numpy as np
import pytest

from oracle import py_ht40 as m


def _capture(rng, mcs, length, sigma, cfo_step=0.0, lead=400):
    ps = [m.add_fcs(rng.integers(0, 256, length - 4, dtype=np.uint8).tobytes()) for _ in range(2)]
    x, nsym, pre = m.tx_frame(ps, mcs)
    y = m.channel(x, [[1.0, 0.2j], [0.15, 0.9]], sigma, rng, cfo_step=cfo_step, lead=lead)
    y = np.concatenate([y, np.zeros((2, 2000, 2), np.int16)], axis=1)
    n = y.shape[1] // 28 * 28
    return y[:, :n], ps, nsym, pre


@pytest.mark.parametrize("mcs", [8, 9, 10])
def test_reference_front_end_parses_the_model_preamble(oracle, mcs):
    rng = np.random.default_rng(100 + mcs)
    for length, sigma, cfo in ((60, 0.0, 0.0), (200, 20.0, 0.0), (1500, 20.0, 25.0), (333, 40.0, -18.0)):
        y, _, nsym, pre = _capture(rng, mcs, length, sigma, cfo)
        ev = oracle.rx11n_capture(m.front_end_view(y[0]), m.front_end_view(y[1]))
        assert len(ev) == 1, (mcs, length, ev)
        e = ev[0]
        assert e["error_code"] != 0x80000005 and e["rate_kbps"] == mcs and e["length"] == length, (mcs, length, hex(e["error_code"]), e["rate_kbps"], e["length"])


def test_without_the_shift_the_header_does_not_parse(oracle):
    """the plain even samples (no x j^n): the two halves of the channel alias onto each other out of phase -- the header fails; so the
    test above really exercises the front-end view"""
    rng = np.random.default_rng(5)
    y, _, _, _ = _capture(rng, 9, 200, 20.0)
    ev = oracle.rx11n_capture(y[0], y[1])
    assert all(e["error_code"] == 0x80000005 for e in ev)


def test_ht_sig_crc_and_fields():
    b = m.ht_sig_bits(13, 1234)
    assert sum(int(b[i]) << i for i in range(7)) == 13 and b[7] == 1 and sum(int(b[8 + i]) << i for i in range(16)) == 1234
    crc = 0xFF
    for i in range(34):
        crc ^= int(b[i]); crc = (crc >> 1) ^ 0xE0 if crc & 1 else crc >> 1
    assert sum(int(b[34 + i]) << i for i in range(8)) == (~crc) & 0xFF and not b[42:].any()
