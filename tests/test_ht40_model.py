"""CPU checks of the independent numpy model of the 40 MHz HT data field (oracle/py_ht40.py; test infrastructure for sora_ht40_*, parity
unpinned -- the reference has no such receiver): the interleaver is a permutation with the standard's structure, puncturing keeps the
reference's patterns, and the model's own floating-point receiver recovers the constellation points its transmitter sent."""
import numpy as np

from oracle import py_ht40 as m


def test_carrier_plan_and_interleaver():
    assert len(m.DATA_CARRIERS) == 108 and all(k not in m.PILOTS and k not in (-1, 0, 1) for k in m.DATA_CARRIERS)
    assert int(np.count_nonzero(m.HTLTF40)) == 114
    for nb in (1, 2, 4, 6):
        for iss in (0, 1):
            r = m.interleave_map(nb, iss)
            assert sorted(r.tolist()) == list(range(108 * nb))
        # stream 1 is stream 0 rotated by 2 N_ROT N_BPSC; adjacent coded bits go to different carriers
        r0, r1 = m.interleave_map(nb, 0), m.interleave_map(nb, 1)
        assert np.array_equal(r1, (r0 - 2 * m.N_ROT * nb) % (108 * nb))
        assert all(int(r0[k]) // nb != int(r0[k + 1]) // nb for k in range(108 * nb - 1))


def test_symbol_counts_and_puncturing():
    assert [m.ndbps(nb, cr) for nb, cr in ((1, 0), (2, 0), (2, 2), (4, 0), (4, 2), (6, 1), (6, 2))] == [54, 108, 162, 216, 324, 432, 486]
    a = np.arange(12, dtype=np.uint8); b = a + 100
    assert m.puncture(a, b, 2).tolist() == [0, 100, 1, 102, 3, 103, 4, 105, 6, 106, 7, 108, 9, 109, 10, 111]     # (A0 B0) (A1) (B2), viterbi.hpp:173-187
    assert m.puncture(a, b, 1).tolist() == [0, 100, 1, 2, 102, 3, 4, 104, 5, 6, 106, 7, 8, 108, 9, 10, 110, 11]


def test_float_receiver_recovers_the_transmitted_points():
    rng = np.random.default_rng(3)
    for nb, cr in ((2, 0), (6, 2)):
        ps = [m.add_fcs(rng.integers(0, 256, 200, dtype=np.uint8).tobytes()) for _ in range(2)]
        x, nsym = m.tx(ps, nb, cr)
        H = np.array([[1.0, 0.3j], [0.2, 0.8 * np.exp(1j)]])
        iq = m.channel(x, H, 0.0, rng, lead=64)
        Y = m.rx_symbols(iq, 64, nsym)
        W = m.mmse_weights(Y, 0.0)
        for s in range(2):
            bits = m.stream_bits(ps[s], nsym, nb, cr, (0x5D, 0x2B)[s])
            a, b = m.encode(bits); coded = m.puncture(a, b, cr)
            imap = m.interleave_map(nb, s)
            for d in (0, nsym - 1):
                blk = coded[d * 108 * nb:(d + 1) * 108 * nb]; il = np.zeros(108 * nb, np.uint8); il[imap] = blk
                want = m.qam(il.astype(float), nb)
                got = np.array([(W[k % 128] @ Y[:, 2 + d, k % 128])[s] for k in m.DATA_CARRIERS]) * 128.0
                assert np.abs(got - want).max() < 2.0, (nb, cr, s, d)
