"""The lane-level model of the 16-lanes-per-frame-pair trellis kernel (tools/emu_trellis16.py, the specification k_vit16.hip was
written against) against the oracle's T11aViterbi: the coset layout's algebra, the role bits, marks, banking, normalisation, the
window schedule of both graphs (256 / 24 and 192 / 36) and the trace-back -- on a CPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import emu_trellis16 as emu  # noqa: E402


def _soft(rng, cr, L):
    per = {0: 2, 1: 3, 2: 4}[cr]
    steps = L * 8 + 16 + 6 + 40
    nsoft = int(np.ceil(steps * per / {0: 1, 1: 2, 2: 3}[cr] / 48.0)) * 48
    nsoft = (nsoft + per * 4 - 1) // (per * 4) * (per * 4)
    s = rng.integers(0, 8, size=nsoft).astype(np.uint8)
    h = nsoft // 2
    s[:h] = np.clip(rng.choice([0, 7], size=h) + rng.integers(-3, 4, size=h), 0, 7)       # a noisy codeword-like half, then pure noise (metrics wrap)
    return s


def test_layout_algebra():
    assert emu.self_check()


@pytest.mark.parametrize("cr", [0, 1, 2])
def test_model_equals_the_oracle(oracle, cr):
    rng = np.random.default_rng(900 + cr)
    for LA, LB in ((33, 70), (1, 4), (64, None), (150, 150)):
        sA = _soft(rng, cr, LA); sB = _soft(rng, cr, LB) if LB is not None else None
        gA, gB = emu.decode_pair(sA, sB, cr, LA, LB or 0)
        assert gA == bytes(oracle.viterbi_frame(sA, cr, LA)), (cr, LA, LB)
        if LB is not None:
            assert gB == bytes(oracle.viterbi_frame(sB, cr, LB)), (cr, LA, LB)


def test_model_with_the_11n_window_schedule(oracle):
    """T11aViterbi<.., 192, 36> (fb11ndemod_config.hpp:199): the same machinery with the other window."""
    rng = np.random.default_rng(77)
    for cr in (0, 2):
        sA = _soft(rng, cr, 90); sB = _soft(rng, cr, 61)
        gA, gB = emu.decode_pair(sA, sB, cr, 90, 61, win=192, look=36)
        assert gA == bytes(oracle.viterbi_frame_ex(sA, cr, 90, 192, 36)) and gB == bytes(oracle.viterbi_frame_ex(sB, cr, 61, 192, 36)), cr
