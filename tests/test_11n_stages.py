"""802.11n (row f1), stage level: T11nDemap* and T11nDeinterleave*_S{0,1} -- the C restatement (oracle/so_11n.c) against the
reference's own bricks (live where oracle/_ref exists, recorded in tests/golden/ref_vectors_11n.npz everywhere), and the GPU
stage kernels (sora_hip_demap11n, sora_hip_deinterleave11n) against both."""
import os

import numpy as np
import pytest

from oracle.pyoracle import Oracle, ReferenceGraph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors_11n.npz")


@pytest.fixture(scope="module")
def o():
    return Oracle()


def test_oracle_11n_bricks_equal_recorded_reference_output(o):
    z = np.load(GOLD)
    for nb in (1, 2, 4, 6):
        for x, want in zip(z["demap_in"], z["demap_out_%d" % nb]):
            assert np.array_equal(o.demap11n(nb, x), want), nb
        for st in (0, 1):
            for x, want in zip(z["deint_in_%d" % nb], z["deint_out_%d_%d" % (nb, st)]):
                assert np.array_equal(o.deinterleave11n(nb, st, x), want), (nb, st)


def test_oracle_11n_bricks_equal_reference_bricks_live(o):
    g = ReferenceGraph()
    if not g.available():
        pytest.skip("oracle/_ref/libsora_refgraph.so not built (needs the reference tree)")
    rng = np.random.default_rng(5)
    for nb in (1, 2, 4, 6):
        for t in range(150):
            amp = (120, 200, 1000, 32767)[t % 4]
            x = rng.integers(-amp, amp + 1, size=(64, 2)).astype(np.int16)
            assert np.array_equal(o.demap11n(nb, x), g.demap11n(nb, x)), nb
        for st in (0, 1):
            idx = np.arange(52 * nb)                                      # the permutation itself, through two byte planes
            lo = g.deinterleave11n(nb, st, (idx & 0xFF).astype(np.uint8)).astype(int)
            hi = g.deinterleave11n(nb, st, (idx >> 8).astype(np.uint8)).astype(int)
            mine = o.deinterleave11n(nb, st, (idx & 0xFF).astype(np.uint8)).astype(int) | (o.deinterleave11n(nb, st, (idx >> 8).astype(np.uint8)).astype(int) << 8)
            assert np.array_equal(lo | (hi << 8), mine) and sorted(mine) == list(idx), (nb, st)


@pytest.mark.gpu
def test_gpu_11n_stage_kernels(o):
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    z = np.load(GOLD)
    rng = np.random.default_rng(6)
    for nb in (1, 2, 4, 6):
        x = np.concatenate([z["demap_in"], rng.integers(-400, 401, size=(1003, 64, 2)).astype(np.int16)])
        got = sora_amd.demap11n(torch.from_numpy(x).cuda(), nb).cpu().numpy()
        assert np.array_equal(got[:len(z["demap_in"])], z["demap_out_%d" % nb])
        for i in range(len(z["demap_in"]), len(x), 37):
            assert np.array_equal(got[i], o.demap11n(nb, x[i])), (nb, i)
        soft = np.concatenate([z["deint_in_%d" % nb], rng.integers(0, 8, size=(777, 52 * nb)).astype(np.uint8)])
        for st in (0, 1):
            gd = sora_amd.deinterleave11n(torch.from_numpy(soft).cuda(), nb, st).cpu().numpy()
            assert np.array_equal(gd[:6], z["deint_out_%d_%d" % (nb, st)])
            for i in range(6, len(soft), 29):
                assert np.array_equal(gd[i], o.deinterleave11n(nb, st, soft[i])), (nb, st, i)
    with pytest.raises(Exception):
        sora_amd.demap11n(torch.zeros((1, 64, 2), dtype=torch.int16).cuda(), 3)
