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


def test_oracle_mimo_bricks(o):
    """TMimoChannelEst (float 2x2 inverse, operation for operation) and TMimoChannelComp against the recorded output of the
    reference's bricks, and live against the bricks on random / singular / zero channels."""
    z = np.load(GOLD)
    for i in range(len(z["mimo_ltf0"])):
        h, hi = o.mimo_est11n(z["mimo_ltf0"][i], z["mimo_ltf1"][i])
        assert np.array_equal(h, z["mimo_h"][i]) and np.array_equal(hi, z["mimo_hinv"][i]), i
        x0, x1 = o.mimo_comp11n(z["mimo_hinv"][i], z["mimo_y0"][i], z["mimo_y1"][i])
        assert np.array_equal(x0, z["mimo_x0"][i]) and np.array_equal(x1, z["mimo_x1"][i]), i
    g = ReferenceGraph()
    if not g.available():
        return
    rng = np.random.default_rng(8)
    for t in range(300):
        amp = (30, 400, 3000, 32767)[t % 4]
        l0 = rng.integers(-amp, amp + 1, size=(128, 2)).astype(np.int16); l1 = rng.integers(-amp, amp + 1, size=(128, 2)).astype(np.int16)
        if t % 7 == 0:
            l1[:] = l0
        h, hi = o.mimo_est11n(l0, l1); rh, rhi = g.mimo_est11n(l0, l1)
        assert np.array_equal(h, rh) and np.array_equal(hi, rhi), t
        y0 = rng.integers(-amp, amp + 1, size=(64, 2)).astype(np.int16); y1 = rng.integers(-amp, amp + 1, size=(64, 2)).astype(np.int16)
        a = o.mimo_comp11n(rhi, y0, y1); b = g.mimo_comp11n(rhi, y0, y1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), t


def test_oracle_phase_bricks(o):
    """TFreqEstimator_11n, TFreqComp_11n, TPilotTrack_11n (dsp_math tables from libm) against recorded reference-brick output."""
    z = np.load(GOLD)
    for i in range(len(z["cfo_l0"])):
        assert np.array_equal(o.cfo_est11n(z["cfo_l0"][i], z["cfo_l1"][i]), z["cfo_state"][i])
        st, o0, o1 = o.freq_comp11n(z["fc_state_in"][i], z["fc_in0"][i], z["fc_in1"][i])
        assert np.array_equal(st, z["fc_state_out"][i]) and np.array_equal(o0, z["fc_out0"][i]) and np.array_equal(o1, z["fc_out1"][i])
    th = np.full(8, 123, np.int16)
    for a, b, want in zip(z["pt_x0"], z["pt_x1"], z["pt_theta"]):
        th = o.pilot_track11n(th, a, b)
        assert np.array_equal(th, want)
    assert len(set(z["cfo_state"][:, 1].tolist())) > 4                     # the recorded offsets really differ


def test_oracle_legacy_preamble_bricks(o):
    """TSisoChannelEst, TSisoChannelComp -> TMrcCombine, T11nSigDemap against recorded reference-brick output, and live where oracle/_ref exists."""
    z = np.load(GOLD)
    for i in range(len(z["siso_l0"])):
        assert np.array_equal(o.siso_est11n(z["siso_l0"][i], z["siso_l1"][i]), z["siso_ch"][i]), i
        x0, x1, m = o.siso_comp11n(z["siso_ch"][i], z["siso_y0"][i], z["siso_y1"][i])
        assert np.array_equal(x0, z["siso_x0"][i]) and np.array_equal(x1, z["siso_x1"][i]) and np.array_equal(m, z["siso_mrc"][i]), i
    for x, want in zip(z["sig_sym"], z["sig_soft"]):
        assert np.array_equal(o.sig_demap11n(x), want)
    assert z["siso_ch"].any() and len(np.unique(z["sig_soft"])) == 8
    g = ReferenceGraph()
    if not g.available():
        return
    rng = np.random.default_rng(77)
    for t in range(300):
        amp = (20, 300, 3000, 32767)[t % 4]
        l0 = rng.integers(-amp, amp + 1, size=(128, 2)).astype(np.int16); l1 = rng.integers(-amp, amp + 1, size=(128, 2)).astype(np.int16)
        ch = g.siso_est11n(l0, l1)
        assert np.array_equal(o.siso_est11n(l0, l1), ch), t
        y0 = rng.integers(-amp, amp + 1, size=(64, 2)).astype(np.int16); y1 = rng.integers(-amp, amp + 1, size=(64, 2)).astype(np.int16)
        c = ch if t % 2 else rng.integers(-amp, amp + 1, size=(2, 64, 2)).astype(np.int16)
        assert all(np.array_equal(a, b) for a, b in zip(o.siso_comp11n(c, y0, y1), g.siso_comp11n(c, y0, y1))), t
        s3 = rng.integers(-min(amp, 400), min(amp, 400) + 1, size=(3, 64, 2)).astype(np.int16)
        assert np.array_equal(o.sig_demap11n(s3), g.sig_demap11n(s3)), t


def _sig_record(ok, out9, fields):
    """the 12-word record sora_hip_sig_decode11n writes, from the (ok, bytes, fields) form of oracle and reference"""
    b = [int(x) for x in out9]
    return [int(x) for x in fields] + [b[0] | b[1] << 8 | b[2] << 16, b[3] | b[4] << 8 | b[5] << 16 | b[6] << 24, b[7] | b[8] << 8]


def test_oracle_sig_decode(o):
    """T11aDeinterleaveBPSK x3 -> T11nViterbiSig -> T11nSigParser: recorded reference-brick output, and live where oracle/_ref exists."""
    from gpu_util import htsig_cases
    z = np.load(GOLD)
    assert np.array_equal(htsig_cases(2024, 120), z["sigdec_soft"])
    for x, ok, by, f in zip(z["sigdec_soft"], z["sigdec_ok"], z["sigdec_bytes"], z["sigdec_fields"]):
        got = o.sig_decode11n(x)
        assert got[0] == ok and np.array_equal(got[1], by) and np.array_equal(got[2], f)
    ok = z["sigdec_ok"] == 1
    assert 10 < ok.sum() < 100 and set(z["sigdec_fields"][ok][:, 3].tolist()) == {8, 9, 10}      # all three MCS decode, failures present
    assert (z["sigdec_fields"][~ok][:, 0] == 0x80000005).all()
    g = ReferenceGraph()
    if not g.available():
        return
    for x in htsig_cases(7, 1500):
        a, b = o.sig_decode11n(x), g.sig_decode11n(x)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.gpu
def test_gpu_sig_decode(o):
    import torch
    import sora_amd
    from gpu_util import htsig_cases
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    z = np.load(GOLD)
    rec = sora_amd.sig_decode11n(torch.from_numpy(z["sigdec_soft"]).cuda()).cpu().numpy().view(np.uint32)
    for i in range(len(rec)):
        assert rec[i].tolist() == _sig_record(z["sigdec_ok"][i], z["sigdec_bytes"][i], z["sigdec_fields"][i]), i
    soft = htsig_cases(99, 3001)                                          # not a multiple of four: a partly empty last block
    rec = sora_amd.sig_decode11n(torch.from_numpy(soft).cuda()).cpu().numpy().view(np.uint32)
    nok = 0
    for i in range(len(soft)):
        want = o.sig_decode11n(soft[i]); nok += want[0]
        assert rec[i].tolist() == _sig_record(*want), i
    assert nok > 500


@pytest.mark.gpu
def test_gpu_legacy_preamble_stage_kernels(o):
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    z = np.load(GOLD)
    ch = sora_amd.siso_est11n(torch.from_numpy(z["siso_l0"]).cuda(), torch.from_numpy(z["siso_l1"]).cuda())
    assert np.array_equal(ch.cpu().numpy(), z["siso_ch"]), np.argwhere(ch.cpu().numpy() != z["siso_ch"])[:8].tolist()
    n = len(z["siso_y0"])
    x0, x1, m = sora_amd.siso_comp11n(ch, torch.from_numpy(z["siso_y0"]).cuda(), torch.from_numpy(z["siso_y1"]).cuda(),
                                      frame_index=torch.arange(n, dtype=torch.int32).cuda())
    assert np.array_equal(x0.cpu().numpy(), z["siso_x0"]) and np.array_equal(x1.cpu().numpy(), z["siso_x1"]) and np.array_equal(m.cpu().numpy(), z["siso_mrc"])
    assert np.array_equal(sora_amd.sig_demap11n(torch.from_numpy(z["sig_sym"]).cuda()).cpu().numpy(), z["sig_soft"])
    rng = np.random.default_rng(13)
    nf = 501                                                             # odd: the last block is half empty
    l0 = rng.integers(-32767, 32768, size=(nf, 128, 2)).astype(np.int16); l1 = rng.integers(-300, 301, size=(nf, 128, 2)).astype(np.int16)
    l0[7] = 0; l1[9, 5] = (-32768, -32768)
    ch = sora_amd.siso_est11n(torch.from_numpy(l0).cuda(), torch.from_numpy(l1).cuda())
    chn = ch.cpu().numpy()
    for f in range(nf):
        assert np.array_equal(chn[f], o.siso_est11n(l0[f], l1[f])), f
    ns = 1003; fi = rng.integers(0, nf, size=ns).astype(np.int32)
    y0 = rng.integers(-32767, 32768, size=(ns, 64, 2)).astype(np.int16); y1 = rng.integers(-2000, 2001, size=(ns, 64, 2)).astype(np.int16)
    x0, x1, m = (t.cpu().numpy() for t in sora_amd.siso_comp11n(ch, torch.from_numpy(y0).cuda(), torch.from_numpy(y1).cuda(), frame_index=torch.from_numpy(fi).cuda()))
    for s_ in range(ns):
        w = o.siso_comp11n(chn[fi[s_]], y0[s_], y1[s_])
        assert np.array_equal(x0[s_], w[0]) and np.array_equal(x1[s_], w[1]) and np.array_equal(m[s_], w[2]), s_
    sig = rng.integers(-400, 401, size=(333, 3, 64, 2)).astype(np.int16)
    soft = sora_amd.sig_demap11n(torch.from_numpy(sig).cuda()).cpu().numpy()
    for f in range(len(sig)):
        assert np.array_equal(soft[f], o.sig_demap11n(sig[f])), f


@pytest.mark.gpu
def test_gpu_phase_stage_kernels(o):
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    z = np.load(GOLD)
    st = sora_amd.cfo_est11n(torch.from_numpy(z["cfo_l0"]).cuda(), torch.from_numpy(z["cfo_l1"]).cuda())
    assert np.array_equal(st.cpu().numpy(), z["cfo_state"])
    rng = np.random.default_rng(12)
    l0 = rng.integers(-32767, 32768, size=(300, 128, 2)).astype(np.int16); l1 = rng.integers(-32767, 32768, size=(300, 128, 2)).astype(np.int16)
    st = sora_amd.cfo_est11n(torch.from_numpy(l0).cuda(), torch.from_numpy(l1).cuda()).cpu().numpy()
    for i in range(300):
        assert np.array_equal(st[i], o.cfo_est11n(l0[i], l1[i])), i
    # frequency compensation: 8 frames laid end to end, different burst counts
    nb = np.array([20, 20, 7, 20, 1, 20, 13, 20], np.int32); first = np.arange(8, dtype=np.int32) * 160
    state = torch.from_numpy(z["fc_state_in"].copy()).cuda()
    o0, o1 = sora_amd.freq_comp11n(torch.from_numpy(z["fc_in0"].reshape(-1, 2)).cuda(), torch.from_numpy(z["fc_in1"].reshape(-1, 2)).cuda(),
                                   torch.from_numpy(first).cuda(), torch.from_numpy(nb).cuda(), state)
    o0 = o0.cpu().numpy().reshape(8, 160, 2); o1 = o1.cpu().numpy().reshape(8, 160, 2); state = state.cpu().numpy()
    for f in range(8):
        wst, w0, w1 = o.freq_comp11n(z["fc_state_in"][f], z["fc_in0"][f][:8 * nb[f]], z["fc_in1"][f][:8 * nb[f]])
        assert np.array_equal(state[f], wst) and np.array_equal(o0[f][:8 * nb[f]], w0) and np.array_equal(o1[f][:8 * nb[f]], w1), f
        if nb[f] == 20:
            assert np.array_equal(o0[f], z["fc_out0"][f]) and np.array_equal(state[f], z["fc_state_out"][f])
    # pilot tracking: the recorded 40-symbol frame plus a long random one (more than one 64-symbol pass)
    x0 = np.concatenate([z["pt_x0"], rng.integers(-9000, 9001, size=(150, 64, 2)).astype(np.int16)])
    x1 = np.concatenate([z["pt_x1"], rng.integers(-9000, 9001, size=(150, 64, 2)).astype(np.int16)])
    state = np.zeros((2, 24), np.int16); state[0, 16:] = 123; state[1, 16:] = -7
    dstate = torch.from_numpy(state.copy()).cuda()
    th = sora_amd.pilot_track11n(torch.from_numpy(x0).cuda(), torch.from_numpy(x1).cuda(), torch.tensor([0, 40], dtype=torch.int32).cuda(),
                                 torch.tensor([40, 150], dtype=torch.int32).cuda(), dstate).cpu().numpy()
    assert np.array_equal(th[:40], z["pt_theta"])
    t = np.full(8, -7, np.int16)
    for s in range(40, 190):
        t = o.pilot_track11n(t, x0[s], x1[s])
        assert np.array_equal(th[s], t), s
    assert np.array_equal(dstate.cpu().numpy()[1, 16:], t) and np.array_equal(dstate.cpu().numpy()[0, 16:], z["pt_theta"][-1])


@pytest.mark.gpu
def test_gpu_mimo_stage_kernels(o):
    import torch
    import sora_amd
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    z = np.load(GOLD)
    rng = np.random.default_rng(9)
    n = 500
    l0 = np.concatenate([z["mimo_ltf0"], np.stack([rng.integers(-a, a + 1, size=(128, 2)) for a in rng.choice([30, 400, 3000, 32767], n)]).astype(np.int16)])
    l1 = np.concatenate([z["mimo_ltf1"], np.stack([rng.integers(-a, a + 1, size=(128, 2)) for a in rng.choice([30, 400, 3000, 32767], n)]).astype(np.int16)])
    l1[20] = l0[20]
    h, hinv = sora_amd.mimo_est11n(torch.from_numpy(l0).cuda(), torch.from_numpy(l1).cuda())
    h = h.cpu().numpy(); hinv = hinv.cpu().numpy()
    k = len(z["mimo_ltf0"])
    assert np.array_equal(h[:k], z["mimo_h"]) and np.array_equal(hinv[:k], z["mimo_hinv"])
    for i in range(k, len(l0)):
        wh, whi = o.mimo_est11n(l0[i], l1[i])
        assert np.array_equal(h[i], wh) and np.array_equal(hinv[i], whi), i
    nsym = 3000
    fidx = rng.integers(0, len(l0), nsym).astype(np.uint32)
    y0 = rng.integers(-4000, 4001, size=(nsym, 64, 2)).astype(np.int16); y1 = rng.integers(-4000, 4001, size=(nsym, 64, 2)).astype(np.int16)
    x0, x1 = sora_amd.mimo_comp11n(torch.from_numpy(hinv).cuda(), torch.from_numpy(y0).cuda(), torch.from_numpy(y1).cuda(),
                                   frame_index=torch.from_numpy(fidx.astype(np.int32)).cuda())
    x0 = x0.cpu().numpy(); x1 = x1.cpu().numpy()
    for s in range(0, nsym, 13):
        w0, w1 = o.mimo_comp11n(hinv[fidx[s]], y0[s], y1[s])
        assert np.array_equal(x0[s], w0) and np.array_equal(x1[s], w1), s


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
